#!/bin/bash
# steps/s of (critics, batch) configurations with the throughput kernels off / on / forward only
for cfg in "3 256" "4 256" "8 256" "3 1024" "8 1024" "4 512"; do
  for mode in "0 0" "1 1" "1 0"; do
    set -- $mode
    r=$(IQLHIP_TP=$1 IQLHIP_TP_BWD=$2 timeout -k 10 100 python tools/ens_run.py $cfg 3000 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_us_events']; print(f\"{d['steps_per_s']:8.0f} steps/s  f/b/u {k['k_forward']:.2f} {k['k_backward']:.2f} {k['k_update']:.2f}\")")
    echo "E B = $cfg  TP fwd=$1 bwd=$2: $r"
  done
done

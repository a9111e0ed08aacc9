#!/bin/bash
# A/B of one environment knob on ONE box: tools/ab_env.sh VAR valA valB [bench args]  (alternates A B A B)
V=$1; A=$2; B=$3; shift 3
for round in 1 2; do
  for val in "$A" "$B"; do
    env $V=$val timeout -k 10 200 python bench.py --steps 20000 --warmup 2000 --no-cpu-baseline --agents-per-gpu 0 --ensemble-q 0 "$@" 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$val', round(d['value']), d['roofline']['step']['kernel_us_events_only'])" || exit 1
  done
done

#!/bin/bash
# A/B of library builds x the throughput-kernel switch: tools/ab_libs.sh <out> lib1.so [lib2.so ...]
# (each: E = 4 / batch 1024 alone, then seed groups of 8, 4 and one seed)
OUT=$1; shift
for round in 1 2; do for lib in "$@"; do for tp in ${AB_TP:-0 1}; do
  echo "== $lib TP=$tp" >> $OUT
  IQLHIP_LIB=$PWD/iqlpref_amd/$lib IQLHIP_TP=$tp timeout -k 10 120 python tools/ens_run.py 4 1024 3000 2>/dev/null | grep "^{" >> $OUT || exit 1
  IQLHIP_LIB=$PWD/iqlpref_amd/$lib IQLHIP_TP=$tp timeout -k 10 200 python tools/group_scan.py ${AB_GROUPS:-8 4 1} 2>/dev/null | grep "^[0-9]" >> $OUT || exit 1
done; done; done
python - $OUT <<'PY'
import json, sys
lab = None
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith("=="):
        lab = l[3:]
        continue
    if l.startswith("{"):
        d = json.loads(l); k = d["kernel_us_events"]
        print(f"{lab:34s} E4/B1024 {d['steps_per_s']:8.0f} steps/s {d['us_per_step']:6.2f} us  f/b/u {k['k_forward']:.2f} {k['k_backward']:.2f} {k['k_update']:.2f}")
    elif l and l[0].isdigit():
        K, rest = l.split(" ", 1); d = json.loads(rest); k = d["kernel_us_events"]
        print(f"{lab:34s} K={K:2s}     {d['steps_per_s']:8.0f} steps/s {d['group_step_us']:6.2f} us  f/b/u {k['k_forward']:.2f} {k['k_backward']:.2f} {k['k_update']:.2f}")
PY

#!/bin/bash
# A/B of two builds of libiqlhip on ONE box (devices differ by ~8 %): tools/ab.sh <libA> <libB> [bench args]
# alternates A B A B and prints steps/s and per-kernel microseconds of each run.
A=$1; B=$2; shift 2
for round in 1 2; do
  for lib in "$A" "$B"; do
    IQLHIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 20000 --warmup 2000 --no-cpu-baseline --no-relabel --agents-per-gpu 0 --ensemble-q 0 "$@" 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']), d['roofline']['step']['kernel_us'])" || exit 1
  done
done

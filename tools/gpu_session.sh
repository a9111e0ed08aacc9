#!/bin/bash
# One GPU-box session (each gpurun call pays ~10 min to get a box: do everything in one):
#   tools/gpu_session.sh <tag> [tests] [stamps] [relabel] [bench] [bench20] [profile]
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
# a step that was killed at its time limit ends the session (no further GPU step after a hang)
stop_if_killed() { if [ "$1" = 124 ] || [ "$1" = 137 ]; then echo "$2 was killed at its limit: session ends"; exit 1; fi; }
for what in "$@"; do
  case $what in
    tests)
      IQL_TEST_DIAG=$OUT/diag.txt timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/tests.log 2>&1
      rc=$?; echo "tests rc=$rc"; stop_if_killed $rc tests; tail -4 $OUT/tests.log | cut -c1-300 ;;
    testsall)
      IQL_TEST_DIAG=$OUT/diag.txt timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
      rc=$?; echo "tests rc=$rc"; stop_if_killed $rc tests; tail -12 $OUT/tests.log | cut -c1-300 ;;
    stamps)
      for k in 0 1 2; do
        STAMP_GRAPH=8 STAMP_KERNEL=$k timeout -k 10 120 python tools/stamps.py bf16 > $OUT/stamps_k$k.txt 2>&1; rc=$?; stop_if_killed $rc stamps
      done
      head -40 $OUT/stamps_k2.txt ;;
    relabel)
      timeout -k 10 300 python tools/bench_relabel.py --cpu > $OUT/relabel.json 2> $OUT/relabel.err
      rc=$?; echo "relabel rc=$rc"; stop_if_killed $rc relabel; cat $OUT/relabel.json | head -120 ;;
    bench)
      timeout -k 10 300 python bench.py --no-cpu-baseline --no-relabel > $OUT/bench.json 2> $OUT/bench.err
      rc=$?; echo "bench rc=$rc"; stop_if_killed $rc bench; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('value', round(d['value']), d['roofline']['step']['kernel_us'], 'frac', round(d['roofline']['frac'],3))
print('agents', d.get('agents_per_gpu',{}).get('value'), 'ensemble', d.get('ensemble_q',{}).get('value'))" ;;
    bench20)
      timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err
      rc=$?; echo "bench20 rc=$rc"; stop_if_killed $rc bench20; tail -c 1500 $OUT/bench20.json ;;
    ckpt)
      timeout -k 10 120 python tools/make_checkpoint.py $OUT/our_checkpoint.pt > $OUT/ckpt.log 2>&1; echo "ckpt rc=$?"; tail -2 $OUT/ckpt.log ;;
    groupscan)
      timeout -k 10 300 python tools/group_scan.py > $OUT/group_scan.txt 2>&1; rc=$?; echo "groupscan rc=$rc"; stop_if_killed $rc groupscan; grep -v Dataset $OUT/group_scan.txt | cut -c1-300 ;;
    profile)
      bash tools/profile.sh $TAG > $OUT/profile.log 2>&1; rc=$?; echo "profile rc=$rc"; tail -5 $OUT/profile.log | cut -c1-300; [ $rc = 0 ] || exit 1 ;;
  esac
done

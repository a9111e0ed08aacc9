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
      python -m iqlpref_amd.build --stamps > $OUT/build_stamps.log 2>&1 || { echo "stamps build failed"; tail -5 $OUT/build_stamps.log; exit 1; }
      for k in 0 1 2; do
        STAMP_GRAPH=8 STAMP_KERNEL=$k timeout -k 10 120 python tools/stamps.py bf16 > $OUT/stamps_k$k.txt 2>&1; rc=$?; stop_if_killed $rc stamps
      done
      head -40 $OUT/stamps_k2.txt ;;
    stampsens)
      # the E = 4 critics / batch 1024 step (BASELINE configs[4]): per-work-group stage timeline of the three kernels
      python -m iqlpref_amd.build --stamps > $OUT/build_stamps.log 2>&1 || { echo "stamps build failed"; tail -5 $OUT/build_stamps.log; exit 1; }
      STAMP_E=4 STAMP_B=1024 STAMP_GRAPH=8 timeout -k 10 120 python tools/stamps.py bf16 > $OUT/stamps_e4_b1024.txt 2>&1; rc=$?; stop_if_killed $rc stampsens
      grep -v Dataset $OUT/stamps_e4_b1024.txt | cut -c1-400 ;;
    relabel)
      timeout -k 10 300 python tools/bench_relabel.py --cpu > $OUT/relabel.json 2> $OUT/relabel.err
      rc=$?; echo "relabel rc=$rc"; stop_if_killed $rc relabel; cat $OUT/relabel.json | head -120 ;;
    bench)
      timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
      rc=$?; echo "bench rc=$rc"; stop_if_killed $rc bench; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('value', round(d['value']), d['roofline']['step']['kernel_us'], 'frac', round(d['roofline']['frac'],3))
print('agents', d.get('agents_per_gpu',{}).get('value'), 'ensemble', d.get('ensemble_q',{}).get('value'))" ;;
    bench20)
      timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err
      rc=$?; echo "bench20 rc=$rc"; stop_if_killed $rc bench20; tail -c 1500 $OUT/bench20.json ;;
    waitmode)
      # host wait policy at the end of a timed block: HIP's active-wait window (us) before it blocks on the interrupt
      for round in 1 2; do for w in 0 100 10000; do
        ROC_ACTIVE_WAIT_TIMEOUT=$w timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-relabel --agents-per-gpu 0 --ensemble-q 0 2>/dev/null |
          python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wait', $w, round(d['value']), d['ms_per_step'], d['timing']['ms_per_step_device'])" || exit 1
      done; done ;;
    unrollscan)
      # steps per hipGraph inside the driver's 20-step timed block: when does the GPU see its first packet
      for round in 1 2; do for u in 20 10 5 4 2 1; do
        timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --unroll $u --no-cpu-baseline --no-relabel --agents-per-gpu 0 --ensemble-q 0 2>/dev/null |
          python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unroll', $u, round(d['value']), d['ms_per_step'], d['timing']['ms_per_step_device'])" || exit 1
      done; done ;;
    mlpscan)
      # reward-MLP kernel: the product build against $EXTRA_LIBS
      for lib in libiqlhip.so $EXTRA_LIBS; do
        IQLHIP_LIB=$PWD/iqlpref_amd/$lib timeout -k 10 120 python tools/mlp_scan.py 2>&1 | grep -v "Dataset\|amdgpu.ids" >> $OUT/mlpscan.txt; rc=$?; stop_if_killed $rc mlpscan
      done
      cat $OUT/mlpscan.txt ;;
    bwdpw)
      # parts of the dZ1 columns per backward work-group in a group of 8 (and of 4, 2): IQLHIP_BWD_PW
      for round in 1 2; do for pw in 1 2 4; do
        echo "== IQLHIP_BWD_PW=$pw" >> $OUT/bwdpw.txt
        IQLHIP_BWD_PW=$pw timeout -k 10 200 python tools/group_scan.py 8 4 2>&1 | grep -v "Dataset\|amdgpu.ids" >> $OUT/bwdpw.txt; rc=$?; stop_if_killed $rc bwdpw
      done; done
      cut -c1-260 $OUT/bwdpw.txt ;;
    envab)
      # A/B of an environment knob ($ENVAB, values 0 / 1): one seed and groups of 2, 4, 8
      for round in 1 2; do for v in ${ENVAB_VALUES:-0 1}; do
        echo "== $ENVAB=$v" >> $OUT/envab.txt
        env $ENVAB=$v timeout -k 10 300 python tools/group_scan.py 1 2 4 8 2>&1 | grep -v "Dataset\|amdgpu.ids" >> $OUT/envab.txt; rc=$?; stop_if_killed $rc envab
      done; done
      cut -c1-260 $OUT/envab.txt ;;
    ckpt)
      timeout -k 10 120 python tools/make_checkpoint.py $OUT/our_checkpoint.pt > $OUT/ckpt.log 2>&1; echo "ckpt rc=$?"; tail -2 $OUT/ckpt.log ;;
    groupscan)
      timeout -k 10 300 python tools/group_scan.py > $OUT/group_scan.txt 2>&1; rc=$?; echo "groupscan rc=$rc"; stop_if_killed $rc groupscan; grep -v Dataset $OUT/group_scan.txt | cut -c1-300 ;;
    membench)
      for k in 8 1; do timeout -k 10 120 tools/membench3 $k > $OUT/membench3_k$k.txt 2>&1; rc=$?; stop_if_killed $rc membench; done
      cat $OUT/membench3_k8.txt; cat $OUT/membench3_k1.txt ;;
    stampsgroup)
      python -m iqlpref_amd.build --stamps > $OUT/build_stamps.log 2>&1 || { echo "stamps build failed"; tail -5 $OUT/build_stamps.log; exit 1; }
      for k in 0 1 2; do
        STAMP_GROUP=8 STAMP_GRAPH=8 STAMP_KERNEL=$k timeout -k 10 120 python tools/stamps.py bf16 > $OUT/stamps_g8_k$k.txt 2>&1; rc=$?; stop_if_killed $rc stampsgroup
      done
      grep -v Dataset $OUT/stamps_g8_k2.txt | cut -c1-400 | head -40 ;;
    groupab)
      # group of 8: the committed build (libiqlhip_prev.so, tools/build_prev.sh) against the working tree
      for round in 1 2; do for lib in libiqlhip_prev.so libiqlhip.so $EXTRA_LIBS; do
        echo "== $lib" >> $OUT/groupab.txt
        IQLHIP_LIB=$PWD/iqlpref_amd/$lib timeout -k 10 200 python tools/group_scan.py 8 2>&1 | grep -v "Dataset\|amdgpu.ids" >> $OUT/groupab.txt; rc=$?; stop_if_killed $rc groupab
      done; done
      cut -c1-300 $OUT/groupab.txt ;;
    groupbisect)
      # which build variants keep a seed group bit-identical to the seeds alone
      for lib in $EXTRA_LIBS; do
        IQLHIP_LIB=$PWD/iqlpref_amd/$lib timeout -k 10 300 python -m pytest tests -m gpu -q -k "seed_group_matches_separate_runs and group" > $OUT/bisect_$lib.log 2>&1
        echo "$lib rc=$?"; tail -2 $OUT/bisect_$lib.log | cut -c1-200
      done ;;
    grouptests)
      timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "seed_group or seeds_per_gpu or continue or group_close or shapes_beyond or trajectory_parity" > $OUT/grouptests.log 2>&1
      rc=$?; echo "grouptests rc=$rc"; stop_if_killed $rc grouptests; tail -5 $OUT/grouptests.log | cut -c1-300; [ $rc = 0 ] || exit 1 ;;
    stampsall)
      python -m iqlpref_amd.build --stamps > $OUT/build_stamps.log 2>&1 || { echo "stamps build failed"; tail -5 $OUT/build_stamps.log; exit 1; }
      STAMP_GROUP=8 STAMP_GRAPH=8 STAMP_ALL=1 timeout -k 10 120 python tools/stamps.py bf16 > $OUT/stamps_g8_all.txt 2>&1; rc=$?; stop_if_killed $rc stampsall
      grep -v Dataset $OUT/stamps_g8_all.txt | cut -c1-200
      for m in 3 6; do
        STAMP_GROUP=8 STAMP_GRAPH=8 STAMP_MEMBER=$m timeout -k 10 120 python tools/stamps.py bf16 > $OUT/stamps_g8_m$m.txt 2>&1; rc=$?; stop_if_killed $rc stampsall
        echo "--- member $m"; grep -v Dataset $OUT/stamps_g8_m$m.txt | cut -c1-420
      done ;;
    varianttests)
      IQLHIP_LIB=$PWD/iqlpref_amd/libiqlhip_$VARIANT.so IQL_TEST_DIAG=$OUT/diag_$VARIANT.txt timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/tests_$VARIANT.log 2>&1
      rc=$?; echo "varianttests rc=$rc"; stop_if_killed $rc varianttests; tail -8 $OUT/tests_$VARIANT.log | cut -c1-300 ;;
    abvariant)
      # A/B of iqlpref_amd/libiqlhip_$VARIANT.so against the product build: one seed, then a group of 8
      bash tools/ab.sh iqlpref_amd/libiqlhip.so iqlpref_amd/libiqlhip_$VARIANT.so > $OUT/ab_$VARIANT.txt 2>&1; rc=$?; stop_if_killed $rc ab
      cat $OUT/ab_$VARIANT.txt
      for round in 1 2; do for lib in libiqlhip.so libiqlhip_$VARIANT.so; do
        echo "== $lib" >> $OUT/ab_${VARIANT}_group.txt
        IQLHIP_LIB=$PWD/iqlpref_amd/$lib timeout -k 10 200 python tools/group_scan.py 8 2>&1 | grep -v Dataset >> $OUT/ab_${VARIANT}_group.txt; rc=$?; stop_if_killed $rc abgroup
      done; done
      cut -c1-300 $OUT/ab_${VARIANT}_group.txt ;;
    profile)
      bash tools/profile.sh $TAG > $OUT/profile.log 2>&1; rc=$?; echo "profile rc=$rc"; tail -5 $OUT/profile.log | cut -c1-300; [ $rc = 0 ] || exit 1 ;;
  esac
done

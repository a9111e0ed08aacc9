#!/bin/bash
# Regenerates the rocprofv3 evidence under profiles/ (run on the GPU box through gpurun):
#   tools/profile.sh <tag>        e.g. r02
# Passes: kernel trace + stats; PMC FETCH_SIZE; PMC WRITE_SIZE; PMC MFMA busy (each its own run:
# the TCC counters do not fit one pass, and --pmc is never combined with other trace domains).
# The profiled program is `python bench.py ...` itself (nothing between rocprofv3 and python).
set -e
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_$TAG
# (--unroll 50: the traced runs replay hipGraphs of 50 steps.  bench.py's default, plain kernel launches,
# is paced by the tracer's per-dispatch interception -- ~6 us per launch, 18.7 us per step instead of
# 14.9 -- and the "durations" of kernels that tile the step then measure the tool, not the kernels.)
ARGS="--unroll 50 --no-cpu-baseline --no-relabel --no-pen --agents-per-gpu 0 --ensemble-q 0 --min-timed-s 0.05"
mkdir -p $OUT
# which library the set was collected on (bench.py labels every profile-derived figure with it)
python -c "from iqlpref_amd import _lib; import json; print(json.dumps({'build': _lib.build_tag(), 'tag': '$TAG'}))" > $OUT/meta.json
export TMPDIR=/tmp
# The tracer's pacing of graph replays is bimodal on this pool: the same library on one box traces at 57-59k or
# at 42-45k steps/s (67k untraced), run by run (tools/ab_trace.sh, gpurun_out of round 4), and in the slow mode
# every kernel measures 5-9 % longer (wider gaps between dispatches: colder caches and clocks).  The pass is
# repeated up to three times and the LEAST perturbed run (highest traced rate) is kept; all rates go to meta.json.
best=0; rates=""
for try in 1 2 3; do
  rm -rf $OUT/trace_try
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_try -- python bench.py --steps 3000 --warmup 300 $ARGS > $OUT/trace_try.json
  rate=$(python -c "import json; print(int(json.loads(open('$OUT/trace_try.json').read().strip().splitlines()[-1])['value']))")
  rates="$rates $rate"
  if [ "$rate" -gt "$best" ]; then best=$rate; rm -rf $OUT/trace; mv $OUT/trace_try $OUT/trace; mv $OUT/trace_try.json $OUT/trace.json; fi
  if [ "$rate" -gt 52000 ]; then break; fi
done
rm -rf $OUT/trace_try $OUT/trace_try.json
python - <<PY
import json
m = json.load(open("$OUT/meta.json"))
m["headline_trace_rates_steps_per_s"] = [int(x) for x in "$rates".split()]
m["headline_trace_kept"] = $best
json.dump(m, open("$OUT/meta.json", "w"))
PY
echo "trace done (traced rates:$rates; kept $best)"
# counter passes: short regions only.  Under a serialising counter pass, a long region of back-to-back
# hipGraph launches ends in HSA_STATUS_ERROR_INVALID_PACKET_FORMAT: the packet the queue dumps is one of
# OUR kernel dispatches whose `setup` field (number of grid dimensions) reads 0 -- it was rewritten on its
# way through the profiler's intercept queue (r2m; again in round 3, r4b, with the library's queue depth
# bounded to 6144 dispatches: depth is not the cause).  Kernel-trace passes are not affected.
PMC="--steps 500 --warmup 100 --no-sustained $ARGS"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python bench.py $PMC > $OUT/fetch.json
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python bench.py $PMC > $OUT/write.json
echo "write done"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -- python bench.py $PMC > $OUT/mfma.json
echo "mfma done"
# the relabel kernels (reward MLP, CVaR, preference transformer, dataset preparation): kernel trace only
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/relabel -- python tools/bench_relabel.py > $OUT/relabel.json
echo "relabel done"
# the seed-group launches (gridDim.y = 8): kernel trace only
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/group -- python tools/group_scan.py 8 > $OUT/group.json
echo "group done"
find $OUT/group -name "*kernel_stats.csv" -exec cp {} $OUT/group8_kernel_stats.csv \;
# ... and their fabric traffic (short region: see above)
GROUP_SCAN_STEPS=500 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/gfetch -- python tools/group_scan.py 8 > $OUT/gfetch.json
GROUP_SCAN_STEPS=500 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/gwrite -- python tools/group_scan.py 8 > $OUT/gwrite.json
echo "group pmc done"
# BASELINE configs[4] (E = 4 critics, batch 1024) and configs[2] (pen shapes + dropout, batch 256): kernel trace
# + stats, and the fabric traffic of the ensemble step (short regions, see above)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ens -- python tools/ens_run.py 4 1024 3000 > $OUT/ens4.json
find $OUT/ens -name "*kernel_stats.csv" -exec cp {} $OUT/ens4_kernel_stats.csv \;
ENS_NO_TIMING=1 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/efetch -- python tools/ens_run.py 4 1024 500 > $OUT/efetch.json
ENS_NO_TIMING=1 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/ewrite -- python tools/ens_run.py 4 1024 500 > $OUT/ewrite.json
python tools/pmc_summary.py $OUT/efetch 1 $OUT/ens4_pmc_fetch_size.json > /dev/null
python tools/pmc_summary.py $OUT/ewrite 1 $OUT/ens4_pmc_write_size.json > /dev/null
python tools/traffic_json.py $OUT/ens4_pmc_fetch_size.json $OUT/ens4_pmc_write_size.json $OUT/ens4_traffic.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pen -- python tools/ens_run.py 2 256 3000 pen > $OUT/pen.json
find $OUT/pen -name "*kernel_stats.csv" -exec cp {} $OUT/pen_kernel_stats.csv \;
rm -rf $OUT/ens $OUT/efetch $OUT/ewrite $OUT/pen
echo "ens4 / pen done"
# the general layer-wise step (three hidden layers of 256 units, batch 256): kernel trace + stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gen -- python tools/general_run.py 256 3 256 2 3000 > $OUT/general.json
find $OUT/gen -name "*kernel_stats.csv" -exec cp {} $OUT/general_kernel_stats.csv \;
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/genf -- python tools/general_run.py 256 3 256 2 300 > /dev/null
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/genw -- python tools/general_run.py 256 3 256 2 300 > /dev/null
python tools/pmc_summary.py $OUT/genf 1 $OUT/general_pmc_fetch_size.json > /dev/null
python tools/pmc_summary.py $OUT/genw 1 $OUT/general_pmc_write_size.json > /dev/null
python tools/traffic_json.py $OUT/general_pmc_fetch_size.json $OUT/general_pmc_write_size.json $OUT/general_traffic.json
rm -rf $OUT/gen $OUT/genf $OUT/genw
echo "general done"
python tools/pmc_summary.py $OUT/gfetch 1 $OUT/group8_pmc_fetch_size.json > /dev/null
python tools/pmc_summary.py $OUT/gwrite 1 $OUT/group8_pmc_write_size.json > /dev/null
python tools/traffic_json.py $OUT/group8_pmc_fetch_size.json $OUT/group8_pmc_write_size.json $OUT/group8_traffic.json
rm -rf $OUT/gfetch $OUT/gwrite
find $OUT/relabel -name "*kernel_stats.csv" -exec cp {} $OUT/relabel_kernel_stats.csv \;
python tools/pmc_summary.py $OUT/fetch 1 $OUT/pmc_fetch_size.json > /dev/null
python tools/pmc_summary.py $OUT/write 1 $OUT/pmc_write_size.json > /dev/null
python tools/pmc_summary.py $OUT/mfma 1 $OUT/pmc_mfma.json > /dev/null
python tools/traffic_json.py $OUT/pmc_fetch_size.json $OUT/pmc_write_size.json $OUT/traffic.json
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# keep only the summaries (the raw per-dispatch CSVs are tens of MB)
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/mfma $OUT/relabel $OUT/group
ls -la $OUT

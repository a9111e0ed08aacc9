#!/bin/bash
# Regenerates the rocprofv3 evidence under profiles/ (run on the GPU box through gpurun):
#   tools/profile.sh <tag>        e.g. r02
# Passes: kernel trace + stats; PMC FETCH_SIZE; PMC WRITE_SIZE; PMC MFMA busy (each its own run:
# the TCC counters do not fit one pass, and --pmc is never combined with other trace domains).
# The profiled program is `python bench.py ...` itself (nothing between rocprofv3 and python).
set -e
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_$TAG
ARGS="--no-cpu-baseline --no-relabel --agents-per-gpu 0 --ensemble-q 0 --min-timed-s 0.05"
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 3000 --warmup 300 $ARGS > $OUT/trace.json
echo "trace done"
# counter passes: short regions only (a counter pass serialises every dispatch and the profiler's
# intercepted queue holds 16k packets: the 20,000-step `sustained` region overran it in r2m)
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
find $OUT/relabel -name "*kernel_stats.csv" -exec cp {} $OUT/relabel_kernel_stats.csv \;
python tools/pmc_summary.py $OUT/fetch 1 $OUT/pmc_fetch_size.json > /dev/null
python tools/pmc_summary.py $OUT/write 1 $OUT/pmc_write_size.json > /dev/null
python tools/pmc_summary.py $OUT/mfma 1 $OUT/pmc_mfma.json > /dev/null
python tools/traffic_json.py $OUT/pmc_fetch_size.json $OUT/pmc_write_size.json $OUT/traffic.json
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# keep only the summaries (the raw per-dispatch CSVs are tens of MB)
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/mfma $OUT/relabel $OUT/group
ls -la $OUT

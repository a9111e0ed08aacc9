# The same traced command several times on one box (is the tracer's pacing stable?):
#   AB_N=6 bash tools/ab_trace.sh        optionally AB_LIB=<variant .so under iqlpref_amd/> alternates with the product build
export TMPDIR=/tmp
ARGS="--unroll 50 --no-cpu-baseline --no-relabel --no-pen --agents-per-gpu 0 --ensemble-q 0 --min-timed-s 0.05"
run() {
  rm -rf /tmp/tr_x
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_x -- python bench.py --steps 3000 --warmup 300 $ARGS > /tmp/tr.json 2>/dev/null
  python - <<PY
import csv,glob,json
d=json.loads(open('/tmp/tr.json').read().strip().splitlines()[-1])
f=glob.glob("/tmp/tr_x/**/*kernel_stats.csv", recursive=True)[0]
print("$1", round(d['value']), [(r["Name"].split("(")[0].replace("void iqlhip::","")[:12], round(float(r["AverageNs"]))) for r in list(csv.DictReader(open(f)))[:3]])
PY
}
for i in $(seq 1 ${AB_N:-6}); do
  unset IQLHIP_LIB
  run product
  if [ -n "$AB_LIB" ]; then export IQLHIP_LIB=$GRAFT_REPO_ROOT/iqlpref_amd/$AB_LIB; run $AB_LIB; fi
done

"""Summarise rocprofv3 --pmc CSV output (counter_collection) into bytes per step.
Usage: python tools/pmc_summary.py <dir with *_counter_collection.csv> <steps> [out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d, steps = sys.argv[1], int(sys.argv[2])
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            tot[row["Counter_Name"]][k] += float(row["Counter_Value"])
            cnt[row["Counter_Name"]][k] += 1
out = {}
for c in tot:
    for k, v in tot[c].items():
        if ("iqlhip::k_" in k or "iqlhip::kd_" in k) and any(w in k for w in ("forward", "backward", "update", "stage")):
            name = k.split("iqlhip::")[1].split("<")[0]
            out.setdefault(name, {})[c] = {"sum": v, "dispatches": cnt[c][k], "per_dispatch": v / cnt[c][k]}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)

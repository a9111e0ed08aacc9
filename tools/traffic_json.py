"""bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 for every step kernel, from the two PMC
summaries of tools/pmc_summary.py (gfx950: FETCH_SIZE counts half of a wide coalesced read,
MI355X_MICROARCH.md section HBM), tagged with the build the counters were collected on.
Usage: python tools/traffic_json.py <pmc_fetch.json> <pmc_write.json> <out.json>"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iqlpref_amd import _lib  # noqa: E402

fetch, write = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
out = {"note": "bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half of a "
               "wide coalesced read; separate --pmc passes of the bench command, tools/profile.sh)",
       "build": _lib.build_tag()}
step = 0.0
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, {}).get("FETCH_SIZE", {}).get("per_dispatch", 0.0)
    w = write.get(k, {}).get("WRITE_SIZE", {}).get("per_dispatch", 0.0)
    out[f"{k}_bytes_per_launch"] = (2.0 * f + w) * 1024.0
    if k != "k_stage":  # launched once per call, not once per step
        step += out[f"{k}_bytes_per_launch"]
out["step_bytes"] = step
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of profiles/run_prof.sh into a record of profiles/traffic.json.

    python profiles/collect_traffic.py <gpurun_out/prof_TAG> <key> [note]

key = the one bench.py looks up: workload/k<k>/<implicit>/<Gr>x<Gc>/hot<hot_buffers>.  Per launch of k_epoch_free:
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 - on gfx950 FETCH_SIZE reports half of a coalesced read stream
(MI355X_MICROARCH.md, HBM section; calibrated in round 1 against the known P-row volume).  The record carries the hash
of the kernel sources (bench.kernel_source_hash), and bench.py quotes it only for a build with the same hash.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402


def counter_mean(d, name, kernel="k_epoch_free"):
    vals = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == name:
                vals[row["Dispatch_Id"]] = vals.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    if not vals:
        raise SystemExit(f"no {name} rows for {kernel} under {d}")
    v = list(vals.values())
    return sum(v) / len(v), len(v)


d, key = sys.argv[1], sys.argv[2]
fetch, nf = counter_mean(os.path.join(d, "fetch"), "FETCH_SIZE")
write, nw = counter_mean(os.path.join(d, "write"), "WRITE_SIZE")
path = os.path.join(ROOT, "profiles", "traffic.json")
db = json.load(open(path))
db[key] = {"fetch_kb": round(fetch, 1), "write_kb": round(write, 1), "traffic_bytes": int((2 * fetch + write) * 1024),
           "launches": [nf, nw], "source_hash": kernel_source_hash(), "round": 3,
           "note": sys.argv[3] if len(sys.argv) > 3 else ""}
json.dump(db, open(path, "w"), indent=2)
print(key, db[key])

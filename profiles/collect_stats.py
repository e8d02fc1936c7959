#!/usr/bin/env python3
"""profiles/r3_shapes_stats.csv from the rocprofv3 runs of one gpurun call (gpurun_out/prof_r3_*): the epoch kernel's row of
each run's kernel_stats.csv next to what bench.py printed in the same run.  python profiles/collect_stats.py"""
import csv, glob, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
runs = (("netflix", "gpurun_out/prof_r3_nf/stats"), ("netflix_frozen", "gpurun_out/prof_r3_nf_frozen/stats"),
        ("netflix_k40", "gpurun_out/prof_r3_nf_k40/stats"), ("ml25m", "gpurun_out/prof_r3_ml25m/stats"),
        ("ml1m", "gpurun_out/prof_r3_other/ml1m"), ("yahoo", "gpurun_out/prof_r3_other/yahoo"), ("nf64", "gpurun_out/prof_r3_other/nf64"),
        ("nf256", "gpurun_out/prof_r3_other/nf256"), ("ml100k", "gpurun_out/prof_r3_other/ml100k"))
TRAFFIC_KEY = {"netflix": "netflix/k128/none/288x254/hot8", "netflix_frozen": "netflix/k128/frozen/288x254/hot8",
               "netflix_k40": "netflix/k40/none/288x254/hot8", "ml25m": "ml25m/k128/none/288x256/hot8"}
rows = []
for tag, d in runs:
    p = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    r = next(r for r in csv.DictReader(open(p)) if "k_epoch_free" in r["Name"])
    log = d + ".log" if os.path.exists(d + ".log") else os.path.dirname(d) + "/stats.log"
    b = [json.loads(line) for line in open(log) if line.startswith("{")][-1]
    rf = b["roofline"]
    if rf.get("compulsory_frac") is None and tag in TRAFFIC_KEY:   # the stats run came before the PMC passes of the same call
        rec = json.load(open("profiles/traffic.json"))[TRAFFIC_KEY[tag]]
        if rec.get("source_hash") == rf.get("source_hash"):
            rf["compulsory_frac"] = rec["traffic_bytes"] / (float(r["AverageNs"]) * 1e-9) / 8e12
    rows.append((tag, r["Name"][:75], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6,
                 b["ms_per_step"], b["value"] / 1e9, rf["frac"], rf["compulsory_frac"], b["config"]["grid"]))
with open("profiles/r3_shapes_stats.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 <args>  (round-3 build, one MI355X): the epoch "
            "kernel's row of each run's kernel_stats.csv (calls include the warm-up launches), next to what bench.py printed in the same run; "
            "compulsory_frac = PMC bytes of that configuration (profiles/traffic.json, same source hash) / launch time / 8 TB/s\n")
    f.write("tag,grid,kernel,calls,avg_ms,min_ms,max_ms,bench_ms_per_step,G_updates_per_s,roofline_frac_algorithmic,compulsory_frac_pmc\n")
    for r in rows:
        f.write('%s,%dx%d,"%s",%s,%.3f,%.3f,%.3f,%.3f,%.3f,%.3f,%s\n' % (r[0], r[10][0], r[10][1], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8],
                                                                         "" if r[9] is None else "%.3f" % r[9]))
print(open("profiles/r3_shapes_stats.csv").read())

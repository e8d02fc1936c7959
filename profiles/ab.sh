#!/bin/bash
# A/B runs of bench.py on the GPU box: bash profiles/ab.sh <outdir> <tag>:<bench args> ...
OUT=$1; shift
mkdir -p $OUT
for spec in "$@"; do
  tag=${spec%%:*}; args=${spec#*:}
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 --warmup 2 $args > $OUT/$tag.json 2> $OUT/$tag.err || { echo "$tag FAILED"; tail -5 $OUT/$tag.err; exit 1; }
  python - "$OUT/$tag.json" "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], d["config"]["grid"], "%.3f ms"%d["ms_per_step"], "%.2f G/s"%(d["value"]/1e9), "frac %.3f"%d["roofline"]["frac"], "hot", d["roofline"]["hot_segment_ratings"], "rmse %.4f"%d["rmse"]["test_after"])
PY
done

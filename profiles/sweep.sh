#!/bin/bash
# bash profiles/sweep.sh <outfile> "<common args>" "<variant 1>" "<variant 2>" ...
# a variant is bench.py arguments, optionally preceded by environment assignments: "MFSGD_STREAM_DEPTH=3 --grid 2048,508"
OUT=$1; COMMON=$2; shift 2
: > $OUT
for v in "$@"; do
  echo "== $v" >> $OUT
  envs=(); args=()
  for w in $v; do
    if [[ ${#args[@]} -eq 0 && "$w" == [A-Z_]*=* ]]; then envs+=("$w"); else args+=("$w"); fi
  done
  ( for e in "${envs[@]}"; do export "$e"; done
    timeout -k 10 300 python -u bench.py --no-cpu-baseline $COMMON "${args[@]}" 2>/dev/null ) | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%.1f Mupd/s  frac=%.3f  ms=%s  %s rmse=%.4f' % (d['value']/1e6, d['roofline']['frac'], [round(x,1) for x in d['roofline']['kernel_ms']], d['config']['grid'], d['rmse']['test_after']))
" >> $OUT || { echo "FAILED rc=$?" >> $OUT; exit 1; }
done
cat $OUT

#!/bin/bash
# bash profiles/sweep.sh <outfile> "<common args>" "<variant args 1>" "<variant args 2>" ...
OUT=$1; COMMON=$2; shift 2
: > $OUT
for v in "$@"; do
  echo "== $v" >> $OUT
  timeout -k 10 300 python -u bench.py --no-cpu-baseline $COMMON $v 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%.1f Mupd/s  frac=%.3f  ms=%s  %s rmse=%.4f' % (d['value']/1e6, d['roofline']['frac'], [round(x,1) for x in d['roofline']['kernel_ms']], d['config']['grid'], d['rmse']['test_after']))
" >> $OUT || { echo "FAILED rc=$?" >> $OUT; exit 1; }
done
cat $OUT

#!/bin/bash
# bash profiles/sweep_env.sh <outfile> "<bench args>" FLAGS1 FLAGS2 ...   (MFSGD_DEBUG_FLAGS timing experiments)
OUT=$1; ARGS=$2; shift 2
: > $OUT
for f in "$@"; do
  echo "== MFSGD_DEBUG_FLAGS=$f" >> $OUT
  MFSGD_DEBUG_FLAGS=$f timeout -k 10 300 python -u bench.py --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%.1f Mupd/s  frac=%.3f  ms=%s  %s rmse=%.4f' % (d['value']/1e6, d['roofline']['frac'], [round(x,1) for x in d['roofline']['kernel_ms']], d['config']['grid'], d['rmse']['test_after']))
" >> $OUT || { echo "FAILED rc=$?" >> $OUT; exit 1; }
done
cat $OUT

#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/run_prof.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{stats,fetch,write}: kernel-trace stats and the two HBM PMC passes (separate runs).
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/stats.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/fetch.log 2>&1 || exit 2
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/write.log 2>&1 || exit 3
find $OUT -name "*.csv" | head -20

set -o pipefail
R=$(pwd)
bash profiles/run_prof.sh r3_nf --steps 6 --warmup 2 || exit 1
bash profiles/run_prof.sh r3_nf_frozen --steps 6 --warmup 2 --implicit frozen || exit 2
bash profiles/run_prof.sh r3_nf_k40 --steps 6 --warmup 2 --k 40 || exit 3
bash profiles/run_prof.sh r3_ml25m --steps 6 --warmup 2 --workload ml25m || exit 3
mkdir -p $R/gpurun_out/prof_r3_other
cd /tmp && export TMPDIR=/tmp
for spec in "ml1m:--workload ml1m --steps 20" "yahoo:--workload yahoo --steps 3" "nf64:--k 64" "nf256:--k 256" "ml100k:--workload ml100k --steps 20"; do
  tag=${spec%%:*}; args=${spec#*:}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3_other/$tag -- python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 $args > $R/gpurun_out/prof_r3_other/$tag.log 2>&1 || exit 4
done

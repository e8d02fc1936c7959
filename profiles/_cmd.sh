mkdir -p gpurun_out/s3j
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/s3j/tests.log 2>&1 ; echo "tests rc=$?" >> gpurun_out/s3j/tests.log
tail -3 gpurun_out/s3j/tests.log
grep -q "rc=0" gpurun_out/s3j/tests.log || exit 1
bash profiles/ab.sh gpurun_out/s3j "nf:" "nf_frozen:--implicit frozen" "ml25m:--workload ml25m" "ml1m:--workload ml1m" "nf64:--k 64" "nf256:--k 256" "yahoo:--workload yahoo --steps 4" "ml100k:--workload ml100k" "nf_k40:--k 40" > gpurun_out/s3j/ab.txt 2>&1 || exit 1

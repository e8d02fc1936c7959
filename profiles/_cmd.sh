mkdir -p gpurun_out/s2l
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/s2l/tests.log 2>&1 ; echo "tests rc=$?" >> gpurun_out/s2l/tests.log
tail -3 gpurun_out/s2l/tests.log
grep -q "rc=0" gpurun_out/s2l/tests.log || exit 1
timeout -k 10 200 python profiles/topn_bench.py 480189 8 128 > gpurun_out/s2l/topn_k128.txt 2>&1 || exit 1
timeout -k 10 200 python profiles/topn_bench.py 480189 8 256 > gpurun_out/s2l/topn_k256.txt 2>&1 || exit 1
timeout -k 10 200 python profiles/topn_bench.py 480189 8 64 > gpurun_out/s2l/topn_k64.txt 2>&1 || exit 1
bash profiles/ab.sh gpurun_out/s2l "nf:" "nf_frozen:--implicit frozen" "ml25m:--workload ml25m" "ml1m:--workload ml1m" "nf64:--k 64" "nf256:--k 256" "yahoo:--workload yahoo --steps 4" "ml100k:--workload ml100k" > gpurun_out/s2l/ab.txt 2>&1 || exit 1
for w in 2 4 8; do timeout -k 10 300 python profiles/dsgd_rank_sim.py $w >> gpurun_out/s2l/rank_sim.txt 2>&1 || exit 1; done

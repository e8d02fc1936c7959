mkdir -p gpurun_out/s2n
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -k "topn or top_n" -x -q --timeout 120 --timeout-method thread > gpurun_out/s2n/tests.log 2>&1; echo "rc=$?" >> gpurun_out/s2n/tests.log; tail -n 3 gpurun_out/s2n/tests.log
grep -q "rc=0" gpurun_out/s2n/tests.log || exit 1
for a in "8 128" "16 128" "4 128" "8 256" "8 64" "8 40"; do timeout -k 10 200 python profiles/topn_bench.py 480189 $a 2>&1 | tail -n 1 >> gpurun_out/s2n/tn.txt; done

mkdir -p gpurun_out/s2d
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -k als -x -q --timeout 300 --timeout-method thread > gpurun_out/s2d/als_tests.log 2>&1; echo "rc=$?" >> gpurun_out/s2d/als_tests.log
tail -5 gpurun_out/s2d/als_tests.log
grep -q "rc=0" gpurun_out/s2d/als_tests.log || exit 1
ALS_BENCH_NO_CPU=1 timeout -k 10 300 python profiles/als_bench.py netflix 256 2 > gpurun_out/s2d/als_k256.json 2>&1 && \
ALS_BENCH_NO_CPU=1 timeout -k 10 300 python profiles/als_bench.py netflix 128 2 > gpurun_out/s2d/als_k128.json 2>&1

mkdir -p gpurun_out/s3c
timeout -k 10 400 python bench.py --steps 10 --warmup 2 --with-random-leg --with-tickets-leg > gpurun_out/s3c/bench_full.json 2> gpurun_out/s3c/bench_full.err || exit 1
timeout -k 10 300 python bench.py > gpurun_out/s3c/bench_default.json 2> gpurun_out/s3c/bench_default.err

timeout -k 10 400 python bench.py --steps 10 --warmup 2 --with-random-leg --with-tickets-leg > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err || exit 5

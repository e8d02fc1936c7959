#!/usr/bin/env python3
"""bench.py - rating-updates/s of the blocked-SGD (FPSGD) matrix-factorization hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one epoch = one conflict-free pass of the epoch kernel over every training rating of the
synthetic data set (already bucketed and resident in HBM when the timed region starts).  value =
steps * n_train / wall time, whole job.  `roofline` prices the epoch kernel against HBM with the
ALGORITHMIC bytes per update B(k) = 16k + 28; `cpu_baseline` times the CPU oracle's
threaded FPSGD (scheduler + block kernel, fp64 like the reference) on a bounded sample of the same data.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n_users, n_items, n_ratings, k, (s_u, q_u), (s_i, q_i))   Zipf-Mandelbrot weights (rank+q)^-s
    "ml100k": (943, 1682, 100_000, 32, (0.6, 20.0), (0.8, 20.0)),
    "ml1m": (6040, 3706, 1_000_000, 64, (0.6, 40.0), (0.8, 30.0)),
    "ml25m": (162_541, 59_047, 25_000_000, 128, (0.8, 60.0), (1.0, 60.0)),
    "netflix": (480_189, 17_770, 100_000_000, 128, (0.7, 100.0), (1.0, 80.0)),
    "yahoo": (1_000_990, 624_961, 250_000_000, 256, (1.1, 400.0), (1.2, 300.0)),
}
HBM_PEAK = 8.0e12      # B/s, MI355X_MICROARCH.md
ALPHA, BETA1, BETA2 = 0.006, 0.05, 0.015   # SGDRecommender.py:33-35


def bytes_per_update(k, implicit):
    return 16 * k + 28 + (4 * k if implicit == "frozen" else 0)


def synth(torch, name, seed, scale=1.0):
    """Seeded synthetic ratings on the GPU: distinct (u,i) pairs, Zipf-Mandelbrot user activity and item
    popularity, planted rank-8 model, integer ratings 1..5, rows sorted by user; 90/10 split."""
    U, I, n, k, (su, qu), (si, qi) = WORKLOADS[name]
    n = int(n * scale)
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)

    def sampler(N, s, q):
        w = (torch.arange(N, device="cuda", dtype=torch.float64) + 1.0 + q) ** (-s)
        cdf = torch.cumsum(w / w.sum(), 0)
        ids = torch.randperm(N, device="cuda", generator=g)
        return lambda m: ids[torch.searchsorted(cdf, torch.rand(m, device="cuda", dtype=torch.float64, generator=g)).clamp_(max=N - 1)]

    draw_u, draw_i = sampler(U, su, qu), sampler(I, si, qi)
    keys = torch.empty(0, dtype=torch.int64, device="cuda")
    want = int(n * 1.02)
    while keys.numel() < want:
        m = min(1 << 25, int((want - keys.numel()) * 1.3) + 1024)
        new = draw_u(m) * I + draw_i(m)
        keys = torch.unique(torch.cat([keys, new]))
    keep = torch.rand(keys.numel(), device="cuda", generator=g) < (n / keys.numel())
    keys = keys[keep]                                   # sorted: by user, then item
    u, i = keys // I, keys % I
    del keys, keep
    pu = torch.randn(U, 8, device="cuda", generator=g) * 0.6
    qi_ = torch.randn(I, 8, device="cuda", generator=g) * 0.6
    r = torch.empty(u.numel(), device="cuda")
    for a in range(0, u.numel(), 1 << 24):
        b = min(a + (1 << 24), u.numel())
        r[a:b] = (pu[u[a:b]] * qi_[i[a:b]]).sum(1)
    r = (3.5 + r + 0.5 * torch.randn(u.numel(), device="cuda", generator=g)).round_().clamp_(1, 5)
    s = torch.stack([u.int(), i.int(), r.int()], 1).contiguous()
    del u, i, r
    test_mask = torch.rand(s.shape[0], device="cuda", generator=g) < 0.1
    return s[~test_mask].contiguous(), s[test_mask].contiguous(), U, I, k


def cpu_baseline(torch, m, k, implicit, seconds_hint=20.0):
    """Threaded FPSGD of the CPU oracle (fp64, P = threads + 1 like RayBlockScheduler.py:185) on the
    ratings of a random subset of users (ids are already randomly permuted: users < U_s)."""
    import numpy as np
    from oracle import oracle as orc
    threads = min(16, os.cpu_count() or 1)
    n = int(m._d_samples.shape[0])
    target = min(n, 60_000_000)
    U_s = max(threads + 1, int(m.n_users * target / n))
    sub = m._d_samples[m._d_samples[:, 0] < U_s].cpu().numpy()
    W = threads + 1
    ubp, ibp = U_s // W + 1, m.n_items // W + 1
    blocks, ptr = orc.bucket(sub, ubp, ibp, -(-U_s // ubp), -(-m.n_items // ibp))
    if -(-U_s // ubp) != W or -(-m.n_items // ibp) != W:
        return None
    rng = np.random.default_rng(0)
    P = rng.normal(scale=1.0 / k, size=(U_s, k))
    Q = rng.normal(scale=1.0 / k, size=(m.n_items, k))
    bu, bi = np.zeros(U_s), np.zeros(m.n_items)
    t0 = time.perf_counter()
    iters = 10
    orc.fpsgd(P, Q, bu, bi, float(m.b), None, blocks, ptr, W, None, None, ALPHA, BETA1, BETA2, 0, iters, threads,
              deterministic=False, seed=1)
    dt = time.perf_counter() - t0
    return {"value": iters * len(sub) / dt, "unit": "updates/s", "cores": threads, "kind": "port",
            "sample": f"{iters} epochs over the {len(sub)} ratings of {U_s} users (all items), fp64, "
                      f"{W}x{W} grid, oracle/mf_oracle.c threaded FPSGD, {dt:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="netflix", choices=sorted(WORKLOADS))
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the rating count (debug)")
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--implicit", default="none", choices=["none", "frozen"])
    ap.add_argument("--grid", default="", help="Gr,Gc of the block grid (default: chosen from the shape)")
    ap.add_argument("--wgs", type=int, default=0)
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--renumber", default="balanced", choices=["balanced", "random"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=20240611)
    ap.add_argument("--hot-batch", type=int, default=8,
                    help="apply the updates of each stripe's most rated item in groups of this size "
                         "(mfsgd_grid_set_hot_batch; 1 = the reference's sequential order inside a block)")
    ap.add_argument("--swap-roles", action="store_true",
                    help="experiment: feed the data with users and items swapped (which side is LDS-stationary)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import recommender_amd as ra
    from recommender_amd import _abi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("MFSGD_DIST_BACKEND", "nccl")   # "gloo": rehearsal of the N>1 leg on a one-GPU box
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    force_dsgd = bool(os.environ.get("MFSGD_FORCE_DSGD"))   # rehearsal: run the multi-GPU leg (RCCL plumbing) with one rank
    if world > 1 or force_dsgd:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    _abi.require_gpu()
    train, test, U, I, k = synth(torch, args.workload, args.seed + sorted(WORKLOADS).index(args.workload), args.scale)
    if args.swap_roles:
        train = train[:, [1, 0, 2]].contiguous(); test = test[:, [1, 0, 2]].contiguous(); U, I = I, U
    if args.k:
        k = args.k
    if world > 1 or force_dsgd:
        from recommender_amd import dsgd
        return dsgd.bench_main(args, rank, world, (train, test, U, I, k), bytes_per_update, HBM_PEAK,
                               (ALPHA, BETA1, BETA2))
    m = ra.ExplicitMF(n_threads=15, n_factors=k, alpha=ALPHA, beta1=BETA1, beta2=BETA2, implicit=args.implicit,
                      grid=tuple(int(x) for x in args.grid.split(",")) if args.grid else None,
                      n_workgroups=args.wgs, waves_per_wg=args.waves, hot_batch=args.hot_batch, verbose=False)
    m.load_samples_from_device(train, test, U, I)
    import numpy as np
    np.random.seed(args.seed % (2 ** 31))
    t0 = time.perf_counter()
    if args.renumber == "balanced":
        m.balanced_renumber_samples()            # D2 with a load-balancing permutation
    else:
        m.random_renumber_samples()              # D2 exactly as the reference (random shuffle)
    grid = m.build_grid()                        # D3 + block-local runs + tickets on the device
    torch.cuda.synchronize()
    prep_s = time.perf_counter() - t0
    m.init_params_device(args.seed)
    m._to_dev()
    n_train = int(m._d_samples.shape[0])
    rmse0 = m.device_errors()                    # before any training (the warm-up epochs train too)
    for _ in range(args.warmup):
        m.run_epochs(1)
    grid.status()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        m.run_epochs(1)
        b.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    grid.status()
    kern_ms = [a.elapsed_time(b) for a, b in evs]
    avg_kern_s = sum(kern_ms) / len(kern_ms) / 1e3
    rmse1 = m.device_errors()
    value = args.steps * n_train / wall
    traffic = None
    try:        # measured per-launch HBM bytes of this exact configuration, if a PMC profile is on file
        key = f"{args.workload}/k{k}/{args.implicit}/{grid.info.Gr}x{grid.info.Gc}"
        traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key, {}).get("traffic_bytes")
    except Exception:
        pass
    B = bytes_per_update(k, args.implicit)
    achieved = B * n_train / avg_kern_s
    out = {
        "metric": "rating-updates/sec", "value": value, "unit": "updates/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}-shape {U}x{I}, {n_train} train ratings, k={k}, implicit={args.implicit}, "
                               f"grid {grid.info.Gr}x{grid.info.Gc}, {args.renumber} renumbering, free-block schedule"
                               + (f", hot-item groups of {args.hot_batch}" if args.hot_batch > 1 else ""),
                   "hot_batch": args.hot_batch,
                   "n_train": n_train, "k": k, "grid": [grid.info.Gr, grid.info.Gc], "prep_seconds": prep_s,
                   "max_col_nnz": int(grid.info.max_col_nnz), "max_block": int(grid.info.max_block)},
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK, "traffic": traffic, "bytes_per_update": B,
                     "kernel": "k_epoch_free", "kernel_ms": kern_ms},
        "rmse": {"train_before": rmse0[0], "test_before": rmse0[1], "train_after": rmse1[0], "test_after": rmse1[1]},
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(torch, m, k, args.implicit)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

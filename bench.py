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

from workloads import ALPHA, BASE_SEED, BETA1, BETA2, WORKLOADS, synth, workload_seed  # noqa: E402

HBM_PEAK = 8.0e12      # B/s, MI355X_MICROARCH.md


def bytes_per_update(k, implicit):
    return 16 * k + 28 + (4 * k if implicit == "frozen" else 0)


def kernel_source_hash():
    """Identifies the build a PMC record in profiles/traffic.json belongs to: sha256 over the sources of the epoch kernel
    (csrc/epoch.hip, csrc/common.h - the GPU box has no .git).  profiles/collect_traffic.py stores the same hash with
    every record it writes."""
    import hashlib
    h = hashlib.sha256()
    for name in ("epoch.hip", "common.h"):
        path = os.path.join(ROOT, "recommender_amd", "csrc", name)
        h.update(name.encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(torch, m, k, implicit):
    """Threaded FPSGD of the CPU oracle (scheduler + block kernel, P = threads + 1 like RayBlockScheduler.py:185) on
    the ratings of a random subset of users (ids are already permuted: users < U_s), in float64 (the reference's
    arithmetic) and float32 (the GPU's).  Threads = this job's CPU share: OMP_NUM_THREADS when the box sets it (16 per
    GPU on the pool; os.cpu_count() shows the whole machine there), else every core."""
    import numpy as np
    from oracle import oracle as orc
    visible = os.cpu_count() or 1
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = max(1, min(visible, share) if share > 0 else visible)
    n = int(m._d_samples.shape[0])
    target = min(n, 40_000_000)
    U_s = max(threads + 1, int(m.n_users * target / n))
    sub = m._d_samples[m._d_samples[:, 0] < U_s].cpu().numpy()
    W = threads + 1
    ubp, ibp = U_s // W + 1, m.n_items // W + 1
    if -(-U_s // ubp) != W or -(-m.n_items // ibp) != W:
        return None
    blocks, ptr = orc.bucket(sub, ubp, ibp, W, W)
    legs = {}
    iters = 6
    for dt in (np.float64, np.float32):
        rng = np.random.default_rng(0)
        P = rng.normal(scale=1.0 / k, size=(U_s, k)).astype(dt)
        Q = rng.normal(scale=1.0 / k, size=(m.n_items, k)).astype(dt)
        bu, bi = np.zeros(U_s, dt), np.zeros(m.n_items, dt)
        t0 = time.perf_counter()
        orc.fpsgd(P, Q, bu, bi, float(m.b), None, blocks, ptr, W, None, None, ALPHA, BETA1, BETA2, 0, iters, threads,
                  deterministic=False, seed=1)
        dt_s = time.perf_counter() - t0
        legs["f64" if dt is np.float64 else "f32"] = {"value": iters * len(sub) / dt_s, "seconds": dt_s}
    return {"value": legs["f64"]["value"], "unit": "updates/s", "cores": threads, "kind": "port",
            "value_f32": legs["f32"]["value"], "cpu_model": cpu_model(), "host_cores_visible": visible,
            "sample": f"{iters} epochs over the {len(sub)} ratings of {U_s} users (all items), {W}x{W} grid "
                      f"(P = threads + 1), {threads} threads, oracle/mf_oracle.c threaded FPSGD: fp64 "
                      f"{legs['f64']['seconds']:.2f} s (value), fp32 {legs['f32']['seconds']:.2f} s (value_f32)"}


def run_leg(args, torch, ra, data, k, hot_buffers, renumber=None):
    """One measured configuration: prep (renumber + grid), warm-up epochs, `steps` timed epochs."""
    import numpy as np
    train, test, U, I = data
    m = ra.ExplicitMF(n_threads=15, n_factors=k, alpha=ALPHA, beta1=BETA1, beta2=BETA2, implicit=args.implicit,
                      grid=tuple(int(x) for x in args.grid.split(",")) if args.grid else None,
                      n_workgroups=args.wgs, waves_per_wg=args.waves, hot_buffers=hot_buffers, hot_mode=args.hot_mode,
                      **({} if args.hot_items < 0 else {"hot_items": args.hot_items}),
                      **({} if not args.user_order else {"user_order": args.user_order}),
                      **({} if args.hot_cost < 0 else {"hot_cost": args.hot_cost}),
                      **({} if args.hot_min < 0 else {"hot_min": args.hot_min}), verbose=False)
    m.load_samples_from_device(train.clone(), test.clone(), U, I)      # renumbering rewrites the ids in place
    np.random.seed(args.seed % (2 ** 31))
    t0 = time.perf_counter()
    if (renumber or args.renumber) == "balanced":
        m.balanced_renumber_samples()            # D2 with a load-balancing permutation
    else:
        m.random_renumber_samples()              # D2 exactly as the reference (random shuffle)
    grid = m.build_grid()                        # D3 + block-local runs + tickets on the device
    torch.cuda.synchronize()
    prep_s = time.perf_counter() - t0
    m.init_params_device(args.seed)
    m._to_dev()
    rmse0 = m.device_errors()                    # before any training (the warm-up epochs train too)
    for _ in range(args.warmup):
        m.run_epochs(1)
    grid.status()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in evs:                             # the epoch kernel runs on torch's current stream, so do the events
        a.record()
        m.run_epochs(1)
        b.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    grid.status()
    kern_ms = [a.elapsed_time(b) for a, b in evs]
    return {"m": m, "grid": grid, "wall": wall, "kern_ms": kern_ms, "prep_s": prep_s, "rmse0": rmse0,
            "rmse1": m.device_errors(), "n_train": int(m._d_samples.shape[0]), "hot_buffers": hot_buffers}


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
    ap.add_argument("--with-random-leg", action="store_true",
                    help="also time the reference's own D2 (random shuffle of the ids, SGDRecommender.py:228-232) instead of "
                         "the load-balancing permutation: same kernel, stripes of unequal weight")
    ap.add_argument("--with-tickets-leg", action="store_true",
                    help="also time the same grid with the hot segments on the ticket path (hot_buffers 0)")
    ap.add_argument("--seed", type=int, default=BASE_SEED)
    ap.add_argument("--hot-cost", type=float, default=-1.0,
                    help="cost of a pipeline rating against a ticket-path one when the items are dealt to stripes")
    ap.add_argument("--hot-items", type=int, default=-1,
                    help="hot items (hot segments per block) a stripe may have where the pipeline runs (default: the library's)")
    ap.add_argument("--hot-min", type=int, default=-1,
                    help="ratings per block an item needs to become a hot item (default: the model's, 64)")
    ap.add_argument("--user-order", default="", choices=["", "heavy", "light", "shuffle", "spread"],
                    help="order of a user stripe's ids under the balanced renumbering (= order of a block's runs)")
    ap.add_argument("--hot-mode", type=int, default=2,
                    help="hot segments: 2 = hot-item walker (default), 1 = matrix-core pipeline, 0 = item tickets")
    ap.add_argument("--hot-buffers", type=int, default=8,
                    help="LDS row buffers of the hot-item pipeline (mfsgd_grid_set_hot_buffers); 0 = every rating "
                         "through the item tickets.  Either way a block is walked sequentially in stored order")
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
    train, test, U, I, k = synth(torch, args.workload, workload_seed(args.workload, args.seed), args.scale)
    if args.k:
        k = args.k
    if world > 1 or force_dsgd:
        from recommender_amd import dsgd
        return dsgd.bench_main(args, rank, world, (train, test, U, I, k), bytes_per_update, HBM_PEAK,
                               (ALPHA, BETA1, BETA2))
    B = bytes_per_update(k, args.implicit)
    try:        # measured per-launch HBM bytes of a configuration, if a PMC profile of it is on file
        traffic_db = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        traffic_db = {}

    src_hash = kernel_source_hash()

    def summarize(leg):
        g = leg["grid"].info
        avg_s = sum(leg["kern_ms"]) / len(leg["kern_ms"]) / 1e3
        key = f"{args.workload}/k{k}/{args.implicit}/{g.Gr}x{g.Gc}/hot{leg['hot_buffers']}"
        rec = traffic_db.get(key, {})
        # PMC bytes are only quoted for the build they were measured on (profiles/collect_traffic.py stores the hash)
        traffic = rec.get("traffic_bytes") if rec.get("source_hash") == src_hash else None
        achieved = B * leg["n_train"] / avg_s
        return {"value": args.steps * leg["n_train"] / leg["wall"], "ms_per_step": leg["wall"] / args.steps * 1e3,
                "grid": [g.Gr, g.Gc], "hot_buffers": leg["hot_buffers"], "avg_kernel_s": avg_s,
                "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK, "traffic": traffic, "bytes_per_update": B,
                             # `achieved` / `frac` price the ALGORITHMIC bytes (16k + 28 per update) - the
                             # contract's definition; the kernel moves about a third of them (the Q stripe lives in LDS, a
                             # run reads and writes its P row once), so the fraction of the HBM peak it really uses is
                             # compulsory_frac = PMC bytes / launch time / peak (null without a PMC record of THIS build)
                             "achieved_kind": "algorithmic bytes per launch / launch time (not measured traffic)",
                             "compulsory_frac": (traffic / avg_s / HBM_PEAK) if traffic else None,
                             "traffic_source": (f"profiles/traffic.json[{key}] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                                f"source_hash {src_hash})") if traffic else None,
                             "source_hash": src_hash,
                             "limiter": "latency of one rating on one wave + LDS ticket hand-offs (DESIGN.md section 4), not HBM bandwidth",
                             "hot_segment_ratings": int(g.n_hot),
                             "kernel": "k_epoch_free", "kernel_ms": leg["kern_ms"]},
                "rmse": {"train_before": leg["rmse0"][0], "test_before": leg["rmse0"][1],
                         "train_after": leg["rmse1"][0], "test_after": leg["rmse1"][1]}}

    leg = run_leg(args, torch, ra, (train, test, U, I), k, args.hot_buffers)
    head = summarize(leg)
    g = leg["grid"].info
    out = {
        "metric": "rating-updates/sec", "value": head["value"], "unit": "updates/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}-shape {U}x{I}, {leg['n_train']} train ratings, k={k}, "
                               f"implicit={args.implicit}, grid {g.Gr}x{g.Gc}, {args.renumber} renumbering, "
                               f"free-block schedule, sequential inside a block"
                               + (" (hot segments on the hot-item pipeline)" if args.hot_buffers else ""),
                   "hot_buffers": args.hot_buffers, "n_train": leg["n_train"], "k": k, "grid": [g.Gr, g.Gc],
                   "prep_seconds": leg["prep_s"], "max_col_nnz": int(g.max_col_nnz), "max_block": int(g.max_block)},
        "roofline": head["roofline"], "rmse": head["rmse"],
    }
    cpu_m = leg["m"]
    if args.with_random_leg and args.renumber == "balanced":
        leg3 = run_leg(args, torch, ra, (train, test, U, I), k, args.hot_buffers, renumber="random")
        s3 = summarize(leg3)
        out["random_renumbering"] = {"value": s3["value"], "unit": "updates/s", "ms_per_step": s3["ms_per_step"],
                                     "grid": s3["grid"], "frac": s3["roofline"]["frac"], "kernel_ms": s3["roofline"]["kernel_ms"],
                                     "rmse": s3["rmse"],
                                     "note": "the reference's D2 (SGDRecommender.py:228-232) instead of the load-balancing permutation"}
        del leg3
    if args.with_tickets_leg and args.hot_buffers:
        leg2 = run_leg(args, torch, ra, (train, test, U, I), k, 0)
        s2 = summarize(leg2)
        out["tickets_only"] = {"value": s2["value"], "unit": "updates/s", "ms_per_step": s2["ms_per_step"],
                               "grid": s2["grid"], "frac": s2["roofline"]["frac"], "kernel_ms": s2["roofline"]["kernel_ms"],
                               "rmse": s2["rmse"]}
        del leg2
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(torch, cpu_m, k, args.implicit)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

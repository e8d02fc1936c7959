"""Sensitivity of the held-out RMSE after epochs 1..5 to what the SCHEDULE leaves open (development tool).

    python profiles/rmse_knobs.py [workload] [what ...]      what: gpu, s1

Every run starts from the same model: the initial factors are drawn once in the ORIGINAL id space and carried into
each run's own numbering.  GPU legs: the free-block schedule under different grids / seeds / arithmetic.  S1 legs: the
CPU oracle's threaded FPSGD under the reference's own free parameters (threads, epoch barrier, float32).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import recommender_amd as ra  # noqa: E402
import workloads  # noqa: E402
from _envelope import s1_envelope  # noqa: E402
from oracle import oracle as orc  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ml25m"
what = sys.argv[2:] or ["gpu", "s1"]
epochs = 5
HP = (workloads.ALPHA, workloads.BETA1, workloads.BETA2)
U, I, _, K = workloads.WORKLOADS[name][:4]
gen = torch.Generator(device="cuda")
gen.manual_seed(1)
P_orig = torch.randn(U, K, device="cuda", generator=gen) / K          # original id space
Q_orig = torch.randn(I, K, device="cuda", generator=gen) / K


def model(renumber="balanced", **kw):
    train, test, U_, I_, k = workloads.synth(torch, name)
    m = ra.ExplicitMF(n_threads=15, n_factors=k, alpha=HP[0], beta1=HP[1], beta2=HP[2], implicit="none",
                      verbose=False, **kw)
    m.load_samples_from_device(train, test, U_, I_)
    np.random.seed(20240611)
    m.random_renumber_samples()
    if renumber == "balanced":
        m.balanced_renumber_samples()
    g = m.build_grid()
    m.init_params_device(1)
    rp = torch.from_numpy(m.row_perm.astype(np.int64)).cuda()
    cp = torch.from_numpy(m.col_perm.astype(np.int64)).cuda()
    dt = torch.float64 if m.dtype == "float64" else torch.float32
    m.P = torch.empty(U_, k, device="cuda", dtype=dt)
    m.P[rp] = P_orig.to(dt)
    m.Q = torch.empty(I_, k, device="cuda", dtype=dt)
    m.Q[cp] = Q_orig.to(dt)
    m._to_dev()
    return m, g


def gpu_leg(tag, reseed=False, one_launch=False, **kw):
    m, g = model(**kw)
    traj = [m.device_errors()[1]]
    t0 = time.perf_counter()
    start = {n: m._dev[n].clone() for n in ("P", "Q", "b_u", "b_i")}
    for e in range(epochs):
        if reseed:
            m.seed = 1000 * (e + 1)
        if one_launch:          # a run of e + 1 passes from the start, like the S1 envelope's legs
            for n in start:
                m._dev[n].copy_(start[n])
            m.run_epochs(e + 1)
        else:
            m.run_epochs(1)
        traj.append(m.device_errors()[1])
    torch.cuda.synchronize()
    print(f"{tag:34s} {g.info.Gr}x{g.info.Gc} hot {int(g.info.n_hot):9d}: " + " ".join(f"{x:.5f}" for x in traj) +
          f"   ({(time.perf_counter() - t0) * 1e3 / epochs:.1f} ms/epoch incl. RMSE)", flush=True)
    del m, g
    torch.cuda.empty_cache()


if "s2" in what:
    gpu_leg("balanced, launch per pass", pass_barrier=True)
    gpu_leg("balanced, schedule 2, run of e passes", one_launch=True)
    gpu_leg("balanced, schedule 2, again", one_launch=True)
    gpu_leg("balanced, schedule 2, seed 7", one_launch=True, seed=7)
    gpu_leg("random, schedule 2, run of e passes", one_launch=True, renumber="random")
    gpu_leg("random, launch per pass", pass_barrier=True, renumber="random")
    # timing: 5 passes in one launch against 5 launches
    for pb in (True, False):
        m, g = model(pass_barrier=pb)
        m.run_epochs(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.run_epochs(5)
        torch.cuda.synchronize()
        print(f"5 passes, pass_barrier={pb}: {(time.perf_counter() - t0) * 200:.2f} ms per pass", flush=True)
        g.status()
        del m, g

if "s2w" in what:
    Gc0 = 256 if name == "ml25m" else 254
    for Gr_ in (288, 320):
        for w in (2, 4, 8, 16, 64):
            gpu_leg(f"balanced, schedule 2, {Gr_} rows, pick {w}", one_launch=True, grid=(Gr_, Gc0), pick_width=w)

if "p3" in what:
    for w in (3, 3, 3, 5):
        gpu_leg(f"balanced, schedule 2, pick {w}", one_launch=True, pick_width=w)

if "s1x" in what:        # a wider sample of the reference's own spread
    m, g = model(renumber="random")
    train, test = m._d_samples.cpu().numpy(), m._d_test.cpu().numpy()
    init = [m._dev[n].cpu().numpy() for n in ("P", "Q", "b_u", "b_i")]
    b = m.b
    del m, g
    torch.cuda.empty_cache()
    env = s1_envelope(orc, train, test, U, I, init, b, HP, n_threads=16, epochs=epochs, seeds=tuple(range(1, 9)))
    for seed, traj in env.items():
        print(f"S1 16 threads 17x17 seed {seed}: " + " ".join(f"{x[1]:.5f}" for x in traj), flush=True)

if "s2g" in what:
    Gc0 = 256 if name == "ml25m" else 254
    for Gr_ in (304, 320, 352, 384):
        gpu_leg(f"balanced, schedule 2, {Gr_} rows", one_launch=True, grid=(Gr_, Gc0))
        for pb in (True, False):
            m, g = model(pass_barrier=pb, grid=(Gr_, Gc0))
            m.run_epochs(2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.run_epochs(5)
            torch.cuda.synchronize()
            print(f"    {Gr_} rows, 5 passes, pass_barrier={pb}: {(time.perf_counter() - t0) * 200:.2f} ms per pass", flush=True)
            g.status()
            del m, g

if "gpu" in what:
    gpu_leg("balanced (default)")
    gpu_leg("balanced, again")
    gpu_leg("balanced, new seed every epoch", reseed=True)
    gpu_leg("balanced, seed 7", seed=7)
    gpu_leg("random renumbering", renumber="random")
    gpu_leg("random, tickets only", renumber="random", hot_buffers=0)
    gpu_leg("balanced, float64", dtype="float64")
    for grid in ((256, 256), (272, 256), (320, 256), (384, 256), (576, 256), (144, 256), (432, 256)):
        gpu_leg(f"balanced, grid {grid}", grid=grid)
    gpu_leg("balanced, 128 workgroups", n_workgroups=128)
    gpu_leg("balanced, 192 workgroups", n_workgroups=192)
    gpu_leg("balanced, 8 waves", waves_per_wg=8)

if "s1" in what:
    m, g = model(renumber="random")
    train, test = m._d_samples.cpu().numpy(), m._d_test.cpu().numpy()
    init = [m._dev[n].cpu().numpy() for n in ("P", "Q", "b_u", "b_i")]
    b = m.b
    del m, g
    torch.cuda.empty_cache()

    def s1_leg(tag, threads, dtype=np.float64, cumulative=False, seeds=(1, 2)):
        t0 = time.perf_counter()
        ini = [a.astype(dtype) for a in init]
        if dtype is np.float32:
            # s1_envelope converts to float64: run the float32 leg by hand
            W = threads + 1
            ubp, ibp = int(U / W) + 1, int(I / W) + 1
            blocks, ptr = orc.bucket(train, ubp, ibp, W, W)
            for seed in seeds:
                traj = []
                for e in range(1, epochs + 1):
                    cur = [a.copy() for a in ini]
                    orc.fpsgd(*cur, b, None, blocks, ptr, W, None, None, *HP, 0, e, threads, deterministic=False, seed=seed)
                    traj.append(orc.rmse(test, None, None, *cur, b, None, implicit=0))
                print(f"S1 {tag:31s} seed {seed}: " + " " * 8 + " ".join(f"{x:.5f}" for x in traj), flush=True)
            return
        env = s1_envelope(orc, train, test, U, I, ini, b, HP, n_threads=threads, epochs=epochs, seeds=seeds,
                          cumulative=cumulative)
        for seed, traj in env.items():
            print(f"S1 {tag:31s} seed {seed}: " + " ".join(f"{x[1]:.5f}" for x in traj) +
                  f"   ({time.perf_counter() - t0:.0f} s so far)", flush=True)

    s1_leg("16 threads 17x17 (the envelope)", 16, seeds=(1, 2, 3))
    s1_leg("16 threads, epoch barrier", 16, cumulative=True)
    s1_leg("8 threads 9x9", 8)
    s1_leg("32 threads 33x33", 32)
    s1_leg("16 threads, float32", 16, dtype=np.float32)

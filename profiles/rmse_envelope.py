"""Held-out RMSE after epochs 1..5: the GPU block schedules against the ENVELOPE of the reference's own schedule.

    python profiles/rmse_envelope.py [workload] [epochs] [n_seeds]

The reference's run length is iters = 5 (RayBlockScheduler.py:185) and its block order is random (S1, :50-58) and
timing dependent, so its held-out RMSE after e passes is a range.  tests/_envelope.py measures that range with the CPU
oracle (threaded FPSGD, S1 random tie-break, 16 threads, 17 x 17 grid, float64, one run of e passes per seed and e),
and this script prints the GPU schedules next to it - same ratings, same initial model (the factor rows are carried
into each run's own numbering):
  fixed16     deterministic=True: the reference's grid, row-major block order (schedule 0)
  gpu1        balanced renumbering (the product's default prep), throughput grid, a launch per pass (schedule 1)
  gpu2        the same, all passes of a run in one launch, overlapped as the reference's scheduler overlaps them
              (schedule 2, least visited of 3): a run of e passes from the start for every e, like the envelope's legs
  gpu2 s=7    the same with another scheduler seed
  gpu1 rand   schedule 1 on the reference's own random renumbering (D2): stripes of unequal weight
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
from _envelope import band_distance, envelope_bounds, s1_envelope  # noqa: E402
from oracle import oracle as orc  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ml25m"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n_seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
HP = (workloads.ALPHA, workloads.BETA1, workloads.BETA2)
threads = 16


def model(renumber="balanced", **kw):
    train, test, U, I, k = workloads.synth(torch, name)
    m = ra.ExplicitMF(n_threads=15, n_factors=k, alpha=HP[0], beta1=HP[1], beta2=HP[2], implicit="none", verbose=False, **kw)
    m.load_samples_from_device(train, test, U, I)
    np.random.seed(20240611)
    m.random_renumber_samples()
    if renumber == "balanced":
        m.balanced_renumber_samples()
    g = m.build_grid()
    m.init_params_device(1)
    m._to_dev()
    return m, g


m, g = model("random", deterministic=True)
host = {"train": m._d_samples.cpu().numpy(), "test": m._d_test.cpu().numpy(), "U": m.n_users, "I": m.n_items, "b": m.b,
        "init": [m._dev[n].cpu().numpy().copy() for n in ("P", "Q", "b_u", "b_i")]}
rp0 = torch.from_numpy(m.row_perm.astype(np.int64)).cuda()
cp0 = torch.from_numpy(m.col_perm.astype(np.int64)).cuda()
P0, Q0 = m._dev["P"].clone(), m._dev["Q"].clone()
out = {"fixed16": [m.device_errors()[1]]}
for _ in range(epochs):
    m.run_epochs(1)
    out["fixed16"].append(m.device_errors()[1])
del m, g


def reset(m):
    rp = torch.from_numpy(m.row_perm.astype(np.int64)).cuda()
    cp = torch.from_numpy(m.col_perm.astype(np.int64)).cuda()
    m._dev["P"][rp] = P0[rp0]
    m._dev["Q"][cp] = Q0[cp0]
    m._dev["b_u"].zero_()
    m._dev["b_i"].zero_()


for tag, kw, one_launch in (("gpu1", dict(pass_barrier=True), False), ("gpu2", dict(), True), ("gpu2 s=7", dict(seed=7), True),
                            ("gpu1 rand", dict(pass_barrier=True, renumber="random"), False)):
    m, g = model(**kw)
    reset(m)
    traj = [m.device_errors()[1]]
    t0 = time.perf_counter()
    for e in range(1, epochs + 1):
        if one_launch:
            reset(m)
            m.run_epochs(e)
        else:
            m.run_epochs(1)
        traj.append(m.device_errors()[1])
    out[tag] = traj
    print(f"# {tag}: grid {g.info.Gr}x{g.info.Gc} ({int(g.info.n_hot)} ratings in hot segments), "
          f"{time.perf_counter() - t0:.1f} s", flush=True)
    del m, g
    torch.cuda.empty_cache()

t0 = time.perf_counter()
env = s1_envelope(orc, host["train"], host["test"], host["U"], host["I"], host["init"], host["b"], HP, n_threads=threads,
                  epochs=epochs, seeds=tuple(range(1, n_seeds + 1)), train_rmse=False,
                  log=lambda s: print(s, flush=True))
print(f"# oracle S1 envelope: {n_seeds} seeds x run lengths 1..{epochs}, {threads} threads, {threads + 1}x{threads + 1} "
      f"grid, float64: {time.perf_counter() - t0:.1f} s")
lo, hi = envelope_bounds(env, 1)
print(f"# {name}, k={workloads.WORKLOADS[name][3]}, alpha {HP[0]}: held-out RMSE after e passes")
print("    e   S1 min    S1 max  " + " ".join(f"{t:>9s}" for t in out))
for e in range(epochs + 1):
    print(f"{e:5d} {lo[e]:.5f} - {hi[e]:.5f} " + " ".join(f"{out[t][e]:9.5f}" for t in out))
for t in out:
    d = [band_distance(out[t][e], out[t][e], lo[e], hi[e]) for e in range(epochs + 1)]
    print(f"# {t:9s}: distance from the reference's band per epoch: " + " ".join(f"{x:.5f}" for x in d) + f"  (max {max(d):.5f})")

"""Held-out RMSE trajectories of the block orders on one BASELINE shape (north_star's 0.002 gate).

    python profiles/rmse_gate.py [workload] [epochs]

Same data, same id space (the reference's random shuffle), same initial factors for every run:
  fixed16   deterministic=True: the reference's (n_threads+1)^2 = 16x16 grid, rows in file order inside a block,
            blocks in row-major order (schedule 0)
  fixedfine the throughput grid (user-sorted blocks), blocks in fixed row-major order (schedule 0)
  free      the throughput grid, on-GPU free-block schedule, sequential inside a block
  grouped   the same with the hot segments on the hot-item pipeline (column name kept from round 2a)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import recommender_amd as ra  # noqa: E402
import workloads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ml25m"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
alpha = float(sys.argv[3]) if len(sys.argv) > 3 else workloads.ALPHA


def model(tag):
    train, test, U, I, k = workloads.synth(torch, name)
    kw = dict(n_threads=15, n_factors=k, alpha=alpha, beta1=workloads.BETA1, beta2=workloads.BETA2,
              implicit="none", verbose=False)
    if tag == "fixed16":
        m = ra.ExplicitMF(deterministic=True, **kw)
    else:
        m = ra.ExplicitMF(hot_buffers=4 if tag == "grouped" else 0, **kw)
    m.load_samples_from_device(train, test, U, I)
    np.random.seed(20240611)
    m.random_renumber_samples()
    g = m.build_grid()
    if tag == "fixedfine":
        m.deterministic = True      # run_epochs: schedule 0 on the grid just built
    m.init_params_device(1)
    m._to_dev()
    return m, g


out = {}
for tag in ("fixed16", "fixedfine", "free", "grouped"):
    m, g = model(tag)
    traj = [m.device_errors()]
    t0 = time.perf_counter()
    for e in range(epochs):
        m.run_epochs(1)
        traj.append(m.device_errors())
    dt = time.perf_counter() - t0
    out[tag] = traj
    print(f"# {tag}: grid {g.info.Gr}x{g.info.Gc} order {g.info.order}, {epochs} epochs + RMSE in {dt:.1f} s", flush=True)
    del m, g
    torch.cuda.empty_cache()
print(f"# {name}, alpha {alpha}: held-out RMSE (train RMSE) after each epoch")
print("epoch " + " ".join(f"{t:>17s}" for t in out))
for e in range(epochs + 1):
    print(f"{e:5d} " + " ".join(f"{out[t][e][1]:.5f} ({out[t][e][0]:.5f})" for t in out))
base = out["fixed16"]
for t in out:
    d = [abs(out[t][e][1] - base[e][1]) for e in range(epochs + 1)]
    print(f"# max |held-out({t}) - held-out(fixed16)| over epochs: {max(d):.5f}; at the last epoch {d[-1]:.5f}")

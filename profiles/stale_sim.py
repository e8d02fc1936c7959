"""Bounded-staleness multi-GPU mode, numerics rehearsal on ONE GPU (development tool for recommender_amd/dsgd.py).

    python profiles/stale_sim.py [workload] [world] [epochs]

W ranks, users cut into W super-stripes.  Every rank runs the single-GPU free-block epoch kernel on ITS users' ratings
against a private copy of the item factors for a whole epoch ("local SGD": no ring, no sub-epochs - an item's chain is
cut W-fold); then the copies are merged per item row.  The ranks run one after the other here (same arithmetic as W
GPUs side by side, since they share nothing during the epoch).  Merge rules tried:
  avg     Q = Q0 + mean_g(dQ_g)                (always stable; rows whose deltas are independent learn W x slower)
  adasum  pairwise tree of  a (+) b = (1 - a.b / 2|a|^2) a + (1 - a.b / 2|b|^2) b   per item row [q_i, b_i]:
          the sum when the deltas are orthogonal, the mean when they are equal
Prints held-out RMSE per epoch next to the single-GPU run from the same start, and one rank's kernel time.
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
from recommender_amd import _abi  # noqa: E402
from recommender_amd.SGDRecommender import Grid, lds_bytes  # noqa: E402
from recommender_amd.numba_methods import rmse_exact  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "netflix"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 5
HP = (workloads.ALPHA, workloads.BETA1, workloads.BETA2)

train, test, U, I, k = workloads.synth(torch, name)
m = ra.ExplicitMF(n_threads=15, n_factors=k, alpha=HP[0], beta1=HP[1], beta2=HP[2], implicit="none", verbose=False,
                  pass_barrier=True)
m.load_samples_from_device(train, test, U, I)
np.random.seed(20240611)
m.balanced_renumber_samples()
g1 = m.build_grid()
m.init_params_device(1)
m._to_dev()
P0, Q0 = m._dev["P"].clone(), m._dev["Q"].clone()
b = m.b
single = [m.device_errors()[1]]
for e in range(epochs):
    m.run_epochs(1)
    single.append(m.device_errors()[1])
print(f"# {name} k={k} single GPU {g1.info.Gr}x{g1.info.Gc}: held-out " + " ".join(f"{x:.5f}" for x in single), flush=True)
s_tr, s_te = m._d_samples, m._d_test
Gc = g1.info.Gc
ibp = g1.info.ibp
edges = [U * g // W for g in range(W + 1)]
grids, locs = [], []
for g in range(W):
    u0, u1 = edges[g], edges[g + 1]
    loc = s_tr[(s_tr[:, 0] >= u0) & (s_tr[:, 0] < u1)].clone()
    loc[:, 0] -= u0
    nu = u1 - u0
    Gr = 288
    ubp = -(-nu // Gr)
    hot = (1, 64) if lds_bytes(ibp, -(-nu // ubp), k, 4, 2) <= 160 * 1024 else 0
    grids.append(Grid(loc.contiguous(), nu, I, ubp, ibp, order=1, hot_segments=hot))
    locs.append((u0, u1))
print(f"# {W} ranks: per rank {grids[0].info.Gr}x{grids[0].info.Gc} blocks, {int(grids[0].info.n)} ratings, "
      f"{int(grids[0].info.n_hot)} in hot segments", flush=True)


def adasum(a, b_):
    ab = (a * b_).sum(1, keepdim=True)
    aa = (a * a).sum(1, keepdim=True).clamp_min(1e-30)
    bb = (b_ * b_).sum(1, keepdim=True).clamp_min(1e-30)
    return (1 - ab / (2 * aa)) * a + (1 - ab / (2 * bb)) * b_


for rule in ("adasum", "avg"):
    P, Q = P0.clone(), Q0.clone()
    bu, bi = torch.zeros(U, device="cuda"), torch.zeros(I, device="cuda")
    traj = [single[0]]
    kms = []
    for e in range(epochs):
        deltas = []
        for g in range(W):
            u0, u1 = locs[g]
            Qg, big = Q.clone(), bi.clone()
            Pg, bug = P[u0:u1], bu[u0:u1]          # views: updated in place
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            _abi.check(_abi.lib().mfsgd_hip_epoch_f32(grids[g].handle, k, _abi.ptr(Pg), _abi.ptr(Qg), _abi.ptr(bug),
                                                      _abi.ptr(big), b, None, *HP, 0, 0, 1, 1, 1000 * e + g, 0, 0, None,
                                                      _abi.stream_ptr()))
            ev1.record()
            torch.cuda.synchronize()
            kms.append(ev0.elapsed_time(ev1))
            deltas.append(torch.cat([Qg - Q, (big - bi)[:, None]], 1))
        if rule == "avg":
            d = sum(deltas) / W
        else:
            lvl = deltas
            while len(lvl) > 1:
                lvl = [adasum(lvl[j], lvl[j + 1]) if j + 1 < len(lvl) else lvl[j] for j in range(0, len(lvl), 2)]
            d = lvl[0]
        Q += d[:, :k]
        bi += d[:, k]
        traj.append(rmse_exact(s_te, None, None, P, Q, bu, bi, b, None))
    for gr in grids:
        gr.status()
    print(f"# {rule:7s} W={W}: held-out " + " ".join(f"{x:.5f}" for x in traj) +
          f" | max |diff to single| {max(abs(a - c) for a, c in zip(traj, single)):.5f}"
          f" | one rank's epoch kernel: median {np.median(kms):.2f} ms, max {max(kms):.2f} ms", flush=True)

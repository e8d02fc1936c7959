"""Bounded staleness for the HOTTEST items only, numerics rehearsal on ONE GPU (development tool for recommender_amd/dsgd.py).

    python profiles/stale_hot_sim.py [workload] [world] [epochs] [merges_per_epoch]

VERDICT r2 item 3: exact DSGD keeps an item's updates serial across the ranks (its shard is on one rank at a time), so the
netflix shape's 140 K-rating chains cap the job near the single-GPU time.  Rehearsed here: the stripes' hot items (one per
item stripe, the ones the hot-item walker takes: 21 % of the netflix shape's ratings) leave the ring; every rank trains
them against a PRIVATE copy of their rows on its own users' ratings, and the copies are merged `merges_per_epoch` times
per epoch (slice j of an epoch = the hot ratings of the users whose stripe index is j mod merges); all other items go
round the ring exactly (simulated by one launch over the rest: the super-blocks of a sub-epoch share no user and no item).
The ranks run one after the other here - same arithmetic as W GPUs side by side.  Merge rules: sum, avg, adasum
(profiles/stale_sim.py).  Prints held-out RMSE per epoch next to the single-GPU run from the same start.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import recommender_amd as ra  # noqa: E402
import workloads  # noqa: E402
from recommender_amd import _abi  # noqa: E402
from recommender_amd.SGDRecommender import Grid  # noqa: E402
from recommender_amd.numba_methods import rmse_exact  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "netflix"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 5
M = int(sys.argv[4]) if len(sys.argv) > 4 else W
HP = (workloads.ALPHA, workloads.BETA1, workloads.BETA2)

train, test, U, I, k = workloads.synth(torch, name)
m = ra.ExplicitMF(n_threads=15, n_factors=k, alpha=HP[0], beta1=HP[1], beta2=HP[2], implicit="none", verbose=False, pass_barrier=True)
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
ubp, ibp = g1.info.ubp, g1.info.ibp
hot_items = torch.from_numpy(g1.col_hot()[:, 0].astype(np.int64)).cuda()
hot_items = hot_items[hot_items >= 0]
is_hot_item = torch.zeros(I, dtype=torch.bool, device="cuda")
is_hot_item[hot_items] = True
hot_mask = is_hot_item[s_tr[:, 1].long()]
rest = s_tr[~hot_mask].contiguous()
hot = s_tr[hot_mask].contiguous()
print(f"# hot items {int(hot_items.numel())}: {int(hot.shape[0])} of {int(s_tr.shape[0])} ratings leave the ring", flush=True)
g_rest = Grid(rest, U, I, ubp, ibp, order=1, hot_segments=0)
edges = [U * g // W for g in range(W + 1)]
stripe_of_user = (hot[:, 0] // ubp)
hot_grids = {}
for g in range(W):
    u0, u1 = edges[g], edges[g + 1]
    for j in range(M):
        sel = (hot[:, 0] >= u0) & (hot[:, 0] < u1) & (stripe_of_user % M == j)
        loc = hot[sel].clone()
        loc[:, 0] -= u0
        hot_grids[g, j] = Grid(loc.contiguous(), u1 - u0, I, ubp, ibp, order=1, hot_segments=(1, 1)) if loc.shape[0] else None


def adasum(a, b_):
    ab = (a * b_).sum(1, keepdim=True)
    aa = (a * a).sum(1, keepdim=True).clamp_min(1e-30)
    bb = (b_ * b_).sum(1, keepdim=True).clamp_min(1e-30)
    return (1 - ab / (2 * aa)) * a + (1 - ab / (2 * bb)) * b_


def epoch_call(grid, P, Q, bu, bi, seed):
    _abi.check(_abi.lib().mfsgd_hip_epoch_f32(grid.handle, k, _abi.ptr(P), _abi.ptr(Q), _abi.ptr(bu), _abi.ptr(bi), b, None, *HP,
                                              0, 0, 1, 1, seed, 0, 0, None, _abi.stream_ptr()))


for rule in ("sum", "adasum", "avg"):
    P, Q = P0.clone(), Q0.clone()
    bu, bi = torch.zeros(U, device="cuda"), torch.zeros(I, device="cuda")
    traj = [single[0]]
    for e in range(epochs):
        for j in range(M):
            deltas = []
            for g in range(W):
                if hot_grids[g, j] is None:
                    continue
                u0, u1 = edges[g], edges[g + 1]
                Qg, big = Q.clone(), bi.clone()
                epoch_call(hot_grids[g, j], P[u0:u1], Qg, bu[u0:u1], big, 1000 * e + 10 * j + g)
                deltas.append(torch.cat([(Qg - Q)[hot_items], (big - bi)[hot_items][:, None]], 1))
            if rule == "avg":
                d = sum(deltas) / len(deltas)
            elif rule == "sum":
                d = sum(deltas)
            else:
                lvl = deltas
                while len(lvl) > 1:
                    lvl = [adasum(lvl[t], lvl[t + 1]) if t + 1 < len(lvl) else lvl[t] for t in range(0, len(lvl), 2)]
                d = lvl[0]
            Q[hot_items] += d[:, :k]
            bi[hot_items] += d[:, k]
        # the rest of the ratings: the ring, exact (a launch per epoch over all of them is one of its valid orders)
        epoch_call(g_rest, P, Q, bu, bi, 77 + e)
        traj.append(rmse_exact(s_te, None, None, P, Q, bu, bi, b, None))
    g_rest.status()
    print(f"# {rule:7s} W={W} merges/epoch={M}: held-out " + " ".join(f"{x:.5f}" for x in traj) +
          f" | max |diff to single| {max(abs(a - c) for a, c in zip(traj, single)):.5f}", flush=True)

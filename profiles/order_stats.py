"""How "Gauss-Seidel" is a block schedule?  One number per schedule, from its block order alone.

    python profiles/order_stats.py [workload]

Near the small-initialisation saddle one SGD epoch is a product of elementary updates (1 + alpha A_t), t = ratings in
processing order.  Orders differ from the third order in alpha on: a chain  q_i -> p_u -> q_i' -> p_u'  through the
ratings a = (u,i), b = (u,i'), c = (u',i') contributes within ONE epoch iff a < b < c.  Summed over the middle rating:
    S = E_b[ x_b (1 - y_b) + y_b (1 - x_b) ],   x_b = share of user(b)'s ratings processed before b,
                                                  y_b = share of item(b)'s ratings processed before b
S = 1/2 for a row-major (or column-major) sequential pass (x, y independent), S = 1/3 for any order in which users and
items progress in step with the clock (x = y = t): S = 1/2 - 2 Cov(x, y).  The script prints S (plain and weighted by
d_u d_i, the number of chains through b) for: the reference's S1 schedule simulated event by event (16 workers, 17 x 17,
block time = its rating count), the fixed row-major order, and the block orders the GPU kernel logged.
"""
import heapq
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import recommender_amd as ra  # noqa: E402
import workloads  # noqa: E402
from recommender_amd.RayBlockScheduler import BlockScheduler  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ml25m"
HP = (workloads.ALPHA, workloads.BETA1, workloads.BETA2)


def stat(u, i, t):
    """u, i: int64 device tensors; t: float64 processing time per rating."""
    n = u.numel()
    tr = torch.empty(n, dtype=torch.int64, device="cuda")
    tr[torch.argsort(t)] = torch.arange(n, device="cuda")

    def share(ids):
        N = int(ids.max()) + 1
        deg = torch.bincount(ids, minlength=N)
        start = torch.cumsum(deg, 0) - deg
        order = torch.argsort(ids * n + tr)
        pos = torch.empty(n, dtype=torch.int64, device="cuda")
        pos[order] = torch.arange(n, device="cuda") - start[ids[order]]
        return pos.double() / deg[ids].double(), deg[ids].double()

    x, du = share(u)
    y, di = share(i)
    s = x * (1 - y) + y * (1 - x)
    w = du * di
    return float(s.mean()), float((s * w).sum() / w.sum())


def block_times_s1(nnz, W, workers, seed):
    """Event-driven S1 run (one epoch): start time of every block, block time = its rating count."""
    random.seed(seed)
    sch = BlockScheduler(W, 1)
    start = np.zeros((W, W))
    heap, now = [], 0.0
    for w in range(workers):
        b = sch.get_next()
        if b is not None:
            start[b] = now
            heapq.heappush(heap, (now + nnz[b], w, b))
    while heap:
        now, w, done = heapq.heappop(heap)
        b = sch.get_next(done)
        if b is not None:
            start[b] = now
            heapq.heappush(heap, (now + nnz[b], w, b))
    return start


def model(tag):
    train, test, U, I, k = workloads.synth(torch, name)
    kw = dict(n_threads=15, n_factors=k, alpha=HP[0], beta1=HP[1], beta2=HP[2], implicit="none", verbose=False)
    if tag == "pipeline379":
        m = ra.ExplicitMF(grid=(288, 384), hot_items=3, **kw)
    else:
        m = ra.ExplicitMF(hot_buffers=0 if tag == "free" else None, **kw)
    m.load_samples_from_device(train, test, U, I)
    np.random.seed(20240611)
    m.random_renumber_samples()
    if tag == "free_bal":
        m.balanced_renumber_samples()
    return m


m = model("free")
s = m._d_samples
u, i = s[:, 0].long(), s[:, 1].long()
n = u.numel()
W = 17
ubp, ibp = int(m.n_users / W) + 1, int(m.n_items / W) + 1
blk = (u // ubp) * W + i // ibp
nnz = torch.bincount(blk, minlength=W * W).cpu().numpy().reshape(W, W).astype(np.float64)
# position of a rating inside its coarse block in file order
order = torch.argsort(blk, stable=True)
bstart = torch.cumsum(torch.bincount(blk, minlength=W * W), 0) - torch.bincount(blk, minlength=W * W)
pos = torch.empty(n, dtype=torch.int64, device="cuda")
pos[order] = torch.arange(n, device="cuda") - bstart[blk[order]]
print(f"# {name}: {n} ratings; S = E[x(1-y) + y(1-x)] (plain / weighted by d_u d_i)")
t_fixed = torch.from_numpy(np.cumsum(nnz.reshape(-1)) - nnz.reshape(-1)).cuda()[blk] + pos.double()
print("fixed16 row-major          S = %.4f / %.4f" % stat(u, i, t_fixed))
for seed in (1, 2, 3):
    st = torch.from_numpy(block_times_s1({(r, c): nnz[r, c] for r in range(W) for c in range(W)}, W, 16, seed).reshape(-1)).cuda()
    print(f"S1 simulated, seed {seed}        S = %.4f / %.4f" % stat(u, i, st[blk] + pos.double()))
# the same S1 with every rating of a block stamped with the block's start: what the coarse order alone contributes
st = torch.from_numpy(block_times_s1({(r, c): nnz[r, c] for r in range(W) for c in range(W)}, W, 16, 1).reshape(-1)).cuda()
print("S1 seed 1, blocks as points   S = %.4f / %.4f" % stat(u, i, st[blk] + 1e-9 * torch.rand(n, device="cuda", dtype=torch.float64)))
del m
for tag in ("free", "free_bal", "pipeline379"):
    m = model(tag)
    g = m.build_grid()
    m.init_params_device(1)
    m._to_dev()
    s = m._d_samples
    u, i = s[:, 0].long(), s[:, 1].long()
    Gr, Gc = g.info.Gr, g.info.Gc
    blk = (u // g.info.ubp) * Gc + i // g.info.ibp
    for e in range(3):
        log = torch.full((Gr * Gc, 2), -1, dtype=torch.int32, device="cuda")
        m.run_epochs(1, order_log=log)
        g.status()
        lg = log.long()
        ok = lg[:, 0] >= 0
        slot = torch.full((Gr * Gc,), -1, dtype=torch.int64, device="cuda")
        slot[(lg[ok, 0] * Gc + lg[ok, 1])] = torch.arange(Gr * Gc, device="cuda")[ok]
        t = slot[blk].double() + torch.rand(u.numel(), device="cuda", dtype=torch.float64)
        tr, te = m.device_errors()
        print(f"{tag:12s} {Gr}x{Gc} epoch {e + 1} S = %.4f / %.4f   held-out RMSE after it {te:.5f}" % stat(u, i, t))
    del m, g
    torch.cuda.empty_cache()

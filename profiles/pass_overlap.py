"""When does a block get its p-th visit?  The reference's scheduler against the GPU's, from their own order logs.

    python profiles/pass_overlap.py [workload] [iters]

The reference runs `iters` passes in ONE scheduler session (RayBlockScheduler.py:185, iters = 5): a worker takes a block
of minimal update count among the FREE rows x FREE columns (:50-58), and with n_threads + 1 stripes it sees one or two
of each - so a block often gets its second visit while others still wait for their first.  This script measures how
far the passes overlap: for every visit its position in the issue order (in units of one pass), by the block's visit
number p, as quantiles.  Legs: the CPU oracle's threaded FPSGD (S1 restatement, 16 threads, 17 x 17, real timing), the
GPU kernel's schedule 2 (one launch, least-visited-of-two), and schedule 1 (a launch per pass: no overlap).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import recommender_amd as ra  # noqa: E402
import workloads  # noqa: E402
from oracle import oracle as orc  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ml25m"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
grid = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else None
HP = (workloads.ALPHA, workloads.BETA1, workloads.BETA2)


def table(tag, order, n_blocks_per_pass, ncols):
    """order: [m][2] issued blocks (r, c) in issue order."""
    order = order[order[:, 0] >= 0]
    ids = order[:, 0].astype(np.int64) * ncols + order[:, 1]
    seen = {}
    p = np.empty(len(ids), dtype=np.int64)
    for j, b in enumerate(ids.tolist()):
        p[j] = seen.get(b, 0)
        seen[b] = p[j] + 1
    t = np.arange(len(ids)) / n_blocks_per_pass
    print(f"{tag}: {len(ids)} visits, {len(seen)} blocks, {len(ids) / n_blocks_per_pass:.2f} passes")
    for q in range(int(p.max()) + 1):
        tq = t[p == q]
        qs = np.quantile(tq, [0, .1, .25, .5, .75, .9, .99, 1])
        print(f"   visit {q}: issued at 0/10/25/50/75/90/99/100 % = " + " ".join(f"{x:.2f}" for x in qs) +
              f"; later than a full pass behind schedule: {np.mean(tq > q + 1):.2f}")


train, test, U, I, k = workloads.synth(torch, name)
m = ra.ExplicitMF(n_threads=15, n_factors=k, alpha=HP[0], beta1=HP[1], beta2=HP[2], implicit="none", verbose=False, grid=grid)
m.load_samples_from_device(train, test, U, I)
np.random.seed(20240611)
m.random_renumber_samples()
h_train = m._d_samples.cpu().numpy()
W = 17
ubp, ibp = int(U / W) + 1, int(I / W) + 1
blocks, ptr = orc.bucket(h_train, ubp, ibp, W, W)
rng = np.random.default_rng(0)
P = rng.normal(scale=1.0 / k, size=(U, k)); Q = rng.normal(scale=1.0 / k, size=(I, k))
bu, bi = np.zeros(U), np.zeros(I)
for seed in (1, 2):
    order = orc.fpsgd(P, Q, bu, bi, 3.5, None, blocks, ptr, W, None, None, *HP, 0, iters, 16, deterministic=False, seed=seed,
                      want_order=True)
    table(f"reference S1 (oracle threads, 16 workers, 17x17, iters={iters}, seed {seed})", order, W * W, W)
m.balanced_renumber_samples()
g = m.build_grid()
m.init_params_device(1)
m._to_dev()
nb = g.info.Gr * g.info.Gc
nonempty = int((np.diff(g.export()["blk_run_ptr"]) > 0).sum())
for barrier in (False, True):
    m.pass_barrier = barrier
    log = torch.full((iters * nb, 2), -1, dtype=torch.int32, device="cuda")
    m.run_epochs(iters, order_log=log)
    g.status()
    table(f"GPU {'schedule 1 (launch per pass)' if barrier else 'schedule 2 (one launch)'}, grid {g.info.Gr}x{g.info.Gc}",
          log.cpu().numpy(), nonempty, g.info.Gc)

#!/usr/bin/env python3
"""How much of each item stripe is one item's serial chain?  python profiles/hot_stats.py [workload] [Gr,Gc]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import recommender_amd as ra

wl = sys.argv[1] if len(sys.argv) > 1 else "netflix"
grid = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 and sys.argv[2] else None
train, test, U, I, k = bench.synth(torch, wl, 20240611 + sorted(bench.WORKLOADS).index(wl))
m = ra.ExplicitMF(n_threads=15, n_factors=k, implicit="none", grid=grid, verbose=False)
m.load_samples_from_device(train, test, U, I)
m.balanced_renumber_samples()
g = m.build_grid()
ubp, ibp, Gr, Gc = g.info.ubp, g.info.ibp, g.info.Gr, g.info.Gc
s = m._d_samples
u, i = s[:, 0].long(), s[:, 1].long()
n = u.numel()
deg = torch.bincount(i, minlength=I)
col = torch.arange(I, device=deg.device) // ibp
col_nnz = torch.zeros(Gc, dtype=torch.long, device=deg.device).index_add_(0, col, deg)
# hottest and second hottest item per column
d2 = torch.full((Gc * ibp,), 0, dtype=torch.long, device=deg.device)
d2[:I] = deg
d2 = d2.view(Gc, ibp)
top2 = torch.topk(d2, 2, dim=1)
hot_deg, sec_deg = top2.values[:, 0], top2.values[:, 1]
hot_item = top2.indices[:, 0] + torch.arange(Gc, device=deg.device) * ibp
share = hot_deg.double() / col_nnz.clamp(min=1).double()
print(f"grid {Gr}x{Gc} ubp {ubp} ibp {ibp} n {n}")
q = lambda t, p: float(torch.quantile(t.double(), p))
print("hot share per column: min %.2f p25 %.2f med %.2f p75 %.2f max %.2f" % (q(share, 0), q(share, .25), q(share, .5), q(share, .75), q(share, 1)))
print("hot deg: max %d med %d min %d ; second deg: max %d med %d" % (hot_deg.max(), hot_deg.median(), hot_deg.min(), sec_deg.max(), sec_deg.median()))
# runs: (user, column) pairs; hot runs contain the column's hot item
rk = u * Gc + (i // ibp)
is_hot = hot_item[i // ibp] == i
uniq, inv, cnt = torch.unique(rk, return_inverse=True, return_counts=True)
run_hot = torch.zeros(uniq.numel(), dtype=torch.bool, device=u.device)
run_hot[inv[is_hot]] = True
nh = int(run_hot.sum())
print(f"runs {uniq.numel()} (avg len {n/uniq.numel():.2f}); hot runs {nh} ({100.0*nh/uniq.numel():.1f}%), avg len {float(cnt[run_hot].double().mean()):.2f}, "
      f"singletons {100.0*float((cnt[run_hot]==1).double().mean()):.1f}%, ratings in hot runs {100.0*float(cnt[run_hot].sum())/n:.1f}% ; hot ratings {100.0*float(is_hot.sum())/n:.1f}%")
rc = uniq % Gc
for c in (0, Gc // 4, Gc // 2, 3 * Gc // 4, Gc - 1):
    mk = rc == c
    hk = mk & run_hot
    print(f"col {c}: nnz {int(col_nnz[c])} hot deg {int(hot_deg[c])} second {int(sec_deg[c])} runs {int(mk.sum())} hot runs {int(hk.sum())} "
          f"avg hot-run len {float(cnt[hk].double().mean()):.2f}")

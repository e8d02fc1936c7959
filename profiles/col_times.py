#!/usr/bin/env python3
"""Per-item-stripe tenure times of one free-block epoch (profiling aid): python profiles/col_times.py [bench args]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import workloads
import recommender_amd as ra
from recommender_amd import _abi

wl = sys.argv[1] if len(sys.argv) > 1 else "netflix"
grid = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 and sys.argv[2] else None
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hot_buffers = int(sys.argv[4]) if len(sys.argv) > 4 else None
train, test, U, I, k = workloads.synth(torch, wl)
m = ra.ExplicitMF(n_threads=15, n_factors=k, implicit="none", grid=grid, waves_per_wg=waves, hot_buffers=hot_buffers, verbose=False)
m.load_samples_from_device(train, test, U, I)
m.balanced_renumber_samples()
g = m.build_grid()
m.init_params_device(1); m._to_dev()
m.run_epochs(1); g.status()
t0 = time.perf_counter(); m.run_epochs(1); g.status(); dt = time.perf_counter() - t0
out = np.zeros((g.info.Gc, 3), dtype=np.int64)
_abi.check(_abi.lib().mfsgd_grid_col_times(g.handle, out.ctypes.data_as(C.c_void_p), None))
nnz, st, en = out[:, 0], out[:, 1], out[:, 2]
ok = en > 0
t_first = st[ok].min()
dur = (en - st) / 100.0          # us
hot = g.col_hot()
n_hot_items = (hot >= 0).sum(1)
hot = hot[:, 0]
deg = torch.bincount(m._d_samples[:, 1].long(), minlength=I).cpu().numpy()
print(f"hot_buffers {m.hot_buffers}, stripes with a hot item {g.info.n_hot_stripes} ({int(g.info.n_hot)} ratings), top item degrees {np.sort(deg)[-5:][::-1].tolist()}")
print(f"epoch {dt*1e3:.1f} ms, grid {g.info.Gr}x{g.info.Gc}, runs {g.info.n_runs}, n {g.info.n}, avg run {g.info.n/g.info.n_runs:.2f}")
print("tenure us: min %.0f med %.0f max %.0f ; end-of-last %.0f us ; start-of-last %.0f us" % (dur[ok].min(), np.median(dur[ok]), dur[ok].max(), (en[ok].max()-t_first)/100.0, (st[ok].max()-t_first)/100.0))
order = np.argsort(-nnz)
slow = np.argsort(-dur)
for c in list(slow[:6]) + list(order[:3]) + list(order[len(order)//2:len(order)//2+3]) + list(order[-3:]):
    print(f"col {c}: {n_hot_items[c]} hot items, top deg {deg[hot[c]] if hot[c] >= 0 else 0} nnz {nnz[c]} start {(st[c]-t_first)/100:.0f} us dur {dur[c]:.0f} us -> {nnz[c]/max(dur[c],1):.2f} ratings/us")
rate = nnz[ok] / np.maximum(dur[ok], 1)
print("ratings/us per tenure: min %.2f med %.2f max %.2f" % (rate.min(), np.median(rate), rate.max()))

#!/usr/bin/env python3
"""One item stripe alone on the GPU: what one rating of a very popular item costs (the serial chain of SGD on
its row).  python profiles/hot_chain.py [n_users] [rows] [extra_items] [waves]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import recommender_amd as ra
U = int(sys.argv[1]) if len(sys.argv) > 1 else 140000
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 640
extra = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # ratings of other items of the stripe, per 100 users
waves = int(sys.argv[4]) if len(sys.argv) > 4 else 0
hot_buffers = int(sys.argv[5]) if len(sys.argv) > 5 else None
rng = np.random.default_rng(3)
u = np.arange(U)
s = [np.column_stack([u, np.zeros(U, np.int64), rng.integers(1, 6, U)])]
I = 35
if extra:
    m = U * extra // 100
    s.append(np.column_stack([rng.integers(0, U, m), rng.integers(1, I, m), rng.integers(1, 6, m)]))
a = np.concatenate(s).astype(np.int32)
a = a[np.unique(a[:, 0].astype(np.int64) * I + a[:, 1], return_index=True)[1]]
k = 128
m = ra.ExplicitMF(n_threads=15, n_factors=k, implicit="none", grid=(rows, 1), waves_per_wg=waves, hot_buffers=hot_buffers, verbose=False)
t = torch.from_numpy(a).cuda()
m.load_samples_from_device(t, t[:10].clone(), U, I)
g = m.build_grid()
m.init_params_device(1); m._to_dev()
m.run_epochs(1); g.status()
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); m.run_epochs(1); g.status(); ts.append(time.perf_counter() - t0)
dt = min(ts)
print(f"one stripe: {len(a)} ratings ({U} of the leading item, {int(g.info.n_hot)} in hot segments, hot_buffers {m.hot_buffers}), grid {g.info.Gr}x{g.info.Gc}: "
      f"{dt*1e3:.2f} ms -> {dt/len(a)*1e6:.3f} us per rating, {len(a)/dt/1e6:.2f} ratings/us")
if os.environ.get("MFSGD_LIBRARY"):
    import ctypes as C
    from recommender_amd import _abi
    out = np.zeros((g.info.Gc, 3), dtype=np.int64)
    _abi.check(_abi.lib().mfsgd_grid_col_times(g.handle, out.ctypes.data_as(C.c_void_p), None))

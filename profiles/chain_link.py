#!/usr/bin/env python3
"""Time per link of ONE item's update chain: n users each rate item 0 once (plus nothing else), one block.
python profiles/chain_link.py [k] [n] [mode] [implicit] [dtype]
mode: tickets | pipeline (matrix cores) | walker (one wave, item row in registers) | spread (n ratings over 64 items: no chain)
The epoch time / n is the serial cost of one rating of a hot item on that path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import recommender_amd as ra
from recommender_amd import _abi
from recommender_amd.SGDRecommender import Grid

k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
mode = sys.argv[3] if len(sys.argv) > 3 else "tickets"
implicit = sys.argv[4] if len(sys.argv) > 4 else "none"
dtype = sys.argv[5] if len(sys.argv) > 5 else "float32"
_abi.require_gpu()
I = 64
u = np.arange(n, dtype=np.int32)
i = (np.arange(n) % I).astype(np.int32) if mode == "spread" else np.zeros(n, np.int32)
s = torch.from_numpy(np.column_stack([u, i, np.random.default_rng(1).integers(1, 6, n).astype(np.int32)])).cuda()
m = ra.ExplicitMF(n_threads=0, n_factors=k, implicit=implicit, grid=(1, 1), verbose=False, dtype=dtype,
                  hot_mode={"pipeline": 1, "walker": 2}.get(mode, 0))
m.load_samples_from_device(s, s[:16].clone(), n, I)
m._grid = Grid(m._d_samples, n, I, n, I, order=1, hot_segments=(1, 1) if mode in ("pipeline", "walker") else 0)
m._grid.set_hot_mode(m.hot_mode)
m.init_params_device(1); m._to_dev()
m.run_epochs(1); m._grid.status()
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); m.run_epochs(1); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
m._grid.status()
dt = min(ts)
print(f"k={k} {dtype} implicit={implicit} n={n} mode={mode} hot={int(m._grid.info.n_hot)}: epoch {dt*1e3:.3f} ms -> {dt/n*1e9:.1f} ns per rating")
if os.environ.get("MFSGD_LIBRARY"):      # the instrumented build prints its phase sums from mfsgd_grid_col_times
    import ctypes as C
    out = np.zeros((m._grid.info.Gc, 3), dtype=np.int64)
    _abi.check(_abi.lib().mfsgd_grid_col_times(m._grid.handle, out.ctypes.data_as(C.c_void_p), None))

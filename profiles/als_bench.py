#!/usr/bin/env python3
"""One ALS iteration (user half-step + item half-step, mfsgd_hip_als_step_f32) on a bench.py workload: time per
half-step, TFLOP/s against the fp32 vector peak, RMSE before/after, and the same half-step done by the CPU
restatement on a sample of rows beside it.  python profiles/als_bench.py [workload] [k] [iters]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from recommender_amd import _abi
from recommender_amd.ALSRecommender import ExplicitMF as ALS, _transpose
from recommender_amd.Recommender import CSRView

wl = sys.argv[1] if len(sys.argv) > 1 else "netflix"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 128
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
_abi.require_gpu()
train, test, U, I, _ = bench.synth(torch, wl, 20240611 + sorted(bench.WORKLOADS).index(wl))
n = int(train.shape[0])
order = torch.argsort(train[:, 0].long(), stable=True)
rp = torch.zeros(U + 1, dtype=torch.int64, device="cuda")
rp[1:] = torch.cumsum(torch.bincount(train[:, 0].long(), minlength=U), 0)
by_user = CSRView(rp, train[order, 1].contiguous(), train[order, 2].float().contiguous(), U, I)
by_item = _transpose(by_user)
g = torch.Generator(device="cuda"); g.manual_seed(3)
X = torch.rand(U, k, device="cuda", generator=g); Y = torch.rand(I, k, device="cuda", generator=g)
lib = _abi.lib()
lam = 0.05

def half(lat, fix, r):
    _abi.check(lib.mfsgd_hip_als_step_f32(_abi.ptr(lat), _abi.ptr(fix), _abi.ptr(r.rowptrs), _abi.ptr(r.colinds),
                                          _abi.ptr(r.values), int(r.nrows), int(fix.shape[0]), n, k, lam, None))

def rmse(s):
    pred = (X[s[:, 0].long()] * Y[s[:, 1].long()]).sum(1)
    return float(torch.sqrt(((s[:, 2].float() - pred) ** 2).mean()))

r0 = (rmse(train), rmse(test))
tu, ti = [], []
for _ in range(iters):
    torch.cuda.synchronize(); t0 = time.perf_counter(); half(X, Y, by_user); tu.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); half(Y, X, by_item); ti.append(time.perf_counter() - t0)     # the entry synchronises
r1 = (rmse(train), rmse(test))
flop = lambda rows: n * (k * (k + 1) + 2 * k) + rows * (k ** 3 / 3 + 2 * k * k)   # Gram triangle + rhs, factor, solves
out = {"kernel": "k_als_rows", "workload": f"{wl}-shape {U}x{I}, {n} train ratings", "k": k, "dtype": "f32", "lambda": lam,
       "user_half_step_ms": min(tu) * 1e3, "item_half_step_ms": min(ti) * 1e3,
       "user_tflops": flop(U) / min(tu) / 1e12, "item_tflops": flop(I) / min(ti) / 1e12,
       "frac_of_fp32_vector_peak": (flop(U) + flop(I)) / (min(tu) + min(ti)) / 1e12 / 157.3,
       "rmse_train_test_before": r0, "rmse_train_test_after": r1, "iterations": iters}
if os.environ.get("ALS_BENCH_NO_CPU"):       # timing experiments with broken numerics
    print(json.dumps(out)); sys.exit(0)
# CPU restatement (fp32, one thread) on the first rows that hold ~2 M ratings
sys.path.insert(0, ROOT)
from oracle import oracle as orc
hrp = by_user.rowptrs.cpu().numpy()
rows = int(np.searchsorted(hrp, 2_000_000))
hci, hv = by_user.colinds[:hrp[rows]].cpu().numpy(), by_user.values[:hrp[rows]].cpu().numpy()
Yh = Y.cpu().numpy()
t0 = time.perf_counter()
orc.als_step(np.zeros((rows, k), np.float32), Yh, hrp[:rows + 1], hci, hv, lam)
dt = time.perf_counter() - t0
out["cpu_port"] = {"rows": rows, "ratings": int(hrp[rows]), "seconds": dt, "ratings_per_s": hrp[rows] / dt, "cores": 1,
                   "gpu_user_step_ratings_per_s": n / min(tu)}
print(json.dumps(out))

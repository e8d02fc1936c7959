#!/usr/bin/env python3
"""Batched scoring + top-N (mfsgd_hip_topn_f32) on the netflix shape: time, TFLOP/s against the fp32 MFMA peak, and
the same job done by torch (matmul in slabs + topk) beside it.  python profiles/topn_bench.py [n_users] [n] [k]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from recommender_amd import _abi
U, I, k = 480189, 17770, 128
m = int(sys.argv[1]) if len(sys.argv) > 1 else U
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
if len(sys.argv) > 3:
    k = int(sys.argv[3])
g = torch.Generator(device="cuda"); g.manual_seed(5)
P = torch.randn(U, k, device="cuda", generator=g) * 0.2
Q = torch.randn(I, k, device="cuda", generator=g) * 0.2
bu = torch.randn(U, device="cuda", generator=g); bi = torch.randn(I, device="cuda", generator=g)
users = torch.randperm(U, device="cuda", generator=g)[:m].int()
items = torch.empty(m, n, dtype=torch.int32, device="cuda"); scores = torch.empty(m, n, device="cuda")
lib = _abi.lib()
def run():
    _abi.check(lib.mfsgd_hip_topn_f32(_abi.ptr(P), _abi.ptr(Q), _abi.ptr(bu), _abi.ptr(bi), 3.5, _abi.ptr(users), m, I, k, 1, n,
                                      _abi.ptr(items), _abi.ptr(scores), None))
run(); torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
for a, b in evs:
    a.record(); run(); b.record()
torch.cuda.synchronize()
ms = min(a.elapsed_time(b) for a, b in evs)
flop = 2.0 * m * I * k
def torch_way():
    out_i = torch.empty(m, n, dtype=torch.int64, device="cuda"); out_s = torch.empty(m, n, device="cuda")
    for s0 in range(0, m, 16384):
        u = users[s0:s0 + 16384].long()
        sc = P[u] @ Q.T + bi[None, :] + bu[u][:, None] + 3.5
        v, ix = torch.topk(sc, n, dim=1)
        out_s[s0:s0 + 16384] = v; out_i[s0:s0 + 16384] = ix
    return out_i, out_s
ti, ts = torch_way(); torch.cuda.synchronize()
t0 = time.perf_counter(); torch_way(); torch.cuda.synchronize(); tms = (time.perf_counter() - t0) * 1e3
agree = float((ti[:, 0].int() == items[:, 0]).float().mean())
print(json.dumps({"kernel": "k_topn", "users": m, "items": I, "k": k, "n": n, "ms": ms, "tflops": flop / ms / 1e9,
                  "frac_of_fp32_mfma_peak": flop / ms / 1e9 / 157.3, "users_per_s": m / ms * 1e3,
                  "torch_matmul_topk_ms": tms, "top1_agreement_with_torch": agree}))

#!/usr/bin/env python3
"""What the memory system gives for the access pattern of the epoch kernel: random 512-byte rows of a 246 MB
matrix read (and written back).  python profiles/gather_bw.py"""
import torch, time
U, k = 480189, 128
P = torch.randn(U, k, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
for n in (1 << 22, 1 << 24):
    idx = torch.randint(0, U, (n,), device="cuda", generator=g)
    out = torch.empty(n, k, device="cuda")
    for name, fn in (("gather  P[idx] -> out", lambda: torch.index_select(P, 0, idx, out=out)),
                     ("scatter out -> P[idx]", lambda: P.index_copy_(0, idx, out))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{name}: n={n} rows {dt*1e3:.2f} ms  {n/dt/1e6:.0f} Mrows/s  random-side {n*k*4/dt/1e12:.2f} TB/s (total {2*n*k*4/dt/1e12:.2f} TB/s)")

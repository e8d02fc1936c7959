#!/usr/bin/env python3
"""Runs (= random row accesses of the streamed factor matrix) per epoch for both orientations of the kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
for wl in sys.argv[1:] or ["netflix"]:
    train, test, U, I, k = bench.synth(torch, wl, 20240611 + sorted(bench.WORKLOADS).index(wl))
    u, i = train[:, 0].long(), train[:, 1].long()
    n = u.numel()
    perm_u = torch.randperm(U, device=u.device); perm_i = torch.randperm(I, device=u.device)
    u, i = perm_u[u], perm_i[i]
    print(f"{wl}: U {U} I {I} n {n} k {k}")
    for w in (35, 70, 140):
        r = torch.unique(u * ((I + w - 1) // w) + i // w).numel()
        print(f"  item-stationary, {w:4d} items/stripe ({(I+w-1)//w} stripes): runs {r/1e6:.1f} M  ({n/r:.2f} ratings/run)")
    for w in (140, 280):
        r = torch.unique(i * ((U + w - 1) // w) + u // w).numel()
        print(f"  user-stationary, {w:4d} users/stripe ({(U+w-1)//w} stripes): runs {r/1e6:.1f} M  ({n/r:.2f} ratings/run)")

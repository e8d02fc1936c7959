#!/usr/bin/env python3
"""What ONE rank of an N-GPU DSGD run computes, timed alone on one GPU (no exchange): the N sub-epoch kernels
of rank `rank` on the netflix shape.  python profiles/dsgd_rank_sim.py <world> [Gr,Gc] [hot_batch] [rank] [workload] [waves]
Tells how much of bench.py --gpus N is kernel time, before RCCL enters (8-GPU runs are the driver's)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from recommender_amd import _abi, dsgd
from recommender_amd.SGDRecommender import Grid, balanced_permutation

world = int(sys.argv[1])
grid = sys.argv[2] if len(sys.argv) > 2 else ""
hot_batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rank = int(sys.argv[4]) if len(sys.argv) > 4 else 0
wl = sys.argv[5] if len(sys.argv) > 5 else "netflix"
waves = int(sys.argv[6]) if len(sys.argv) > 6 else 0
_abi.require_gpu()
train, test, U, I, k = bench.synth(torch, wl, 20240611 + sorted(bench.WORKLOADS).index(wl))
n = int(train.shape[0])
if grid:
    Gr, Gc = (int(x) for x in grid.split(","))
    ubp, ibp = -(-U // Gr), -(-I // Gc)
else:
    ubp, ibp = dsgd.choose_multi_grid(U, I, n, world, k, hot_batch=hot_batch)
du = torch.bincount(train[:, 0].long(), minlength=U).cpu().numpy()
di = torch.bincount(train[:, 1].long(), minlength=I).cpu().numpy()
rp = torch.from_numpy(balanced_permutation(du, ubp).astype(np.int32)).cuda()
cp = torch.from_numpy(balanced_permutation(di, ibp).astype(np.int32)).cuda()
lib = _abi.lib()
_abi.check(lib.mfsgd_hip_remap(_abi.ptr(train), n, 3, _abi.ptr(rp), _abi.ptr(cp), U, I, None))
plan = dsgd.DSGDPlan(U, I, world, ubp, ibp)
u0, u1 = plan.user_range(rank)
tl = train[(train[:, 0] >= u0) & (train[:, 0] < u1)].contiguous()
grids = []
for q in range(world):
    i0, i1 = plan.item_range(q)
    grids.append(Grid(tl[(tl[:, 1] >= i0) & (tl[:, 1] < i1)].contiguous(), U, I, ubp, ibp, order=1))
    if hot_batch > 1:
        grids[-1].set_hot_batch(hot_batch)
P = torch.randn(U, k, device="cuda") / k
Q = torch.randn(I, k, device="cuda") / k
b_u = torch.zeros(U, device="cuda"); b_i = torch.zeros(I, device="cuda")
b = float(train[:, 2].float().mean())

def compute(q):
    _abi.check(lib.mfsgd_hip_epoch_f32(grids[q].handle, k, _abi.ptr(P), _abi.ptr(Q), _abi.ptr(b_u), _abi.ptr(b_i), b, None,
                                       0.006, 0.05, 0.015, 0, 0, 1, 1, 7 + q, 0, waves, None, None))

for q in range(world):
    compute(q)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(world)]
t0 = time.perf_counter()
for s in range(world):
    q = plan.shard_at(rank, s)
    ev[s][0].record(); compute(q); ev[s][1].record()
    torch.cuda.synchronize()          # the ring exchange waits for the kernel
wall = time.perf_counter() - t0
for gr in grids:
    gr.status()
if os.environ.get("MFSGD_LIBRARY"):     # instrumented build: phase counters of one sub-epoch grid (stderr)
    import ctypes as C
    gq = grids[plan.shard_at(rank, 1)]
    out = np.zeros((gq.info.Gc, 3), dtype=np.int64)
    _abi.check(lib.mfsgd_grid_col_times(gq.handle, out.ctypes.data_as(C.c_void_p), None))
ms = [a.elapsed_time(b_) for a, b_ in ev]
nn = [int(g.info.n) for g in grids]
print(f"world {world} rank {rank} grid {plan.Gr}x{plan.Gc} (per sub-epoch {-(-plan.Gr//world)}x{-(-plan.Gc//world)}) hot_batch {hot_batch} waves {waves}: "
      f"ratings per sub-epoch {min(nn)}..{max(nn)}, kernel ms {[round(x, 2) for x in ms]}, epoch wall {wall*1e3:.2f} ms "
      f"-> {n / wall / 1e9:.2f} G updates/s if every rank does the same and the exchange is free")

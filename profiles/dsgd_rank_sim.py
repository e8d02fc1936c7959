#!/usr/bin/env python3
"""What ONE rank of an N-GPU DSGD run computes, timed alone on one GPU (no exchange): the N sub-epoch kernels
of rank `rank` (a recommender_amd.dsgd.RankWorker, exactly what bench.py --gpus N runs per rank).
python profiles/dsgd_rank_sim.py <world> [Gr,Gc] [hot_buffers] [rank] [workload] [waves]
Tells how much of bench.py --gpus N is kernel time, before RCCL enters (8-GPU runs are the driver's)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import workloads
from recommender_amd import _abi, dsgd
from recommender_amd.SGDRecommender import balanced_permutation, interleave_stripes, item_costs

world = int(sys.argv[1])
grid = sys.argv[2] if len(sys.argv) > 2 else ""
hot_buffers = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rank = int(sys.argv[4]) if len(sys.argv) > 4 else 0
wl = sys.argv[5] if len(sys.argv) > 5 else "netflix"
waves = int(sys.argv[6]) if len(sys.argv) > 6 else 0
_abi.require_gpu()
train, test, U, I, k = workloads.synth(torch, wl)
n = int(train.shape[0])
if grid:
    Gr, Gc = (int(x) for x in grid.split(","))
    ubp, ibp = -(-U // Gr), -(-I // Gc)
else:
    ubp, ibp = dsgd.choose_multi_grid(U, I, n, world, k, max_item_degree=int(torch.bincount(train[:, 1].long(), minlength=I).max()))
du = torch.bincount(train[:, 0].long(), minlength=U).cpu().numpy()
di = torch.bincount(train[:, 1].long(), minlength=I).cpu().numpy()
plan0 = dsgd.DSGDPlan(U, I, world, ubp, ibp)
rows = max(1, -(-(plan0.user_range(0)[1] - plan0.user_range(0)[0]) // ubp))
cost = item_costs(di, rows, plan0.Gc, dsgd.pipeline_fits(I, ibp, rows, k, hot_buffers), 1, walker=True)
rp = torch.from_numpy(balanced_permutation(du, ubp).astype(np.int32)).cuda()
cp = torch.from_numpy(interleave_stripes(balanced_permutation(di, ibp, cost), ibp, world).astype(np.int32)).cuda()
_abi.check(_abi.lib().mfsgd_hip_remap(_abi.ptr(train), n, 3, _abi.ptr(rp), _abi.ptr(cp), U, I, None))
plan = dsgd.DSGDPlan(U, I, world, ubp, ibp)
u0, u1 = plan.user_range(rank)
b = float(train[:, 2].float().mean())
tl = train[(train[:, 0] >= u0) & (train[:, 0] < u1)].contiguous()
del train, test
w = dsgd.RankWorker(plan, rank, tl, k, b, (0.006, 0.05, 0.015), seed=7, hot_buffers=hot_buffers, waves_per_wg=waves)
for q in range(world):
    w.compute(q)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(world)]
t0 = time.perf_counter()
for s in range(world):
    q = plan.shard_at(rank, s)
    ev[s][0].record(); w.compute(q); ev[s][1].record()
    torch.cuda.synchronize()          # the ring exchange waits for the kernel
wall = time.perf_counter() - t0
# the ranks meet at every exchange: a sub-epoch lasts as long as its slowest shard, and some rank always holds it
worst = max(a.elapsed_time(b_) for a, b_ in ev) * world * 1e-3
w.status()
ms = [a.elapsed_time(b_) for a, b_ in ev]
nn = [int(g.info.n) for g in w.grids]
gi = w.grids[0].info
print(f"world {world} rank {rank} {wl} grid {plan.Gr}x{plan.Gc} (per sub-epoch {gi.Gr}x{-(-plan.Gc//world)}, ~{n // (plan.Gr * plan.Gc)} ratings per block) "
      f"hot_buffers {hot_buffers} ({sum(int(g.info.n_hot) for g in w.grids)} hot-segment ratings) waves {waves}: "
      f"ratings per sub-epoch {min(nn)}..{max(nn)}, kernel ms {[round(x, 2) for x in ms]}, epoch wall {wall*1e3:.2f} ms "
      f"-> {n / wall / 1e9:.2f} G updates/s if every rank does the same and the exchange is free; "
      f"{n / max(worst, 1e-9) / 1e9:.2f} G updates/s if every sub-epoch takes as long as this rank's slowest")

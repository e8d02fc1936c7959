#!/usr/bin/env python3
"""Epochs per launch (schedule 2 keeps all passes of a launch in one kernel, overlapped): ms per epoch for 1 / 4 / 16 passes per launch.
python profiles/multi_pass.py <workload> [k]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import workloads
import recommender_amd as ra

wl = sys.argv[1] if len(sys.argv) > 1 else "ml1m"
train, test, U, I, k = workloads.synth(torch, wl)
if len(sys.argv) > 2:
    k = int(sys.argv[2])
m = ra.ExplicitMF(n_threads=15, n_factors=k, implicit="none", verbose=False)
m.load_samples_from_device(train, test, U, I)
m.balanced_renumber_samples()
g = m.build_grid()
m.init_params_device(1); m._to_dev()
m.run_epochs(2); g.status()
for per in (1, 4, 16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(16 // per):
        m.run_epochs(per)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    g.status()
    print(f"{wl} k={k} grid {g.info.Gr}x{g.info.Gc}: {per} epochs per launch: {dt / 16 * 1e3:.3f} ms per epoch, {16 * g.info.n / dt / 1e9:.3f} G updates/s, held-out RMSE {m.device_errors()[1]:.4f}", flush=True)

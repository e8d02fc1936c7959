"""Development check of the hot-item walker on a k = 128 float32 build: oracle replay + time per link of one chain.
MFSGD_LIBRARY=<dev .so> python profiles/dev_walker.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recommender_amd as ra
from oracle import oracle as orc
import test_gpu_parity as tp
for kw in (dict(), dict(variant=1), dict(lead=True, grid=(2, 8), U=6000, I=600, n=250000), dict(waves=2), dict(mode="frozen")):
    kw = dict(kw)
    U, I, n, grid, waves, mode = kw.pop("U", 3000), kw.pop("I", 900), kw.pop("n", 200000), kw.pop("grid", (64, 16)), kw.pop("waves", 8), kw.pop("mode", "none")
    t0 = time.perf_counter()
    info = tp._replay_case(ra, orc, 128, U, I, n, grid, waves, mode, hot_bufs=8, **kw)
    print("replay ok", kw, mode, "hot ratings", int(info.n_hot), f"{time.perf_counter()-t0:.1f}s", flush=True)
import subprocess
for a in (["128", "20000", "walker"], ["128", "20000", "pipeline"], ["128", "2000", "walker"], ["128", "200", "walker"], ["128", "20000", "walker", "frozen"]):
    print(subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "chain_link.py")] + a, capture_output=True, text=True).stdout.strip().split("\n")[-1], flush=True)

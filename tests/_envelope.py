"""Held-out RMSE of the reference's OWN parallel schedule, as an envelope (test infrastructure).

The reference's parallel run (RayBlockScheduler.py:183-199) is: (n_threads+1)^2 grid (D3), `n_threads` workers, the
S1 scheduler handing out a free block of minimal update count with a RANDOM tie-break (:50-58), `iters` passes.  Its
result depends on the random draws and on the workers' timing, so "the reference's held-out RMSE after e epochs" is a
RANGE.  `s1_envelope` measures that range with the CPU oracle's threaded FPSGD (oracle/mf_oracle_impl.h orc_fpsgd:
the S1 restatement pinned by tests/golden/scheduler.npz + the A2 kernel pinned by tests/golden/kernels_*.npz) in the
reference's arithmetic (float64): for every seed and every run length e = 1..epochs one run of `iters = e` from the
same initial factors, then the exact RMSE on train and held-out ratings.
"""
import numpy as np


def s1_envelope(orc, train, test, n_users, n_items, init, b, hp, *, n_threads=16, epochs=5, seeds=(1, 2, 3),
                cumulative=False, log=None, train_rmse=True):
    """train / test: host int32 [n][3] in ONE id space; init = (P, Q, b_u, b_i) host arrays (any float dtype).
    Returns rmse[seed][e] = (train, held-out) for e = 0..epochs (e = 0: the initial factors).
    cumulative=True runs `epochs` times iters=1 on the same factors instead (an epoch barrier the reference does not
    have; a third of the work) - for quick looks only."""
    W = n_threads + 1                                  # RayBlockScheduler.py:185: BlockScheduler(n_threads + 1)
    ubp, ibp = int(n_users / W) + 1, int(n_items / W) + 1       # SGDRecommender.py:191-192
    assert -(-n_users // ubp) == W and -(-n_items // ibp) == W, "the reference's D3 needs a square grid here"
    blocks, ptr = orc.bucket(train, ubp, ibp, W, W)
    alpha, beta1, beta2 = hp
    f64 = [np.ascontiguousarray(a, dtype=np.float64) for a in init]

    def errs(P, Q, bu, bi):
        return (orc.rmse(train, None, None, P, Q, bu, bi, b, None, implicit=0) if train_rmse else float("nan"),
                orc.rmse(test, None, None, P, Q, bu, bi, b, None, implicit=0))

    out = {}
    e0 = errs(*f64)
    for seed in seeds:
        traj = [e0]
        cur = [a.copy() for a in f64]
        for e in range(1, epochs + 1):
            if cumulative:
                orc.fpsgd(*cur, b, None, blocks, ptr, W, None, None, alpha, beta1, beta2, 0, 1, n_threads,
                          deterministic=False, seed=seed * 1000 + e)
            else:
                cur = [a.copy() for a in f64]
                orc.fpsgd(*cur, b, None, blocks, ptr, W, None, None, alpha, beta1, beta2, 0, e, n_threads,
                          deterministic=False, seed=seed)
            traj.append(errs(*cur))
            if log:
                log(f"# S1 seed {seed} iters={e}: train {traj[-1][0]:.5f} held-out {traj[-1][1]:.5f}")
        out[seed] = traj
    return out


def band_distance(lo_a, hi_a, lo_b, hi_b):
    """Distance between the intervals [lo_a, hi_a] and [lo_b, hi_b] (0 when they overlap)."""
    return max(0.0, lo_a - hi_b, lo_b - hi_a)


def envelope_bounds(env, which=1):
    """min / max over the seeds, per epoch, of train (which=0) or held-out (which=1) RMSE."""
    trajs = list(env.values())
    n = len(trajs[0])
    lo = [min(t[e][which] for t in trajs) for e in range(n)]
    hi = [max(t[e][which] for t in trajs) for e in range(n)]
    return lo, hi

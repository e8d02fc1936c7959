#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE, runs only in the build container).

Imports the reference's own functions from /root/reference under three identity
stubs (numba / ray / csr are not installed here) and dumps
small input/output vectors to tests/golden/*.npz.  Nothing from the reference is
copied: the fixtures hold arrays only.  The stubs make the njit kernels run as
plain strict-IEEE NumPy float64, which is the better-defined parity target (the
real Numba build uses fastmath and is not bit-defined).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Skips itself when /root/reference is absent (GPU box).
"""
import ast
import io
import os
import random
import sys
import tempfile
import types
import contextlib

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.dont_write_bytecode = True


# ----------------------------------------------------------------------------- stubs
def _install_stubs():
    nb = types.ModuleType("numba")

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    nb.njit = njit
    nb.prange = range
    nb.int64 = np.int64
    sys.modules["numba"] = nb

    ray = types.ModuleType("ray")

    def remote(*a, **k):
        if len(a) == 1 and (callable(a[0]) or isinstance(a[0], type)) and not k:
            return a[0]
        return lambda f: f

    ray.remote = remote
    ray.get = lambda x: x
    sys.modules["ray"] = ray

    csr = types.ModuleType("csr")

    class CSR:
        """COO -> CSR, stable (input order) inside a row.  Within-row order of the
        real `csr` package is unpinned; this is the build's definition."""

        def __init__(self, nrows, ncols, nnz, rowptrs, colinds, values):
            self.nrows, self.ncols, self.nnz = nrows, ncols, nnz
            self.rowptrs, self.colinds, self.values = rowptrs, colinds, values

        @classmethod
        def from_coo(cls, rows, cols, vals, shape=None):
            rows, cols, vals = np.asarray(rows), np.asarray(cols), np.asarray(vals)
            nrows = int(rows.max()) + 1 if shape is None else shape[0]
            ncols = int(cols.max()) + 1 if shape is None else shape[1]
            order = np.argsort(rows, kind="stable")
            rp = np.zeros(nrows + 1, dtype=np.int64)
            np.cumsum(np.bincount(rows, minlength=nrows), out=rp[1:])
            return cls(nrows, ncols, len(rows), rp, cols[order].astype(np.int32), vals[order])

        # what ALSRecommender.als_step asks of the package (ALSRecommender.py:131-152)
        def row_cs(self, r):
            return self.colinds[self.rowptrs[r]:self.rowptrs[r + 1]]

        def row(self, r):
            out = np.zeros(self.ncols, dtype=self.values.dtype)
            out[self.row_cs(r)] = self.values[self.rowptrs[r]:self.rowptrs[r + 1]]
            return out

        def transpose(self):
            rows = np.repeat(np.arange(self.nrows), np.diff(self.rowptrs))
            return CSR.from_coo(self.colinds, rows, self.values, shape=(self.ncols, self.nrows))

    csr.CSR = CSR
    sys.modules["csr"] = csr
    return CSR


def _load_ray_block_scheduler_parts():
    """RayBlockScheduler.py runs a training job at import (module-level driver,
    lines 183-199).  Parse it and exec only `class BlockScheduler` and `def sgd`."""
    src = open(os.path.join(REF, "RayBlockScheduler.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body
            if (isinstance(n, ast.ClassDef) and n.name == "BlockScheduler")
            or (isinstance(n, ast.FunctionDef) and n.name == "sgd")]
    mod = ast.Module(body=keep, type_ignores=[])
    ns = {"np": np, "numba": sys.modules["numba"], "ray": sys.modules["ray"], "random": random}
    exec(compile(mod, "RayBlockScheduler.py[BlockScheduler,sgd]", "exec"), ns)
    return ns["BlockScheduler"], ns["sgd"]


def _load_mp_scheduler():
    src = open(os.path.join(REF, "MPBlockScheduler.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("BlockScheduler", "SubSample")]
    ns = {}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "MPBlockScheduler.py[BlockScheduler]", "exec"), ns)
    return ns["BlockScheduler"]


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


# ----------------------------------------------------------------------------- data
def synth_ratings(rng, U, I, n, user_sorted=True):
    """Distinct (u,i) pairs, integer ratings 1..5, every user and item present."""
    pairs = set()
    for u in range(U):
        pairs.add((u, int(rng.integers(I))))
    for i in range(I):
        pairs.add((int(rng.integers(U)), i))
    while len(pairs) < n:
        u = int(U * rng.random() ** 1.6)
        i = int(I * rng.random() ** 1.8)
        pairs.add((u, i))
    arr = np.array(sorted(pairs) if user_sorted else list(pairs), dtype=np.int64)
    if not user_sorted:
        rng.shuffle(arr)
    r = rng.integers(1, 6, size=len(arr))
    return np.column_stack([arr, r]).astype(np.int32)


def csr_of(CSR, samples):
    c = CSR.from_coo(samples[:, 0], samples[:, 1], samples[:, 2])
    return c.rowptrs.astype(np.int64), c.colinds.astype(np.int32)


def exact_rmse(samples, rp, ci, P, Q, bu, bi, b, y):
    """Independent evaluation of the numba_methods.py:14-17 predictor over ALL rows
    (mse2's exact=True path is broken in the reference, numba_methods.py:87-91)."""
    se = 0.0
    for u, i, r in samples[:, :3]:
        rated = ci[rp[u]:rp[u + 1]]
        R = np.sqrt(rated.shape[0])
        ys = y[rated].sum(axis=0)
        pred = Q[i].dot(P[u] + ys / R) + bu[u] + bi[i] + b
        se += (r - pred) ** 2
    return float(np.sqrt(se / samples.shape[0]))


def main():
    if not os.path.isdir(REF):
        print("reference absent: nothing to do")
        return 0
    CSR = _install_stubs()
    sys.path.insert(0, REF)
    import numba_methods as nm
    import SGDRecommender as sg
    RayScheduler, block_sgd = _load_ray_block_scheduler_parts()
    MPScheduler = _load_mp_scheduler()
    sgd2 = nm.sgd2.__func__ if isinstance(nm.sgd2, staticmethod) else nm.sgd2
    os.makedirs(OUT, exist_ok=True)
    alpha, beta1, beta2 = 0.006, 0.05, 0.015

    # ---------------------------------------------------------------- G1/G2/G4: A1, A3, RMSE
    for k in (4, 32, 128):
        rng = np.random.default_rng(1000 + k)
        U, I, n = 60, 45, 700
        samples = synth_ratings(rng, U, I, n)
        # a long run of one user and a repeated item are guaranteed by the skewed draw
        rp, ci = csr_of(CSR, samples)
        P = rng.normal(scale=1.0 / k, size=(U, k))
        Q = rng.normal(scale=1.0 / k, size=(I, k))
        y = rng.normal(scale=1.0 / k, size=(I, k))
        bu = rng.normal(scale=0.1, size=U)
        bi = rng.normal(scale=0.1, size=I)
        b = float(samples[:, 2].mean())
        # a pass in a shuffled (non user-grouped) order too
        order = rng.permutation(len(samples))
        out = dict(samples=samples, shuffled=samples[order], row_ptrs=rp, col_inds=ci,
                   P=P, Q=Q, y=y, b_u=bu, b_i=bi, b=b, alpha=alpha, beta1=beta1, beta2=beta2)
        for tag, fn, smp in (("a1", nm.sgd, samples), ("a3", sgd2, samples), ("a1s", nm.sgd, samples[order])):
            P2, Q2, y2, bu2, bi2 = (a.copy() for a in (P, Q, y, bu, bi))
            fn(P2, Q2, bu2, bi2, b, y2, smp, rp, ci, alpha, beta1, beta2)
            out.update({f"{tag}_P": P2, f"{tag}_Q": Q2, f"{tag}_y": y2, f"{tag}_b_u": bu2, f"{tag}_b_i": bi2})
            if tag == "a1":
                out["a1_rmse_exact"] = exact_rmse(samples, rp, ci, P2, Q2, bu2, bi2, b, y2)
        # y == 0 and beta-decay of zero stays zero only if (e/R)Q is dropped: implicit="none"
        # is emulated by restoring y to zero after every single-rating call.
        P2, Q2, bu2, bi2 = (a.copy() for a in (P, Q, bu, bi))
        y0 = np.zeros_like(y)
        for t in range(len(samples)):
            nm.sgd(P2, Q2, bu2, bi2, b, y0, samples[t:t + 1], rp, ci, alpha, beta1, beta2)
            y0[:] = 0.0
        out.update(none_P=P2, none_Q=Q2, none_b_u=bu2, none_b_i=bi2)
        out["none_rmse_exact"] = exact_rmse(samples, rp, ci, P2, Q2, bu2, bi2, b, y0)
        # frozen: y read-only (restore after each rating)
        P2, Q2, bu2, bi2 = (a.copy() for a in (P, Q, bu, bi))
        yf = y.copy()
        for t in range(len(samples)):
            nm.sgd(P2, Q2, bu2, bi2, b, yf, samples[t:t + 1], rp, ci, alpha, beta1, beta2)
            yf[:] = y
        out.update(frozen_P=P2, frozen_Q=Q2, frozen_b_u=bu2, frozen_b_i=bi2)
        # mse2 (always subsamples n//10 rows with the legacy global RNG)
        np.random.seed(77)
        out["mse2_seed77"] = float(nm.mse2(samples, rp, ci, P, Q, bu, bi, b, y))
        np.random.seed(77)
        out["mse2_seed77_idx"] = np.random.choice(samples.shape[0], size=samples.shape[0] // 10, replace=False)
        out["rmse_exact_init"] = exact_rmse(samples, rp, ci, P, Q, bu, bi, b, y)
        np.savez_compressed(os.path.join(OUT, f"kernels_k{k}.npz"), **out)
        print("kernels k=%d: a1-vs-a3 max|dP|=%.3g" % (k, np.abs(out["a1_P"] - out["a3_P"]).max()))

    # ---------------------------------------------------------------- G5/G6/G3/G8/G9: pipeline
    def pipeline(tag, U, I, n, n_threads, k, seed, iters=5, user_sorted=True):
        rng = np.random.default_rng(seed)
        raw = synth_ratings(rng, U, I, n, user_sorted=user_sorted)
        out = dict(raw=raw, n_threads=n_threads, k=k, seed=seed, iters=iters)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "r.npy")
            np.save(path, raw)
            m = sg.ExplicitMF(n_threads=n_threads, n_factors=k)
            np.random.seed(seed)
            with quiet():
                m.load_samples_from_npy(path, "all")
            out.update(d1_samples=m.samples.copy(), d1_test=m.test_samples.copy(),
                       n_users=m.n_users, n_items=m.n_items,
                       d1_rowptrs=np.asarray(m.ratings.rowptrs), d1_colinds=np.asarray(m.ratings.colinds))
            with quiet():
                groups = m.generate_indpendent_samples_new()
            row_perm = np.array([m.row_converter[i] for i in range(m.n_users)], dtype=np.int32)
            col_perm = np.array([m.col_converter[i] for i in range(m.n_items)], dtype=np.int32)
            out.update(row_perm=row_perm, col_perm=col_perm, d2_samples=m.samples.copy(),
                       d2_test=m.test_samples.copy(), u_bins=np.array(m.u_bins), i_bins=np.array(m.i_bins),
                       d4_rowptrs=np.asarray(m.ratings.rowptrs, dtype=np.int64),
                       d4_colinds=np.asarray(m.ratings.colinds, dtype=np.int32))
            Pw = len(groups)
            out["grid_shape"] = np.array([Pw, len(groups[0])])
            ok = all(len(g) == Pw for g in groups) and Pw == n_threads + 1
            out["grid_square"] = ok
            blocks = [g for row in groups for g in row]
            out["block_sizes"] = np.array([len(g) for g in blocks])
            out["blocks_concat"] = np.concatenate([g.reshape(-1, 3) for g in blocks]).astype(np.int32)
            if not ok:
                np.savez_compressed(os.path.join(OUT, f"pipeline_{tag}.npz"), **out)
                print(f"pipeline {tag}: grid {Pw}x{[len(g) for g in groups]} (ragged; bucketing only)")
                return
            with quiet():
                m.train(1, multithreaded=True)
            out.update(m1_P=m.P.copy(), m1_Q=m.Q.copy(), m1_y=m.y.copy(), m1_b=float(m.b))
            rp, ci = m.get_rated_by_user()
            rp = np.asarray(rp, dtype=np.int64); ci = np.asarray(ci, dtype=np.int32)
            P0, Q0, y0 = m.P.copy(), m.Q.copy(), m.y.copy()

            def run(mode):
                m.P, m.Q, m.y = P0.copy(), Q0.copy(), y0.copy()
                m.b_u = np.zeros(m.n_users); m.b_i = np.zeros(m.n_items)
                if mode == "none":
                    m.y[:] = 0.0
                for _ in range(iters):
                    for r in range(Pw):
                        for c in range(Pw):
                            ss = m.make_subsample((r, c))
                            g = groups[r][c]
                            if mode == "full":
                                block_sgd(ss.idx_offset, ss.P, ss.Q, ss.b_u, ss.b_i, ss.b, m.y, g, rp, ci,
                                          ss.alpha, ss.beta1, ss.beta2)     # slices are views: in place
                            else:
                                keep = m.y.copy()
                                for t in range(len(g)):
                                    block_sgd(ss.idx_offset, ss.P, ss.Q, ss.b_u, ss.b_i, ss.b, m.y, g[t:t + 1],
                                              rp, ci, ss.alpha, ss.beta1, ss.beta2)
                                    m.y[:] = keep
                res = {f"g8_{mode}_P": m.P.copy(), f"g8_{mode}_Q": m.Q.copy(), f"g8_{mode}_y": m.y.copy(),
                       f"g8_{mode}_b_u": m.b_u.copy(), f"g8_{mode}_b_i": m.b_i.copy(),
                       f"g8_{mode}_train_rmse": exact_rmse(m.samples, rp, ci, m.P, m.Q, m.b_u, m.b_i, m.b, m.y),
                       f"g8_{mode}_test_rmse": exact_rmse(m.test_samples, rp, ci, m.P, m.Q, m.b_u, m.b_i, m.b, m.y)}
                return res

            for mode in ("full", "none", "frozen"):
                out.update(run(mode))
            # one A2 call on a single off-diagonal block from the initial state (G3)
            m.P, m.Q, m.y = P0.copy(), Q0.copy(), y0.copy()
            m.b_u = np.zeros(m.n_users); m.b_i = np.zeros(m.n_items)
            ss = m.make_subsample((1, 2))
            Pb, Qb, bub, bib, yb = ss.P.copy(), ss.Q.copy(), ss.b_u.copy(), ss.b_i.copy(), m.y.copy()
            block_sgd(ss.idx_offset, Pb, Qb, bub, bib, ss.b, yb, groups[1][2], rp, ci, ss.alpha, ss.beta1, ss.beta2)
            out.update(g3_off=np.array(ss.idx_offset), g3_P=Pb, g3_Q=Qb, g3_b_u=bub, g3_b_i=bib, g3_y=yb)
            # G9 bold driver: increment() every 2*P^2 ticks calls mse2 (legacy global RNG)
            m.P, m.Q, m.y = out["g8_full_P"].copy(), out["g8_full_Q"].copy(), out["g8_full_y"].copy()
            m.b_u, m.b_i = out["g8_full_b_u"].copy(), out["g8_full_b_i"].copy()
            m.alpha, m.previous_mse, m.update_counter = 0.006, 0, 0
            np.random.seed(seed + 1)
            traj = []
            with quiet():
                for tick in range(2 * Pw * Pw * 6):
                    m.increment()
                    if (tick + 1) % (2 * Pw * Pw) == 0:
                        traj.append(m.alpha)
                        m.P *= 1.02 if len(traj) % 3 else 0.9      # perturb so that both branches fire
            out["g9_alpha"] = np.array(traj)
        np.savez_compressed(os.path.join(OUT, f"pipeline_{tag}.npz"), **out)
        print(f"pipeline {tag}: grid {Pw}x{Pw}, n_train={len(out['d2_samples'])}, "
              f"rmse full/none train = {out['g8_full_train_rmse']:.4f}/{out['g8_none_train_rmse']:.4f}")

    pipeline("a", U=90, I=70, n=1500, n_threads=3, k=8, seed=11)
    pipeline("b", U=96, I=64, n=1400, n_threads=3, k=16, seed=12)          # U % (c+1) == 0
    pipeline("shuf", U=80, I=75, n=1300, n_threads=3, k=8, seed=13, user_sorted=False)
    pipeline("ragged", U=100, I=9, n=500, n_threads=3, k=4, seed=14)       # ceil(U/ubp) != ceil(I/ibp)

    # ---------------------------------------------------------------- G7: scheduler traces
    traces = {}
    for width, iters, seed in ((4, 2, 5), (9, 3, 6)):
        # single consumer
        random.seed(seed)
        with quiet():
            s = RayScheduler(width, iters)
            seq, blk = [], s.get_next()
            while blk is not None:
                seq.append(blk)
                blk = s.get_next(completed=blk)
        traces[f"s1_single_w{width}_i{iters}_seed{seed}"] = np.array(seq)
        traces[f"s1_single_w{width}_i{iters}_seed{seed}_counter"] = np.array(s.get_update_counter())
        # scripted 3-consumer interleaving: round-robin completion order, consumer stops at first None
        random.seed(seed)
        with quiet():
            s = RayScheduler(width, iters)
            held = [s.get_next() for _ in range(3)]
            log = [(w, *held[w]) for w in range(3)]
            alive = [True] * 3
            while any(alive):
                for w in range(3):
                    if not alive[w]:
                        continue
                    nxt = s.get_next(completed=held[w])
                    if nxt is None:
                        alive[w] = False
                        log.append((w, -1, -1))
                    else:
                        held[w] = nxt
                        log.append((w, *nxt))
        traces[f"s1_multi3_w{width}_i{iters}_seed{seed}"] = np.array(log)
        traces[f"s1_multi3_w{width}_i{iters}_seed{seed}_counter"] = np.array(s.get_update_counter())
        # S2 deterministic tie-break (MP prototype)
        with quiet():
            s2 = MPScheduler([[0] * width for _ in range(width)], iters)
            seq, ss = [], s2.get_next()
            while ss is not None:
                seq.append(ss.block_pos)
                ss = s2.get_next(completed=ss.block_pos)
        traces[f"s2_single_w{width}_i{iters}"] = np.array(seq)
        traces[f"s2_single_w{width}_i{iters}_counter"] = np.array(s2.update_counter)
    np.savez_compressed(os.path.join(OUT, "scheduler.npz"), **traces)
    print("scheduler traces:", {k: v.shape for k, v in traces.items() if "counter" not in k})

    # ---------------------------------------------------------------- G10: real data, RMSE only
    csvp = os.path.join(REF, "NCF", "ratings.csv")
    import pandas as pd
    df = pd.read_csv(csvp)
    u = pd.factorize(df["userId"], sort=True)[0]
    i = pd.factorize(df["movieId"], sort=True)[0]
    r = df["rating"].to_numpy()
    raw = np.column_stack([u, i, r]).astype(np.int32)          # int truncation as Recommender.py:54
    # keep a 20k-row prefix so that the pure-Python oracle finishes in seconds
    raw = raw[:20000]
    raw[:, 1] = pd.factorize(raw[:, 1], sort=True)[0]
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "r.npy"); np.save(path, raw)
        m = sg.ExplicitMF(n_threads=3, n_factors=32)
        np.random.seed(2024)
        with quiet():
            m.load_samples_from_npy(path, "all")
            groups = m.generate_indpendent_samples_new()
            m.train(1, multithreaded=True)
        rp, ci = m.get_rated_by_user(); rp = np.asarray(rp, dtype=np.int64); ci = np.asarray(ci, dtype=np.int32)
        m.y[:] = 0.0
        Pw = len(groups)
        for _ in range(5):
            for rr in range(Pw):
                for cc in range(Pw):
                    ss = m.make_subsample((rr, cc)); g = groups[rr][cc]
                    for t in range(len(g)):
                        block_sgd(ss.idx_offset, ss.P, ss.Q, ss.b_u, ss.b_i, ss.b, m.y, g[t:t + 1], rp, ci,
                                  ss.alpha, ss.beta1, ss.beta2)
                        m.y[:] = 0.0
        res = dict(raw=raw, seed=2024, n_threads=3, k=32, iters=5,
                   train_rmse=exact_rmse(m.samples, rp, ci, m.P, m.Q, m.b_u, m.b_i, m.b, m.y),
                   test_rmse=exact_rmse(m.test_samples, rp, ci, m.P, m.Q, m.b_u, m.b_i, m.b, m.y))
        np.savez_compressed(os.path.join(OUT, "real_ml_small_20k.npz"), **res)
        print("real data 20k rows: train/test rmse = %.4f / %.4f" % (res["train_rmse"], res["test_rmse"]))

    # ---------------------------------------------------------------- ingestion: Recommender.renumber_column
    # (Recommender.py:18-25) on the first 5000 rows of the real file (sparse 64-bit-ish ids) and on signed ids
    import Recommender as refrec
    head = df.iloc[:5000][["userId", "movieId", "rating"]].copy()
    ren = refrec.Recommender.renumber_column(head.copy(), "userId")
    ren = refrec.Recommender.renumber_column(ren, "movieId")
    rng = np.random.default_rng(99)
    odd = pd.DataFrame({"a": rng.integers(-2**40, 2**40, size=3000) // 7 * 7, "b": rng.integers(0, 50, size=3000)})
    odd_r = refrec.Recommender.renumber_column(odd.copy(), "a")
    np.savez_compressed(os.path.join(OUT, "renumber.npz"),
                        user_raw=head["userId"].to_numpy(np.int64), item_raw=head["movieId"].to_numpy(np.int64),
                        rating_raw=head["rating"].to_numpy(np.float64),
                        user_new=ren["userId"].to_numpy(np.int64), item_new=ren["movieId"].to_numpy(np.int64),
                        odd_raw=odd["a"].to_numpy(np.int64), odd_new=odd_r["a"].to_numpy(np.int64))
    print("renumber_column: %d users, %d items in 5000 rows" % (ren["userId"].max() + 1, ren["movieId"].max() + 1))

    # ---------------------------------------------------------------- scoring: ExplicitMF.predict (SGDRecommender.py:82-83)
    rng = np.random.default_rng(4242)
    Pp = rng.normal(scale=0.3, size=(70, 64)).astype(np.float32)
    Qq = rng.normal(scale=0.3, size=(333, 64)).astype(np.float32)
    Qq[17] = Qq[5]                                     # an exact tie: the lower item id must come first
    us = rng.permutation(70)[:45].astype(np.int32)
    dense = sg.ExplicitMF.predict(us, np.arange(333), Pp.astype(np.float64), Qq.astype(np.float64))   # (45, 333)
    order = np.argsort(-dense, axis=1, kind="stable")[:, :16]
    np.savez_compressed(os.path.join(OUT, "topn.npz"), P=Pp, Q=Qq, users=us, items=order.astype(np.int32),
                        scores=np.take_along_axis(dense, order, axis=1))
    print("predict/top-16: %d users x %d items" % dense.shape)

    # ---------------------------------------------------------------- ALS: ExplicitMF.als_step / partial_train
    # (ALSRecommender.py:118-183).  The class is imported whole; njit is the identity stub, so als_step is the
    # reference's NumPy code run by the interpreter (numpy.linalg.solve, float64).
    import ALSRecommender as refals
    als_fn = refals.ExplicitMF.als_step
    als_fn = getattr(als_fn, "__func__", als_fn)
    out = {}
    for k, U, I, n in ((8, 60, 45, 900), (40, 90, 70, 2500)):
        rng = np.random.default_rng(1000 + k)
        s3 = synth_ratings(rng, U, I, n)                       # unique (user, item) pairs, user-sorted
        s3 = s3[s3[:, 0] != 7]                                 # a user without ratings
        s3[:, 2] *= 2                                          # the 10-point scale of load_samples_from_npy (:103)
        R = CSR.from_coo(s3[:, 0], s3[:, 1], s3[:, 2].astype(np.float64), shape=(U, I))
        X0, Y0 = rng.random((U, k)), rng.random((I, k))
        X1 = als_fn(X0.copy(), Y0, R, 0.05, "user")
        Y1 = als_fn(Y0.copy(), X1, R, 0.1, "item")
        Xn, Yn = X0.copy(), Y0.copy()
        for _ in range(3):                                     # partial_train's loop (:167-180)
            Xn = als_fn(Xn, Yn, R, 0.05, "user")
            Yn = als_fn(Yn, Xn, R, 0.1, "item")
        mse_fn = getattr(refals.ExplicitMF.mse, "__func__", refals.ExplicitMF.mse)
        out.update({f"k{k}_samples": s3.astype(np.int32), f"k{k}_shape": np.array([U, I]), f"k{k}_X0": X0, f"k{k}_Y0": Y0,
                    f"k{k}_X1": X1, f"k{k}_Y1": Y1, f"k{k}_X3": Xn, f"k{k}_Y3": Yn,
                    f"k{k}_rmse3": np.float64(mse_fn(s3, Xn, Yn))})
        print("als k=%d: rmse after 3 iterations %.4f" % (k, out[f"k{k}_rmse3"]))
    np.savez_compressed(os.path.join(OUT, "als.npz"), user_reg=0.05, item_reg=0.1, **out)
    return 0


if __name__ == "__main__":
    sys.exit(main())

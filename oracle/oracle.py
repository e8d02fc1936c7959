"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Pinned against tests/golden/*.npz (vectors produced by the reference's own functions,
oracle/gen_golden.py) in tests/test_oracle_golden.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmforacle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_sched_new.restype = C.c_void_p
        _LIB.orc_sched_counter.restype = C.POINTER(C.c_int32)
        _LIB.orc_filter_test.restype = C.c_int64
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _suf(dt):
    return {np.dtype(np.float64): ("f64", C.c_double), np.dtype(np.float32): ("f32", C.c_float)}[np.dtype(dt)]


def _chk(a, dt):
    assert a.dtype == dt and a.flags.c_contiguous, (a.dtype, dt)
    return a


def sgd(P, Q, b_u, b_i, b, y, samples, row_ptrs, col_inds, alpha, beta1, beta2,
        variant=0, implicit=2, idx_offset=(0, 0), z=None):
    """In place.  variant 0 = sgd (A1/A2), 1 = sgd2 (A3); implicit 0 none / 1 frozen / 2 full."""
    suf, ct = _suf(P.dtype)
    for a in (P, Q, b_u, b_i):
        _chk(a, P.dtype)
    samples = _chk(np.ascontiguousarray(samples, dtype=np.int32), np.int32)
    rp = None if row_ptrs is None else np.ascontiguousarray(row_ptrs, dtype=np.int64)
    ci = None if col_inds is None else np.ascontiguousarray(col_inds, dtype=np.int32)
    if implicit and z is None:
        _chk(y, P.dtype)
    rc = getattr(lib(), "orc_sgd_" + suf)(
        _p(P), _p(Q), _p(b_u), _p(b_i), ct(b), _p(y), _p(z), _p(samples), C.c_int64(samples.shape[0]),
        C.c_int32(samples.shape[1]), _p(rp), _p(ci), C.c_int32(P.shape[1]),
        C.c_int32(idx_offset[0]), C.c_int32(idx_offset[1]), ct(alpha), ct(beta1), ct(beta2),
        C.c_int32(variant), C.c_int32(implicit))
    assert rc == 0
    return P, Q, y, b_u, b_i


def rmse(samples, row_ptrs, col_inds, P, Q, b_u, b_i, b, y, implicit=2, z=None):
    suf, ct = _suf(P.dtype)
    samples = np.ascontiguousarray(samples, dtype=np.int32)
    rp = None if row_ptrs is None else np.ascontiguousarray(row_ptrs, dtype=np.int64)
    ci = None if col_inds is None else np.ascontiguousarray(col_inds, dtype=np.int32)
    out = C.c_double(0)
    rc = getattr(lib(), "orc_rmse_" + suf)(
        _p(P), _p(Q), _p(b_u), _p(b_i), ct(b), _p(y), _p(z), _p(samples), C.c_int64(samples.shape[0]),
        C.c_int32(samples.shape[1]), _p(rp), _p(ci), C.c_int32(P.shape[1]), C.c_int32(implicit), C.byref(out))
    assert rc == 0
    return out.value


def implicit_z(y, row_ptrs, col_inds, n_users):
    suf, _ = _suf(y.dtype)
    rp = np.ascontiguousarray(row_ptrs, dtype=np.int64)
    ci = np.ascontiguousarray(col_inds, dtype=np.int32)
    z = np.zeros((n_users, y.shape[1]), dtype=y.dtype)
    getattr(lib(), "orc_implicit_z_" + suf)(_p(y), _p(rp), _p(ci), C.c_int64(n_users), C.c_int32(y.shape[1]), _p(z))
    return z


def remap(samples, row_perm, col_perm):
    samples = _chk(samples, np.int32)
    lib().orc_remap(_p(samples), C.c_int64(samples.shape[0]), C.c_int32(samples.shape[1]),
                    _p(np.ascontiguousarray(row_perm, dtype=np.int32)),
                    _p(np.ascontiguousarray(col_perm, dtype=np.int32)))
    return samples


def filter_test(test, train, n_users, n_items):
    test = np.ascontiguousarray(test, dtype=np.int32)
    train = np.ascontiguousarray(train, dtype=np.int32)
    out = np.empty_like(test)
    m = lib().orc_filter_test(_p(test), C.c_int64(test.shape[0]), C.c_int32(test.shape[1]),
                              _p(train), C.c_int64(train.shape[0]), C.c_int32(train.shape[1]),
                              C.c_int64(n_users), C.c_int64(n_items), _p(out))
    return out[:m]


def bucket(samples, ubp, ibp, Pu, Pi):
    samples = np.ascontiguousarray(samples, dtype=np.int32)
    out = np.empty((samples.shape[0], 3), dtype=np.int32)
    ptr = np.zeros(Pu * Pi + 1, dtype=np.int64)
    rc = lib().orc_bucket(_p(samples), C.c_int64(samples.shape[0]), C.c_int32(samples.shape[1]),
                          C.c_int32(ubp), C.c_int32(ibp), C.c_int32(Pu), C.c_int32(Pi), _p(out), _p(ptr))
    if rc:
        raise ValueError("sample outside the grid")
    return out, ptr


def csr_from_coo(samples, n_users):
    samples = np.ascontiguousarray(samples, dtype=np.int32)
    rp = np.zeros(n_users + 1, dtype=np.int64)
    ci = np.empty(samples.shape[0], dtype=np.int32)
    lib().orc_csr_from_coo(_p(samples), C.c_int64(samples.shape[0]), C.c_int32(samples.shape[1]),
                           C.c_int64(n_users), _p(rp), _p(ci))
    return rp, ci


def renumber_column(values):
    values = np.ascontiguousarray(values, dtype=np.int64)
    out = np.empty(len(values), dtype=np.int32)
    lib().orc_renumber_column.restype = C.c_int64
    m = lib().orc_renumber_column(_p(values), C.c_int64(len(values)), _p(out))
    return out, int(m)


def topn(P, Q, users, n, b_u=None, b_i=None, b=0.0):
    P = np.ascontiguousarray(P, dtype=np.float32); Q = np.ascontiguousarray(Q, dtype=np.float32)
    users = np.ascontiguousarray(users, dtype=np.int32)
    wb = b_u is not None and b_i is not None
    bu = np.ascontiguousarray(b_u, dtype=np.float32) if wb else np.zeros(1, np.float32)
    bi = np.ascontiguousarray(b_i, dtype=np.float32) if wb else np.zeros(1, np.float32)
    items = np.empty((len(users), n), dtype=np.int32)
    scores = np.empty((len(users), n), dtype=np.float64)
    lib().orc_topn(_p(P), _p(Q), _p(bu), _p(bi), C.c_double(b), _p(users), C.c_int64(len(users)),
                   C.c_int64(Q.shape[0]), C.c_int32(P.shape[1]), C.c_int32(1 if wb else 0), C.c_int32(n),
                   _p(items), _p(scores))
    return items, scores


def als_step(latent, fixed, row_ptrs, col_inds, vals, lam):
    """In place: one ALS half-step (ALSRecommender.py:118-155) over the rows of the given CSR."""
    suf, ct = _suf(latent.dtype)
    _chk(latent, latent.dtype); _chk(fixed, latent.dtype)
    rp = np.ascontiguousarray(row_ptrs, dtype=np.int64)
    ci = np.ascontiguousarray(col_inds, dtype=np.int32)
    v = np.ascontiguousarray(vals, dtype=latent.dtype)
    rc = getattr(lib(), "orc_als_step_" + suf)(_p(latent), _p(fixed), _p(rp), _p(ci), _p(v), C.c_int64(latent.shape[0]),
                                               C.c_int32(latent.shape[1]), ct(lam))
    assert rc == 0, rc
    return latent


def bold_driver(train_rmse, alpha, previous):
    a, p = C.c_double(alpha), C.c_double(previous)
    lib().orc_bold_driver(C.c_double(train_rmse), C.byref(a), C.byref(p))
    return a.value, p.value


class Scheduler:
    """C restatement of BlockScheduler (RayBlockScheduler.py:14-116)."""

    def __init__(self, width, iters, deterministic=True, seed=0):
        self.W = width
        self._s = C.c_void_p(lib().orc_sched_new(C.c_int32(width), C.c_int32(iters),
                                                 C.c_int32(int(deterministic)), C.c_uint64(seed)))

    def get_next(self, completed=None):
        nr, nc = C.c_int32(0), C.c_int32(0)
        have = completed is not None
        r, c = completed if have else (0, 0)
        got = lib().orc_sched_get_next(self._s, C.c_int(int(have)), C.c_int32(r), C.c_int32(c),
                                       C.byref(nr), C.byref(nc))
        return (nr.value, nc.value) if got else None

    def check_completion(self):
        return bool(lib().orc_sched_all_done(self._s))

    def get_update_counter(self):
        p = lib().orc_sched_counter(self._s)
        return np.ctypeslib.as_array(p, shape=(self.W, self.W)).copy()

    def __del__(self):
        try:
            lib().orc_sched_free(self._s)
        except Exception:
            pass


def fpsgd(P, Q, b_u, b_i, b, y, blocks, block_ptr, W, row_ptrs, col_inds, alpha, beta1, beta2,
          implicit, iters, n_threads, deterministic=True, seed=0, z=None, want_order=False):
    """Threaded FPSGD run = the CPU baseline (scheduler + block kernel)."""
    suf, ct = _suf(P.dtype)
    blocks = np.ascontiguousarray(blocks, dtype=np.int32)
    block_ptr = np.ascontiguousarray(block_ptr, dtype=np.int64)
    rp = None if row_ptrs is None else np.ascontiguousarray(row_ptrs, dtype=np.int64)
    ci = None if col_inds is None else np.ascontiguousarray(col_inds, dtype=np.int32)
    order = np.full((iters * W * W, 2), -1, dtype=np.int32) if want_order else None
    rc = getattr(lib(), "orc_fpsgd_" + suf)(
        _p(P), _p(Q), _p(b_u), _p(b_i), ct(b), _p(y), _p(z), _p(blocks), _p(block_ptr), C.c_int32(W),
        _p(rp), _p(ci), C.c_int32(P.shape[1]), ct(alpha), ct(beta1), ct(beta2), C.c_int32(implicit),
        C.c_int32(iters), C.c_int32(n_threads), C.c_int32(int(deterministic)), C.c_uint64(seed), _p(order))
    assert rc == 0, rc
    return order

"""Drop-in for the reference's update / error functions (numba_methods.py, RayBlockScheduler.sgd).

Same names, same positional order, same in-place-and-return convention; the work is done by the HIP
kernels of libmfsgd.so on an MI355X.  Arrays may be NumPy (copied to the GPU, result copied back into
the caller's arrays) or torch ROCm tensors (zero-copy).  Arithmetic follows the dtype of the factor
arrays: float64 (what the reference uses) or float32.

reference                                         here
numba_methods.sgd   (numba_methods.py:7-24)       sgd
numba_methods.sgd2  (numba_methods.py:25-57)      sgd2
numba_methods.mse   (numba_methods.py:59-82)      mse   (the reference's version raises IndexError; this
                                                         evaluates the predictor of mse2 over all rows)
numba_methods.mse2  (numba_methods.py:84-103)     mse2  (exact=True works; exact=False draws the same
                                                         subsample sizes with np.random.choice)
RayBlockScheduler.sgd (RayBlockScheduler.py:118-145)  block_sgd
"""
import math

import numpy as np

from . import _abi

_IMPLICIT = {"none": 0, "frozen": 1, "full": 2}


def _run_sgd(variant, idx_offset, P, Q, b_u, b_i, b, y, samples, row_ptrs, col_inds, alpha, beta1, beta2,
             implicit="full"):
    _abi.require_gpu()
    imp = _IMPLICIT[implicit] if isinstance(implicit, str) else int(implicit)
    if y is None or (hasattr(y, "shape") and y.shape[0] == 0):
        imp, y = 0, None
    kind = _abi.float_kind(P, Q, b_u, b_i, y)
    td, ct = _abi.tdtype_of(kind), _abi.ctype_of(kind)
    dP, dQ, dbu, dbi = (_abi.to_device(a, td) for a in (P, Q, b_u, b_i))
    dy = _abi.to_device(y, td) if imp else None
    ds = _abi.to_device(samples, "int32")
    if ds.dim() != 2 or ds.shape[1] < 3:
        raise ValueError("samples must be an (n, >=3) array of [user, item, rating]")
    drp = _abi.to_device(row_ptrs, "int64") if imp else None
    dci = _abi.to_device(col_inds, "int32") if imp else None
    k = int(dP.shape[1])
    if dQ.shape[1] != k or (dy is not None and dy.shape[1] != k):
        raise ValueError("P, Q and y must have the same number of factors")
    fn = getattr(_abi.lib(), "mfsgd_hip_sgd_" + kind)
    _abi.check(fn(_abi.ptr(dP), _abi.ptr(dQ), _abi.ptr(dbu), _abi.ptr(dbi), ct(float(b)), _abi.ptr(dy), None,
                  _abi.ptr(ds), ds.shape[0], ds.shape[1], _abi.ptr(drp), _abi.ptr(dci), k,
                  int(idx_offset[0]), int(idx_offset[1]), ct(float(alpha)), ct(float(beta1)), ct(float(beta2)),
                  variant, imp, _abi.stream_ptr()))
    _abi.torch_mod().cuda.current_stream().synchronize()
    for dst, t in ((P, dP), (Q, dQ), (b_u, dbu), (b_i, dbi), (y, dy)):
        _abi.write_back(dst, t)
    return P, Q, y, b_u, b_i       # the reference's return order (numba_methods.py:24)


def sgd(P, Q, b_u, b_i, b, y, samples, row_ptrs, col_inds, alpha, beta1, beta2, implicit="full"):
    """One sequential SVD++ SGD pass over `samples`, global ids (numba_methods.py:8-24)."""
    return _run_sgd(0, (0, 0), P, Q, b_u, b_i, b, y, samples, row_ptrs, col_inds, alpha, beta1, beta2, implicit)


def sgd2(P, Q, b_u, b_i, b, y, samples, row_ptrs, col_inds, alpha, beta1, beta2, implicit="full"):
    """The scalar-loop variant with its own numerics: R=sqrt(d)+1, old p and q on the right-hand sides
    (numba_methods.py:27-57)."""
    return _run_sgd(1, (0, 0), P, Q, b_u, b_i, b, y, samples, row_ptrs, col_inds, alpha, beta1, beta2, implicit)


def block_sgd(idx_offset, P, Q, b_u, b_i, b, y, samples, row_ptrs, col_inds, alpha, beta1, beta2, implicit="full"):
    """Block-local pass: P,Q,b_u,b_i are the block's slices, samples carry global ids, y and the CSR stay
    global (RayBlockScheduler.py:119-145)."""
    return _run_sgd(0, idx_offset, P, Q, b_u, b_i, b, y, samples, row_ptrs, col_inds, alpha, beta1, beta2, implicit)


def _rmse(samples, row_ptrs, col_inds, P, Q, b_u, b_i, b, y, z=None):
    _abi.require_gpu()
    torch = _abi.torch_mod()
    imp = 0 if (y is None and z is None) else 1
    kind = _abi.float_kind(P, Q, b_u, b_i, y if imp else None)
    td, ct = _abi.tdtype_of(kind), _abi.ctype_of(kind)
    dP, dQ, dbu, dbi = (_abi.to_device(a, td) for a in (P, Q, b_u, b_i))
    dz = _abi.to_device(z, td) if z is not None else None
    dy = _abi.to_device(y, td) if (imp and dz is None) else None
    ds = _abi.to_device(samples, "int32")
    n = int(ds.shape[0])
    if n == 0:
        return float("nan")
    drp = _abi.to_device(row_ptrs, "int64") if dy is not None else None
    dci = _abi.to_device(col_inds, "int32") if dy is not None else None
    part = torch.empty(int(_abi.lib().mfsgd_rmse_partials(n)), dtype=torch.float64, device="cuda")
    sse = torch.zeros(1, dtype=torch.float64, device="cuda")
    fn = getattr(_abi.lib(), "mfsgd_hip_rmse_" + kind)
    _abi.check(fn(_abi.ptr(dP), _abi.ptr(dQ), _abi.ptr(dbu), _abi.ptr(dbi), ct(float(b)), _abi.ptr(dy), _abi.ptr(dz),
                  _abi.ptr(ds), n, ds.shape[1], _abi.ptr(drp), _abi.ptr(dci), int(dP.shape[1]), imp,
                  _abi.ptr(part), _abi.ptr(sse), _abi.stream_ptr()))
    return math.sqrt(float(sse.item()) / n)


def mse2(samples, row_ptrs, col_inds, P, Q, b_u, b_i, b, y, exact=False):
    """RMSE of the SVD++ predictor (numba_methods.py:85-103).  exact=False evaluates a random subsample of
    n//10 rows (50 000 from 50 000 rows up), drawn with np.random.choice like the reference."""
    if not exact:
        n = samples.shape[0]
        dropsize = int(n / 10) if n < 50000 else 50000
        idx = np.random.choice(n, size=dropsize, replace=False)
        if _abi.is_tensor(samples):
            samples = samples[_abi.torch_mod().as_tensor(idx, device=samples.device)]
        else:
            samples = samples[idx, :]
    return _rmse(samples, row_ptrs, col_inds, P, Q, b_u, b_i, b, y)


def mse(samples, row_ptrs, col_inds, P, Q, b_u, b_i, b, y, exact=False):
    """numba_methods.mse (numba_methods.py:60-82) cannot run as written (users[j+1], :76); it maps to the
    predictor of mse2 with the reference's subsample rule for this function (n//10, 10 000 from 10 000 up)."""
    if not exact:
        n = samples.shape[0]
        dropsize = int(n / 10) if n < 10000 else 10000
        idx = np.random.choice(n, size=dropsize, replace=False)
        if _abi.is_tensor(samples):
            samples = samples[_abi.torch_mod().as_tensor(idx, device=samples.device)]
        else:
            samples = samples[idx, :]
    return _rmse(samples, row_ptrs, col_inds, P, Q, b_u, b_i, b, y)


def rmse_exact(samples, row_ptrs, col_inds, P, Q, b_u, b_i, b, y=None, z=None):
    """Exact RMSE over all rows; y=None and z=None gives the biased-MF predictor."""
    return _rmse(samples, row_ptrs, col_inds, P, Q, b_u, b_i, b, y, z)

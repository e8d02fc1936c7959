"""ctypes binding of libmfsgd.so (include/mfsgd.h) and the tensor hand-off.

PyTorch-ROCm is used for device memory and streams only; every computation is a hand-written HIP
kernel behind the C-ABI.  There is NO CPU fallback: without the library or without a GPU every
entry point raises.
"""
import ctypes as C

MAX_HOT = 8          # MFSGD_MAX_HOT (include/mfsgd.h): hot items an item stripe may have
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MFSGD_LIBRARY: another build of the same C-ABI (profiles/ uses an instrumented one); never a CPU stand-in
LIB_PATH = os.environ.get("MFSGD_LIBRARY") or os.path.join(_HERE, "libmfsgd.so")
_lib = None

c_f32, c_f64, c_i32, c_i64, c_u64, c_vp = C.c_float, C.c_double, C.c_int32, C.c_int64, C.c_uint64, C.c_void_p


class GridInfo(C.Structure):
    _fields_ = [("n", c_i64), ("n_runs", c_i64), ("n_users", c_i64), ("n_items", c_i64),
                ("Gr", c_i32), ("Gc", c_i32), ("ubp", c_i32), ("ibp", c_i32),
                ("order", c_i32), ("user_unique", c_i32),
                ("max_block", c_i64), ("max_col_nnz", c_i64), ("max_row_nnz", c_i64),
                ("n_hot", c_i64), ("n_hot_stripes", c_i32), ("n_hot_items", c_i32)]


def _sig_sgd(ct):
    return [c_vp, c_vp, c_vp, c_vp, ct, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32,
            ct, ct, ct, c_i32, c_i32, c_vp]


def _sig_rmse(ct):
    return [c_vp, c_vp, c_vp, c_vp, ct, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]


def _sig_epoch(ct):
    return [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, ct, c_vp, ct, ct, ct, c_i32, c_i32, c_i32, c_i32, c_u64,
            c_i32, c_i32, c_vp, c_vp]


# every symbol include/mfsgd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mfsgd_last_error": (C.c_char_p, []),
    "mfsgd_version": (C.c_int, []),
    "mfsgd_device_check": (C.c_int, []),
    "mfsgd_hip_sgd_f32": (C.c_int, _sig_sgd(c_f32)),
    "mfsgd_hip_sgd_f64": (C.c_int, _sig_sgd(c_f64)),
    "mfsgd_rmse_partials": (c_i64, [c_i64]),
    "mfsgd_hip_rmse_f32": (C.c_int, _sig_rmse(c_f32)),
    "mfsgd_hip_rmse_f64": (C.c_int, _sig_rmse(c_f64)),
    "mfsgd_hip_rmse_noalloc_f32": (C.c_int, _sig_rmse(c_f32)[:14] + [c_vp, c_vp]),
    "mfsgd_hip_implicit_z_f32": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "mfsgd_hip_implicit_z_f64": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "mfsgd_hip_remap": (C.c_int, [c_vp, c_i64, c_i32, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "mfsgd_hip_filter_test": (C.c_int, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_i64, c_i64, c_vp,
                                        C.POINTER(c_i64), c_vp]),
    "mfsgd_hip_bucket": (C.c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "mfsgd_hip_csr_from_coo": (C.c_int, [c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp]),
    "mfsgd_hip_renumber_column": (C.c_int, [c_vp, c_i64, c_vp, C.POINTER(c_i64), c_vp]),
    "mfsgd_hip_topn_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp,
                                     c_vp]),
    "mfsgd_hip_als_step_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_f32, c_vp]),
    "mfsgd_hip_als_step_f64": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_f64, c_vp]),
    "mfsgd_grid_create": (C.c_int, [c_vp, c_i64, c_i32, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp, C.POINTER(c_vp)]),
    "mfsgd_grid_create_opt": (C.c_int, [c_vp, c_i64, c_i32, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, C.POINTER(c_vp)]),
    "mfsgd_grid_get_info": (C.c_int, [c_vp, C.POINTER(GridInfo)]),
    "mfsgd_grid_destroy": (None, [c_vp]),
    "mfsgd_grid_export": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mfsgd_lds_bytes": (c_i64, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    "mfsgd_grid_set_hot_buffers": (C.c_int, [c_vp, c_i32]),
    "mfsgd_grid_set_pick_width": (C.c_int, [c_vp, c_i32]),
    "mfsgd_grid_set_hot_mode": (C.c_int, [c_vp, c_i32]),
    "mfsgd_dsgd_plan": (C.c_int, [c_i64, c_i64, c_i32, c_i32, c_i32, C.POINTER(c_i64), C.POINTER(c_i64),
                                  C.POINTER(c_i32), C.POINTER(c_i32)]),
    "mfsgd_dsgd_schedule": (C.c_int, [c_i32, c_i32, c_i32, C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_i32)]),
    "mfsgd_dsgd_choose_grid": (C.c_int, [c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_i64, c_i32, C.POINTER(c_i32),
                                         C.POINTER(c_i32)]),
    "mfsgd_grid_col_hot": (C.c_int, [c_vp, c_vp, c_vp]),
    "mfsgd_grid_status": (C.c_int, [c_vp, c_vp]),
    "mfsgd_grid_col_times": (C.c_int, [c_vp, c_vp, c_vp]),
    "mfsgd_hip_epoch_f32": (C.c_int, _sig_epoch(c_f32)),
    "mfsgd_hip_epoch_f64": (C.c_int, _sig_epoch(c_f64)),
}


class MfsgdError(RuntimeError):
    pass


def lib():
    """Load libmfsgd.so (built in-tree by __graft_entry__.build() / make -C recommender_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MfsgdError(f"{LIB_PATH} is missing: build it with `make -C recommender_amd/csrc` "
                             "(there is no CPU fallback)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)       # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise MfsgdError(f"libmfsgd error {rc}: {lib().mfsgd_last_error().decode()}")


def torch_mod():
    import torch
    return torch


_device_ok = False


def require_gpu():
    """Raise unless a gfx950 GPU is usable.  Never falls back to the CPU."""
    global _device_ok
    if _device_ok:
        return
    torch = torch_mod()
    if not torch.cuda.is_available():
        raise MfsgdError("no ROCm GPU visible: recommender_amd runs on MI355X only (no CPU fallback)")
    n = lib().mfsgd_device_check()
    if n < 0:
        raise MfsgdError(lib().mfsgd_last_error().decode())
    _device_ok = True


def stream_ptr():
    torch = torch_mod()
    return c_vp(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return c_vp(0) if t is None else c_vp(t.data_ptr())


_NP2T = {np.dtype(np.float32): "float32", np.dtype(np.float64): "float64",
         np.dtype(np.int32): "int32", np.dtype(np.int64): "int64"}


def is_tensor(x):
    torch = torch_mod()
    return isinstance(x, torch.Tensor)


def to_device(a, dtype=None):
    """numpy array or torch tensor -> contiguous device tensor of `dtype` (torch dtype name)."""
    torch = torch_mod()
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        t = a if a.is_cuda else a.cuda()
    else:
        arr = np.ascontiguousarray(a)
        if arr.dtype not in _NP2T:
            arr = arr.astype(np.float64 if arr.dtype.kind == "f" else np.int64)
        t = torch.from_numpy(arr).cuda()
    if dtype is not None:
        t = t.to(getattr(torch, dtype))
    return t.contiguous()


def float_kind(*arrays):
    """'f32' if every given factor array is float32, else 'f64' (the reference computes in float64)."""
    torch = torch_mod()
    for a in arrays:
        if a is None:
            continue
        dt = a.dtype
        if dt in (np.float32, torch.float32) or str(dt) in ("float32", "torch.float32"):
            continue
        return "f64"
    return "f32"


def ctype_of(kind):
    return c_f32 if kind == "f32" else c_f64


def tdtype_of(kind):
    return "float32" if kind == "f32" else "float64"


def write_back(dst, t):
    """In-place semantics of the reference: copy the device result into the caller's array."""
    torch = torch_mod()
    if dst is None or t is None:
        return
    if isinstance(dst, torch.Tensor):
        if dst.data_ptr() != t.data_ptr():
            dst.copy_(t.to(dst.dtype))
    else:
        dst[...] = t.cpu().numpy().astype(dst.dtype, copy=False)

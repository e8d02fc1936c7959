"""The allocation-free entry point mfsgd_hip_rmse_noalloc_f32 (include/mfsgd.h) against mfsgd_hip_rmse_f32.

Collected last on purpose (file name): the round's end runs `pytest -x`.  Called through the C-ABI (ctypes), two
calls in a row, the second with a smaller input; the result must be bit-identical to the existing entry point's
(same kernels, same fixed reduction order) - the only difference is where the per-block partial sums live."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs(torch, rng, U, I, n, k):
    u = torch.from_numpy(rng.integers(0, U, size=n).astype(np.int32))
    i = torch.from_numpy(rng.integers(0, I, size=n).astype(np.int32))
    r = torch.from_numpy(rng.integers(1, 6, size=n).astype(np.int32))
    s = torch.stack([u, i, r], 1).contiguous().cuda()
    P = torch.from_numpy(rng.normal(scale=0.3, size=(U, k)).astype(np.float32)).cuda()
    Q = torch.from_numpy(rng.normal(scale=0.3, size=(I, k)).astype(np.float32)).cuda()
    bu = torch.from_numpy(rng.normal(scale=0.1, size=U).astype(np.float32)).cuda()
    bi = torch.from_numpy(rng.normal(scale=0.1, size=I).astype(np.float32)).cuda()
    return s, P, Q, bu, bi


def test_rmse_noalloc_matches_rmse_two_calls_second_smaller():
    import torch
    from recommender_amd import _abi
    _abi.require_gpu()
    lib = _abi.lib()
    rng = np.random.default_rng(2)
    parts = torch.empty(int(lib.mfsgd_rmse_partials(1)), dtype=torch.float64, device="cuda")
    # warm both paths once (code-object load, symbol lookup), then watch the device's free memory
    first = True
    for (U, I, n, k) in ((3000, 2000, 200_000, 128), (300, 200, 1_500, 128), (50, 40, 3, 37)):
        s, P, Q, bu, bi = _inputs(torch, rng, U, I, n, k)
        z = torch.from_numpy(rng.normal(scale=0.05, size=(U, k)).astype(np.float32)).cuda()
        ref = torch.zeros(1, dtype=torch.float64, device="cuda")
        got = torch.full((1,), -1.0, dtype=torch.float64, device="cuda")
        for zz, imp in ((None, 0), (z, 1)):
            _abi.check(lib.mfsgd_hip_rmse_f32(_abi.ptr(P), _abi.ptr(Q), _abi.ptr(bu), _abi.ptr(bi), 3.25, None, _abi.ptr(zz),
                                              _abi.ptr(s), n, 3, None, None, k, imp, _abi.ptr(parts), _abi.ptr(ref), None))
            torch.cuda.synchronize()
            if first:
                _abi.check(lib.mfsgd_hip_rmse_noalloc_f32(_abi.ptr(P), _abi.ptr(Q), _abi.ptr(bu), _abi.ptr(bi), 3.25, None,
                                                          _abi.ptr(zz), _abi.ptr(s), n, 3, None, None, k, imp, _abi.ptr(got),
                                                          None))
                torch.cuda.synchronize()
                first = False
            free0 = torch.cuda.mem_get_info()[0]
            _abi.check(lib.mfsgd_hip_rmse_noalloc_f32(_abi.ptr(P), _abi.ptr(Q), _abi.ptr(bu), _abi.ptr(bi), 3.25, None,
                                                      _abi.ptr(zz), _abi.ptr(s), n, 3, None, None, k, imp, _abi.ptr(got), None))
            torch.cuda.synchronize()
            assert torch.cuda.mem_get_info()[0] == free0          # nothing was allocated (or freed) by the call
            assert float(got.item()) == float(ref.item()), (n, imp)   # bit-identical
            assert np.isfinite(float(got.item())) and float(got.item()) > 0


def test_rmse_noalloc_limits():
    import torch
    from recommender_amd import _abi
    _abi.require_gpu()
    lib = _abi.lib()
    rng = np.random.default_rng(3)
    s, P, Q, bu, bi = _inputs(torch, rng, 10, 10, 20, 8)
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    rc = lib.mfsgd_hip_rmse_noalloc_f32(_abi.ptr(P), _abi.ptr(Q), _abi.ptr(bu), _abi.ptr(bi), 0.0, None, None, _abi.ptr(s),
                                        1 << 31, 3, None, None, 8, 0, _abi.ptr(out), None)
    assert rc != 0 and b"limit" in lib.mfsgd_last_error()
    rc = lib.mfsgd_hip_rmse_noalloc_f32(_abi.ptr(P), _abi.ptr(Q), _abi.ptr(bu), _abi.ptr(bi), 0.0, None, None, _abi.ptr(s),
                                        20, 3, None, None, 5000, 0, _abi.ptr(out), None)
    assert rc != 0 and b"limit" in lib.mfsgd_last_error()

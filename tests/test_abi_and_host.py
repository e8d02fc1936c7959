"""CPU-only checks: the C-ABI library loads and exports every symbol include/mfsgd.h declares, the host
mirror of the scheduler reproduces the reference's traces, and the product never reaches for the oracle."""
import os
import random
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    from recommender_amd import _abi
    hdr = open(os.path.join(ROOT, "include", "mfsgd.h")).read()
    declared = set(re.findall(r"\b(mfsgd_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    assert declared == set(_abi.SIGNATURES), declared ^ set(_abi.SIGNATURES)
    lib = _abi.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mfsgd_version() >= 100
    assert lib.mfsgd_rmse_partials(10) > 0


def test_lds_bytes_mirror_matches_library():
    """The grid chooser's LDS arithmetic (Python) is the library's (a pure host function: no device touched)."""
    from recommender_amd import _abi
    from recommender_amd.SGDRecommender import lds_bytes
    lib = _abi.lib()
    for ibp, Gr, k, esz, grp in [(35, 640, 128, 4, 4), (243, 320, 128, 4, 2), (35, 640, 128, 4, 0), (119, 640, 256, 4, 1),
                                 (50, 16, 200, 4, 11), (7, 3, 5, 8, 2), (1, 1, 1, 4, 1), (100, 1000, 64, 8, 0)]:
        assert lib.mfsgd_lds_bytes(ibp, Gr, k, esz, grp) == lds_bytes(ibp, Gr, k, esz, grp), (ibp, Gr, k, esz, grp)
    assert lib.mfsgd_lds_bytes(0, 1, 1, 4, 1) == -1 and lib.mfsgd_lds_bytes(1, 1, 1, 2, 1) == -1 and lib.mfsgd_lds_bytes(1, 1, 1, 4, -1) == -1


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import recommender_amd as ra
    from recommender_amd._abi import MfsgdError
    a = np.zeros((2, 4))
    with pytest.raises(MfsgdError):
        ra.sgd(a, a.copy(), np.zeros(2), np.zeros(2), 0.0, a.copy(), np.zeros((1, 3), np.int32),
               np.zeros(3, np.int64), np.zeros(1, np.int32), 0.1, 0.1, 0.1)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "recommender_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower().replace("test oracle", ""), f


@pytest.mark.parametrize("W,it,seed", [(4, 2, 5), (9, 3, 6)])
def test_host_scheduler_reproduces_reference_traces(W, it, seed):
    import recommender_amd as ra
    g = np.load(os.path.join(G, "scheduler.npz"))
    random.seed(seed)
    s = ra.BlockScheduler(W, it)
    seq, blk = [], s.get_next()
    while blk is not None:
        seq.append(blk)
        blk = s.get_next(completed=blk)
    np.testing.assert_array_equal(np.array(seq), g[f"s1_single_w{W}_i{it}_seed{seed}"])
    np.testing.assert_array_equal(np.array(s.get_update_counter()), g[f"s1_single_w{W}_i{it}_seed{seed}_counter"])
    assert s.check_completion() and s.get_test() == it * W * W
    # scripted 3-consumer interleaving
    random.seed(seed)
    s = ra.BlockScheduler(W, it)
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
                assert all(nxt[0] != held[x][0] and nxt[1] != held[x][1] for x in range(3) if x != w and alive[x])
                held[w] = nxt
                log.append((w, *nxt))
    np.testing.assert_array_equal(np.array(log), g[f"s1_multi3_w{W}_i{it}_seed{seed}"])
    # deterministic tie-break = the multiprocessing prototype's
    s = ra.BlockScheduler(W, it, deterministic=True)
    seq, blk = [], s.get_next()
    while blk is not None:
        seq.append(blk)
        blk = s.get_next(completed=blk)
    np.testing.assert_array_equal(np.array(seq), g[f"s2_single_w{W}_i{it}"])


def test_scheduler_desync_guard():
    import recommender_amd as ra
    s = ra.BlockScheduler(3, 1)
    with pytest.raises(Exception, match="Desync"):
        s.completed_chunk((0, 0))           # never handed out


def test_subsample_struct():
    import recommender_amd as ra
    ss = ra.SubSample((1, 2), (10, 20), "P", "Q", "bu", "bi", 3.5, False, 0.1, 0.2, 0.3)
    assert ss.block_pos == (1, 2) and ss.idx_offset == (10, 20) and ss.b == 3.5 and ss.beta2 == 0.3

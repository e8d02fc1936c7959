"""The multi-GPU DSGD ring (recommender_amd/dsgd.py) on CPU: world_size 2 and 3 under gloo, with the oracle's
block kernel standing in for the HIP epoch kernel.  Checks that the ring + ownership plan is equivalent
to a single worker replaying the super-blocks sub-epoch by sub-epoch, bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from recommender_amd.dsgd import DSGDPlan, choose_multi_grid, gather_items, ring_epoch

HP = (0.006, 0.05, 0.015)


def _data(U=120, I=90, n=4000, k=8, seed=3):
    rng = np.random.default_rng(seed)
    key = np.unique(rng.integers(0, U * I, size=n))
    s = np.column_stack([key // I, key % I, rng.integers(1, 6, size=len(key))]).astype(np.int32)
    P = rng.normal(scale=1 / k, size=(U, k))
    Q = rng.normal(scale=1 / k, size=(I, k))
    return s, P, Q, float(s[:, 2].mean())


def _superblock(s, plan, g, q):
    u0, u1 = plan.user_range(g)
    i0, i1 = plan.item_range(q)
    m = (s[:, 0] >= u0) & (s[:, 0] < u1) & (s[:, 1] >= i0) & (s[:, 1] < i1)
    return s[m]


def _worker(rank, world, port, epochs, out):
    from oracle import oracle as orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, P, Q, b = _data()
    U, I = P.shape[0], Q.shape[0]
    plan = DSGDPlan(U, I, world, ubp=7, ibp=5)
    bu, bi = np.zeros(U), np.zeros(I)
    Qt, bit = torch.from_numpy(Q), torch.from_numpy(bi)          # share memory with the numpy arrays

    def compute(q):
        orc.sgd(P, Q, bu, bi, b, None, _superblock(s, plan, rank, q), None, None, *HP, implicit=0)

    for _ in range(epochs):
        ring_epoch(plan, rank, Qt, bit, compute, dist)
    gather_items(plan, rank, Qt, bit, dist)
    u0, u1 = plan.user_range(rank)
    Pt, but = torch.from_numpy(P), torch.from_numpy(bu)
    for g in range(world):                                       # collect the user rows for the check
        a0, a1 = plan.user_range(g)
        dist.broadcast(Pt[a0:a1], src=g)
        dist.broadcast(but[a0:a1], src=g)
    if rank == 0:
        np.savez(out, P=P, Q=Q, bu=bu, bi=bi)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_ring_equals_serial_replay(world, tmp_path):
    from oracle import oracle as orc
    out = str(tmp_path / "res.npz")
    epochs = 2
    mp.spawn(_worker, args=(world, _free_port(), epochs, out), nprocs=world, join=True)
    got = np.load(out)
    s, P, Q, b = _data()
    plan = DSGDPlan(P.shape[0], Q.shape[0], world, ubp=7, ibp=5)
    bu, bi = np.zeros(P.shape[0]), np.zeros(Q.shape[0])
    for _ in range(epochs):
        for g, q in plan.serial_order():
            orc.sgd(P, Q, bu, bi, b, None, _superblock(s, plan, g, q), None, None, *HP, implicit=0)
    for name, ref in (("P", P), ("Q", Q), ("bu", bu), ("bi", bi)):
        np.testing.assert_array_equal(got[name], ref, err_msg=name)


def test_plan_covers_everything_once():
    for U, I, W, ubp, ibp in ((120, 90, 2, 7, 5), (1000, 77, 4, 13, 3), (50, 50, 8, 2, 2), (10, 3, 4, 5, 1)):
        plan = DSGDPlan(U, I, W, ubp, ibp)
        assert plan.user_edges[0] == 0 and plan.user_edges[-1] == U and plan.item_edges[-1] == I
        assert all(a <= b for a, b in zip(plan.user_edges, plan.user_edges[1:]))
        assert all(e % ubp == 0 or e == U for e in plan.user_edges)
        assert all(e % ibp == 0 or e == I for e in plan.item_edges)
        for s in range(W):                       # in one sub-epoch every shard is held by exactly one rank
            assert sorted(plan.shard_at(g, s) for g in range(W)) == list(range(W))
        for g in range(W):                       # what a rank receives is what its neighbour just held
            for s in range(W):
                assert plan.shard_at(g, s + 1) == plan.shard_at(plan.recv_from(g), s)
                assert plan.shard_at(plan.send_to(g), s + 1) == plan.shard_at(g, s)
        assert sorted(plan.serial_order()) == [(g, q) for g in range(W) for q in range(W)]


def test_choose_multi_grid_shapes():
    for world in (1, 2, 4, 8):
        ubp, ibp = choose_multi_grid(480189, 17770, 90_000_000, world, 128)
        plan = DSGDPlan(480189, 17770, world, ubp, ibp)
        assert ibp * (128 + 1) * 4 + ibp * 4 <= 150 * 1024
        per_block = 90_000_000 / (plan.Gr * plan.Gc)
        assert per_block >= 30, (world, plan.Gr, plan.Gc)


def test_dsgd_plan_c_abi_matches_the_python_plan():
    """mfsgd_dsgd_plan / _schedule / _choose_grid (include/mfsgd.h; host-only calls) against DSGDPlan / choose_multi_grid."""
    import ctypes as C
    from recommender_amd import _abi
    lib = _abi.lib()
    shapes = [(480189, 17770, 90_000_000, 128, 139853), (162541, 59047, 22_500_000, 128, 33797),
              (1000990, 624961, 225_000_000, 256, 0), (6040, 3706, 900_000, 64, 3085), (943, 1682, 90_000, 32, 500),
              (50, 7, 300, 8, 0)]
    for U, I, n, k, top in shapes:
        for world in (1, 2, 3, 4, 8):
            ubp, ibp = choose_multi_grid(U, I, n, world, k, max_item_degree=top or None)
            cu, ci = C.c_int32(), C.c_int32()
            _abi.check(lib.mfsgd_dsgd_choose_grid(U, I, n, world, k, 4, top, 256, C.byref(cu), C.byref(ci)))
            assert (cu.value, ci.value) == (ubp, ibp), (U, world, (cu.value, ci.value), (ubp, ibp))
            plan = DSGDPlan(U, I, world, ubp, ibp)
            ue, ie = (C.c_int64 * (world + 1))(), (C.c_int64 * (world + 1))()
            gr, gc = C.c_int32(), C.c_int32()
            _abi.check(lib.mfsgd_dsgd_plan(U, I, world, ubp, ibp, ue, ie, C.byref(gr), C.byref(gc)))
            assert list(ue) == plan.user_edges and list(ie) == plan.item_edges and (gr.value, gc.value) == (plan.Gr, plan.Gc)
            for rank in range(world):
                for step in range(2 * world + 1):
                    q, to, frm = C.c_int32(), C.c_int32(), C.c_int32()
                    _abi.check(lib.mfsgd_dsgd_schedule(world, rank, step, C.byref(q), C.byref(to), C.byref(frm)))
                    assert (q.value, to.value, frm.value) == (plan.shard_at(rank, step), plan.send_to(rank), plan.recv_from(rank))
    assert lib.mfsgd_dsgd_schedule(4, 4, 0, None, None, None) != 0          # rank outside the world
    assert lib.mfsgd_dsgd_plan(10, 10, 0, 1, 1, None, None, None, None) != 0

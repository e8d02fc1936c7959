"""The multi-GPU DSGD ring with the HIP epoch kernel: two (and three) ranks as separate processes sharing the ONE GPU of
the test box - fresh `spawn` children, torch.distributed over gloo (the shard exchange is staged through host buffers;
with RCCL the same `ring_epoch` sends the device rows in place).  Every rank is a `RankWorker`: it holds only its own
user super-stripe (ratings, block grids per item shard, P rows).  The free-block kernel logs the order in which it issued
the blocks of every sub-epoch; the parent replays exactly that - sub-epoch by sub-epoch, rank by rank
(`plan.serial_order()`), block by block in logged order, ratings in the grid's stored order - on the CPU oracle's plain
sequential kernel and compares every factor (1e-4 relative, north_star's bar).  Also: RMSE of the job, computed by the
ranks with one all-reduce, equals the oracle's on the gathered factors."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HP = (0.006, 0.05, 0.015)
K = 128
SEED = 5


def _data(U=4000, I=1500, n=300_000, seed=11):
    rng = np.random.default_rng(seed)
    u = (U * rng.random(n) ** 1.5).astype(np.int64)
    i = (I * rng.random(n) ** 2.0).astype(np.int64)
    key = np.unique(u * I + i)
    s = np.column_stack([key // I, key % I, rng.integers(1, 6, size=len(key))]).astype(np.int32)
    test = rng.random(len(s)) < 0.1
    return s[~test], s[test], U, I


def _worker(rank, world, port, epochs, ubp, ibp, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from recommender_amd import _abi
    from recommender_amd.dsgd import DSGDPlan, RankWorker
    torch.cuda.set_device(0)
    _abi.require_gpu()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    train, test, U, I = _data()
    plan = DSGDPlan(U, I, world, ubp, ibp)
    u0, u1 = plan.user_range(rank)
    mine = lambda a: torch.from_numpy(a[(a[:, 0] >= u0) & (a[:, 0] < u1)]).cuda()
    w = RankWorker(plan, rank, mine(train), K, float(train[:, 2].mean()), HP, seed=SEED, test_local=mine(test),
                   log_blocks=True)
    init = {"P": w.P.cpu().numpy().copy(), "Q": w.Q.cpu().numpy().copy()}
    for _ in range(epochs):
        w.epoch(dist)
    w.status()
    rm = w.errors(dist)                      # gathers every shard of Q / b_i onto every rank
    np.savez(f"{out}.{rank}.npz", P0=init["P"], Q0=init["Q"], P=w.P.cpu().numpy(), b_u=w.b_u.cpu().numpy(),
             Q=w.Q.cpu().numpy(), b_i=w.b_i.cpu().numpy(), rmse=np.array(rm),
             shards=np.array([q for q, _ in w.logs]), logs=np.stack([l.cpu().numpy() for _, l in w.logs]),
             col_hot=np.stack([g.col_hot() for g in w.grids]), n_hot=np.array([int(g.info.n_hot) for g in w.grids]),
             grid=np.array([w.grids[0].info.Gr, w.grids[0].info.Gc]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("world,ubp,ibp", [(2, 125, 50), (3, 100, 40)])
def test_ring_of_hip_ranks_replays_on_the_oracle(world, ubp, ibp, tmp_path):
    import torch.multiprocessing as mp
    from _order import stored_order
    from oracle import oracle as orc
    from recommender_amd.dsgd import DSGDPlan
    epochs = 2
    out = str(tmp_path / "rank")
    mp.spawn(_worker, args=(world, _free_port(), epochs, ubp, ibp, out), nprocs=world, join=True)
    res = [np.load(f"{out}.{r}.npz") for r in range(world)]
    train, test, U, I = _data()
    plan = DSGDPlan(U, I, world, ubp, ibp)
    b = float(train[:, 2].mean())
    # the starting point: every rank drew its own users' rows, all drew the same item rows
    P = np.concatenate([res[r]["P0"][:plan.user_range(r)[1] - plan.user_range(r)[0]] for r in range(world)]).copy()
    Q = res[0]["Q0"].copy()
    for r in range(1, world):
        np.testing.assert_array_equal(res[r]["Q0"], Q)
    bu, bi = np.zeros(U, np.float32), np.zeros(I, np.float32)
    Gc = plan.Gc
    # per rank and shard: the super-block's ratings in the grid's stored order, with LOCAL user ids as the rank sees them
    blocks = {}
    for g in range(world):
        u0, u1 = plan.user_range(g)
        loc = train[(train[:, 0] >= u0) & (train[:, 0] < u1)].copy()
        loc[:, 0] -= u0
        for q in range(world):
            i0, i1 = plan.item_range(q)
            sub = loc[(loc[:, 1] >= i0) & (loc[:, 1] < i1)]
            perm, key, hot = stored_order(sub, ubp, ibp, Gc, res[g]["col_hot"][q])
            assert int(hot.sum()) == int(res[g]["n_hot"][q])
            ss, ks = sub[perm], key[perm]
            Gr_l = int(res[g]["grid"][0])
            blocks[g, q] = (ss, np.searchsorted(ks, np.arange(Gr_l * Gc + 1)))
    calls = [0] * world
    for e in range(epochs):
        for s_ in range(world):
            for g in range(world):
                q = plan.shard_at(g, s_)
                assert int(res[g]["shards"][calls[g]]) == q
                log = res[g]["logs"][calls[g]]
                calls[g] += 1
                ss, ptr = blocks[g, q]
                u0, u1 = plan.user_range(g)
                issued = [(r, c) for r, c in log if r >= 0]
                nonempty = {(x // Gc, x % Gc) for x in np.flatnonzero(np.diff(ptr) > 0)}
                assert len(issued) == len(set(issued)) and set(issued) == nonempty
                Pl, bul = P[u0:u1], bu[u0:u1]           # views: the rank's own rows
                for r, c in issued:
                    orc.sgd(Pl, Q, bul, bi, b, None, ss[ptr[r * Gc + c]:ptr[r * Gc + c + 1]], None, None, *HP, implicit=0)
    rel = lambda a, ref: float(np.abs(a.astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))
    for g in range(world):
        u0, u1 = plan.user_range(g)
        assert rel(res[g]["P"][:u1 - u0], P[u0:u1].astype(np.float64)) <= 1e-4, g
        assert rel(res[g]["b_u"][:u1 - u0], bu[u0:u1].astype(np.float64)) <= 1e-4, g
        assert rel(res[g]["Q"], Q.astype(np.float64)) <= 1e-4, g          # gathered: every rank holds every shard
        assert rel(res[g]["b_i"], bi.astype(np.float64)) <= 1e-4, g
    tr = orc.rmse(train, None, None, P, Q, bu, bi, b, None, implicit=0)
    te = orc.rmse(test, None, None, P, Q, bu, bi, b, None, implicit=0)
    for g in range(world):
        assert abs(res[g]["rmse"][0] - tr) < 1e-4 and abs(res[g]["rmse"][1] - te) < 1e-4
    assert tr < 1.41      # uniform random ratings 1..5: sqrt(2) before any training

"""Multi-GPU DSGD on top of the single-GPU FPSGD epoch kernel: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

No counterpart in the reference (it has no multi-device path); the block-exclusivity
rule it implements is the same FPSGD invariant (RayBlockScheduler.py:50-62) one level up:

  * users are cut into `world` contiguous super-stripes, items into `world` contiguous shards, both on
    stripe boundaries of the block grid;
  * rank g owns the P / b_u rows of user super-stripe g for good and only ever sees the ratings of those
    users, bucketed by item shard (one block grid per shard);
  * an epoch is `world` sub-epochs; in sub-epoch s rank g runs the free-block epoch kernel on the
    super-block (g, (g+s) mod world) - no two ranks touch the same user or item rows - and then the ring
    turns: every rank sends the Q / b_i rows of the shard it has just updated to rank g-1 and receives the
    next shard from rank g+1 (paired send/recv, in place, no host staging, no all-reduce);
  * held-out RMSE: shard owners broadcast their rows, every rank sums squared errors over its own users,
    one all-reduce of four numbers.

The plan (who holds what when) is pure Python and runs anywhere; the compute step is a callable, so the
CPU tests drive the same ring with a CPU block kernel of their own under gloo.
"""
import os
import sys
import time

import numpy as np


class DSGDPlan:
    def __init__(self, n_users, n_items, world, ubp, ibp):
        self.n_users, self.n_items, self.world, self.ubp, self.ibp = n_users, n_items, world, ubp, ibp
        self.Gr, self.Gc = -(-n_users // ubp), -(-n_items // ibp)
        rows = -(-self.Gr // world)          # row stripes per rank
        cols = -(-self.Gc // world)          # column stripes per shard
        self.user_edges = [min(n_users, g * rows * ubp) for g in range(world + 1)]
        self.item_edges = [min(n_items, q * cols * ibp) for q in range(world + 1)]
        self.user_edges[-1], self.item_edges[-1] = n_users, n_items

    def shard_at(self, rank, step):
        """Item shard that `rank` holds (and updates) in sub-epoch `step`."""
        return (rank + step) % self.world

    def send_to(self, rank):
        return (rank - 1) % self.world

    def recv_from(self, rank):
        return (rank + 1) % self.world

    def user_range(self, rank):
        return self.user_edges[rank], self.user_edges[rank + 1]

    def item_range(self, shard):
        return self.item_edges[shard], self.item_edges[shard + 1]

    def serial_order(self):
        """The (rank, shard) super-blocks of one epoch in an order a single worker could replay:
        sub-epoch by sub-epoch (the blocks inside one sub-epoch share no rows and no columns)."""
        return [(g, self.shard_at(g, s)) for s in range(self.world) for g in range(self.world)]


def ring_epoch(plan, rank, Q, b_i, compute, dist=None):
    """One DSGD epoch on this rank: world x (compute on the held shard; pass it on, take the next one).
    `compute(shard)` updates P/b_u of this rank's users and the shard's rows of Q/b_i in place."""
    W = plan.world
    for s in range(W):
        q = plan.shard_at(rank, s)
        compute(q)
        if W == 1:
            continue
        nq = plan.shard_at(rank, s + 1)
        a0, a1 = plan.item_range(q)
        r0, r1 = plan.item_range(nq)
        staged = getattr(Q, "is_cuda", False) and dist.get_backend() == "gloo"
        if staged:   # rehearsal on one GPU box: gloo moves host buffers only (RCCL sends the device rows in place)
            sq, sb = Q[a0:a1].cpu(), b_i[a0:a1].cpu()
            rq, rb = Q[r0:r1].cpu(), b_i[r0:r1].cpu()
        else:
            sq, sb, rq, rb = Q[a0:a1], b_i[a0:a1], Q[r0:r1], b_i[r0:r1]
        ops = [dist.P2POp(dist.isend, sq, plan.send_to(rank)),
               dist.P2POp(dist.isend, sb, plan.send_to(rank)),
               dist.P2POp(dist.irecv, rq, plan.recv_from(rank)),
               dist.P2POp(dist.irecv, rb, plan.recv_from(rank))]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if staged:
            Q[r0:r1].copy_(rq)
            b_i[r0:r1].copy_(rb)


def gather_items(plan, rank, Q, b_i, dist):
    """After a whole epoch shard q is current on rank q: everyone fetches every shard (evaluation only)."""
    staged = getattr(Q, "is_cuda", False) and dist.get_backend() == "gloo"   # rehearsal on one GPU box (see ring_epoch)
    for q in range(plan.world):
        a0, a1 = plan.item_range(q)
        if a1 > a0:
            if staged:
                hq, hb = Q[a0:a1].cpu(), b_i[a0:a1].cpu()
                dist.broadcast(hq, src=q)
                dist.broadcast(hb, src=q)
                if rank != q:
                    Q[a0:a1].copy_(hq)
                    b_i[a0:a1].copy_(hb)
            else:
                dist.broadcast(Q[a0:a1], src=q)
                dist.broadcast(b_i[a0:a1], src=q)


# what profiles/dsgd_rank_sim.py measured for ONE rank's sub-epoch kernel on an MI355X (netflix shape, k = 128)
SUB_EPOCH_BLOCK_US = 15.0       # a block's fixed cost on the stripe's critical path: transition + pipeline fill
SUB_EPOCH_RATE = 15.0           # ratings per microsecond and workgroup when nothing waits
CHAIN_LINK_US = (0.09, 0.22)    # one link of an item's chain: hot-item walker, item tickets (profiles/dev_walker.py, r2_chain_link.txt)


def choose_multi_grid(n_users, n_items, n_train, world, k, esz=4, max_item_degree=None, n_cus=256):
    """Block grid for `world` ranks.  A rank's sub-epoch walks `rows x cols` blocks, one workgroup per column stripe,
    and a stripe's tenure is bound by latency, not by throughput: the most rated item's chain (its ratings by this
    rank's users, serial) plus `rows` block transitions; the ratings themselves only matter when the stripes are few:
        t(cols) = max(chain + rows * 15 us, ratings per sub-epoch / (cols * 15 per us)),  rows = cols + cols / 8
    (fits the measured kernels to 5-10 %: netflix shape, world 2 / 4 / 8: 127, 127 and 64 stripes per shard take 11.4,
    6.2 and 3.0 ms per sub-epoch; 254, 159 and 80 stripes - round 2's earlier rule, "fill every CU" - 13.5, 7.2, 3.5).
    So: the FEWEST stripes per shard whose throughput term does not exceed the latency term, never more than CUs."""
    ibp_max = max(1, (120 * 1024) // ((k + 1) * esz + 4))
    per = n_train / float(world * world)
    if max_item_degree is None:
        max_item_degree = 0.0016 * n_train          # netflix-like head
    from .SGDRecommender import lds_bytes
    best = None
    cmin = max(1, -(-n_items // (world * ibp_max)))
    for cols in range(max(cmin, 8), min(n_cus, max(1, n_items // world)) + 1, 4):
        rows = min(cols + max(1, cols // 8), max(1, n_users // world))
        ibp = -(-n_items // (cols * world))
        piped = k in (64, 128, 256) and esz == 4 and lds_bytes(ibp, rows, k, 4) <= 160 * 1024
        chain = max_item_degree / world * CHAIN_LINK_US[0 if piped else 1]
        t = max(chain + rows * SUB_EPOCH_BLOCK_US, per / (cols * SUB_EPOCH_RATE))
        if best is None or t < best[0] - 1e-9:
            best = (t, rows, cols)
    if best is None:   # stripes limited by the LDS, more of them than CUs (Yahoo shape): a workgroup takes several in turn
        best = (0.0, min(n_cus + n_cus // 8, max(1, n_users // world)), cmin)
    _, rows, cols = best
    ibp = max(1, min(ibp_max, -(-n_items // (cols * world))))
    ubp = max(1, -(-n_users // (rows * world)))
    return ubp, ibp


def pipeline_fits(n_items, ibp, rows, k, hot_buffers=8):
    """Hot segments in a rank's shard grids: only where the free-block launch will walk them with the hot-item walker
    (SGDRecommender.ExplicitMF._walker_runs: fp32 here, k = 64 / 128 / 256, the stripe within a CU's LDS; a shard never
    has more non-empty stripes than the chooser's 256, so one workgroup runs per CU)."""
    from .SGDRecommender import lds_bytes
    return k in (64, 128, 256) and hot_buffers > 0 and lds_bytes(ibp, rows, k, 4) <= 160 * 1024


class RankWorker:
    """One rank of the DSGD ring on its GPU.  It holds ONLY its own user super-stripe: the stripe's ratings (user
    ids made local), one block grid per item shard built from them, and P / b_u rows for those users alone.  Q and
    b_i are allocated whole (they are small next to P and the ring moves shards in place); only the shard the rank
    currently holds is current."""

    def __init__(self, plan, rank, train_local, k, b, hp, seed=0, hot_buffers=8, n_workgroups=0, waves_per_wg=0,
                 test_local=None, log_blocks=False):
        import torch
        from . import _abi
        from .SGDRecommender import Grid
        _abi.require_gpu()
        self.plan, self.rank, self.k, self.b, self.hp = plan, rank, k, float(b), hp
        self.seed, self.n_workgroups, self.waves_per_wg = seed, n_workgroups, waves_per_wg
        u0, u1 = plan.user_range(rank)
        self.u0, self.nu = u0, u1 - u0
        loc = train_local.clone()
        if loc.shape[0]:
            assert int(loc[:, 0].min()) >= u0 and int(loc[:, 0].max()) < u1, "train_local holds another rank's users"
        loc[:, 0] -= u0
        self.train = loc
        self.test = None
        if test_local is not None:
            self.test = test_local.clone()
            self.test[:, 0] -= u0
        rows = max(1, -(-self.nu // plan.ubp))
        hot = (1, 64) if pipeline_fits(plan.n_items, plan.ibp, rows, k, hot_buffers) else 0
        self.grids = []
        for q in range(plan.world):
            i0, i1 = plan.item_range(q)
            sel = (loc[:, 1] >= i0) & (loc[:, 1] < i1)
            g = Grid(loc[sel].contiguous(), max(1, self.nu), plan.n_items, plan.ubp, plan.ibp, order=1, hot_segments=hot)
            if hot_buffers != 8:
                g.set_hot_buffers(hot_buffers)
            self.grids.append(g)
        gen = torch.Generator(device="cuda")
        gen.manual_seed(int(seed) + 7919 * (rank + 1))          # this rank's users
        self.P = torch.randn(max(1, self.nu), k, device="cuda", generator=gen) / k
        self.b_u = torch.zeros(max(1, self.nu), device="cuda")
        gen.manual_seed(int(seed))                              # the items: the same draw on every rank
        self.Q = torch.randn(plan.n_items, k, device="cuda", generator=gen) / k
        self.b_i = torch.zeros(plan.n_items, device="cuda")
        self.logs = [] if log_blocks else None                  # (shard, int32 [Gr*Gc][2] device tensor) per compute()
        self._calls = 0

    def compute(self, q):
        """One sub-epoch: the free-block epoch kernel on super-block (rank, q)."""
        import torch
        from . import _abi
        g = self.grids[q]
        log = None
        if self.logs is not None:
            log = torch.full((g.info.Gr * g.info.Gc, 2), -1, dtype=torch.int32, device="cuda")
            self.logs.append((q, log))
        a, b1, b2 = self.hp
        _abi.check(_abi.lib().mfsgd_hip_epoch_f32(g.handle, self.k, _abi.ptr(self.P), _abi.ptr(self.Q), _abi.ptr(self.b_u),
                                                  _abi.ptr(self.b_i), self.b, None, a, b1, b2, 0, 0, 1, 1,
                                                  int(self.seed) + self._calls, self.n_workgroups, self.waves_per_wg,
                                                  _abi.ptr(log), _abi.stream_ptr()))
        self._calls += 1

    def epoch(self, dist, check=True):
        """One epoch of the ring.  check: ask every shard grid whether a bounded wait ran out (mfsgd_grid_status) before
        the epoch's results are used - a kernel that gave up leaves a shard half updated, and the ring would carry it on."""
        ring_epoch(self.plan, self.rank, self.Q, self.b_i, self.compute, dist)
        if check:
            self.status()

    def status(self):
        for g in self.grids:
            g.status()

    def errors(self, dist):
        """(train RMSE, held-out RMSE) of the whole job: shard owners broadcast, local sums, one all-reduce."""
        import torch
        from .numba_methods import rmse_exact
        gather_items(self.plan, self.rank, self.Q, self.b_i, dist)
        acc = torch.zeros(4, dtype=torch.float64, device="cuda")
        for j, s in enumerate((self.train, self.test)):
            if s is not None and s.shape[0]:
                r = rmse_exact(s, None, None, self.P, self.Q, self.b_u, self.b_i, self.b)
                acc[2 * j] = r * r * s.shape[0]
                acc[2 * j + 1] = s.shape[0]
        staged = dist.get_backend() == "gloo"
        red = acc.cpu() if staged else acc
        dist.all_reduce(red)
        return float((red[0] / red[1]).sqrt()), float((red[2] / red[3].clamp(min=1)).sqrt())


def bench_main(args, rank, world, data, bytes_per_update, hbm_peak, hp):
    """bench.py's multi-GPU leg (strong scaling: the same data set on 1..N GPUs)."""
    import json
    import torch
    import torch.distributed as dist
    from . import _abi
    from .SGDRecommender import balanced_permutation, interleave_stripes, item_costs

    train, test, U, I, k = data
    n_train_total = int(train.shape[0])
    verbose = bool(os.environ.get("MFSGD_BENCH_VERBOSE"))

    def note(msg):      # progress on stderr (rehearsals: a silent minute looks like a hang)
        if verbose:
            print(f"[dsgd bench rank {rank}] {msg} (+{time.perf_counter() - t_start:.1f} s)", file=sys.stderr, flush=True)
    t_start = time.perf_counter()
    if args.grid:
        Gr, Gc = (int(x) for x in args.grid.split(","))
        ubp, ibp = -(-U // Gr), -(-I // Gc)
    else:
        ubp, ibp = choose_multi_grid(U, I, n_train_total, world, k,
                                     max_item_degree=int(torch.bincount(train[:, 1].long(), minlength=I).max()))
    t0 = time.perf_counter()
    # The bench generates the SAME synthetic data set on every rank (so that 1, 2, 4 and 8 GPUs train on identical
    # ratings) and computes the same load-balancing permutation everywhere; from here on a rank keeps only its own
    # users' ratings, P rows and biases (RankWorker).  A real job hands each rank its stripe and nothing else.
    du = torch.bincount(train[:, 0].long(), minlength=U).cpu().numpy()
    di = torch.bincount(train[:, 1].long(), minlength=I).cpu().numpy()
    note(f"grid: {ubp} users x {ibp} items per block")
    plan0 = DSGDPlan(U, I, world, ubp, ibp)
    rows = max(1, -(-(plan0.user_range(0)[1] - plan0.user_range(0)[0]) // ubp))
    cost = item_costs(di, rows, plan0.Gc, pipeline_fits(I, ibp, rows, k, args.hot_buffers), 1,
                      None if args.hot_cost < 0 else args.hot_cost, walker=True)
    rp = torch.from_numpy(balanced_permutation(du, ubp).astype(np.int32)).cuda()
    cp = torch.from_numpy(interleave_stripes(balanced_permutation(di, ibp, cost), ibp, world).astype(np.int32)).cuda()
    note("permutations done")
    lib = _abi.lib()
    for t in (train, test):
        _abi.check(lib.mfsgd_hip_remap(_abi.ptr(t), int(t.shape[0]), 3, _abi.ptr(rp), _abi.ptr(cp), U, I, _abi.stream_ptr()))
    plan = DSGDPlan(U, I, world, ubp, ibp)
    u0, u1 = plan.user_range(rank)
    b = float(train[:, 2].double().mean().item())
    train_l = train[(train[:, 0] >= u0) & (train[:, 0] < u1)].contiguous()
    test_l = test[(test[:, 0] >= u0) & (test[:, 0] < u1)].contiguous()
    del train, test, rp, cp
    data = None
    torch.cuda.empty_cache()
    w = RankWorker(plan, rank, train_l, k, b, hp, seed=args.seed, hot_buffers=args.hot_buffers, n_workgroups=args.wgs,
                   waves_per_wg=args.waves, test_local=test_l)
    del train_l, test_l
    torch.cuda.synchronize()
    prep_s = time.perf_counter() - t0
    note(f"rank worker built: {[ (g.info.Gr, g.info.Gc) for g in w.grids ]}")
    rm0 = w.errors(dist)                         # before any training (the warm-up epochs train too)
    note(f"rmse before: {rm0}")
    for _ in range(args.warmup):
        w.epoch(dist)
        note("warm-up epoch done")
    w.status()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        w.epoch(dist)
    torch.cuda.synchronize()
    note("timed epochs done")
    dist.barrier()
    wall = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    if dist.get_backend() == "gloo":
        wall = wall.cpu()
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    wall = float(wall.item())
    w.status()
    rm1 = w.errors(dist)
    if rank == 0:
        value = args.steps * n_train_total / wall
        Bk = bytes_per_update(k, "none")
        gi = w.grids[0].info
        out = {
            "metric": "rating-updates/sec", "value": value, "unit": "updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}-shape {U}x{I}, {n_train_total} train ratings, k={k}, implicit=none, "
                                   f"DSGD ring over {world} GPUs, grid {plan.Gr}x{plan.Gc} ({gi.Gr} user stripes per "
                                   f"rank), balanced renumbering, sequential inside a block",
                       "hot_buffers": args.hot_buffers, "hot_segment_ratings_rank0": int(sum(g.info.n_hot for g in w.grids)),
                       "n_train": n_train_total, "k": k, "grid": [plan.Gr, plan.Gc], "prep_seconds": prep_s},
            "roofline": {"bound": "hbm", "achieved": Bk * value / world / 1e9, "peak": hbm_peak / 1e9, "unit": "GB/s",
                         "frac": Bk * value / world / hbm_peak, "traffic": None, "bytes_per_update": Bk,
                         "kernel": "k_epoch_free", "note": "per GPU, whole sub-epoch incl. ring exchange"},
            "rmse": {"train_before": rm0[0], "test_before": rm0[1], "train_after": rm1[0], "test_after": rm1[1]},
        }
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()

"""ExplicitMF: the reference's model / parameter-server class (SGDRecommender.py:28-325) with the same
method names and semantics, driving the HIP path of libmfsgd.so.

What is the same: constructor arguments and defaults (:30-44), `train(iters, multithreaded)` (:49-79)
including the initialisation order and scales (:50-57) and the bold-driver schedules (:70-77, :318-324),
`predict` (:82-83), `make_subsample` / `update_params` (:88-121), `generate_indpendent_samples_new`
(:187-226), `random_renumber_samples` (:228-268), `unrandomize_samples` (:270-284), `save_all_factors`
(:290-294), `get_ratings` / `get_rated_by_user` / `geterrors` / `increment` (:298-325), and the
attributes P, Q, y, b_u, b_i, b, samples, test_samples, ratings, n_users, n_items, u_bins, i_bins, alpha,
update_counter.

What differs, on purpose (each is a defect of the reference):
  * the class is a plain class, not a Ray actor; `train(..., multithreaded=True)` initialises like the
    reference AND then runs the FPSGD epochs on the GPU (the reference leaves that to Ray workers);
    `fit` is an alias;
  * `update_params` merges y without raising (`if subsample.y:` on an ndarray, :96);
  * `unrandomize_samples` really reorders the factors (inverse row permutation for P,b_u; inverse column
    permutation for Q,y,b_i) (:280-282 only rebinds a local);
  * `save_all_factors` writes an .npz (np.save with keywords is invalid, :292);
  * `mse` (used at :71,:79) is broken in the reference; errors are computed by the mse2 predictor.
"""
import multiprocessing
import os
from time import perf_counter

import numpy as np

from . import _abi
from .numba_methods import mse, mse2, rmse_exact, sgd, sgd2  # noqa: F401  (re-exported like `from numba_methods import *`)
from .Recommender import Recommender
from .SubSample import SubSample

_IMPLICIT = {"none": 0, "frozen": 1, "full": 2}


def balanced_permutation(deg, width, cost=None):
    """new_id[old_id] such that the stripes [s*width, (s+1)*width) of the new numbering carry near-equal
    sums of `cost` (default: `deg`): ids by decreasing degree, each to the least-loaded stripe that still has a
    free slot."""
    import heapq
    deg = np.asarray(deg)
    N = len(deg)
    if cost is not None:
        return _balanced_by_cost(deg, np.asarray(cost, dtype=np.float64), width)
    G = -(-N // width)
    cap = [width] * (G - 1) + [N - (G - 1) * width]
    order = np.argsort(-deg, kind="stable")
    new_id = np.empty(N, dtype=np.int64)
    if N <= (1 << 21):
        heap = [(0, s) for s in range(G)]
        used = [0] * G
        dl = deg[order].tolist()
        for old, d in zip(order.tolist(), dl):
            load, s = heapq.heappop(heap)
            new_id[old] = s * width + used[s]
            used[s] += 1
            if used[s] < cap[s]:
                heapq.heappush(heap, (load + d, s))
    else:       # boustrophedon deal: rank r -> stripe r mod G, direction flipping every round
        rnk = np.arange(N)
        rnd, pos = rnk // G, rnk % G
        stripe = np.where(rnd % 2 == 0, pos, G - 1 - pos)
        # the last stripe may be narrower: overflow there is re-dealt to the stripes with free slots
        slot = np.zeros(N, dtype=np.int64)
        for s_ in range(G):
            idx = np.flatnonzero(stripe == s_)
            slot[idx] = np.arange(len(idx))
        over = slot >= np.asarray(cap)[stripe]
        if over.any():
            free = [(s_, c_) for s_ in range(G) for c_ in range(int((stripe == s_).sum() - (over & (stripe == s_)).sum()), cap[s_])]
            for t_, (s_, c_) in zip(np.flatnonzero(over), free):
                stripe[t_], slot[t_] = s_, c_
        new_id[order] = stripe * width + slot
    return new_id


def _balanced_by_cost(deg, cost, width):
    import heapq
    N = len(deg)
    G = -(-N // width)
    cap = [width] * (G - 1) + [N - (G - 1) * width]
    order = np.argsort(-deg, kind="stable")
    new_id = np.empty(N, dtype=np.int64)
    heap = [(0.0, s) for s in range(G)]
    used = [0] * G
    for old, c in zip(order.tolist(), cost[order].tolist()):
        load, s = heapq.heappop(heap)
        new_id[old] = s * width + used[s]
        used[s] += 1
        if used[s] < cap[s]:
            heapq.heappush(heap, (load + c, s))
    return new_id


def reorder_within_stripes(new_id, width, mode, seed=0):
    """The deal numbers a stripe's ids in the order it hands them out: most rated first.  Inside a block the runs are
    stored by user id, so that is also the order in which a block's runs are walked.  mode "heavy" keeps it, "light"
    reverses it inside every stripe, "shuffle" permutes every stripe's ids at random (seeded), "spread" interleaves the
    stripe's heavy half with its light half.  The stripes' contents - and their balance - do not change."""
    if mode == "heavy":
        return new_id
    new_id = np.asarray(new_id)
    N = len(new_id)
    stripe, slot = new_id // width, new_id % width
    size = np.bincount(stripe, minlength=int(stripe.max()) + 1)[stripe]           # ids in this id's stripe
    if mode == "light":
        slot2 = size - 1 - slot
    elif mode == "spread":      # slot s of the heavy half -> 2 s, slot s of the light half -> 2 (s - half) + 1
        half = (size + 1) // 2
        slot2 = np.where(slot < half, 2 * slot, 2 * (slot - half) + 1)
    elif mode == "shuffle":
        rng = np.random.default_rng(seed)
        key = stripe.astype(np.float64) + rng.random(N)                            # random order inside every stripe
        order = np.argsort(key, kind="stable")
        first = np.zeros(int(stripe.max()) + 2, dtype=np.int64)
        np.cumsum(np.bincount(stripe, minlength=int(stripe.max()) + 1), out=first[1:])
        slot2 = np.empty(N, dtype=np.int64)
        slot2[order] = np.arange(N) - first[stripe[order]]
    else:
        raise ValueError(f"reorder_within_stripes: unknown mode {mode!r}")
    return stripe * width + slot2


CHAIN_RATING_COST = 5.0     # a rating of a stripe's most rated item on the ticket path against any other rating there
HOT_RATING_COST = 0.8       # a rating on the matrix-core pipeline against one on the ticket path, as the dealing should count
                            # it (netflix shape, k = 128, swept 0.25 .. 1.0: 25.4 ms per epoch at 0.25, 24.1 at 0.7 - 0.85,
                            # 24.4 at 1.0; the pipeline's own time per rating is a third of the ticket path's, but the
                            # users of a hot segment hold their other runs back until the pipeline has passed them)
WALKER_RATING_COST = 0.15   # the same for the hot-item walker (two waves instead of six, the frontier moves every 64 links),
                            # final build (the apply wave 16 links behind the chain): netflix shape, k = 128: 17.87-17.89 ms per
                            # epoch at 0.05-0.1, 17.85-17.91 at 0.15, 17.92 at 0.2, 18.03 at 0.3, 18.07 at 0.35; ML-25M shape
                            # 6.93 at 0.1-0.15, 6.94-6.97 at 0.3, 7.04 at 0.4


def item_costs(deg, Gr, Gc, pipeline_runs, hot_items=1, weight=None, hot_min=64, walker=False):
    """What an item's ratings cost in time, for dealing items to stripes (balanced_permutation): its degree, except for
    the items that lead their stripes.  Where the hot-item pipeline runs, the stripes' hot items - the hot_items * Gc
    most rated ones, as far as they qualify (ExplicitMF.build_grid) - count HOT_RATING_COST per rating; through the tickets
    a stripe's most rated item is a serial chain the stripe's other ratings wait behind, and its ratings cost more than
    the others' (ML-25M shape, 288 x 256: the 33 K-rating items' stripes took 11.6 ms against a median of 8.6 ms with
    equal counts; 10.0 ms per epoch with the chain counted five-fold).  None = plain degrees."""
    deg = np.asarray(deg)
    if len(deg) > (1 << 21):
        return None
    top = np.argsort(-deg, kind="stable")
    if pipeline_runs and hot_items > 0:
        top = top[:hot_items * Gc]
        top = top[deg[top] >= max(256, hot_min * Gr)]
        w = (WALKER_RATING_COST if walker else HOT_RATING_COST) if weight is None else float(weight)
    else:
        top = top[:Gc]
        w = CHAIN_RATING_COST if weight is None else float(weight)
    if w == 1.0 or not len(top):
        return None
    cost = deg.astype(np.float64)
    cost[top] *= w
    return cost


def interleave_stripes(new_id, width, world):
    """Renumber the stripes of a dealing so that stripe c of the deal becomes stripe (c mod world) * per + c // world
    (per = stripes per shard): balanced_permutation fills stripe 0 with the most rated item, stripe 1 with the next,
    and so on, and the DSGD ring cuts the items into `world` CONTIGUOUS shards - without this the first shard holds
    all the longest chains and every sub-epoch waits for the rank that has it (netflix shape, 8 ranks, one rank's
    sub-epochs: 4.5 ms on shard 0 against 1.9 ms on shard 7)."""
    new_id = np.asarray(new_id)
    N = len(new_id)
    G = -(-N // width)
    per = -(-G // world)
    last_full = N - (G - 1) * width == width        # a narrower last stripe has to stay last
    c = new_id // width
    movable = G if last_full else G - 1
    order = [s_ * per + j for j in range(per) for s_ in range(world) if s_ * per + j < movable]   # targets, deal order
    target = np.arange(G)
    target[:len(order)] = np.asarray(order) if len(order) == movable else target[:len(order)]
    if len(order) != movable:                       # shards of unequal size: leave the deal as it is
        return new_id
    return target[c] * width + new_id % width


def lds_bytes(ibp, Gr, k, esz, hot_bufs=0):
    """LDS a free-block workgroup needs for item stripes of `ibp` items (lds_plan in csrc/epoch.hip, kept in
    step with it by tests/test_abi_and_host.py through mfsgd_lds_bytes); `hot_bufs` row buffers of the hot-item
    pipeline (float32 only, 32 rows of k + 4 floats each) on top."""
    up = lambda x: (x + 15) & ~15
    o = up((16 + 128 + 5 * _abi.MAX_HOT + 4) * 4)
    o = up(o + 4 * 64 * esz)            # the hot-item walker's error words (HOT_EBUFS chunks of 64)
    o = up(o + ibp * 4)
    o = up(o + Gr)
    o = up(o + Gr * 16)
    o = up(o + ibp * k * esz)
    o = up(o + ibp * esz)
    o = up(o + 1026 * 4)
    o = up(o + 1026 * 4)
    if esz == 4:
        o += min(hot_bufs, 8) * up(32 * (k + 4) * 4)
    return o


class Grid:
    """Owner of a device block grid (mfsgd_grid_*)."""

    def __init__(self, d_samples, n_users, n_items, ubp, ibp, order, hot_segments=None):
        """hot_segments: None = the rule of mfsgd_grid_create (one hot item per stripe when it carries 1/16 of it);
        0 / False = none; (items, min_per_block) = mfsgd_grid_create_opt: up to `items` hot items per stripe, each
        with at least `min_per_block` ratings per block on average."""
        _abi.require_gpu()
        h = _abi.c_vp(0)
        if hot_segments is None:
            _abi.check(_abi.lib().mfsgd_grid_create(_abi.ptr(d_samples), int(d_samples.shape[0]), int(d_samples.shape[1]),
                                                    int(n_users), int(n_items), int(ubp), int(ibp), int(order),
                                                    _abi.stream_ptr(), _abi.C.byref(h)))
        else:
            items, per_block = hot_segments if hot_segments else (0, 0)
            _abi.check(_abi.lib().mfsgd_grid_create_opt(_abi.ptr(d_samples), int(d_samples.shape[0]),
                                                        int(d_samples.shape[1]), int(n_users), int(n_items), int(ubp),
                                                        int(ibp), int(order), int(items), int(per_block),
                                                        _abi.stream_ptr(), _abi.C.byref(h)))
        self.handle = h
        info = _abi.GridInfo()
        _abi.check(_abi.lib().mfsgd_grid_get_info(self.handle, _abi.C.byref(info)))
        self.info = info

    def status(self):
        """Synchronise and raise if an epoch kernel gave up on a bounded wait."""
        _abi.check(_abi.lib().mfsgd_grid_status(self.handle, _abi.stream_ptr()))

    def set_hot_buffers(self, n):
        """mfsgd_grid_set_hot_buffers: LDS row buffers (0..8) the hot-item pipeline may use in the free-block kernel;
        0 sends the hot segments through the item tickets like every other rating (same results up to rounding)."""
        _abi.check(_abi.lib().mfsgd_grid_set_hot_buffers(self.handle, int(n)))

    def set_hot_mode(self, mode):
        """mfsgd_grid_set_hot_mode: 2 = hot-item walker (one wave, item row in registers; default), 1 = matrix-core
        pipeline, 0 = item tickets."""
        _abi.check(_abi.lib().mfsgd_grid_set_hot_mode(self.handle, int(mode)))

    def set_pick_width(self, n):
        """mfsgd_grid_set_pick_width: schedule 2 takes the least visited of the first `n` free user stripes."""
        _abi.check(_abi.lib().mfsgd_grid_set_pick_width(self.handle, int(n)))

    def col_hot(self):
        """[Gc][MFSGD_MAX_HOT]: per item stripe the global ids of its hot items, most rated first (item s's ratings form
        hot segment s of the stripe's blocks); -1 = none."""
        out = np.full((self.info.Gc, _abi.MAX_HOT), -1, dtype=np.int32)
        _abi.check(_abi.lib().mfsgd_grid_col_hot(self.handle, out.ctypes.data_as(_abi.c_vp), _abi.stream_ptr()))
        return out

    def export_device(self):
        """Device copies of the grid arrays (mfsgd_grid_export): dict of ROCm tensors."""
        torch = _abi.torch_mod()
        i = self.info
        nb, n, nr = i.Gr * i.Gc, int(i.n), int(i.n_runs)
        t = {"blk_run_ptr": torch.zeros(nb + 1, dtype=torch.int32, device="cuda"),
             "run_start": torch.zeros(nr + 1, dtype=torch.int32, device="cuda"),
             "run_user": torch.zeros(max(nr, 1), dtype=torch.int32, device="cuda"),
             "item": torch.zeros(max(n, 1), dtype=torch.int32, device="cuda"),
             "val": torch.zeros(max(n, 1), dtype=torch.float32, device="cuda"),
             "tick": torch.zeros(max(n, 1), dtype=torch.int32, device="cuda")}
        _abi.check(_abi.lib().mfsgd_grid_export(self.handle, *(_abi.ptr(t[k]) for k in
                                                ("blk_run_ptr", "run_start", "run_user", "item", "val", "tick")),
                                                _abi.stream_ptr()))
        torch.cuda.synchronize()
        t["run_user"], t["item"], t["val"], t["tick"] = t["run_user"][:nr], t["item"][:n], t["val"][:n], t["tick"][:n]
        return t

    def export(self):
        """Host copies of the grid arrays (tests: bucketing must be bit-exact)."""
        return {k: v.cpu().numpy() for k, v in self.export_device().items()}

    def __del__(self):
        try:
            if self.handle:
                _abi.lib().mfsgd_grid_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class ExplicitMF(Recommender):
    def __init__(self,
                 n_threads=multiprocessing.cpu_count(),
                 n_factors=40,
                 alpha=0.006,
                 beta1=.05,
                 beta2=.015,
                 *,
                 implicit=None,          # "none" | "frozen" | "full"; None = "full" single-thread, "frozen" FPSGD
                 deterministic=False,    # True: reference (n_threads+1)^2 grid, fixed row-major block order
                 dtype="float32",        # device arithmetic of the FPSGD path ("float32" | "float64")
                 grid=None,              # (Gr, Gc) of the throughput grid; None = chosen from the shapes
                 exact_errors=True,      # geterrors over all rows (the reference subsamples, mse2)
                 hot_buffers=None,       # LDS row buffers of the hot-item pipeline (None = 8, as many as fit; 0 = ticket path only)
                 hot_items=1,            # hot items (hot segments per block) a stripe may have where the pipeline runs (<= 8)
                 hot_mode=2,             # hot segments: 2 = register-resident walker (one hot item per stripe), 1 = matrix-core pipeline, 0 = tickets
                 hot_cost=None,          # cost of a pipeline rating against a ticket-path one in balanced_renumber_samples
                 hot_min=64,             # ratings per block (on average) an item needs to become a hot item (two pipeline batches)
                 user_order="heavy",     # balanced renumbering: order of a stripe's users (= of a block's runs), reorder_within_stripes
                 pad_factors=True,       # throughput mode: device copies of P / Q / y get zero columns up to a multiple of 64
                 pass_barrier=False,     # False: the passes of one run_epochs call overlap like the reference's scheduler lets
                                         # them (schedule 2: one launch, least-visited-of-two); True: a launch per pass
                 pick_width=None,        # schedule 2: free user stripes among which the least visited is taken (None: library default)
                 n_workgroups=0, waves_per_wg=0, seed=0, verbose=True):
        self.n_threads = n_threads
        self.n_factors = n_factors
        self.alpha = alpha
        self.beta1 = beta1
        self.beta2 = beta2
        self.update_counter = 0
        self.previous_mse = 0
        self.timestamp = perf_counter()
        self.implicit = implicit
        self.deterministic = deterministic
        self.dtype = dtype
        self.grid_shape = grid
        self.exact_errors = exact_errors
        self.hot_buffers = 8 if hot_buffers is None else int(hot_buffers)
        self.hot_items = int(hot_items)
        self.hot_mode = int(hot_mode)
        self.hot_cost = hot_cost
        self.hot_min = int(hot_min)
        self.user_order = user_order
        self.pad_factors = bool(pad_factors)
        self.pass_barrier = bool(pass_barrier)
        self.pick_width = pick_width
        self._launches = 0
        self.n_workgroups, self.waves_per_wg, self.seed, self.verbose = n_workgroups, waves_per_wg, seed, verbose
        self.history = []            # (epoch, train_rmse, test_rmse, alpha) at every bold-driver checkpoint
        self.row_perm = self.col_perm = None
        self._grid = None
        self._dev = None             # device copies of P,Q,y,b_u,b_i while the FPSGD loop owns them

    def _log(self, *a):
        if self.verbose:
            print(*a)

    # ------------------------------------------------------------------ M1: init + training
    def init_params(self):
        """SGDRecommender.py:50-57: P, Q, y ~ N(0, scale=1/k) drawn in this order from the legacy global
        RNG, zero biases, b = mean training rating."""
        k = self.n_factors
        self.P = np.random.normal(scale=1. / k, size=(self.n_users, k))
        self.Q = np.random.normal(scale=1. / k, size=(self.n_items, k))
        self.y = np.random.normal(scale=1. / k, size=(self.n_items, k))
        self.y_owner = False
        self.b_u = np.zeros(self.n_users)
        self.b_i = np.zeros(self.n_items)
        self.b = float(self._d_samples[:, 2].double().mean().item())

    def init_params_device(self, seed=0):
        """Same distributions as init_params, drawn on the GPU (for inputs too large to draw on the host:
        not the legacy-RNG stream, so not comparable draw by draw with the reference)."""
        torch = _abi.torch_mod()
        g = torch.Generator(device="cuda")
        g.manual_seed(int(seed))
        k = self.n_factors
        dt = torch.float64 if self.dtype == "float64" else torch.float32
        self.P = torch.randn(self.n_users, k, device="cuda", dtype=dt, generator=g) / k
        self.Q = torch.randn(self.n_items, k, device="cuda", dtype=dt, generator=g) / k
        self.y = torch.randn(self.n_items, k, device="cuda", dtype=dt, generator=g) / k
        self.y_owner = False
        self.b_u = torch.zeros(self.n_users, device="cuda", dtype=dt)
        self.b_i = torch.zeros(self.n_items, device="cuda", dtype=dt)
        self.b = float(self._d_samples[:, 2].double().mean().item())

    def train(self, iters, multithreaded=True, run=True):
        self.init_params()
        self._log("items,users:", self.n_users, self.n_items)
        if not multithreaded:
            return self._train_single(iters)
        if run:
            return self.run_fpsgd(iters)
        return None

    fit = train

    def _train_single(self, iters):
        """SGDRecommender.py:62-79: `iters` exact-order passes over the samples in file order, bold driver
        (x0.5 / x1.05) on every odd epoch, test error at the end.  Factors stay on the GPU between passes."""
        torch = _abi.torch_mod()
        imp = self.implicit or "full"
        rp, ci = (self._ensure_csr() if imp != "none" else (None, None))
        td = "float64" if self.dtype == "float64" else "float32"
        dev = [_abi.to_device(a, td) for a in (self.P, self.Q, self.b_u, self.b_i)]
        dy = _abi.to_device(self.y, td) if imp != "none" else None
        previous_mse = 0
        for i in range(1, iters + 1):
            t0 = perf_counter()
            sgd(dev[0], dev[1], dev[2], dev[3], self.b, dy, self._d_samples, rp, ci, self.alpha, self.beta1,
                self.beta2, implicit=imp)
            self._log(f"SGD time: {perf_counter()-t0}")
            if i % 2:
                train_mse = rmse_exact(self._d_samples, rp, ci, dev[0], dev[1], dev[2], dev[3], self.b, dy)
                if train_mse > previous_mse and previous_mse:
                    self.alpha *= .5
                else:
                    self.alpha *= 1.05
                self._log(f"Changed alpha to {self.alpha}")
                previous_mse = train_mse - .0001
                self._log("Iteration: %d ; train error = %.4f" % (i, train_mse))
                self.history.append((i, train_mse, None, self.alpha))
        test = rmse_exact(self._d_test, rp, ci, dev[0], dev[1], dev[2], dev[3], self.b, dy)
        self._log("Test error = %.4f" % test)
        self.P, self.Q, self.b_u, self.b_i = (t.double().cpu().numpy() for t in dev)
        if dy is not None:
            self.y = dy.double().cpu().numpy()
        torch.cuda.synchronize()
        return test

    # ------------------------------------------------------------------ the GPU FPSGD path
    def _device_k(self):
        """Row width of the device copies of P / Q / y.  The free-block kernel has straight-line builds (exact waits, the
        hot-item pipeline) for k = 64, 128, 192, 256 only; any other k runs the general build with lane guards and
        optional loads, at about half the rate (netflix shape: k = 40 37.1 ms per epoch against 20.4 at k = 64, k = 100
        45.3 against 23.7 at k = 128).  Zero columns stay zero under the update (p' = a p + alpha e q with p = q = 0) and
        add zeros to every dot product, so in throughput mode the device rows are padded to the next multiple of 64 and
        cut back when they return to the host: same factors, same errors."""
        k = int(self.n_factors)
        kd = -(-k // 64) * 64
        if self.deterministic or not self.pad_factors or kd > 256:
            return k
        return kd

    def _walker_runs(self, Gr, Gc):
        """Will the free-block launch walk hot segments with the hot-item walker (csrc/epoch.hip phase_a_serial)?  Any
        straight-line build (device rows a multiple of 64 factors, float32 or float64, with or without the frozen
        term) whose stripes sit in LDS, one hot item per stripe."""
        if self.deterministic or self.hot_mode != 2 or self.hot_items != 1 or self.hot_buffers <= 0:
            return False
        kd = self._device_k()
        esz = 8 if self.dtype == "float64" else 4
        # a ring slot is one 4 / 8 / 16-byte load per lane: k = 64, 128, 256 in float32, 64 and 128 in float64
        if kd not in (64, 128, 256) or (kd // 64) * esz > 16 or (self.implicit or "frozen") == "full":
            return False
        return lds_bytes(-(-self.n_items // Gc), Gr, kd, esz) <= 160 * 1024

    def _hot_path_runs(self, Gr, Gc):
        return self._walker_runs(Gr, Gc) or self._pipeline_runs(Gr, Gc)

    def _pipeline_kernel(self):
        """The hot-item pipeline is worth its hot segments for fp32, k = 64 or 128, no implicit term.  The kernel also
        exists for k = 256 (csrc/epoch.hip, FASTA; parity-tested), but a batch's row buffer is 33 KB there, two fit
        next to a stripe, and two pipeline waves are slower than the tickets (netflix shape, k = 256: 74 ms per epoch
        against 56 ms), so the chooser does not ask for hot segments at k = 256."""
        return (self.dtype != "float64" and self._device_k() in (64, 128) and (self.implicit or "frozen") == "none"
                and self.hot_buffers > 0 and self.hot_mode >= 1)

    def _pipeline_runs(self, Gr, Gc):
        """Will the free-block launch walk hot segments with the pipeline on a Gr x Gc grid?  It does when the kernel
        exists and at least two of its row buffers fit behind the stripe in a CU's LDS (one wave alone is no pipeline);
        the launch then runs one workgroup per CU on the 128-register build (launch_epochs)."""
        if not self._pipeline_kernel():
            return False
        ibp = -(-self.n_items // Gc)
        return lds_bytes(ibp, Gr, self._device_k(), 4, min(2, self.hot_buffers)) <= 160 * 1024

    def _choose_grid(self):
        """Throughput grid.  A workgroup keeps its item stripe (rows of Q, b_i, tickets, run table) in LDS, and a CU has
        160 KB of it.  One 16-wave workgroup per CU on the 128-register build is the default shape: 256 item stripes
        (one round of the 256 CUs) whenever the stripe fits, else 512 (two stripes per workgroup, handed out heaviest
        first).  User stripes: an eighth more than workgroups in flight (row locks are taken at random; swept 256..320:
        flat, 272-288 best).  Small inputs shrink both so that a block still holds ~200 ratings: a stripe's tenure is the
        chain of its most rated item plus one transition (~7 us: lock, fences, run table, first rows, barrier) per
        user stripe, so few large blocks beat many small ones until the workgroups run out of ratings (ML-1M shape:
        93 x 75 blocks 1.26 ms per epoch, 68 x 60 1.18, 56 x 48 1.20)."""
        if self.grid_shape is not None:
            Gr, Gc = self.grid_shape
        else:
            k = self._device_k()
            esz = 8 if self.dtype == "float64" else 4
            n = int(self._d_samples.shape[0])
            U, I = self.n_users, self.n_items
            cap = 160 * 1024
            fits = lambda Gr_, Gc_: lds_bytes(-(-I // Gc_), Gr_, k, esz) <= cap
            if U >= 288 and I >= 256 and n >= 128 * 288 * 256 and fits(288, 256):
                Gr, Gc = 288, 256
            elif U >= 288 and I >= 512 and n >= 128 * 288 * 512 and fits(288, 512):
                Gr, Gc = 288, 512
            else:
                ibp_max = max(1, (120 * 1024) // ((k + 1) * esz + 4))
                Gc_min = -(-I // ibp_max)
                Gc = min(I, 512)
                ibp0 = -(-I // max(Gc, Gc_min))
                two = esz == 4 and k <= 128 and lds_bytes(ibp0, 640, k, esz) <= 80 * 1024
                Gr = min(U, 640 if two else 272)      # Yahoo shape, k=256: 640 -> 320 rows took 276 -> 201 ms; round 3: 320 rows 186.5, 288 181.6, 272 179.6
                want = max(1.0, n / 200.0)
                if Gr * Gc > want:
                    f = (want / (Gr * Gc)) ** 0.5
                    Gr, Gc = max(1, int(Gr * f)), max(1, int(Gc * f))
                Gc = min(I, max(Gc, Gc_min))
        ubp = -(-self.n_users // Gr)
        ibp = -(-self.n_items // Gc)
        return ubp, ibp

    def build_grid(self, force=False):
        if self._grid is not None and not force:
            return self._grid
        if self.deterministic:
            c = self.n_threads
            ubp = int(self.n_users / (c + 1)) + 1      # SGDRecommender.py:191-192
            ibp = int(self.n_items / (c + 1)) + 1
            self._grid = Grid(self._d_samples, self.n_users, self.n_items, ubp, ibp, order=0)
        else:
            ubp, ibp = self._choose_grid()
            # hot segments only where the pipeline will walk them: through the item tickets they cost time
            # (ML-25M shape, 288 x 256: 11.8 ms per epoch without them, 13.0 ms with); and only items that fill at least
            # two pipeline batches (64 ratings) per block on average: a batch costs ~9 us of row gather and Gram tile
            # before its first link (ML-1M shape: 93 x 75 blocks 1.26 ms per epoch without hot segments, 1.70 ms with
            # items of 32+ ratings per block; 64 x 56 blocks 1.18 against 1.41)
            runs = self._hot_path_runs(-(-self.n_users // ubp), -(-self.n_items // ibp))
            hot = (self.hot_items, self.hot_min) if runs and self.hot_items > 0 else 0
            self._grid = Grid(self._d_samples, self.n_users, self.n_items, ubp, ibp, order=1, hot_segments=hot)
        if self.hot_buffers != 8:
            self._grid.set_hot_buffers(self.hot_buffers)
        if self.pick_width is not None:
            self._grid.set_pick_width(self.pick_width)
        if self.hot_mode != 2:
            self._grid.set_hot_mode(self.hot_mode)
        return self._grid

    def _to_dev(self):
        td = "float64" if self.dtype == "float64" else "float32"
        self._dev = {n: _abi.to_device(getattr(self, n), td) for n in ("P", "Q", "b_u", "b_i")}
        k, kd = int(self.n_factors), self._device_k()
        self._kd = kd
        pad = (lambda t: _abi.torch_mod().nn.functional.pad(t, (0, kd - k)).contiguous()) if kd != k else (lambda t: t)
        self._dev["P"], self._dev["Q"] = pad(self._dev["P"]), pad(self._dev["Q"])
        self._dev["z"] = None
        imp = self.implicit or "frozen"
        if imp == "full":
            raise ValueError('implicit="full" rewrites y globally on every rating and cannot run block-parallel; '
                             'use train(iters, multithreaded=False) or implicit="frozen"/"none"')
        if imp == "frozen":
            torch = _abi.torch_mod()
            rp, ci = self._ensure_csr()
            dy = pad(_abi.to_device(self.y, td))
            z = torch.empty_like(self._dev["P"])
            fn = getattr(_abi.lib(), "mfsgd_hip_implicit_z_" + ("f64" if td == "float64" else "f32"))
            _abi.check(fn(_abi.ptr(dy), _abi.ptr(rp), _abi.ptr(ci), self.n_users, kd, 0, _abi.ptr(z),
                          _abi.stream_ptr()))
            self._dev["z"] = z
        return imp

    def _from_dev(self):
        k = int(self.n_factors)
        for n in ("P", "Q", "b_u", "b_i"):
            t = self._dev[n]
            if n in ("P", "Q") and int(t.shape[1]) != k:
                t = t[:, :k]                                    # the zero columns of _device_k stay on the device
            setattr(self, n, t.double().cpu().numpy())

    def run_epochs(self, n_epochs, order_log=None):
        """`n_epochs` FPSGD epochs on the device-resident factors (no host traffic)."""
        d = self._dev
        kind = "f64" if self.dtype == "float64" else "f32"
        ct = _abi.ctype_of(kind)
        fn = getattr(_abi.lib(), "mfsgd_hip_epoch_" + kind)
        imp = 1 if d["z"] is not None else 0
        _abi.check(fn(self._grid.handle, int(d["P"].shape[1]), _abi.ptr(d["P"]), _abi.ptr(d["Q"]), _abi.ptr(d["b_u"]),
                      _abi.ptr(d["b_i"]), ct(self.b), _abi.ptr(d["z"]), ct(self.alpha), ct(self.beta1),
                      ct(self.beta2), 0, imp, int(n_epochs), 0 if self.deterministic else (1 if self.pass_barrier else 2),
                      int(self.seed + self.update_counter + 7919 * self._launches), int(self.n_workgroups),
                      int(self.waves_per_wg), _abi.ptr(order_log), _abi.stream_ptr()))
        self._launches += 1

    def device_errors(self):
        self._grid.status()
        d = self._dev
        tr = rmse_exact(self._d_samples, None, None, d["P"], d["Q"], d["b_u"], d["b_i"], self.b, None, d["z"])
        te = rmse_exact(self._d_test, None, None, d["P"], d["Q"], d["b_u"], d["b_i"], self.b, None, d["z"])
        return tr, te

    def run_fpsgd(self, iters):
        """The worker loop + scheduler + parameter server of RayBlockScheduler.py:148-181 as GPU epochs.
        Bold driver as `increment` (:315-325): after every 2 grid passes compare the train error."""
        self.build_grid()
        self._to_dev()
        blocks_per_epoch = (self.n_threads + 1) ** 2
        done = 0
        while done < iters:
            step = min(2, iters - done)
            t0 = perf_counter()
            self.run_epochs(step)
            done += step
            self.update_counter += step * blocks_per_epoch
            if step == 2:
                _abi.torch_mod().cuda.synchronize()
                self._log(f"Full grid update took {perf_counter()-t0} s.")
                tr, te = self.device_errors()
                self._log("Iteration: %d ; train error = %.4f ; test error = %.4f ; lr = %.4f" % (done, tr, te, self.alpha))
                if tr > self.previous_mse and self.previous_mse:
                    self.alpha *= .6
                else:
                    self.alpha *= 1.06
                self.previous_mse = tr - .0001
                self.history.append((done, tr, te, self.alpha))
        self._grid.status()
        self._from_dev()
        return self.history

    # ------------------------------------------------------------------ P1
    @staticmethod
    def predict(row, col, user_vecs, item_vecs):
        """SGDRecommender.py:82-83: plain dot of factor rows (no biases, no implicit term)."""
        return user_vecs[row, :].dot(item_vecs[col, :].T)

    @staticmethod
    def top_n(users, user_vecs, item_vecs, n=10, b_u=None, b_i=None, b=0.0):
        """The `n` best items for each of `users`: what `argsort(-predict(users, all_items, ...))[:, :n]` gives,
        without ever forming the users x items matrix (mfsgd_hip_topn_f32: MFMA tiles + top-n on the fly).
        With b_u/b_i the scores are the training predictor's (numba_methods.py:17, no implicit term), else the
        plain dot of `predict`.  Arrays may be NumPy or ROCm tensors; returns (items [len(users)][n] int32,
        scores [len(users)][n] float32), best first, ties to the lower item id."""
        _abi.require_gpu()
        torch = _abi.torch_mod()
        if not 1 <= n <= 16:
            raise ValueError("top_n: n must be in [1, 16]")
        nn = 4 if n <= 4 else (8 if n <= 8 else 16)
        dP, dQ = _abi.to_device(user_vecs, "float32"), _abi.to_device(item_vecs, "float32")
        if int(dP.shape[1]) != int(dQ.shape[1]) or not 32 <= int(dP.shape[1]) <= 256:
            raise ValueError("top_n: factor matrices must share k in [32, 256]")
        with_bias = b_u is not None and b_i is not None
        dbu = _abi.to_device(b_u, "float32") if with_bias else None
        dbi = _abi.to_device(b_i, "float32") if with_bias else None
        du = _abi.to_device(np.asarray(users).reshape(-1) if not hasattr(users, "is_cuda") else users, "int32")
        m = int(du.shape[0])
        if m and (int(du.min()) < 0 or int(du.max()) >= int(dP.shape[0])):
            raise IndexError("top_n: a user id is outside the factor matrix")
        items = torch.empty((m, nn), dtype=torch.int32, device="cuda")
        scores = torch.empty((m, nn), dtype=torch.float32, device="cuda")
        _abi.check(_abi.lib().mfsgd_hip_topn_f32(_abi.ptr(dP), _abi.ptr(dQ), _abi.ptr(dbu), _abi.ptr(dbi), float(b),
                                                 _abi.ptr(du), m, int(dQ.shape[0]), int(dP.shape[1]),
                                                 1 if with_bias else 0, nn, _abi.ptr(items), _abi.ptr(scores),
                                                 _abi.stream_ptr()))
        torch.cuda.synchronize()
        return items[:, :n].cpu().numpy(), scores[:, :n].cpu().numpy()

    def recommend(self, users, n=10):
        """Top-n items (current id space) for `users` under the trained model's predictor without the implicit term."""
        return ExplicitMF.top_n(users, self.P, self.Q, n, self.b_u, self.b_i, self.b)

    def predict_rating(self, user, item):
        """The training predictor (numba_methods.py:17) for one (user, item) in the current id space."""
        rp, ci = self.ratings.rowptrs, self.ratings.colinds
        rated = ci[rp[user]:rp[user + 1]]
        ys = self.y[rated].sum(axis=0) / np.sqrt(len(rated)) if len(rated) and (self.implicit or "full") != "none" else 0.0
        return float(self.Q[item].dot(self.P[user] + ys) + self.b_u[user] + self.b_i[item] + self.b)

    # ------------------------------------------------------------------ M2
    def _block_ranges(self, block_pos):
        r, c = block_pos
        return ((self.u_bins[r], min(self.u_bins[r + 1], self.n_users)),
                (self.i_bins[c], min(self.i_bins[c + 1], self.n_items)))

    def update_params(self, subsample: SubSample) -> SubSample:
        (u0, u1), (i0, i1) = self._block_ranges(subsample.block_pos)
        self.P[u0:u1] = subsample.P
        self.Q[i0:i1] = subsample.Q
        self.b_u[u0:u1] = subsample.b_u
        self.b_i[i0:i1] = subsample.b_i
        if subsample.y is not None and subsample.y is not False:
            self.y = subsample.y
        return subsample

    def make_subsample(self, block_pos) -> SubSample:
        y = self.y if self.y_owner is False else False
        (u0, u1), (i0, i1) = self._block_ranges(block_pos)
        return SubSample(tuple(block_pos), (u0, i0), self.P[u0:u1], self.Q[i0:i1], self.b_u[u0:u1],
                         self.b_i[i0:i1], self.b, y, self.alpha, self.beta1, self.beta2)

    # ------------------------------------------------------------------ D2 / D3 / D4
    def random_renumber_samples(self):
        """SGDRecommender.py:228-268: shuffle item ids, then user ids (legacy global RNG, this order), remap
        train and test on the GPU, drop test rows whose user or item is unseen in train, rebuild the CSR."""
        col_index = np.arange(self.n_items)
        row_index = np.arange(self.n_users)
        np.random.shuffle(col_index)
        np.random.shuffle(row_index)
        self._apply_renumbering(row_index, col_index)

    def balanced_renumber_samples(self):
        """Throughput-mode alternative to the random shuffle: a permutation that deals users and items into
        the equal-width stripes of the block grid so that the stripes carry about equal numbers of ratings
        (longest-processing-time packing).  The reference relies on the random shuffle alone for balance
        (SGDRecommender.py:228-232); with thousands of stripes a popular item would otherwise share its stripe
        with other popular items and that stripe's serial chain would set the epoch time."""
        torch = _abi.torch_mod()
        ubp, ibp = self._choose_grid()
        du = torch.bincount(self._d_samples[:, 0].long(), minlength=self.n_users).cpu().numpy()
        di = torch.bincount(self._d_samples[:, 1].long(), minlength=self.n_items).cpu().numpy()
        Gr, Gc = -(-self.n_users // ubp), -(-self.n_items // ibp)
        cost = item_costs(di, Gr, Gc, self._hot_path_runs(Gr, Gc), self.hot_items, self.hot_cost, self.hot_min,
                          walker=self._walker_runs(Gr, Gc))
        self._apply_renumbering(reorder_within_stripes(balanced_permutation(du, ubp), ubp, self.user_order, self.seed),
                                balanced_permutation(di, ibp, cost))

    def _apply_renumbering(self, row_index, col_index):
        """Remap ids by `row_index` / `col_index` (current id -> new id).  self.row_perm / col_perm always map the
        ORIGINAL ids to the current ones: a second renumbering composes with the first, so that
        unrandomize_samples undoes all of them (SGDRecommender.py:270-284 undoes its single shuffle)."""
        torch = _abi.torch_mod()
        step_c, step_r = np.asarray(col_index).astype(np.int32), np.asarray(row_index).astype(np.int32)
        d_rp, d_cp = _abi.to_device(step_r, "int32"), _abi.to_device(step_c, "int32")
        lib = _abi.lib()
        before = torch.bincount(self._d_samples[:, 0].long(), minlength=self.n_users).sort().values
        for t in (self._d_samples, self._d_test):
            _abi.check(lib.mfsgd_hip_remap(_abi.ptr(t), int(t.shape[0]), int(t.shape[1]), _abi.ptr(d_rp),
                                           _abi.ptr(d_cp), self.n_users, self.n_items, _abi.stream_ptr()))
        after = torch.bincount(self._d_samples[:, 0].long(), minlength=self.n_users).sort().values
        assert bool((before == after).all()), "permutation changed the per-user rating counts"
        out = torch.empty_like(self._d_test)
        cnt = _abi.c_i64(0)
        _abi.check(lib.mfsgd_hip_filter_test(_abi.ptr(self._d_test), int(self._d_test.shape[0]),
                                             int(self._d_test.shape[1]), _abi.ptr(self._d_samples),
                                             int(self._d_samples.shape[0]), int(self._d_samples.shape[1]),
                                             self.n_users, self.n_items, _abi.ptr(out), _abi.C.byref(cnt),
                                             _abi.stream_ptr()))
        self._d_test = out[:cnt.value].contiguous()
        # only now that the device arrays are remapped: original id -> current id
        self.row_perm = step_r if self.row_perm is None else step_r[self.row_perm]
        self.col_perm = step_c if self.col_perm is None else step_c[self.col_perm]
        if self.samples is not None:
            self.samples = self._d_samples.cpu().numpy()
            self.test_samples = self._d_test.cpu().numpy()
        self._build_csr()
        self._grid = None

    def generate_indpendent_samples_new(self):
        """SGDRecommender.py:187-226: permute ids, cut users and items into n_threads+1 equal-width bins and
        bucket the ratings, on the GPU; returns the list-of-lists of (n_b, 3) arrays the reference returns."""
        self.random_renumber_samples()
        torch = _abi.torch_mod()
        c = self.n_threads
        ubp = int(self.n_users / (c + 1)) + 1
        ibp = int(self.n_items / (c + 1)) + 1
        self.u_bins = [*range(0, self.n_users + ubp, ubp)]
        self.i_bins = [*range(0, self.n_items + ibp, ibp)]
        Pu, Pi = -(-self.n_users // ubp), -(-self.n_items // ibp)
        n = int(self._d_samples.shape[0])
        out = torch.empty((n, 3), dtype=torch.int32, device="cuda")
        ptr = torch.zeros(Pu * Pi + 1, dtype=torch.int64, device="cuda")
        _abi.check(_abi.lib().mfsgd_hip_bucket(_abi.ptr(self._d_samples), n, int(self._d_samples.shape[1]), ubp, ibp,
                                               Pu, Pi, _abi.ptr(out), _abi.ptr(ptr), _abi.stream_ptr()))
        self._d_blocks, self._d_block_ptr = out, ptr
        h, p = out.cpu().numpy(), ptr.cpu().numpy()
        groups = [[h[p[r * Pi + cc]:p[r * Pi + cc + 1]] for cc in range(Pi)] for r in range(Pu)]
        self._log("Shape of FPSGD grid:", Pu, Pi)
        self._log(f"Total grid samples: {n}")
        if Pu != Pi:
            self._log(f"warning: {Pu} user bins vs {Pi} item bins - the reference would mis-bucket here "
                      "(SGDRecommender.py:208); every block above is still exact")
        return groups

    def get_ratings(self):
        return self.ratings

    def get_rated_by_user(self):
        return (self.ratings.rowptrs, self.ratings.colinds)

    # ------------------------------------------------------------------ M3
    def geterrors(self, update_num):
        self._log(f"Full grid update took {perf_counter()-self.timestamp} s.")
        self.timestamp = perf_counter()
        start = perf_counter()
        rp, ci = self._ensure_csr()
        y = None if (self.implicit == "none") else self.y
        fn = (lambda s: rmse_exact(s, rp, ci, self.P, self.Q, self.b_u, self.b_i, self.b, y)) if self.exact_errors \
            else (lambda s: mse2(s, rp, ci, self.P, self.Q, self.b_u, self.b_i, self.b, y if y is not None else np.zeros_like(self.Q)))
        train_mse, test_mse = fn(self._d_samples), fn(self._d_test)
        self._log("Iteration: %d ; train error = %.4f ; test error = %.4f ; lr = %.4f" % (update_num, train_mse, test_mse, self.alpha))
        self._log(f"Calculated error in {perf_counter()-start} s.")
        self.history.append((update_num, train_mse, test_mse, self.alpha))
        return train_mse

    def increment(self):
        """SGDRecommender.py:315-325: one tick per finished block; every 2*P^2 ticks the bold driver."""
        self.update_counter += 1
        P2 = (self.n_threads + 1) ** 2
        if self.update_counter % (P2 * 2) == 0:
            train_mse = self.geterrors(self.update_counter / P2)
            if train_mse > self.previous_mse and self.previous_mse:
                self.alpha *= .6
            else:
                self.alpha *= 1.06
            self.previous_mse = train_mse - .0001
        return self.update_counter

    # ------------------------------------------------------------------ persistence
    def unrandomize_samples(self):
        """Undo random_renumber_samples: ids AND factor rows go back to the original numbering."""
        if self.row_perm is None:
            return
        inv_r = np.argsort(self.row_perm).astype(np.int32)      # new id -> old id
        inv_c = np.argsort(self.col_perm).astype(np.int32)
        d_r, d_c = _abi.to_device(inv_r, "int32"), _abi.to_device(inv_c, "int32")
        for t in (self._d_samples, self._d_test):
            _abi.check(_abi.lib().mfsgd_hip_remap(_abi.ptr(t), int(t.shape[0]), int(t.shape[1]), _abi.ptr(d_r),
                                                  _abi.ptr(d_c), self.n_users, self.n_items, _abi.stream_ptr()))
        if self.samples is not None:
            self.samples, self.test_samples = self._d_samples.cpu().numpy(), self._d_test.cpu().numpy()
        # row `old` of the un-permuted factors is row perm[old] of the current ones
        self.P, self.b_u = self.P[self.row_perm], self.b_u[self.row_perm]
        self.Q, self.y, self.b_i = self.Q[self.col_perm], self.y[self.col_perm], self.b_i[self.col_perm]
        self.row_perm = self.col_perm = None
        self._build_csr()
        self._grid = None

    def save_all_factors(self, name):
        os.makedirs("./factors", exist_ok=True)
        path = f"./factors/{name},{self.n_factors}{self.alpha},{self.beta1},{self.beta2}.npz"
        try:
            np.savez(path, P=self.P, Q=self.Q, y=self.y, b_u=self.b_u, b_i=self.b_i, b=self.b)
        except AttributeError:
            print("Factors were not initialized")
            return None
        return path

    def load_factors(self, path):
        f = np.load(path)
        self.P, self.Q, self.y, self.b_u, self.b_i, self.b = f["P"], f["Q"], f["y"], f["b_u"], f["b_i"], float(f["b"])

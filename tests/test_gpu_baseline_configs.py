"""`-m gpu` tests of the BASELINE.json configurations at FULL size (synthetic data of workloads.py, the same
generator and seeds bench.py uses), through the C-ABI.

 * ML-1M shape, k=64: the free-block schedule's logged block order replayed rating by rating on the CPU oracle
   reproduces the GPU factors (1e-4 relative, north_star's bar for the factors).
 * ML-25M shape, k=128: north_star's held-out-RMSE gate.  Same data, same id space, same initial factors, same
   epochs: the deterministic fixed-order run on the reference's (n_threads+1)^2 grid and the free-block schedule
   (every rating through the item tickets, and hot segments through the hot-item pipeline) agree pairwise to 0.002 in held-out
   RMSE on the trained model.  One epoch of each free-block run is also replayed on the oracle.
 * Netflix shape k=128 and Yahoo!Music shape k=256: size-independent properties - every non-empty block issued
   exactly once per epoch, no bounded wait ran out, block sizes sum to n, the tickets of a sampled block are the
   ranks of its ratings per item, train and held-out RMSE fall epoch after epoch.
The numbers these tests print are kept in profiles/r2_rmse_parity.txt."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HP = (0.006, 0.05, 0.015)


def relerr(a, ref):
    return float(np.abs(np.asarray(a, dtype=np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.fixture(scope="module")
def ra():
    import recommender_amd
    from recommender_amd import _abi
    _abi.require_gpu()
    return recommender_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _model(ra, name, *, hot_buffers=None, deterministic=False, n_threads=15, renumber="balanced", grid=None, k=None,
           init_seed=1, hot_items=1, **kw):
    """ExplicitMF on the named workload: data generated on the GPU, ids renumbered, grid built, factors drawn."""
    import torch
    import workloads
    train, test, U, I, kk = workloads.synth(torch, name)
    m = ra.ExplicitMF(n_threads=n_threads, n_factors=k or kk, alpha=HP[0], beta1=HP[1], beta2=HP[2], implicit="none",
                      deterministic=deterministic, hot_buffers=hot_buffers, grid=grid, hot_items=hot_items, verbose=False,
                      **kw)
    m.load_samples_from_device(train, test, U, I)
    np.random.seed(20240611)
    m.random_renumber_samples()              # D2 exactly as the reference (SGDRecommender.py:228-268)
    if renumber == "balanced":
        m.balanced_renumber_samples()        # composes with it: row_perm / col_perm map the ORIGINAL ids
    g = m.build_grid()
    m.init_params_device(init_seed)
    m._to_dev()
    return m, g


def _check_order_log(log, g, blk_run_ptr):
    """Every non-empty block exactly once per epoch, nothing else."""
    Gr, Gc = g.info.Gr, g.info.Gc
    nonempty = np.flatnonzero(np.diff(blk_run_ptr) > 0)
    for e in range(log.shape[0]):
        issued = log[e][log[e][:, 0] >= 0]
        ids = issued[:, 0].astype(np.int64) * Gc + issued[:, 1]
        assert len(ids) == len(nonempty), (e, len(ids), len(nonempty))
        assert np.array_equal(np.sort(ids), nonempty), e
        assert issued[:, 0].max() < Gr and issued[:, 1].max() < Gc


def _replay(orc, m, g, log, P, Q, bu, bi):
    """Replay the logged block order on the CPU oracle's plain sequential kernel, in place on the given host factors;
    inside a block the grid's stored order (tests/_order.py: hot segment first, then by user)."""
    from _order import stored_order
    s = m._d_samples.cpu().numpy()
    Gc, ubp, ibp = g.info.Gc, g.info.ubp, g.info.ibp
    perm, key, hot = stored_order(s, ubp, ibp, Gc, g.col_hot())
    assert int(hot.sum()) == g.info.n_hot
    ss, ks = s[perm], key[perm]
    ptr = np.searchsorted(ks, np.arange(g.info.Gr * Gc + 1))
    for e in range(log.shape[0]):
        for r, c in log[e]:
            if r < 0:
                continue
            orc.sgd(P, Q, bu, bi, m.b, None, ss[ptr[r * Gc + c]:ptr[r * Gc + c + 1]], None, None, *HP, implicit=0)


def _host_factors(m):
    return [m._dev[n].cpu().numpy().copy() for n in ("P", "Q", "b_u", "b_i")]


# ------------------------------------------------------------------ config 2: ML-1M shape, k=64
@pytest.mark.parametrize("pass_barrier", [True, False])
def test_ml1m_k64_full_size_replays_on_oracle(ra, orc, pass_barrier):
    """pass_barrier=True: a launch per pass (schedule 1); False: all passes in one launch, overlapped like the
    reference's scheduler overlaps them (schedule 2)."""
    import torch
    m, g = _model(ra, "ml1m", pass_barrier=pass_barrier)
    assert m.n_factors == 64 and g.info.n > 890_000
    iters = 2 if pass_barrier else 3
    ref = _host_factors(m)
    log = torch.full((iters * g.info.Gr * g.info.Gc, 2), -1, dtype=torch.int32, device="cuda")
    m.run_epochs(iters, order_log=log)
    g.status()
    log = log.cpu().numpy().reshape(iters, -1, 2)
    if pass_barrier:
        _check_order_log(log, g, g.export()["blk_run_ptr"])
    else:       # every non-empty block exactly `iters` times in all
        issued = log.reshape(-1, 2)
        issued = issued[issued[:, 0] >= 0]
        ids, counts = np.unique(issued[:, 0].astype(np.int64) * g.info.Gc + issued[:, 1], return_counts=True)
        assert np.array_equal(ids, np.flatnonzero(np.diff(g.export()["blk_run_ptr"]) > 0)) and (counts == iters).all()
    _replay(orc, m, g, log, *ref)
    for name, r_ in zip(("P", "Q", "b_u", "b_i"), ref):
        assert relerr(m._dev[name].cpu().numpy(), r_.astype(np.float64)) <= 1e-4, name


# ------------------------------------------------------------------ config 3: ML-25M shape, k=128 - the RMSE gate
def test_ml25m_k128_rmse_gate_fixed_vs_free_block(ra, orc):
    """north_star: 'held-out RMSE must match within 0.002'.  The reference's own parallel run is any conflict-free
    block order with a sequential pass inside each block (RayBlockScheduler.py:148-181, :58).

    Three runs on the full-size shape, same data, same id space (the reference's random shuffle under one seed),
    same initial factors, same number of epochs at the reference's constant alpha = 0.006:
      fixed     deterministic=True: the reference's 16 x 16 grid, file order inside a block, row-major block order;
      tickets   on-GPU free-block schedule on a 288 x 384 grid, every rating through the item tickets (no hot segments);
      pipeline  the same schedule and grid with up to three hot segments per block, walked by the hot-item pipeline.
    All three are sequential inside a block; they differ in the block grid and the block order only.
    The gate is asserted on the trained model: at every epoch of the last quarter.
    profiles/r2_rmse_parity.txt holds 30-epoch trajectories (profiles/rmse_gate.py): on the plateau the orders agree to
    3e-4; WHILE the model leaves the small-initialisation saddle (epochs 2-8 here) the free-block curves run ~0.15
    epoch behind the fixed order (up to 0.015 apart at epoch 3) - a shift in time, not in where they arrive.
    One epoch of each free-block run is also replayed on the CPU oracle (factors to 1e-4)."""
    import torch
    epochs = 16
    tail = 4
    runs, grids = {}, {}
    for tag, kw in (("fixed", dict(deterministic=True, n_threads=15, renumber="random")),
                    ("tickets", dict(hot_buffers=0, renumber="random", grid=(288, 384))),
                    ("pipeline", dict(renumber="random", grid=(288, 384), hot_items=3))):
        m, g = _model(ra, "ml25m", **kw)
        assert m.n_factors == 128
        if tag == "fixed":
            assert (g.info.Gr, g.info.Gc, g.info.order, g.info.n_hot) == (16, 16, 0, 0)
        elif tag == "tickets":      # no pipeline, so the grid is built without hot segments (ExplicitMF.build_grid)
            assert g.info.order == 1 and g.info.n_hot == 0
        else:                       # 384 stripes of 154 items leave LDS for the pipeline's row buffers
            assert g.info.order == 1 and g.info.n_hot > 1_000_000 and m._pipeline_runs(g.info.Gr, g.info.Gc)
            assert g.info.n_hot_items > g.info.n_hot_stripes            # several hot items per stripe
        traj = [m.device_errors()]
        ref = log = None
        if tag != "fixed":
            ref = _host_factors(m)
            log = torch.full((g.info.Gr * g.info.Gc, 2), -1, dtype=torch.int32, device="cuda")
        m.run_epochs(1, order_log=log)
        if tag != "fixed":
            g.status()
            log = log.cpu().numpy().reshape(1, -1, 2)
            _check_order_log(log, g, g.export_device()["blk_run_ptr"].cpu().numpy())
            _replay(orc, m, g, log, *ref)
            for name, r_ in zip(("P", "Q", "b_u", "b_i"), ref):
                assert relerr(m._dev[name].cpu().numpy(), r_.astype(np.float64)) <= 1e-4, (tag, name)
        traj.append(m.device_errors())
        for _ in range(epochs - 1):
            m.run_epochs(1)
            traj.append(m.device_errors())
        runs[tag], grids[tag] = traj, (g.info.Gr, g.info.Gc, int(g.info.n_hot))
        del m, g
        torch.cuda.empty_cache()
    for tag, traj in runs.items():
        print(f"ml25m k=128 {tag:8s} grid {grids[tag][0]}x{grids[tag][1]} ({grids[tag][2]} ratings in hot segments): "
              "held-out " + " ".join(f"{te:.4f}" for _, te in traj))
    assert runs["fixed"][0] == runs["tickets"][0] == runs["pipeline"][0]      # same start
    assert runs["fixed"][-1][1] < runs["fixed"][0][1] - 0.4                   # trained: 1.059 -> ~0.608
    for a, b in (("fixed", "tickets"), ("fixed", "pipeline"), ("tickets", "pipeline")):
        for e in range(epochs - tail + 1, epochs + 1):
            assert abs(runs[a][e][1] - runs[b][e][1]) < 0.002, (a, b, e, runs[a][e], runs[b][e])
            assert abs(runs[a][e][0] - runs[b][e][0]) < 0.002, (a, b, e, runs[a][e], runs[b][e])
    # The two ways through a hot segment are the same computation (the replay above pins both to 1e-4 on the factors),
    # but their timing differs, so the free-block schedule issues the blocks in a different order: the two curves stay
    # inside north_star's 0.002 at EVERY epoch, the transient included (measured: 7e-4 at epoch 4, 1e-4 at the end).
    assert max(abs(x[1] - y[1]) for x, y in zip(runs["tickets"], runs["pipeline"])) < 0.002


def test_ml25m_k128_held_out_rmse_inside_the_reference_schedulers_envelope(ra, orc):
    """north_star: 'held-out RMSE within 0.002 of the reference', at the reference's own run length (iters = 5,
    RayBlockScheduler.py:185), epoch by epoch.

    The reference's parallel run is NOT the fixed row-major order: its scheduler hands out a free block of minimal
    update count with a random tie-break (RayBlockScheduler.py:50-58), 16 workers on a 17 x 17 grid, all passes in one
    session - a different order every run, so its held-out RMSE after e passes is a RANGE.  tests/_envelope.py measures
    that range with the CPU oracle's threaded FPSGD (float64, 6 seeds, one run of `e` passes per seed and e = 1..5).
    Against it, from the same ratings and the same initial model (the factor rows are carried into each run's own
    numbering):
      gpu2   the product's default: balanced renumbering, throughput grid, schedule 2 (all passes of a run in one
             launch, overlapped like the reference's scheduler overlaps them), three runs with different seeds;
      gpu1   the same with a launch per pass (schedule 1), cumulative epochs;
      fixed  the deterministic row-major order on the reference's 16 x 16 grid - shown to be what the reference's
             parallel run does NOT follow (0.011 off at epoch 3), i.e. the test can tell orders apart.
    Asserted: at every epoch 1..5 the band of the gpu2 runs lies within 0.002 of the reference's band.  Measured
    (profiles/r3_rmse_parity.txt): 0.0011-0.0020 at epoch 3 - the steepest epoch, held-out RMSE falls by 0.10 per epoch
    there, so 0.002 is 2 % of an epoch - and <= 0.0013 at the others; Netflix shape <= 0.0015 at epoch 1 and inside the
    reference's band from epoch 3 on.  gpu1 sits 0.0003 higher at epoch 3 and is held to 0.003."""
    import torch
    from _envelope import band_distance, envelope_bounds, s1_envelope
    epochs = 5
    # the reference's id space (D2) and the start every run shares
    m, g = _model(ra, "ml25m", renumber="random", deterministic=True)
    assert m.n_factors == 128 and (g.info.Gr, g.info.Gc) == (16, 16)
    U, I, b = m.n_users, m.n_items, m.b
    train, test = m._d_samples.cpu().numpy(), m._d_test.cpu().numpy()
    init = _host_factors(m)
    rp0 = torch.from_numpy(m.row_perm.astype(np.int64)).cuda()
    cp0 = torch.from_numpy(m.col_perm.astype(np.int64)).cuda()
    P0, Q0 = m._dev["P"].clone(), m._dev["Q"].clone()
    fixed = [m.device_errors()[1]]
    for _ in range(epochs):
        m.run_epochs(1)
        fixed.append(m.device_errors()[1])
    del m, g
    torch.cuda.empty_cache()

    def carried(m):
        """The shared initial factors in m's numbering: original id o sits at row_perm[o] there and at rp0[o] here."""
        rp = torch.from_numpy(m.row_perm.astype(np.int64)).cuda()
        cp = torch.from_numpy(m.col_perm.astype(np.int64)).cuda()
        P, Q = torch.empty_like(P0), torch.empty_like(Q0)
        P[rp] = P0[rp0]
        Q[cp] = Q0[cp0]
        return P, Q

    gpu2, gpu1 = [], []
    for seed in (0, 7, 1234):
        m, g = _model(ra, "ml25m", seed=seed)
        assert g.info.order == 1 and not m.pass_barrier and abs(m.b - b) < 1e-9
        P, Q = carried(m)
        traj = [None]
        for e in range(1, epochs + 1):          # a run of e passes from the start, like each leg of the envelope
            m._dev["P"].copy_(P); m._dev["Q"].copy_(Q); m._dev["b_u"].zero_(); m._dev["b_i"].zero_()
            if e == 1:
                traj[0] = m.device_errors()[1]
            m.run_epochs(e)
            traj.append(m.device_errors()[1])
        gpu2.append(traj)
        if seed == 0:
            m.pass_barrier = True
            m._dev["P"].copy_(P); m._dev["Q"].copy_(Q); m._dev["b_u"].zero_(); m._dev["b_i"].zero_()
            traj = [m.device_errors()[1]]
            for e in range(epochs):
                m.run_epochs(1)
                traj.append(m.device_errors()[1])
            gpu1.append(traj)
        del m, g
        torch.cuda.empty_cache()
    env = s1_envelope(orc, train, test, U, I, init, b, HP, n_threads=16, epochs=epochs, seeds=(1, 2, 3, 4, 5, 6),
                      train_rmse=False)
    lo, hi = envelope_bounds(env, 1)
    print("ml25m k=128 held-out RMSE after e passes: reference S1 band (6 seeds) | gpu2 x3 | gpu1 | fixed16")
    for e in range(epochs + 1):
        print(f"  e={e}: [{lo[e]:.5f}, {hi[e]:.5f}] | " + " ".join(f"{t[e]:.5f}" for t in gpu2) +
              f" | {gpu1[0][e]:.5f} | {fixed[e]:.5f}")
    assert abs(gpu2[0][0] - lo[0]) < 1e-5 and abs(fixed[0] - lo[0]) < 1e-5          # same start (fp32 against fp64)
    assert hi[epochs] < lo[0] - 0.2                                                  # the run trains: 1.059 -> ~0.789
    d2 = [band_distance(min(t[e] for t in gpu2), max(t[e] for t in gpu2), lo[e], hi[e]) for e in range(epochs + 1)]
    d1 = [band_distance(gpu1[0][e], gpu1[0][e], lo[e], hi[e]) for e in range(epochs + 1)]
    df = [band_distance(fixed[e], fixed[e], lo[e], hi[e]) for e in range(epochs + 1)]
    print("  distance from the reference's band: gpu2 " + " ".join(f"{x:.5f}" for x in d2) + " | gpu1 " +
          " ".join(f"{x:.5f}" for x in d1) + " | fixed16 " + " ".join(f"{x:.5f}" for x in df))
    assert max(d2) <= 0.002, d2
    assert max(d1) <= 0.003, d1
    assert max(df) > 0.005, df          # the envelope tells block orders apart: the fixed order is NOT what S1 does


# ------------------------------------------------------------------ configs 4 and 5: full-size properties
@pytest.mark.parametrize("name,k,hot_buffers", [("netflix", 128, 8), ("netflix", 128, 0), ("yahoo", 256, 8)])
def test_full_size_properties(ra, name, k, hot_buffers):
    import torch
    m, g = _model(ra, name, hot_buffers=hot_buffers)
    assert m.n_factors == k
    n = int(m._d_samples.shape[0])
    assert g.info.n == n and g.info.user_unique == 1
    d = g.export_device()
    brp = d["blk_run_ptr"].cpu().numpy()
    run_start = d["run_start"]
    assert int(run_start[-1]) == n and int(run_start[0]) == 0
    blk_ptr = run_start[torch.from_numpy(brp.astype(np.int64)).cuda()].cpu().numpy()     # rating offsets of the blocks
    sizes = np.diff(blk_ptr)
    assert sizes.min() >= 0 and sizes.sum() == n and sizes.max() == g.info.max_block
    # a sampled block (the largest): its ratings lie in the block, the hot segment comes first with one rating per
    # user, users ascend inside each segment, tickets are per-item ranks
    bsel = int(np.argmax(sizes))
    a, b_ = int(blk_ptr[bsel]), int(blk_ptr[bsel + 1])
    items = d["item"][a:b_].cpu().numpy()
    ticks = d["tick"][a:b_].cpu().numpy()
    r_, c_ = divmod(bsel, g.info.Gc)
    assert items.min() >= c_ * g.info.ibp and items.max() < (c_ + 1) * g.info.ibp
    ru = d["run_user"][int(brp[bsel]):int(brp[bsel + 1])].cpu().numpy()
    pos = 0
    for hot_item in g.col_hot()[c_]:            # segment after segment: one rating per user, users ascending
        if hot_item >= 0:
            nh = int((items == hot_item).sum())
            assert (items[pos:pos + nh] == hot_item).all() and (np.diff(ru[pos:pos + nh]) > 0).all()
            pos += nh
    # the rest: first the runs of the users who are in no hot segment of the block, then the others', each part by user
    rest = ru[pos:]
    in_seg = np.isin(rest, ru[:pos])
    n_free = int((~in_seg).sum())
    assert not np.isin(items[pos:], g.col_hot()[c_]).any() and len(np.unique(rest)) == len(rest)
    assert not in_seg[:n_free].any() and in_seg[n_free:].all()
    assert (np.diff(rest[:n_free]) > 0).all() and (np.diff(rest[n_free:]) > 0).all()
    assert ru.min() >= r_ * g.info.ubp and ru.max() < (r_ + 1) * g.info.ubp
    seen = {}
    for it, tk in zip(items.tolist(), ticks.tolist()):
        assert tk == seen.get(it, 0)
        seen[it] = tk + 1
    del d
    epochs = 3
    log = torch.full((epochs * g.info.Gr * g.info.Gc, 2), -1, dtype=torch.int32, device="cuda")
    errs = [m.device_errors()]
    for e in range(epochs):
        m.run_epochs(1, order_log=log[e * g.info.Gr * g.info.Gc:])
        errs.append(m.device_errors())          # device_errors() checks grid.status() first
    _check_order_log(log.cpu().numpy().reshape(epochs, -1, 2), g, brp)
    print(f"{name} k={k} hot_buffers={hot_buffers} grid {g.info.Gr}x{g.info.Gc}, {int(g.info.n_hot)} ratings in hot segments: " +
          "  ".join(f"{tr:.4f}/{te:.4f}" for tr, te in errs))
    for (tr0, te0), (tr1, te1) in zip(errs[:-1], errs[1:]):
        assert tr1 < tr0 and te1 < te0
    for t in (m._dev["P"], m._dev["Q"], m._dev["b_u"], m._dev["b_i"]):
        assert bool(torch.isfinite(t).all())


def test_grid_rejects_ids_in_the_padding_of_the_last_stripe(ra):
    """ids in [n_items, Gc*ibp) or [n_users, Gr*ubp) fall inside the grid's last stripes but outside the
    factor matrices - they must be refused, not followed."""
    import torch
    from recommender_amd import _abi
    from recommender_amd.SGDRecommender import Grid
    U, I, ubp, ibp = 1000, 1000, 334, 334          # 3 stripes of 334: ids 1000 and 1001 are in the padding
    ok = torch.tensor([[0, 0, 3], [999, 999, 4]], dtype=torch.int32, device="cuda")
    Grid(ok, U, I, ubp, ibp, order=1)
    for bad in ([[0, 1001, 3]], [[1000, 5, 3]], [[0, -1, 3]]):
        s = torch.cat([ok, torch.tensor(bad, dtype=torch.int32, device="cuda")])
        with pytest.raises(_abi.MfsgdError, match="outside"):
            Grid(s, U, I, ubp, ibp, order=1)
    # the test filter's `seen` tables have n_users / n_items entries: a train id beyond them is refused too
    out = torch.empty((2, 3), dtype=torch.int32, device="cuda")
    cnt = _abi.c_i64(0)
    bad_train = torch.tensor([[0, 1001, 3]], dtype=torch.int32, device="cuda")
    rc = _abi.lib().mfsgd_hip_filter_test(_abi.ptr(ok), 2, 3, _abi.ptr(bad_train), 1, 3, U, I, _abi.ptr(out),
                                          _abi.C.byref(cnt), None)
    assert rc != 0 and b"outside" in _abi.lib().mfsgd_last_error()
    # remap validates before it writes: a failed call leaves the array as it was
    s = torch.tensor([[1, 2, 3], [0, 1001, 3]], dtype=torch.int32, device="cuda")
    perm = torch.arange(1000, dtype=torch.int32, device="cuda").flip(0).contiguous()
    rc = _abi.lib().mfsgd_hip_remap(_abi.ptr(s), 2, 3, _abi.ptr(perm), _abi.ptr(perm), U, I, None)
    assert rc != 0 and s.cpu().tolist() == [[1, 2, 3], [0, 1001, 3]]

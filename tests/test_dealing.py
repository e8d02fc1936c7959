"""Host logic that deals users / items to the stripes of the block grid (recommender_amd/SGDRecommender.py:
balanced_permutation, item_costs, interleave_stripes).  It replaces the reference's random permutation
(SGDRecommender.py:228-268) by "a permutation followed by D3" of its own, so what has to hold is: it IS a permutation,
every stripe gets exactly its capacity, the loads come out near-equal, and the DSGD shards each get their share of the
most rated items.  CPU only."""
import numpy as np
import pytest

from recommender_amd.SGDRecommender import (CHAIN_RATING_COST, HOT_RATING_COST, balanced_permutation, interleave_stripes,
                                            item_costs)


def _zipf(n, s, seed, q=3.0):
    rng = np.random.default_rng(seed)
    w = (np.arange(n) + q) ** (-s)
    return rng.permutation(np.maximum(1, (w / w.sum() * 40 * n).astype(np.int64)))


@pytest.mark.parametrize("n,width", [(1000, 64), (1000, 37), (17770, 70), (64, 64), (65, 64), (5, 1)])
def test_balanced_permutation_is_a_permutation_with_full_stripes(n, width):
    deg = _zipf(n, 1.1, n)
    new_id = balanced_permutation(deg, width)
    assert sorted(new_id.tolist()) == list(range(n))
    G = -(-n // width)
    counts = np.bincount(new_id // width, minlength=G)
    assert counts[:-1].tolist() == [width] * (G - 1) and counts[-1] == n - (G - 1) * width
    # local ids of a stripe ascend with falling degree (the grid's "most rated item first" relies on nothing else, but
    # the hot item of a stripe is its local id 0)
    for s in range(G):
        members = np.flatnonzero(new_id // width == s)
        by_local = members[np.argsort(new_id[members])]
        assert (np.diff(deg[by_local]) <= 0).all()


def test_balanced_permutation_balances_loads():
    deg = _zipf(17770, 0.9, 3, q=200.0)         # a netflix-like head: the top item is a quarter of a stripe's share
    width = 70
    assert deg.max() < 0.5 * deg.sum() / (len(deg) / width)
    new_id = balanced_permutation(deg, width)
    G = -(-len(deg) // width)
    load = np.bincount(new_id // width, weights=deg, minlength=G)[:-1]      # the last stripe is narrower
    assert load.max() <= 1.05 * load.mean()
    # the G most rated items lead G different stripes
    top = np.argsort(-deg, kind="stable")[:G]
    assert len(set((new_id[top] // width).tolist())) == G
    # against the reference's random permutation: a factor of imbalance, not per cent
    rnd = np.random.default_rng(0).permutation(len(deg))
    rl = np.bincount(rnd // width, weights=deg, minlength=G)[:-1]
    assert rl.max() / rl.mean() > 1.5 * load.max() / load.mean()


def test_item_costs_weights_the_stripe_leaders_only():
    deg = _zipf(5000, 1.0, 5) * 50
    Gr, Gc = 16, 20
    order = np.argsort(-deg, kind="stable")
    c = item_costs(deg, Gr, Gc, pipeline_runs=False)
    assert np.allclose(c[order[:Gc]], CHAIN_RATING_COST * deg[order[:Gc]]) and np.array_equal(c[order[Gc:]], deg[order[Gc:]])
    c = item_costs(deg, Gr, Gc, pipeline_runs=True, hot_items=2, hot_min=8)
    lead = order[:2 * Gc]
    hot = lead[deg[lead] >= max(256, 8 * Gr)]
    assert len(hot) > 0
    assert np.allclose(c[hot], HOT_RATING_COST * deg[hot])
    rest = np.setdiff1d(np.arange(len(deg)), hot)
    assert np.array_equal(c[rest], deg[rest])
    assert item_costs(deg, Gr, Gc, True, 1, weight=1.0) is None                # nothing to re-weight
    assert item_costs(np.ones(100, np.int64), Gr, Gc, True) is None            # nobody qualifies as a hot item
    # dealing by cost: stripes with a leader that counts five-fold get fewer ratings, the COSTS come out equal
    width = 250
    by_cost = balanced_permutation(deg, width, item_costs(deg, Gr, Gc, False))
    assert sorted(by_cost.tolist()) == list(range(len(deg)))
    cost = item_costs(deg, Gr, Gc, False)
    cl = np.bincount(by_cost // width, weights=cost, minlength=Gc)
    assert cl.max() <= max(1.1 * cl.mean(), 1.05 * cost.max())      # a leader heavier than the mean cannot be balanced away


@pytest.mark.parametrize("n,width,world", [(1600, 100, 4), (1600, 100, 2), (1590, 100, 4), (1600, 100, 3), (800, 100, 8)])
def test_interleave_stripes_spreads_the_deal_over_the_shards(n, width, world):
    deg = _zipf(n, 1.0, 11)
    dealt = balanced_permutation(deg, width)
    il = interleave_stripes(dealt, width, world)
    assert sorted(il.tolist()) == list(range(n))
    G = -(-n // width)
    per = -(-G // world)
    # the members of a stripe stay together, in the same local order
    for s in range(G):
        m = np.flatnonzero(dealt // width == s)
        assert len(set((il[m] // width).tolist())) == 1 and np.array_equal(il[m] % width, dealt[m] % width)
    if G % world == 0 or (n % width != 0 and (G - 1) % world == 0 and False):
        # equal shards: stripe c of the deal lands in shard c mod world - every shard gets one of the `world` most rated items
        top = np.argsort(-deg, kind="stable")[:world]
        shards = sorted((il[top] // width // per).tolist())
        if n % width == 0:
            assert shards == list(range(world))
    # a narrower last stripe stays last (ids must stay below n)
    if n % width:
        last = np.flatnonzero(dealt // width == G - 1)
        assert (il[last] // width == G - 1).all()


def test_reorder_within_stripes_keeps_the_stripes():
    """reorder_within_stripes only renumbers inside a stripe: same stripe for every id, still a permutation; "light"
    reverses the deal's heavy-first order, "spread" alternates the heavy half with the light half."""
    from recommender_amd.SGDRecommender import reorder_within_stripes
    rng = np.random.default_rng(5)
    deg = rng.integers(1, 500, size=1003)
    width = 100
    base = balanced_permutation(deg, width)
    assert reorder_within_stripes(base, width, "heavy") is base
    for mode in ("light", "shuffle", "spread"):
        q = reorder_within_stripes(base, width, mode, seed=3)
        assert sorted(q.tolist()) == list(range(len(deg))), mode
        assert np.array_equal(q // width, base // width), mode
    # stripe 0: the deal hands out ids by decreasing degree, so slot order = decreasing degree
    in0 = np.flatnonzero(base // width == 0)
    by_slot = in0[np.argsort(base[in0])]
    assert (np.diff(deg[by_slot]) <= 0).all()
    light = reorder_within_stripes(base, width, "light")
    assert (np.diff(deg[in0[np.argsort(light[in0])]]) >= 0).all()
    spread = reorder_within_stripes(base, width, "spread")
    s_slot = spread[by_slot] % width                      # heavy half -> even slots, light half -> odd slots
    half = (len(by_slot) + 1) // 2
    assert (s_slot[:half] % 2 == 0).all() and (s_slot[half:] % 2 == 1).all()
    with pytest.raises(ValueError):
        reorder_within_stripes(base, width, "nope")

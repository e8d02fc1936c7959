"""Synthetic rating sets in the shapes BASELINE.json names, generated on the GPU.

Shared by bench.py and the `-m gpu` tests of the BASELINE configurations (tests/test_gpu_baseline_configs.py), so
that both run on exactly the same data.  Not part of the product path: recommender_amd/ never imports it.
"""

WORKLOADS = {
    # name: (n_users, n_items, n_ratings, k, (s_u, q_u), (s_i, q_i))   Zipf-Mandelbrot weights (rank+q)^-s
    "ml100k": (943, 1682, 100_000, 32, (0.6, 20.0), (0.8, 20.0)),
    "ml1m": (6040, 3706, 1_000_000, 64, (0.6, 40.0), (0.8, 30.0)),
    "ml25m": (162_541, 59_047, 25_000_000, 128, (0.8, 60.0), (1.0, 60.0)),
    "netflix": (480_189, 17_770, 100_000_000, 128, (0.7, 100.0), (1.0, 80.0)),
    "yahoo": (1_000_990, 624_961, 250_000_000, 256, (1.1, 400.0), (1.2, 300.0)),
}
BASE_SEED = 20240611
ALPHA, BETA1, BETA2 = 0.006, 0.05, 0.015   # SGDRecommender.py:33-35


def workload_seed(name, seed=BASE_SEED):
    return seed + sorted(WORKLOADS).index(name)


def synth(torch, name, seed=None, scale=1.0):
    """Seeded synthetic ratings on the GPU: distinct (u,i) pairs, Zipf-Mandelbrot user activity and item
    popularity, planted rank-8 model, integer ratings 1..5, rows sorted by user; 90/10 split.
    Returns (train[n,3] int32, test[m,3] int32, n_users, n_items, k) as device tensors / ints."""
    U, I, n, k, (su, qu), (si, qi) = WORKLOADS[name]
    if seed is None:
        seed = workload_seed(name)
    n = int(n * scale)
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)

    def sampler(N, s, q):
        w = (torch.arange(N, device="cuda", dtype=torch.float64) + 1.0 + q) ** (-s)
        cdf = torch.cumsum(w / w.sum(), 0)
        ids = torch.randperm(N, device="cuda", generator=g)
        return lambda m: ids[torch.searchsorted(cdf, torch.rand(m, device="cuda", dtype=torch.float64, generator=g)).clamp_(max=N - 1)]

    draw_u, draw_i = sampler(U, su, qu), sampler(I, si, qi)
    keys = torch.empty(0, dtype=torch.int64, device="cuda")
    want = int(n * 1.02)
    while keys.numel() < want:
        m = min(1 << 25, int((want - keys.numel()) * 1.3) + 1024)
        new = draw_u(m) * I + draw_i(m)
        keys = torch.unique(torch.cat([keys, new]))
    keep = torch.rand(keys.numel(), device="cuda", generator=g) < (n / keys.numel())
    keys = keys[keep]                                   # sorted: by user, then item
    u, i = keys // I, keys % I
    del keys, keep
    pu = torch.randn(U, 8, device="cuda", generator=g) * 0.6
    qi_ = torch.randn(I, 8, device="cuda", generator=g) * 0.6
    r = torch.empty(u.numel(), device="cuda")
    for a in range(0, u.numel(), 1 << 24):
        b = min(a + (1 << 24), u.numel())
        r[a:b] = (pu[u[a:b]] * qi_[i[a:b]]).sum(1)
    r = (3.5 + r + 0.5 * torch.randn(u.numel(), device="cuda", generator=g)).round_().clamp_(1, 5)
    s = torch.stack([u.int(), i.int(), r.int()], 1).contiguous()
    del u, i, r
    test_mask = torch.rand(s.shape[0], device="cuda", generator=g) < 0.1
    return s[~test_mask].contiguous(), s[test_mask].contiguous(), U, I, k

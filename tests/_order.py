"""The stored order of an order-1 block grid (include/mfsgd.h, mfsgd_grid_create), restated in NumPy for the tests.

Blocks row-major; inside a block first the ratings of the stripe's hot item (one per user, by user: the hot
segment), then all other ratings by user, ties in input order.  Moving a user's hot-item rating in front of OTHER
users' ratings changes nothing (updates that share neither user nor item commute exactly), so this is the
sequential pass over "each user's ratings of the block, hot item first"."""
import numpy as np

HOT_SHARE_DIV, HOT_MIN_DEG = 16, 256


def col_hot_expected(s, n_items, ibp):
    """Per item stripe: its most rated item (lowest id among equals) if that item carries at least 1/16 of the
    stripe's ratings and at least 256 of them, else -1."""
    Gc = -(-n_items // ibp)
    deg = np.bincount(s[:, 1], minlength=n_items)
    out = np.full(Gc, -1, dtype=np.int64)
    for c in range(Gc):
        d = deg[c * ibp:min((c + 1) * ibp, n_items)]
        if len(d) and d.max() >= HOT_MIN_DEG and d.max() * HOT_SHARE_DIV >= d.sum():
            out[c] = c * ibp + int(np.argmax(d))
    return out


def stored_order(s, ubp, ibp, Gc, col_hot):
    """(perm, block key of every rating, hot flag of every rating), so that s[perm] is the grid's stored order."""
    key = (s[:, 0] // ubp).astype(np.int64) * Gc + s[:, 1] // ibp
    hot = s[:, 1] == np.asarray(col_hot)[s[:, 1] // ibp]
    perm = np.lexsort((s[:, 0], ~hot, key))          # stable: block, hot first, user, input order
    return perm, key, hot

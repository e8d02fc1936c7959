"""The stored order of an order-1 block grid (include/mfsgd.h, mfsgd_grid_create / _opt), restated in NumPy for the tests.

Blocks row-major; inside a block first the ratings of the stripe's hot items, one segment per item (segment s = hot
item s, one rating per user, by user), then the other ratings: first those of the users who have NO rating in a hot
segment of this block, by user, then those of the users who have one, by user.  One user's ratings inside such a run:
most rated item first (degree over the grid's samples), ties in input order."""
import numpy as np

HOT_SHARE_DIV, HOT_MIN_DEG, MAX_HOT = 16, 256, 8


def col_hot_expected(s, n_items, ibp, items=1, min_deg=0, need_share=True):
    """[Gc][MAX_HOT]: per item stripe its `items` most rated items (lower id among equals first), each with at least 256
    ratings, at least `min_deg`, and - the rule of mfsgd_grid_create - at least 1/16 of the stripe's ratings; -1 = none
    (nothing qualifies after the first item that does not)."""
    Gc = -(-n_items // ibp)
    deg = np.bincount(s[:, 1], minlength=n_items)
    out = np.full((Gc, MAX_HOT), -1, dtype=np.int64)
    for c in range(Gc):
        d = deg[c * ibp:min((c + 1) * ibp, n_items)]
        order = np.argsort(-d, kind="stable")
        for j, i in enumerate(order[:items]):
            if d[i] >= HOT_MIN_DEG and d[i] >= min_deg and (not need_share or d[i] * HOT_SHARE_DIV >= d.sum()):
                out[c, j] = c * ibp + int(i)
            else:
                break
    return out


def stored_order(s, ubp, ibp, Gc, col_hot):
    """(perm, block key of every rating, hot flag of every rating), so that s[perm] is the grid's stored order."""
    col_hot = np.asarray(col_hot).reshape(Gc, -1)
    key = (s[:, 0] // ubp).astype(np.int64) * Gc + s[:, 1] // ibp
    cls = np.full(len(s), col_hot.shape[1], dtype=np.int64)
    for j in range(col_hot.shape[1]):
        cls[s[:, 1] == col_hot[s[:, 1] // ibp, j]] = j
    hot = cls < col_hot.shape[1]
    pair = s[:, 0].astype(np.int64) * Gc + s[:, 1] // ibp        # (user, item stripe) = the user inside one block
    cls[~hot & np.isin(pair, np.unique(pair[hot]))] += 1        # the rest of a user who is in a hot segment goes last
    deg = np.bincount(s[:, 1])
    perm = np.lexsort((-deg[s[:, 1]], s[:, 0], cls, key))   # stable: block, class, user, item degree (descending), input order
    return perm, key, hot

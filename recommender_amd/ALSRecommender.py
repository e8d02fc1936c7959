"""Host-side mirror of the reference's ALSRecommender.py: the same class name, methods
and argument order, with `als_step` running on the GPU (mfsgd_hip_als_step_*, csrc/als.hip).

Reference: `ExplicitMF(n_factors, item_reg, user_reg, verbose)` (ALSRecommender.py:26-59),
`load_samples_from_npy` (:99-114), `als_step(latent_vectors, fixed_vecs, ratings, _lambda, type)` (:116-155),
`train` / `partial_train` (:157-190), `mse` (:191-201), `predict` (:245-247).
No CPU fallback: without the HIP library and a GPU every computing call raises."""
from time import perf_counter

import numpy as np

from . import _abi
from .Recommender import CSRView, Recommender


def _csr_arrays(ratings):
    """(rowptrs, colinds, values, nrows, ncols) of a CSRView / csr.CSR-like object."""
    return ratings.rowptrs, ratings.colinds, ratings.values, int(ratings.nrows), int(ratings.ncols)


def _transpose(ratings):
    """CSR of the transposed matrix, entries of a row in the order they have in the input (what the reference
    gets from `ratings.transpose()`, ALSRecommender.py:142, up to the within-row order its package leaves open)."""
    rp, ci, v, nr, nc = _csr_arrays(ratings)
    if hasattr(rp, "is_cuda"):
        torch = _abi.torch_mod()
        rows = torch.repeat_interleave(torch.arange(nr, device=rp.device), (rp[1:] - rp[:-1]).long())
        order = torch.argsort(ci.long(), stable=True)
        trp = torch.zeros(nc + 1, dtype=torch.int64, device=rp.device)
        trp[1:] = torch.cumsum(torch.bincount(ci.long(), minlength=nc), 0)
        return CSRView(trp, rows[order].int(), v[order], nc, nr)
    rp, ci, v = np.asarray(rp), np.asarray(ci), np.asarray(v)
    rows = np.repeat(np.arange(nr), np.diff(rp))
    order = np.argsort(ci, kind="stable")
    trp = np.zeros(nc + 1, dtype=np.int64)
    np.cumsum(np.bincount(ci, minlength=nc), out=trp[1:])
    return CSRView(trp, rows[order].astype(np.int32), v[order], nc, nr)


class ExplicitMF:
    def __init__(self, n_factors=40, item_reg=0.0, user_reg=0.0, verbose=False, *, dtype="float64"):
        """ALSRecommender.py:26-59.  `dtype` is the device arithmetic ("float64" like the reference, or "float32")."""
        self.n_factors = n_factors
        self.item_reg = item_reg
        self.user_reg = user_reg
        self._v = verbose
        self.dtype = dtype
        self._ratings_T = None

    renumber_column = staticmethod(Recommender.renumber_column)        # ALSRecommender.py:63-70

    # ------------------------------------------------------------------ data
    def load_samples_from_array(self, a, n="all"):
        """The body of load_samples_from_npy (ALSRecommender.py:101-114) on an array already in memory:
        ratings doubled to the 10-point integer scale, int32, 90/10 split drawn with np.random.choice."""
        start = perf_counter()
        a = np.array(a, dtype=np.float64, copy=True)
        a[:, 2] *= 2
        a = a.astype(np.int32)
        if n != "all":
            a = a[:n]
        self.n_users = int(a[:, 0].max()) + 1
        self.n_items = int(a[:, 1].max()) + 1
        drop_indices = np.random.choice(a.shape[0], size=int(a.shape[0] / 10), replace=False)
        self.test_samples = a[drop_indices, :]
        self.samples = np.delete(a, drop_indices, axis=0)
        if self._v:
            print(f"# of train samples: {self.samples.shape[0]}, # of test samples: {self.test_samples.shape[0]}")
        self.set_ratings(self.samples)
        if self._v:
            print(f"Done loading samples in {perf_counter()-start} s.")

    def load_samples_from_npy(self, path, n):
        self.load_samples_from_array(np.load(path), n)

    def set_ratings(self, samples):
        """`self.ratings = csr.CSR.from_coo(users, items, values)` (ALSRecommender.py:113), stable inside a row."""
        s = np.asarray(samples)
        order = np.argsort(s[:, 0], kind="stable")
        rp = np.zeros(self.n_users + 1, dtype=np.int64)
        np.cumsum(np.bincount(s[:, 0], minlength=self.n_users), out=rp[1:])
        self.ratings = CSRView(rp, s[order, 1].astype(np.int32), s[order, 2].astype(np.float64), self.n_users, self.n_items)
        self._ratings_T = None

    # ------------------------------------------------------------------ the solver
    @staticmethod
    def als_step(latent_vectors, fixed_vecs, ratings, _lambda, type):
        """ALSRecommender.py:116-155.  `latent_vectors` is overwritten and returned (NumPy array or ROCm tensor);
        `ratings` is the users' CSR in both cases (the 'item' step transposes it, like the reference)."""
        _abi.require_gpu()
        if type not in ("user", "item"):
            return latent_vectors                                   # the reference falls through both branches
        r = ratings if type == "user" else _transpose(ratings)
        rp, ci, v, nr, nc = _csr_arrays(r)
        kind = "f32" if str(getattr(latent_vectors, "dtype", "")).endswith("float32") else "f64"
        td = _abi.tdtype_of(kind)
        d_lat, d_fix = _abi.to_device(latent_vectors, td), _abi.to_device(fixed_vecs, td)
        if int(d_lat.shape[0]) != nr or int(d_fix.shape[0]) < nc or int(d_lat.shape[1]) != int(d_fix.shape[1]):
            raise ValueError(f"als_step: latent {tuple(d_lat.shape)} / fixed {tuple(d_fix.shape)} do not fit a "
                             f"{nr} x {nc} ratings matrix")
        d_rp, d_ci, d_v = _abi.to_device(rp, "int64"), _abi.to_device(ci, "int32"), _abi.to_device(v, td)
        if int(d_ci.numel()) and (int(d_ci.min()) < 0 or int(d_ci.max()) >= int(d_fix.shape[0])):
            raise IndexError("als_step: a column index is outside fixed_vecs")
        fn = getattr(_abi.lib(), "mfsgd_hip_als_step_" + kind)
        _abi.check(fn(_abi.ptr(d_lat), _abi.ptr(d_fix), _abi.ptr(d_rp), _abi.ptr(d_ci), _abi.ptr(d_v), nr,
                      int(d_fix.shape[0]), int(d_ci.numel()), int(d_lat.shape[1]), float(_lambda), _abi.stream_ptr()))
        _abi.write_back(latent_vectors, d_lat)
        return latent_vectors

    def train(self, n_iter=30):
        """ALSRecommender.py:157-165: uniform [0,1) factors from the legacy global RNG, then partial_train."""
        self.user_vecs = np.random.random((self.n_users, self.n_factors)).astype("float64")
        self.item_vecs = np.random.random((self.n_items, self.n_factors)).astype("float64")
        self.user_biases = np.random.normal(size=self.n_users).astype("float64")
        self.item_biases = np.random.normal(size=self.n_items).astype("float64")
        self.partial_train(n_iter)

    def partial_train(self, n_iter):
        """ALSRecommender.py:167-190.  Factors and both CSRs stay on the GPU between the half-steps."""
        torch = _abi.torch_mod()
        td = "float32" if self.dtype == "float32" else "float64"
        X, Y = _abi.to_device(self.user_vecs, td).clone(), _abi.to_device(self.item_vecs, td).clone()
        dev = lambda r: CSRView(_abi.to_device(r.rowptrs, "int64"), _abi.to_device(r.colinds, "int32"),
                                _abi.to_device(r.values, td), r.nrows, r.ncols)
        by_user = dev(self.ratings)
        if self._ratings_T is None:
            self._ratings_T = _transpose(by_user)
        by_item = self._ratings_T
        kind = "f32" if td == "float32" else "f64"
        fn = getattr(_abi.lib(), "mfsgd_hip_als_step_" + kind)

        def half(lat, fix, r, lam):
            _abi.check(fn(_abi.ptr(lat), _abi.ptr(fix), _abi.ptr(r.rowptrs), _abi.ptr(r.colinds), _abi.ptr(r.values),
                          int(r.nrows), int(fix.shape[0]), int(r.colinds.numel()), self.n_factors, float(lam),
                          _abi.stream_ptr()))

        ctr = 1
        while ctr <= n_iter:
            half(X, Y, by_user, self.user_reg)
            if ctr % 5 == 0 and self._v:
                print("Iteration: %d ; train error after user update= %.4f" % (ctr, ExplicitMF.mse(self.samples, X, Y)))
            half(Y, X, by_item, self.item_reg)
            if ctr % 5 == 0 and self._v:
                print("Iteration: %d ; train error after item updates= %.4f" % (ctr, ExplicitMF.mse(self.samples, X, Y)))
            ctr += 1
        torch.cuda.synchronize()
        self.user_vecs, self.item_vecs = X.double().cpu().numpy(), Y.double().cpu().numpy()
        self.test_error = ExplicitMF.mse(self.test_samples, X, Y)
        if self._v:
            print("Test error = %.4f" % self.test_error)

    @staticmethod
    def mse(samples, user_vecs, item_vecs):
        """ALSRecommender.py:191-201: RMSE of the plain dot-product predictor over `samples` (user, item, rating)."""
        torch = _abi.torch_mod()
        if hasattr(user_vecs, "is_cuda"):
            s = _abi.to_device(samples, "int64")
            if s.shape[0] == 0:
                return float("nan")
            pred = (user_vecs[s[:, 0]].double() * item_vecs[s[:, 1]].double()).sum(1)
            return float(torch.sqrt(((s[:, 2].double() - pred) ** 2).mean()))
        s = np.asarray(samples)
        pred = (np.asarray(user_vecs)[s[:, 0]] * np.asarray(item_vecs)[s[:, 1]]).sum(1)
        return float(np.sqrt(np.sum(np.square(s[:, 2] - pred)) / s.shape[0]))

    def predict(self, u, i):
        """ALSRecommender.py:245-247: single user and item prediction."""
        return self.user_vecs[u, :].dot(self.item_vecs[i, :].T)

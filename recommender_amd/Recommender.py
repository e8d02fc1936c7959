"""Data-loading base class: the part of the reference's Recommender (Recommender.py:11-65) that the
blocked-SGD path uses.  The SQL loader (Recommender.py:27-48) needs a database the reference does not
ship and is out of scope."""
from time import perf_counter

import numpy as np

from . import _abi


class CSRView:
    """What the path reads from `csr.CSR` (Recommender.py:64): rowptrs / colinds / values / nnz / nrows.
    Order inside a row = input order (the real package's order is unpinned)."""

    def __init__(self, rowptrs, colinds, values, nrows, ncols):
        self.rowptrs, self.colinds, self.values = rowptrs, colinds, values
        self.nrows, self.ncols, self.nnz = nrows, ncols, len(colinds)


class Recommender:
    def __init__(self) -> None:
        pass

    # --- device residency helpers (not in the reference) ---------------------------------------
    def _upload_samples(self):
        self._d_samples = _abi.to_device(self.samples, "int32")
        self._d_test = _abi.to_device(self.test_samples, "int32")

    def _build_csr(self):
        """COO -> CSR on the device (mfsgd_hip_csr_from_coo), mirrored to the host for `self.ratings`."""
        _abi.require_gpu()
        torch = _abi.torch_mod()
        n = int(self._d_samples.shape[0])
        rp = torch.zeros(self.n_users + 1, dtype=torch.int64, device="cuda")
        ci = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda")[:n]
        _abi.check(_abi.lib().mfsgd_hip_csr_from_coo(_abi.ptr(self._d_samples), n, self._d_samples.shape[1],
                                                     self.n_users, _abi.ptr(rp), _abi.ptr(ci), _abi.stream_ptr()))
        self._d_rowptrs, self._d_colinds = rp, ci
        self._ratings = None

    @property
    def ratings(self):
        """Host mirror of the train CSR (fetched on first use)."""
        if getattr(self, "_ratings", None) is None:
            if getattr(self, "_d_rowptrs", None) is None:
                self._build_csr()
            self._ratings = CSRView(self._d_rowptrs.cpu().numpy(), self._d_colinds.cpu().numpy(), None,
                                    self.n_users, self.n_items)
        return self._ratings

    def _ensure_csr(self):
        if getattr(self, "_d_rowptrs", None) is None:
            self._build_csr()
        return self._d_rowptrs, self._d_colinds

    # --- reference API -----------------------------------------------------------------------
    @staticmethod
    def renumber_ids(values):
        """Dense ids for a column of raw ids on the device: value -> rank among the sorted distinct values
        (what Recommender.renumber_column does with a dict, Recommender.py:18-25).  Returns (ids int32, n_unique)."""
        _abi.require_gpu()
        torch = _abi.torch_mod()
        v = np.ascontiguousarray(np.asarray(values), dtype=np.int64)
        d_v = torch.from_numpy(v).cuda()
        out = torch.empty(max(len(v), 1), dtype=torch.int32, device="cuda")[:len(v)]
        m = _abi.c_i64(0)
        _abi.check(_abi.lib().mfsgd_hip_renumber_column(_abi.ptr(d_v), len(v), _abi.ptr(out), _abi.C.byref(m),
                                                        _abi.stream_ptr()))
        return out.cpu().numpy(), int(m.value)

    @staticmethod
    def renumber_column(df, column):
        """Recommender.renumber_column (Recommender.py:18-25): same call shape - a pandas DataFrame and a column
        name, the column is replaced by dense ids and the frame returned.  Integer ids only (the reference's SQL
        path also feeds strings; that path is out of scope)."""
        ids, _ = Recommender.renumber_ids(df[column].to_numpy())
        df[column] = ids.astype(np.int64)
        return df

    def load_samples_from_csv(self, path, n="all", columns=("userId", "movieId", "rating")):
        """a ratings file such as the reference's NCF/ratings.csv -> dense ids -> the same
        (n,3) int32 array load_samples_from_npy takes (ratings truncate to int like Recommender.py:54), then the
        holdout split and CSR of load_samples_from_array."""
        import pandas as pd
        df = pd.read_csv(path, usecols=list(columns))
        if n != "all":
            df = df.iloc[:n]
        u, _ = Recommender.renumber_ids(df[columns[0]].to_numpy())
        i, _ = Recommender.renumber_ids(df[columns[1]].to_numpy())
        a = np.column_stack([u, i, df[columns[2]].to_numpy()]).astype(np.int32)
        self.load_samples_from_array(a, "all")

    def load_samples_from_npy(self, path, n):
        """Recommender.py:50-65: np.load -> int32 (ratings truncate) -> first n rows -> n_users/n_items =
        max id + 1 -> random 10 % hold-out (np.random.choice, legacy global RNG) -> train CSR."""
        start = perf_counter()
        a = np.load(path)
        self.load_samples_from_array(a, n)
        print(f"Done loading samples from npz file in {perf_counter()-start} s.")

    def load_samples_from_array(self, a, n="all"):
        a = np.asarray(a).astype(np.int32)
        if n != "all":
            a = a[:n]
        self.n_users = int(a[:, 0].max()) + 1
        self.n_items = int(a[:, 1].max()) + 1
        drop_indices = np.random.choice(a.shape[0], size=int(a.shape[0] / 10), replace=False)
        self.test_samples = a[drop_indices, :]
        self.samples = np.delete(a, drop_indices, axis=0)
        print(f"# of train samples: {self.samples.shape[0]}, # of test samples: {self.test_samples.shape[0]}")
        self._upload_samples()
        self._build_csr()

    def load_samples_from_device(self, train, test, n_users, n_items):
        """Adopt int32 [n][>=3] ROCm tensors that are already split (bench / large synthetic inputs):
        nothing crosses PCIe."""
        _abi.require_gpu()
        self.n_users, self.n_items = int(n_users), int(n_items)
        self._d_samples, self._d_test = train.contiguous(), test.contiguous()
        self.samples = self.test_samples = None       # fetched lazily by host_samples()
        self._d_rowptrs = self._d_colinds = self._ratings = None   # built on first use

    def host_samples(self):
        if self.samples is None:
            self.samples = self._d_samples.cpu().numpy()
            self.test_samples = self._d_test.cpu().numpy()
        return self.samples, self.test_samples

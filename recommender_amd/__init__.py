"""recommender_amd - MI355X-native blocked-SGD matrix factorization behind the Python surface of
dtrager02/Recommender (ExplicitMF, BlockScheduler, SubSample, sgd / sgd2 / mse2).

Only the hot path lives here: csrc/ (hand-written HIP kernels + the C-ABI of include/mfsgd.h) and the
host-side mirror of the reference's interface.  There is no CPU fallback."""
from .SubSample import SubSample  # noqa: F401
from .RayBlockScheduler import BlockScheduler, consumer  # noqa: F401
from .SGDRecommender import ExplicitMF  # noqa: F401
from .Recommender import Recommender  # noqa: F401
from .numba_methods import sgd, sgd2, block_sgd, mse, mse2, rmse_exact  # noqa: F401

__all__ = ["ExplicitMF", "Recommender", "BlockScheduler", "SubSample", "consumer",
           "sgd", "sgd2", "block_sgd", "mse", "mse2", "rmse_exact"]

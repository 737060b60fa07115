"""MI355X-native batched PCT bin-packing environment (hot path only).

Exposes the reference's VecEnv surface (wrapper/vec_env.py, envs.py make_vec_envs /
VecPyTorch) over hand-written gfx950 kernels reached through the C ABI of
include/pct_env.h.  There is no CPU implementation in this package.
"""
from . import _lib  # noqa: F401  (fails loudly when libpct_hip.so is absent)
from .vec_env import PctVecEnv, PctEnvError, VecEnv, LazyInfos, make_vec_envs, evaluate_heuristic, HEURISTICS  # noqa: F401
from .sharding import shard_envs, gather_rollout  # noqa: F401
from .rollout import DeviceRollout, RolloutSlots, collect, get_leaf_nodes  # noqa: F401
from .lstsq_mode import numpy_lstsq_mode, numpy_blas  # noqa: F401

__all__ = ["PctVecEnv", "PctEnvError", "VecEnv", "LazyInfos", "make_vec_envs", "shard_envs", "gather_rollout",
           "DeviceRollout", "RolloutSlots", "collect", "get_leaf_nodes", "evaluate_heuristic", "HEURISTICS", "numpy_lstsq_mode", "numpy_blas"]

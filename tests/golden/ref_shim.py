"""Test-only shim that makes the UNMODIFIED reference importable in this container.

Used only by the golden-vector generator (tests/golden/gen_golden.py) and by the
oracle-vs-reference pinning tests that run where /root/reference exists.  Nothing in
the product path, in `-m gpu` tests, in smoke() or in bench.py imports this.

What it does:
  * injects a stub `gym` module (gym.Env, gym.Wrapper, gym.spaces.Box, registry) --
    `gym` is not installed here and the reference imports it at
    pct_envs/PctDiscrete0/bin3D.py:3, envs.py:1, tools.py:8, wrapper/monitor.py:3;
  * restores the NumPy aliases `np.float` / `np.bool` removed in NumPy >= 1.24
    (convex_hull.py:42, wrapper/shmem_vec_env.py:17, wrapper/dummy_vec_env.py:25);
  * puts /root/reference on sys.path with bytecode writing disabled so the import does
    not drop __pycache__/ into the read-only reference tree;
  * in an interpreter without PyTorch, injects a stub `torch` (the env modules import it for torch.load of datasets only).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PCT_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pct_envs"))


def install():
    import numpy as np

    sys.dont_write_bytecode = True
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "bool"):
        np.bool = bool

    # An interpreter without PyTorch (the image's second Python: /opt/conda, NumPy 1.26.4 / OpenBLAS 0.3.23, used by
    # tests/golden/check_other_numpy.py to run the reference on ANOTHER NumPy release): the env modules import torch at
    # bin3D.py:5 / binCreator.py:3 and touch it only in LoadBoxCreator (torch.load of a dataset), which scripted streams never reach
    try:
        import torch  # noqa: F401
    except ImportError:
        stub = types.ModuleType("torch")

        def _no_torch(*a, **k):
            raise RuntimeError("torch is stubbed in this interpreter (tests/golden/ref_shim.py)")
        stub.load = _no_torch
        sys.modules["torch"] = stub

    if "gym" not in sys.modules:
        gym = types.ModuleType("gym")

        class Env(object):
            metadata = {}
            reward_range = (-float("inf"), float("inf"))
            action_space = None
            observation_space = None

            def close(self):
                pass

            @property
            def unwrapped(self):
                return self

        class Wrapper(Env):
            def __init__(self, env):
                self.env = env
                self.action_space = getattr(env, "action_space", None)
                self.observation_space = getattr(env, "observation_space", None)
                self.reward_range = getattr(env, "reward_range", None)
                self.metadata = getattr(env, "metadata", None)

            def __getattr__(self, name):
                if name.startswith("_"):
                    raise AttributeError(name)
                return getattr(self.env, name)

            def step(self, action):
                return self.env.step(action)

            def reset(self, **kwargs):
                return self.env.reset(**kwargs)

            def close(self):
                return self.env.close()

            @property
            def unwrapped(self):
                return self.env.unwrapped

        class ObservationWrapper(Wrapper):
            pass

        class RewardWrapper(Wrapper):
            pass

        class ActionWrapper(Wrapper):
            pass

        class Box(object):
            def __init__(self, low=None, high=None, shape=None, dtype=None):
                self.low = low
                self.high = high
                self.shape = tuple(shape) if shape is not None else None
                self.dtype = np.dtype(dtype if dtype is not None else np.float32)

        _registry = {}

        def register(id, entry_point=None, **kwargs):
            _registry[id] = entry_point

        def make(id, **kwargs):
            entry = _registry[id]
            mod_name, cls_name = entry.split(":")
            import importlib
            mod = importlib.import_module(mod_name)
            return getattr(mod, cls_name)(**kwargs)

        spaces = types.ModuleType("gym.spaces")
        spaces.Box = Box
        spaces_box = types.ModuleType("gym.spaces.box")
        spaces_box.Box = Box
        spaces.box = spaces_box
        core = types.ModuleType("gym.core")
        core.Wrapper = Wrapper
        core.Env = Env
        envs_mod = types.ModuleType("gym.envs")
        reg_mod = types.ModuleType("gym.envs.registration")
        reg_mod.register = register
        envs_mod.registration = reg_mod

        gym.Env = Env
        gym.Wrapper = Wrapper
        gym.ObservationWrapper = ObservationWrapper
        gym.RewardWrapper = RewardWrapper
        gym.ActionWrapper = ActionWrapper
        gym.spaces = spaces
        gym.core = core
        gym.envs = envs_mod
        gym.make = make
        gym.register = register
        sys.modules["gym"] = gym
        sys.modules["gym.spaces"] = spaces
        sys.modules["gym.spaces.box"] = spaces_box
        sys.modules["gym.core"] = core
        sys.modules["gym.envs"] = envs_mod
        sys.modules["gym.envs.registration"] = reg_mod

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def load_reference_envs():
    """Returns (PackingDiscrete, PackingContinuous, item_size_set) from the reference."""
    install()
    from pct_envs.PctDiscrete0.bin3D import PackingDiscrete
    from pct_envs.PctContinuous0.bin3D import PackingContinuous
    import givenData
    return PackingDiscrete, PackingContinuous, givenData.item_size_set

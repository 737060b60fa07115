"""Host-side mirror of the reference VecEnv surface over the HIP C ABI.

`PctVecEnv` is what `envs.make_vec_envs` returns in the reference -- a `VecPyTorch` around
`ShmemVecEnv` (envs.py:75-116,159-182; wrapper/shmem_vec_env.py:20-117) -- collapsed into
one object whose N envs live on one MI355X:

  reset()            -> obs  float32 [N,(I+L+1)*9] on `device`          (envs.py:166-169)
  step_async(a)      a: float32 [N,9|6|3] leaf rows (numpy / torch, host or device;
                        train_tools.py:66-67, bin3D.py:152-153) or an int64 [N] / [N,1]
                        tensor of leaf indices (extension: no D2H of the selected row)
  step_wait()        -> (obs on device, reward float32 [N,1] on CPU, done numpy bool [N],
                         infos)                                         (envs.py:178-182)
  step(a), close(), num_envs, observation_space, action_space, reset_specific(ids)

Auto-reset: a finished env is reset inside the step; its observation is the reset
observation while reward/done/info are the terminal step's (shmem_vec_env.py:139-143).
`infos[i]` is `{'counter': n}` or, for a finished env, `{'counter','ratio','reward',
'episode': {'r','l','t'}}` (bin3D.py:163-164,186-187; wrapper/monitor.py:64-75); the
dicts are built lazily so that 65k envs do not cost 65k Python dicts per step.

Ownership: the returned `obs` tensor is the handle's output buffer -- valid until the next
step_wait()/reset(); the reference trainer copies it into its rollout storage right away
(storage.py:33-38).

Errors: what the reference raises inside a worker process (IndexError on holder overflow,
ValueError on a malformed action) is recorded per env in sticky device flags; step_wait()
raises `PctEnvError` when any flag is set (strict=True, default) or exposes them as
`error_flags`.
"""
import ctypes
import time
from abc import ABC, abstractmethod

import numpy as np
import torch

from . import _lib


class PctEnvError(RuntimeError):
    pass


class Box(object):
    """Minimal stand-in for gym.spaces.Box (bin3D.py:41-42)."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)


class VecEnv(ABC):
    """Same abstract surface as wrapper/vec_env.py:29-108."""
    closed = False
    viewer = None
    metadata = {"render.modes": ["human", "rgb_array"]}

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space

    @abstractmethod
    def reset(self):
        pass

    @abstractmethod
    def step_async(self, actions):
        pass

    @abstractmethod
    def step_wait(self):
        pass

    def close_extras(self):
        pass

    def close(self):
        if self.closed:
            return
        self.close_extras()
        self.closed = True

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    @property
    def unwrapped(self):
        return self


class LazyInfos(object):
    """Sequence of per-env info dicts, materialised on access."""

    def __init__(self, counter, ratio, done, ep_r, ep_l, ep_t):
        self._c, self._ratio, self._d = counter, ratio, done
        self._r, self._l, self._t = ep_r, ep_l, ep_t

    def __len__(self):
        return len(self._c)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not self._d[i]:
            return {"counter": int(self._c[i])}
        ratio = float(self._ratio[i])
        info = {"counter": int(self._c[i]), "ratio": ratio, "reward": ratio * 10}
        if self._r is not None:
            info["episode"] = {"r": round(float(self._r[i]), 6), "l": int(self._l[i]), "t": round(float(self._t), 6)}
        return info

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


_LNES = {"EMS": _lib.LNES_EMS, "EV": _lib.LNES_EV, "EP": _lib.LNES_EP, "CP": _lib.LNES_CP, "FC": _lib.LNES_FC}


HEURISTICS = {"LSAH": 0, "HM": 1, "OnlineBPH": 2, "DBL": 3, "BR": 4, "MACS": 5, "RANDOM": 6}  # include/pct_env.h PCT_HEUR_*


class PctVecEnv(VecEnv):
    """N independent PCT packing envs on one GPU behind the reference VecEnv surface."""

    def __init__(self, num_envs, setting=2, container_size=(10, 10, 10), item_set=None, data_name=None,
                 load_test_data=False, internal_node_holder=80, leaf_node_holder=50, LNES="EMS", shuffle=False,
                 sample_from_distribution=False, sample_left_bound=None, sample_right_bound=None,
                 device="cuda:0", seed=0, env_id_base=0, item_stream=None, continuous=False, monitor=True,
                 strict=True, ems_capacity=0, candidate_capacity=0, overflow_retry=True, rng="counter", lstsq="gelsd"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise PctEnvError("PctVecEnv runs on an AMD GPU only (device=%r); there is no CPU path" % (device,))
        if not torch.cuda.is_available():
            raise PctEnvError("no GPU visible to PyTorch-ROCm: the env hot path has no CPU fallback")
        self._dataset = None
        if load_test_data:
            if data_name is None:
                raise ValueError("load_test_data=True needs data_name (a torch.save'd list of trajectories)")
            self._dataset = [np.asarray(t, dtype=np.float64) for t in torch.load(data_name)]  # binCreator.py:48-49; [len, 3 or 4]
        self.continuous = bool(continuous)
        # continuous env: items ~ U(a,b) on the 1e-3 lattice when sample_from_distribution (C/bin3D.py:24-27,103-112;
        # bounds default to 0.1 / 0.5 of the smallest bin side, tools.py:178-181), else drawn from item_set by
        # RandomBoxCreator (C/bin3D.py:29,36-39).  Giving bounds implies sampling from the distribution.
        self._cont_from_set = False
        if self.continuous and sample_left_bound is None and sample_right_bound is None:
            if sample_from_distribution:
                sample_left_bound = 0.1 * min(container_size)
                sample_right_bound = 0.5 * min(container_size)
            elif item_set is not None and not load_test_data:
                self._cont_from_set = True
            elif not load_test_data:
                raise ValueError("the continuous env needs sample bounds (sample_from_distribution) or an item_set")
            else:
                sample_left_bound, sample_right_bound = 0.1 * min(container_size), 0.5 * min(container_size)
        if LNES not in _LNES:
            raise NotImplementedError("LNES=%r" % (LNES,))
        self._L = _lib.load()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._dev_index = dev_index
        cfg = _lib.PctConfig()
        cfg.struct_size = ctypes.sizeof(_lib.PctConfig)
        cfg.env_kind = _lib.ENV_CONTINUOUS if self.continuous else _lib.ENV_DISCRETE
        cfg.setting = int(setting)
        cfg.num_envs = int(num_envs)
        scale = 1000 if self.continuous else 1  # continuous geometry is on the 1e-3 lattice
        cfg.container[:] = [int(round(c * scale)) for c in container_size]
        cfg.internal_node_holder = int(internal_node_holder)
        cfg.leaf_node_holder = int(leaf_node_holder)
        cfg.lnes = _LNES[LNES]
        cfg.env_id_base = int(env_id_base)
        cfg.shuffle = 1 if shuffle else 0  # bin3D.py:114-115; see include/pct_env.h pct_shuffle_priority
        cfg.ems_capacity = int(ems_capacity)
        cfg.candidate_capacity = int(candidate_capacity)
        cfg.reserved[0] = 0 if overflow_retry else 1  # include/pct_env.h PCT_OVERFLOW_RETRY_*
        self._h = ctypes.c_void_p()
        with torch.cuda.device(dev_index):
            torch.cuda.init()
            _lib.check(self._L.pct_create(ctypes.byref(cfg), dev_index, ctypes.byref(self._h)))
        self.cfg = cfg
        self.N, self.I, self.Lh = int(num_envs), int(internal_node_holder), int(leaf_node_holder)
        self.row_len = (self.I + self.Lh + 1) * 9
        self.bin_size = tuple(container_size)
        self.setting = int(setting)
        self.env_id_base = int(env_id_base)
        self.strict = strict

        if self.continuous and self._cont_from_set:
            items = np.ascontiguousarray(np.rint(np.asarray(item_set, dtype=np.float64).reshape(-1, 3) * 1000).astype(np.int32))
            _lib.check(self._L.pct_set_item_set(self._h, items.ctypes.data, items.shape[0]))
            self.item_set = items
        elif self.continuous:
            _lib.check(self._L.pct_set_sample_bounds(self._h, int(round(sample_left_bound * 1000)),
                                                     int(round(sample_right_bound * 1000))))
            self.item_set = None
        else:
            if item_set is None:
                raise ValueError("item_set is required (givenData.py:10-14)")
            items = np.ascontiguousarray(np.asarray(item_set, dtype=np.int32).reshape(-1, 3))
            _lib.check(self._L.pct_set_item_set(self._h, items.ctypes.data, items.shape[0]))
            self.item_set = items
        if rng not in ("counter", "numpy"):
            raise ValueError("rng must be 'counter' (counter-keyed draws, the default) or 'numpy' (the reference's own "
                             "per-env MT19937 stream, np.random.seed(seed + rank))")
        self.rng = rng
        if self._dataset is not None:
            self.set_item_dataset(self._dataset)
        elif item_stream is not None:
            self.set_item_stream(item_stream)
        elif rng == "numpy":
            if self.continuous:
                # the RandomBoxCreator behind the sampling mode still draws (and ignores) an index into the item_set
                # the env was handed (C/bin3D.py:36-39,73,202): only its length matters
                n_set = len(np.asarray(item_set).reshape(-1, 3)) if item_set is not None else 125
                _lib.check(self._L.pct_set_numpy_item_count(self._h, int(n_set)))
            _lib.check(self._L.pct_set_numpy_rng(self._h, int(seed) & 0xFFFFFFFF))
        else:
            _lib.check(self._L.pct_set_sampler(self._h, int(seed)))

        if shuffle:
            _lib.check(self._L.pct_set_shuffle_seed(self._h, int(seed)))
        # the solver behind np.linalg.lstsq in the stability check (settings 1 / 3): "gelsd" (default since round 5: LAPACK dgelsd as the
        # reference's NumPy executes it on AVX-512 hosts -- bit-identical to the reference) or "jacobi" (the stand-in of rounds 1-4: the
        # reference's solution up to the last bits, ~1.35 x the throughput) -- pct_env.h
        # ("gelsd_avx2": as NumPy executes it on AVX2 hosts, AMD Zen included -- OpenBLAS' other kernel set)
        # ("numpy": whichever of the two this process's NumPy runs on this host -- lstsq_mode.py; for comparing against a reference
        # that runs in the same Python)
        modes = {"jacobi": _lib.LSTSQ_JACOBI, "gelsd": _lib.LSTSQ_GELSD, "gelsd_avx2": _lib.LSTSQ_GELSD_AVX2}
        if lstsq == "numpy":
            from .lstsq_mode import numpy_lstsq_mode
            lstsq = numpy_lstsq_mode(strict=bool(strict))  # (a BLAS outside the checked releases: an error under strict, else a warning)
            if lstsq is None:
                raise RuntimeError("lstsq='numpy': this process's NumPy runs a BLAS kernel set that is not restated (lstsq_mode.py)")
        if lstsq not in modes:
            raise ValueError("lstsq must be 'jacobi', 'gelsd', 'gelsd_avx2' or 'numpy'")
        self.lstsq = lstsq
        _lib.check(self._L.pct_set_lstsq_mode(self._h, modes[lstsq]))
        # outputs live in torch tensors bound into the handle (zero copy)
        dev = self.device
        self._obs = torch.zeros(self.N, self.row_len, dtype=torch.float32, device=dev)
        # the small per-step outputs live in ONE device block (ratio f64 | reward f32 | counter i32 | flags u32 | done u8),
        # so that step_wait hands them to the host with a single copy into a pinned mirror of the same layout
        N = self.N
        offs, o = {}, 0
        for nm, dt, w in (("ratio", torch.float64, 8), ("reward", torch.float32, 4), ("counter", torch.int32, 4),
                          ("flags", torch.int32, 4), ("done", torch.uint8, 1)):
            offs[nm] = (o, o + N * w, dt)
            o += N * w
        self._pack = torch.zeros(o, dtype=torch.uint8, device=dev)
        self._h_pack = torch.zeros(o, dtype=torch.uint8).pin_memory()
        self._pack_offs = offs
        self._h_ring = None  # two more pinned mirrors + events, made on the first step_outputs_async()
        view = lambda buf, nm: buf[offs[nm][0]:offs[nm][1]].view(offs[nm][2])
        self._ratio, self._reward, self._counter = view(self._pack, "ratio"), view(self._pack, "reward"), view(self._pack, "counter")
        self._flags, self._done = view(self._pack, "flags"), view(self._pack, "done")
        self._h_ratio, self._h_reward, self._h_counter = view(self._h_pack, "ratio"), view(self._h_pack, "reward"), view(self._h_pack, "counter")
        self._h_flags, self._h_done = view(self._h_pack, "flags"), view(self._h_pack, "done")
        self._own = (self._obs, self._reward)
        self._slot_keepalive = None
        self._bind_own_views()
        self._actions_keepalive = None
        self.waiting_step = False

        # Monitor emulation (wrapper/monitor.py:51-77)
        self._monitor = monitor
        self._ep_r = np.zeros(self.N, np.float64)
        self._ep_l = np.zeros(self.N, np.int64)
        self._tstart = time.time()

        obs_space = Box(low=0.0, high=float(self.bin_size[2]), shape=(self.row_len,))
        VecEnv.__init__(self, self.N, obs_space, None)

    # ------------------------------------------------------------------ item sources
    def set_item_stream(self, items):
        items = np.ascontiguousarray(np.asarray(items, dtype=np.int32))
        if items.ndim != 3 or items.shape[0] != self.N or items.shape[2] != 3:
            raise ValueError("item_stream must be int [num_envs, T, 3]")
        _lib.check(self._L.pct_set_item_stream(self._h, items.ctypes.data, items.shape[1]))

    def set_item_dataset(self, trajectories):
        """The reference's dataset format (README.md:75-77, binCreator.py:41-72): a list of
        trajectories, each [len,3] item sizes (the continuous env takes bin units and stores them
        on its 1e-3 lattice) or, for setting 3, [len,4] = size + density (bin3D.py:76).  Episode k
        (1-based) plays trajectory k."""
        scale = 1000 if self.continuous else 1
        n = len(trajectories)
        max_len = max(len(t) for t in trajectories)
        items = np.zeros((n, max_len, 3), np.int32)
        dens = np.ones((n, max_len), np.float64)
        lengths = np.zeros(n, np.int32)
        have_density = False
        for i, t in enumerate(trajectories):
            a = np.asarray(t, dtype=np.float64).reshape(len(t), -1)
            items[i, :len(a)] = np.rint(a[:, :3] * scale).astype(np.int32)
            if a.shape[1] > 3:
                dens[i, :len(a)] = a[:, 3]
                have_density = True
            lengths[i] = len(a)
        _lib.check(self._L.pct_set_item_dataset(self._h, items.ctypes.data, lengths.ctypes.data, n, max_len))
        if self.setting == 3:
            if not have_density:
                raise ValueError("setting 3 reads the density from the dataset items' fourth column (bin3D.py:76)")
            _lib.check(self._L.pct_set_dataset_density(self._h, dens.ctypes.data))

    def set_density_stream(self, den):
        """setting 3 with scripted densities: den float64 [N,T]; the env's c-th observation (counting
        every observation it ever produced) shows den[e, c % T].  Without it the densities come from
        the counter-based pct_density(seed, env, c)."""
        den = np.ascontiguousarray(np.asarray(den, dtype=np.float64))
        if den.ndim != 2 or den.shape[0] != self.num_envs:
            raise ValueError("density stream must be [num_envs, T]")
        _lib.check(self._L.pct_set_density_stream(self._h, den.ctypes.data, den.shape[1]))

    def set_sampler(self, seed):
        _lib.check(self._L.pct_set_sampler(self._h, int(seed)))

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def error_flags(self):
        """Sticky per-env PCT_FLAG_* ERROR bits (synchronises the stream); the non-fatal notice PCT_FLAG_ILL_CONDITIONED
        is reported by `ill_conditioned` instead."""
        return self._flags.cpu().numpy().view(np.uint32) & np.uint32(_lib.FLAG_ERROR_MASK)

    @property
    def ill_conditioned(self):
        """bool [N]: the env has taken a >= 3-supporter load split whose rank decision lay within a factor 1000 of the
        least-squares cut (PCT_FLAG_ILL_CONDITIONED, sticky): from there on the reference's own verdicts depend on its
        LAPACK build and may differ."""
        return (self._flags.cpu().numpy().view(np.uint32) & np.uint32(_lib.FLAG_ILL_CONDITIONED)) != 0

    @property
    def ill_commit(self):
        """bool [N]: ... and a solve of the env's own COMMIT walks raised it (PCT_FLAG_ILL_COMMIT): the part of the notice that does
        not depend on the order in which a candidate's virtual check examines its supporters."""
        return (self._flags.cpu().numpy().view(np.uint32) & np.uint32(_lib.FLAG_ILL_COMMIT)) != 0

    # ------------------------------------------------------------------ VecEnv surface
    def reset(self):
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.pct_reset(self._h, None, 0, self._stream()))
        self.waiting_step = False
        self._ep_r[:] = 0
        self._ep_l[:] = 0
        return self._obs

    def reset_specific(self, indexs):
        ids = torch.as_tensor(np.asarray(indexs, dtype=np.int32), device=self.device)
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.pct_reset(self._h, ids.data_ptr(), ids.numel(), self._stream()))
        self._ids_keepalive = ids
        idx = np.asarray(indexs, dtype=np.int64)
        self._ep_r[idx] = 0
        self._ep_l[idx] = 0
        return self._obs[ids.long()]

    def step_async(self, actions):
        if isinstance(actions, np.ndarray):
            actions = torch.from_numpy(actions)
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions))
        if actions.shape[0] != self.N:  # shmem_vec_env.py:71
            raise AssertionError("expected %d actions, got %d" % (self.N, actions.shape[0]))
        with torch.cuda.device(self._dev_index):
            if actions.dtype in (torch.int64, torch.int32) and (actions.dim() == 1 or actions.shape[-1] == 1):
                idx = actions.reshape(self.N).to(device=self.device, dtype=torch.int64).contiguous()
                self._actions_keepalive = idx
                _lib.check(self._L.pct_step_index(self._h, idx.data_ptr(), self._stream()))
            else:
                rows = actions.to(device=self.device, dtype=torch.float32).contiguous()
                if rows.dim() != 2 or rows.shape[1] not in (9, 6, 3):
                    raise ValueError("actions must be [N,9], [N,6] or [N,3] leaf rows, or [N] leaf indices")
                self._actions_keepalive = rows
                _lib.check(self._L.pct_step_rows(self._h, rows.data_ptr(), rows.shape[1], self._stream()))
        self.waiting_step = True

    def step_hash_policy(self, n_steps=1):
        """n_steps transitions with the on-device stand-in policy (benchmarks / parity)."""
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.pct_step_hash_policy(self._h, int(n_steps), self._stream()))
        self.waiting_step = True

    def step_heuristic(self, name, n_steps=1):
        """n_steps transitions with a heuristic baseline of heuristic.py as the in-env policy
        (`name` in HEURISTICS: LSAH, HM, OnlineBPH, DBL, BR, MACS, RANDOM); follow with step_wait()."""
        if self.rng == "numpy":
            raise PctEnvError("the heuristic policies are not available in strict NumPy-stream mode (rng='numpy')")
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.pct_step_heuristic(self._h, HEURISTICS[name], int(n_steps), self._stream()))
        self.waiting_step = True

    def current_obs(self):
        """The handle's observation buffer (valid until the next transition)."""
        return self._obs

    def step_device(self, leaf_index):
        """Device-resident step: `leaf_index` int64 [N] / [N,1] on the
        device; returns (obs, reward float32 [N,1], mask = 1 - done float32 [N,1]) as DEVICE
        tensors with no host synchronisation and no info dicts (terminal statistics stay
        readable through `terminal_stats()`).  Replaces the D2H of the selected row, the host
        `done` round trip and the CPU reward tensor of train_tools.py:66-70 / envs.py:181."""
        idx = leaf_index.reshape(self.N).to(device=self.device, dtype=torch.int64).contiguous()
        self._actions_keepalive = idx
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.pct_step_index(self._h, idx.data_ptr(), self._stream()))
        return self._obs, self._reward.unsqueeze(1), (1 - self._done.to(torch.float32)).unsqueeze(1)

    def step_into(self, leaf_index, obs_next, reward=None, mask=None):
        """Fused rollout edge (storage.py:33-39 + train_tools.py:66-70 in ONE launch): steps every env with the
        device int64 leaf index and has the transition kernel write the new observation (every row) straight
        into `obs_next` (float32, N * (I+L+1) * 9 contiguous elements, e.g. rollout.obs[t + 1]), the reward into
        `reward` (float32 [N] / [N,1]) and 1 - done into `mask` (float32 [N] / [N,1]).  No host synchronisation,
        no copy kernels; `current_obs()` then refers to `obs_next`."""
        idx = leaf_index.reshape(self.N).to(device=self.device, dtype=torch.int64).contiguous()
        for t in (obs_next, reward, mask):
            if t is not None and not (t.is_contiguous() and t.dtype == torch.float32 and t.device == self.device):
                raise ValueError("rollout slots must be contiguous float32 tensors on the env's device")
        if obs_next.numel() != self.N * self.row_len:
            raise ValueError("obs_next must hold N x (I+L+1) x 9 floats")
        self._actions_keepalive = idx
        # the binding outlives this step (pct_bind_rollout_slot: until the next bind): later plain steps keep writing the
        # observation / reward / mask into these tensors, so they stay alive in their own attribute until
        # unbind_rollout_slot() or the next step_into()
        self._slot_keepalive = (obs_next, reward, mask)
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.pct_bind_rollout_slot(self._h, obs_next.data_ptr(),
                                                     reward.data_ptr() if reward is not None else None,
                                                     mask.data_ptr() if mask is not None else None))
            _lib.check(self._L.pct_step_index(self._h, idx.data_ptr(), self._stream()))
        self._obs = obs_next.view(self.N, self.row_len)
        if reward is not None:
            self._reward = reward.view(self.N)
        return self._obs

    def _bind_own_views(self):
        """(re)binds the handle's outputs to this object's own tensors"""
        self._obs, self._reward = self._own
        _lib.check(self._L.pct_bind_outputs(self._h, self._obs.data_ptr(), self._reward.data_ptr(),
                                            self._done.data_ptr(), self._counter.data_ptr(), self._ratio.data_ptr(),
                                            self._flags.data_ptr()))

    def unbind_rollout_slot(self):
        """Back to the handle's own observation / reward buffers (and no mask output) after a run of step_into()
        calls: the next transition rewrites every observation row there.  RolloutSlots may be dropped afterwards."""
        with torch.cuda.device(self._dev_index):
            # the live observation / reward are in the slot: carry them over, so that current_obs() and a policy that
            # reads it after the unbind see the env's CURRENT leaf list
            if self._obs.data_ptr() != self._own[0].data_ptr():
                self._own[0].copy_(self._obs.view_as(self._own[0]))
            if self._reward.data_ptr() != self._own[1].data_ptr():
                self._own[1].copy_(self._reward.view_as(self._own[1]))
            torch.cuda.current_stream(self.device).synchronize()
        self._bind_own_views()
        self._slot_keepalive = None

    def terminal_stats(self):
        """(done bool [N], counter int32 [N], ratio float64 [N]) of the last step (one sync)."""
        return self._done.bool().cpu().numpy(), self._counter.cpu().numpy(), self._ratio.cpu().numpy()

    def policy_hash_rows(self, out=None):
        """Stand-in policy as its own kernel: float32 [N,9] leaf rows on the device."""
        if out is None:
            out = torch.empty(self.N, 9, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.pct_policy_hash_rows(self._h, out.data_ptr(), self._stream()))
        return out

    def policy_hash_index(self, out=None):
        """The stand-in policy as an INDEX tensor in one launch (pct_policy_hash_index): int64 [N] leaf indices
        pct_mix32(global id, t) % k over the k valid leaves of the current observation -- what a trained policy hands
        `step(index)` / `RolloutSlots.step_env` (train_tools.py:63-66: `selected_leaf_node` is gathered by this index).
        `out`: a contiguous int64 [N] (or [N,1]) device tensor to write into, e.g. the rollout's actions[t]."""
        if out is None:
            out = torch.empty(self.N, dtype=torch.int64, device=self.device)
        if not (out.is_contiguous() and out.dtype == torch.int64 and out.device == self.device and out.numel() == self.N):
            raise ValueError("out must be a contiguous int64 [N] tensor on the env's device")
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.pct_policy_hash_index(self._h, out.data_ptr(), self._stream()))
        return out

    def bind_policy_rows(self, rows):
        """Stand-in policy as an EPILOGUE of every following launch (pct_bind_policy_rows): the transition kernel also
        writes, for the observation it has just produced, the float32 [N,9] leaf row `policy_hash_rows` would gather
        from it into `rows` -- `step_rows_device(rows)` can then follow `step_rows_device(rows)` with no policy
        dispatch in between.  None switches it off."""
        if rows is not None and not (rows.is_contiguous() and rows.dtype == torch.float32 and rows.device == self.device
                                     and rows.numel() == self.N * 9):
            raise ValueError("rows must be a contiguous float32 [N,9] tensor on the env's device")
        self._policy_rows_keepalive = rows
        _lib.check(self._L.pct_bind_policy_rows(self._h, rows.data_ptr() if rows is not None else None))

    def step_rows_device(self, rows):
        """Enqueue one step from device-resident float32 [N,9|6|3] rows; no host work."""
        with torch.cuda.device(self._dev_index):
            _lib.check(self._L.pct_step_rows(self._h, rows.data_ptr(), rows.shape[1], self._stream()))
        self.waiting_step = True

    def profile_enable(self, on=True):
        """True / 1: every transition launch carries the HIP event pair; K > 1: every K-th one; False: none."""
        _lib.check(self._L.pct_profile_enable(self._h, int(on)))

    def profile_read(self):
        """(launches, total_ms) of the transition kernels since the last read (HIP events
        recorded by the library on the launch stream)."""
        n, ms = ctypes.c_int64(), ctypes.c_double()
        _lib.check(self._L.pct_profile_read(self._h, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    def phase_timing(self, on=True):
        """Start/stop per-phase cycle accounting; returns the uint64 [N,44] gathered so far
        (columns: load, drop, genems, set, feas, obs, store, steps, set-gen, set-dedup, set-match, set-rebuild, set statistics
        12..29, stability counters 30..43: csrc/pct_set.cuh)."""
        out = np.zeros((self.N, int(self._L.pct_debug_timing_slots())), np.uint64)  # (the library's PCT_TIMING_SLOTS: 44)
        _lib.check(self._L.pct_debug_phase_timing(self._h, int(bool(on)), out.ctypes.data))
        return out

    def step_outputs_async(self):
        """The small per-step outputs WITHOUT a stream synchronisation: enqueues the packed D2H copy of the step just launched into
        one of two pinned mirrors and returns a ticket; `ticket.wait()` -> (reward [N,1] CPU, done bool [N], infos) blocks only
        until THAT copy has landed.  A caller whose next action needs only the device-resident observation (the reference's trainer:
        train_tools.py:63-79 reads reward / done / infos for its episode statistics, the policy reads obs) launches step t + 1 first
        and consumes step t's ticket while the GPU works: the host's share of a step is hidden behind the kernel.  At most two tickets
        may be outstanding."""
        if self._h_ring is None:
            self._h_ring = [torch.zeros_like(self._h_pack).pin_memory() for _ in range(2)]
            self._ring_ev = [torch.cuda.Event() for _ in range(2)]
            self._ring_i = 0
            self._ticket_issued = 0    # tickets handed out / consumed so far: ticket t lives in buffer t % 2 and must be
            self._ticket_consumed = 0  # waited in order (the monitor's episode sums are accumulated at wait() time)
        if self._ticket_issued - self._ticket_consumed >= 2:
            raise PctEnvError("step_outputs_async: two tickets are outstanding -- wait() the older one first (its pinned buffer "
                              "would be overwritten)")
        buf, ev = self._h_ring[self._ring_i], self._ring_ev[self._ring_i]
        self._ring_i ^= 1
        seq = self._ticket_issued
        self._ticket_issued += 1
        buf.copy_(self._pack, non_blocking=True)
        if self._reward.data_ptr() != self._own[1].data_ptr():  # a rollout slot holds the reward (step_into)
            o = self._pack_offs["reward"]
            buf[o[0]:o[1]].view(o[2]).copy_(self._reward.reshape(-1), non_blocking=True)
        ev.record(torch.cuda.current_stream(self.device))
        self.waiting_step = False
        return _StepTicket(self, buf, ev, seq)

    def _outputs_from(self, buf):
        offs = self._pack_offs
        view = lambda nm: buf[offs[nm][0]:offs[nm][1]].view(offs[nm][2])
        reward = view("reward").clone().unsqueeze(1)
        done = view("done").numpy().astype(bool)
        counter = view("counter").numpy().copy()
        ratio = view("ratio").numpy().copy()
        if self.strict:
            f = view("flags").numpy().view(np.uint32) & np.uint32(_lib.FLAG_ERROR_MASK)
            if f.any():
                bad = int(np.nonzero(f)[0][0])
                raise PctEnvError("env %d raised error flags 0x%x (include/pct_env.h PCT_FLAG_*)" % (bad, int(f[bad])))
        ep_r = ep_l = None
        if self._monitor:
            self._ep_r += reward[:, 0].numpy()
            self._ep_l += 1
            ep_r, ep_l = self._ep_r.copy(), self._ep_l.copy()
            self._ep_r[done] = 0
            self._ep_l[done] = 0
        return reward, done, LazyInfos(counter, ratio, done, ep_r, ep_l, time.time() - self._tstart)

    def step_wait(self):
        # ONE async D2H of the packed output block, then a single stream sync (envs.py:178-182)
        self._h_pack.copy_(self._pack, non_blocking=True)
        if self._reward.data_ptr() != self._own[1].data_ptr():  # a rollout slot holds the reward (step_into)
            self._h_reward.copy_(self._reward, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        self.waiting_step = False
        reward = self._h_reward.clone().unsqueeze(1)
        done = self._h_done.numpy().astype(bool)
        counter = self._h_counter.numpy().copy()
        ratio = self._h_ratio.numpy().copy()
        if self.strict:
            f = self._h_flags.numpy().view(np.uint32) & np.uint32(_lib.FLAG_ERROR_MASK)
            if f.any():
                bad = int(np.nonzero(f)[0][0])
                raise PctEnvError("env %d raised error flags 0x%x (include/pct_env.h PCT_FLAG_*)" % (bad, int(f[bad])))
        ep_r = ep_l = None
        if self._monitor:
            self._ep_r += reward[:, 0].numpy()
            self._ep_l += 1
            ep_r, ep_l = self._ep_r.copy(), self._ep_l.copy()
            self._ep_r[done] = 0
            self._ep_l[done] = 0
        infos = LazyInfos(counter, ratio, done, ep_r, ep_l, time.time() - self._tstart)
        return self._obs, reward, done, infos

    def debug_work_keys(self):
        """uint32 [N]: cycles of every env's last step / 256 << 12 | live EMS count (the heavy-first dispatch's keys)."""
        out = np.zeros(self.N, np.uint32)
        _lib.check(self._L.pct_debug_work_keys(self._h, out.ctypes.data))
        return out

    def debug_retry_count(self, totals=False):
        """envs the last launch handed to the large-capacity retry pass; with `totals` also (envs re-run since creation,
        launches in which the retry pass found work)."""
        last, envs, launches = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self._L.pct_debug_retry_count(self._h, ctypes.byref(last), ctypes.byref(envs), ctypes.byref(launches)))
        return (last.value, envs.value, launches.value) if totals else last.value

    def debug_state(self, e, cap_ems=1024):
        if self.continuous:
            ems = np.zeros((cap_ems, 6), np.float64)
            n_ems, n_boxes, cur = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
            nxt = np.zeros(3, np.float64)
            _lib.check(self._L.pct_debug_state_f64(self._h, int(e), ems.ctypes.data, cap_ems, ctypes.byref(n_ems),
                                                   ctypes.byref(n_boxes), nxt.ctypes.data, ctypes.byref(cur)))
            return dict(ems=ems[:n_ems.value].copy(), n_boxes=n_boxes.value, next_item=nxt, cursor=cur.value)
        A = max(self.bin_size[0], self.bin_size[1])
        hm = np.zeros(A * A, np.int32)
        ems = np.zeros((cap_ems, 6), np.int32)
        n_ems, n_boxes, cur = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
        nxt = np.zeros(3, np.int32)
        _lib.check(self._L.pct_debug_state(self._h, int(e), hm.ctypes.data, ems.ctypes.data, cap_ems,
                                           ctypes.byref(n_ems), ctypes.byref(n_boxes), nxt.ctypes.data,
                                           ctypes.byref(cur)))
        return dict(heightmap=hm.reshape(A, A), ems=ems[:n_ems.value].copy(), n_boxes=n_boxes.value,
                    next_item=nxt, cursor=cur.value)

    def close_extras(self):
        if self._h:
            torch.cuda.synchronize(self.device)
            self._L.pct_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _StepTicket(object):
    """What PctVecEnv.step_outputs_async returns: the packed outputs of one step on their way to a pinned host buffer."""

    def __init__(self, env, buf, ev, seq):
        self._env, self._buf, self._ev, self._seq = env, buf, ev, seq

    def wait(self):
        env = self._env
        if self._seq < env._ticket_consumed:
            raise PctEnvError("step ticket %d was already consumed" % self._seq)
        if self._seq != env._ticket_consumed:
            raise PctEnvError("step tickets must be waited in order: ticket %d is next, this is %d" % (env._ticket_consumed, self._seq))
        self._ev.synchronize()
        out = env._outputs_from(self._buf)
        env._ticket_consumed += 1
        return out


def make_vec_envs(args, log_dir=None, allow_early_resets=True):
    """Drop-in for envs.make_vec_envs (envs.py:75-116): same `args` namespace
    (tools.py:130-198), returns the object the trainer steps (train_tools.py:39,67)."""
    kind = str(getattr(args, "id", "PctDiscrete-v0"))
    return PctVecEnv(
        num_envs=args.num_processes,
        setting=args.setting,
        container_size=args.container_size,
        item_set=args.item_size_set,
        data_name=getattr(args, "dataset_path", None) if getattr(args, "load_dataset", False) else None,
        load_test_data=getattr(args, "load_dataset", False),
        internal_node_holder=args.internal_node_holder,
        leaf_node_holder=args.leaf_node_holder,
        LNES=getattr(args, "lnes", "EMS"),
        shuffle=getattr(args, "shuffle", False),
        sample_from_distribution=getattr(args, "sample_from_distribution", False),
        sample_left_bound=getattr(args, "sample_left_bound", None),
        sample_right_bound=getattr(args, "sample_right_bound", None),
        device=getattr(args, "device", "cuda:0"),
        seed=getattr(args, "seed", 0),
        continuous=kind.startswith("PctContinuous"),
        item_stream=getattr(args, "item_stream", None),
        rng=getattr(args, "rng", "counter"),  # "numpy": the reference's own per-env MT19937 stream (discrete env)
        # the stability settings' np.linalg.lstsq: "gelsd" = the reference's own (default), "gelsd_avx2" / "numpy" = as the NumPy of an
        # AVX2 host / of this process executes it, "jacobi" = the faster stand-in of rounds 1-4
        lstsq=getattr(args, "lstsq", "gelsd"),
    )


def evaluate_heuristic(env, name, episodes):
    """heuristic.py's evaluation loop on the batched env: runs heuristic `name` until every env has finished its
    quota of ceil(episodes / num_envs) episodes -- each env contributes the same number, its FIRST ones, so short
    episodes are not over-represented the way "first `episodes` completions overall" would -- and returns (mean
    utilisation, variance of the utilisation, mean number of packed items) over those, what
    heuristic.py:226,298,425,498,569 return for one env.  Observations are not read."""
    quota = -(-int(episodes) // env.num_envs)
    got = np.zeros(env.num_envs, np.int64)
    util, length = [], []
    env.reset()
    while (got < quota).any():
        env.step_heuristic(name, 1)
        _, _, done, infos = env.step_wait()
        for i in np.nonzero(done & (got < quota))[0]:
            util.append(infos[i]["ratio"])
            length.append(infos[i]["counter"])
            got[i] += 1
    util, length = np.asarray(util), np.asarray(length)
    return float(util.mean()), float(util.var()), float(length.mean())

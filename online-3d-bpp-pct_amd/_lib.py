"""ctypes binding of the C ABI declared in include/pct_env.h (libpct_hip.so).

There is no CPU path behind this module: if the library is missing or does not load, the
import of the product fails loudly.  `import torch` happens BEFORE the dlopen on purpose:
PyTorch-ROCm ships its own libamdhip64.so.7 and the dynamic linker then resolves our
dependency on that SONAME to the copy torch already loaded, so device pointers and
hipStream_t handles are shared between torch and the kernels.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the dlopen, see above)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCT_HIP_LIB") or os.path.join(HERE, "libpct_hip.so")  # PCT_HIP_LIB: kernel experiments only

PCT_OK = 0
ENV_DISCRETE, ENV_CONTINUOUS = 0, 1
LNES_EMS, LNES_EV, LNES_EP, LNES_CP, LNES_FC = 0, 1, 2, 3, 4
FLAG_INTERNAL_OVERFLOW, FLAG_EMS_OVERFLOW, FLAG_CANDIDATE_OVERFLOW, FLAG_BAD_ACTION = 1, 2, 4, 8
FLAG_STABILITY_OVERFLOW, FLAG_DATASET_EXHAUSTED = 16, 32
FLAG_ILL_CONDITIONED = 64  # non-fatal notice (include/pct_env.h)
LSTSQ_JACOBI, LSTSQ_GELSD, LSTSQ_GELSD_AVX2 = 0, 1, 2  # pct_set_lstsq_mode
FLAG_ILL_COMMIT = 128  # non-fatal: ... raised by a solve of a commit walk
FLAG_ERROR_MASK = 0xFFFFFFFF & ~(FLAG_ILL_CONDITIONED | FLAG_ILL_COMMIT)

# every symbol include/pct_env.h declares
ABI_SYMBOLS = [
    "pct_abi_version", "pct_last_error", "pct_create", "pct_destroy", "pct_set_item_set",
    "pct_set_sample_bounds", "pct_set_item_stream", "pct_set_item_dataset", "pct_set_sampler", "pct_set_numpy_rng", "pct_set_numpy_item_count",
    "pct_set_shuffle_seed", "pct_set_lstsq_mode",
    "pct_set_density_stream", "pct_set_dataset_density", "pct_bind_outputs", "pct_bind_rollout_slot", "pct_obs",
    "pct_reward", "pct_done", "pct_info_counter", "pct_info_ratio", "pct_error_flags", "pct_obs_row_len",
    "pct_reset", "pct_step_rows", "pct_step_index", "pct_step_hash_policy", "pct_step_heuristic", "pct_debug_state",
    "pct_policy_hash_rows", "pct_bind_policy_rows", "pct_profile_enable", "pct_profile_read", "pct_debug_phase_timing", "pct_debug_state_f64",
    "pct_debug_work_keys", "pct_debug_retry_count", "pct_debug_timing_slots", "pct_policy_hash_index",
]


class PctConfig(ctypes.Structure):
    """`pct_config` of include/pct_env.h."""
    _fields_ = [
        ("struct_size", ctypes.c_int32),
        ("env_kind", ctypes.c_int32),
        ("setting", ctypes.c_int32),
        ("num_envs", ctypes.c_int32),
        ("container", ctypes.c_int32 * 3),
        ("internal_node_holder", ctypes.c_int32),
        ("leaf_node_holder", ctypes.c_int32),
        ("lnes", ctypes.c_int32),
        ("env_id_base", ctypes.c_int32),
        ("ems_capacity", ctypes.c_int32),
        ("candidate_capacity", ctypes.c_int32),
        ("shuffle", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 3),
    ]


_LIB = None


def load():
    """dlopen libpct_hip.so and declare the prototypes.  Raises if it is not there."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; "
            "g.build()'`).  There is no CPU fallback for the env hot path." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64
    L.pct_abi_version.restype = ctypes.c_int
    L.pct_last_error.restype = ctypes.c_char_p
    L.pct_create.argtypes = [ctypes.POINTER(PctConfig), ctypes.c_int, ctypes.POINTER(vp)]
    L.pct_destroy.argtypes = [vp]
    L.pct_set_item_set.argtypes = [vp, vp, i32]
    L.pct_set_sample_bounds.argtypes = [vp, i32, i32]
    L.pct_set_item_stream.argtypes = [vp, vp, i64]
    L.pct_set_item_dataset.argtypes = [vp, vp, vp, i32, i32]
    L.pct_set_sampler.argtypes = [vp, u64]
    L.pct_set_numpy_rng.argtypes = [vp, ctypes.c_uint32]
    L.pct_set_numpy_item_count.argtypes = [vp, ctypes.c_int32]
    L.pct_set_shuffle_seed.argtypes = [vp, u64]
    L.pct_set_lstsq_mode.argtypes = [vp, i32]
    L.pct_step_heuristic.argtypes = [vp, i32, i32, vp]
    L.pct_set_density_stream.argtypes = [vp, vp, i64]
    L.pct_set_dataset_density.argtypes = [vp, vp]
    L.pct_bind_outputs.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.pct_bind_rollout_slot.argtypes = [vp, vp, vp, vp]
    for name in ("pct_obs", "pct_reward", "pct_done", "pct_info_counter", "pct_info_ratio", "pct_error_flags"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = vp
    L.pct_obs_row_len.argtypes = [vp]
    L.pct_obs_row_len.restype = i32
    L.pct_reset.argtypes = [vp, vp, i32, vp]
    L.pct_step_rows.argtypes = [vp, vp, i32, vp]
    L.pct_step_index.argtypes = [vp, vp, vp]
    L.pct_step_hash_policy.argtypes = [vp, i32, vp]
    L.pct_policy_hash_rows.argtypes = [vp, vp, vp]
    L.pct_policy_hash_index.argtypes = [vp, vp, vp]
    try:
        L.pct_bind_policy_rows.argtypes = [vp, vp]
    except AttributeError:  # an older library selected with PCT_HIP_LIB for an A/B run (kernel experiments only)
        if not os.environ.get("PCT_HIP_LIB"):
            raise
    L.pct_profile_enable.argtypes = [vp, i32]
    L.pct_profile_read.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(ctypes.c_double)]
    L.pct_debug_phase_timing.argtypes = [vp, i32, vp]
    L.pct_debug_timing_slots.argtypes = []
    L.pct_debug_timing_slots.restype = i32
    L.pct_debug_state_f64.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp]
    for name, at in (("pct_debug_work_keys", [vp, vp]), ("pct_debug_retry_count", [vp, vp, vp, vp])):
        try:
            getattr(L, name).argtypes = at
        except AttributeError:  # an older library selected with PCT_HIP_LIB (kernel experiments only)
            if not os.environ.get("PCT_HIP_LIB"):
                raise
    L.pct_debug_state.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp, vp]
    _LIB = L
    return L


def check(rc):
    if rc != PCT_OK:
        raise RuntimeError("pct error %d: %s" % (rc, load().pct_last_error().decode()))

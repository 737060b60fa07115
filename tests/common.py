"""Shared helpers for the parity tests (host-side stand-in policy, fixtures, item sets)."""
import ast
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
M32 = 0xFFFFFFFF


def mix32(g, t):
    """include/pct_env.h pct_mix32 (vectorised over numpy uint64 arrays)."""
    g = np.asarray(g, dtype=np.uint64)
    t = np.asarray(t, dtype=np.uint64)
    h = (g * np.uint64(0x9E3779B1) + t * np.uint64(0x85EBCA77) + np.uint64(0xC2B2AE3D)) & np.uint64(M32)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7FEB352D)) & np.uint64(M32)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x846CA68B)) & np.uint64(M32)
    h ^= h >> np.uint64(16)
    return h


def item_set_range(lo, hi):
    return [(i, j, k) for i in range(lo, hi + 1) for j in range(lo, hi + 1) for k in range(lo, hi + 1)]


def case_items(c):
    """a fixture's item set: (lo..hi)^3, or -- `flat` -- footprints (lo..hi)^2 of height 1 (gen_golden.py case_items)"""
    if c.get("flat"):
        return [(i, j, 1) for i in range(c["lo"], c["hi"] + 1) for j in range(c["lo"], c["hi"] + 1)]
    return item_set_range(c["lo"], c["hi"])


def make_stream(seed, n_envs, T, item_set):
    rng = np.random.RandomState(seed)
    items = np.asarray(item_set, dtype=np.int32)
    return items[rng.randint(0, len(items), size=(n_envs, T))]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = ast.literal_eval(str(z["meta"]))
    return meta, z


def hash_policy_index(obs, I, L, base, t_vec):
    """leaf index per env from a [N,row_len] observation: mix32(g, t) % k, 0 if k == 0."""
    obs = np.asarray(obs)
    N = obs.shape[0]
    leaf = obs.reshape(N, -1, 9)[:, I:I + L]
    k = (leaf[:, :, 8] != 0).sum(1).astype(np.uint64)
    g = np.arange(N, dtype=np.uint64) + np.uint64(base)
    h = mix32(g, t_vec)
    return np.where(k > 0, h % np.maximum(k, np.uint64(1)), np.uint64(0)).astype(np.int64)


def gather_rows(obs, I, idx):
    obs = np.asarray(obs)
    N = obs.shape[0]
    return obs.reshape(N, -1, 9)[np.arange(N), I + idx].copy()


GOLDEN_CASES = ["discrete_s2_10_80_50", "discrete_s2_rect_60_30", "discrete_s2_10_80_5", "discrete_s2_20_120_400",
                "discrete_s2_cp_10_80_50", "discrete_s2_cp_rect_60_16", "discrete_s1_10_80_50", "discrete_s1_rect_60_30",
                "discrete_s2_fc_10_80_50", "discrete_s1_fc_rect_60_24",
                "discrete_s3_10_80_50", "discrete_s3_rect_60_30",
                "discrete_s2_ep_10_80_50", "discrete_s2_ep_rect_60_16", "discrete_s1_ep_10_80_50",
                "discrete_s2_ev_10_80_50", "discrete_s1_ev_rect_60_24", "discrete_s2_ev_small_bin",
                "discrete_s1_flat_lstsq", "discrete_s3_flat_lstsq", "discrete_s1_flat20"]
LNES_CODE = {"EMS": 0, "EV": 1, "EP": 2, "CP": 3, "FC": 4}

CONT_CASES = ["continuous_s2_10_80_50", "continuous_s2_100_200_200", "continuous_s2_rect_60_20",
              "continuous_s1_10_80_50", "continuous_s1_unit_80_50",
              "continuous_s3_unit_80_50", "continuous_s3_10_80_50", "continuous_s1_flat_lstsq"]
CONT_STAB_CASES = ["continuous_s1_10_80_50", "continuous_s1_unit_80_50", "continuous_s3_unit_80_50", "continuous_s3_10_80_50",
                   "continuous_s1_flat_lstsq"]

# CPU-only fixtures (stability, settings 1/3: restated in the oracle, not yet on the GPU)
ORACLE_ONLY_CASES = []
STAB_CASES = ["discrete_s1_10_80_50", "discrete_s1_rect_60_30", "discrete_s3_10_80_50", "discrete_s3_rect_60_30",
              "discrete_s1_flat_lstsq", "discrete_s3_flat_lstsq", "discrete_s1_flat20"]

DATASET_CASES = ["discrete_s2_dataset", "continuous_s2_dataset", "discrete_s3_dataset", "continuous_s3_dataset"]


def case_density(z):
    """scripted densities [N,T] of a setting-3 fixture, else None"""
    return z["density"] if "density" in z.files else None


def dataset_trajectories(z):
    """list of [len,3] (or [len,4] = size + density, setting 3) float arrays from a dataset fixture"""
    out, o = [], 0
    for n in z["traj_len"]:
        out.append(np.asarray(z["traj_items"][o:o + int(n)], np.float64))
        o += int(n)
    return out


HEURISTIC_CASES = ["heur_s2_10", "heur_s1_10", "heur_s2_rect"]
HEURISTIC_CONT_CASES = ["heur_cont_s2_10", "heur_cont_s1_unit"]  # heuristic.py on PackingContinuous: LSAH, OnlineBPH, BR
MACS_CASES = ["heur_macs_s2_10", "heur_macs_s1_rect"]
HEUR_CODE = {"LSAH": 0, "HM": 1, "OnlineBPH": 2, "DBL": 3, "BR": 4, "MACS": 5, "RANDOM": 6}

"""Fixtures for splits over MORE THAN 16 supporters (VERDICT r5 "lift the 16-supporter cap"): scripted placements of the
UNMODIFIED reference -- unit tiles in a grid, then a plate on top of them, then boxes on the plate.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_plate_golden.py

The reference env takes a 3-vector action (rotation flag, lx, ly) for the next box as it is (D/bin3D.py:151-153), so a
scripted item stream + scripted 3-vector actions build any stack.  A plate whose centre of mass no tile holds strictly
inside (D/space.py:88-94: 5 x 5 tiles minus the centre one -> 24 supporters; a 5 x 4 plate, centre on a tile edge -> 20)
makes calculated_impact / calculated_impact_virtual solve np.linalg.lstsq over 24 / 20 unknowns (277 / 191 rows) -- for the
plate's own commit, for every candidate position of the plate in the observation before it (virtual checks), and again
whenever a box lands on the plate.  N <= 25 keeps LAPACK dgelsd on the path the restatement covers (dlalsd: n <= SMLSIZ = 25
-> dlasdq; beyond it dgelsd switches to the divide-and-conquer dlasda / dlalsa, which is NOT restated).
Recorded as in gen_golden.py (float32 observation, reward, done, counter, ratio per step) + the actions; the oracle must agree
bit for bit before the fixture is written (tests/golden/plate_*.npz)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (installs the lstsq counter, the reference shim)
import ref_shim  # noqa: E402


def script_grid_plate(holes, plate, on_top, filler, steps):
    """items [T,3] and actions [T,3] of one env: unit tiles on the 5 x 5 grid minus `holes`, the plate (sx, sy) at the origin,
    `on_top` = [(item, (lx, ly))...] on the plate, then `filler` unit tiles along x = 9 up to `steps` placements"""
    items, acts = [], []
    for i in range(plate[0]):
        for j in range(plate[1]):
            if (i, j) in holes:
                continue
            items.append((1, 1, 1))
            acts.append((0, i, j))
    items.append((plate[0], plate[1], 1))
    acts.append((0, 0, 0))
    for it, (lx, ly) in on_top:
        items.append(it)
        acts.append((0, lx, ly))
    j = 0
    while len(items) < steps:
        items.append((1, 1, 1))
        acts.append((0, 9, j % 10) if j < 10 else (0, 8, j % 10))
        j += 1
    assert len(items) == steps, (len(items), steps)
    return np.array(items, np.int32), np.array(acts, np.int32)


STEPS = 40
ENVS = {
    # 24 supporters, none holds the plate's centre (2.5, 2.5): least squares over 24 unknowns
    "plate_5x5_hole": dict(holes={(2, 2)}, plate=(5, 5), on_top=[((2, 2, 2), (0, 0)), ((3, 2, 1), (2, 0)), ((2, 3, 1), (0, 2)), ((1, 1, 3), (4, 4)), ((3, 3, 2), (2, 2))]),
    # 20 supporters, the centre (2.5, 2.0) lies on tile edges
    "plate_5x4": dict(holes=set(), plate=(5, 4), on_top=[((2, 2, 1), (3, 2)), ((2, 2, 2), (0, 0)), ((1, 4, 1), (2, 0)), ((2, 1, 1), (3, 0))]),
    # 22 supporters: two holes off centre
    "plate_5x5_two_holes": dict(holes={(2, 2), (0, 4), (4, 0)}, plate=(5, 5), on_top=[((4, 4, 1), (0, 0)), ((1, 1, 1), (4, 4)), ((2, 2, 1), (1, 1))]),
    # 25 supporters, the centre tile holds the centre of mass: the direct split (no solve) with a 25-supporter hull
    "plate_5x5_full": dict(holes=set(), plate=(5, 5), on_top=[((5, 5, 1), (0, 0)), ((2, 2, 2), (0, 0))]),
}
CASE = dict(setting=1, container=(10, 10, 10), I=80, L=50, N=len(ENVS), steps=STEPS, base=0)


def scripts():
    items = np.zeros((len(ENVS), STEPS + 1, 3), np.int32)
    acts = np.zeros((STEPS, len(ENVS), 3), np.int32)
    for e, (name, s) in enumerate(ENVS.items()):
        it, ac = script_grid_plate(s["holes"], s["plate"], s["on_top"], None, STEPS)
        items[e, :STEPS] = it
        items[e, STEPS] = (1, 1, 1)
        acts[:, e] = ac
    return items, acts


def run_reference(items, acts):
    PD, _, _ = ref_shim.load_reference_envs()
    c = CASE
    N, I, L = c["N"], c["I"], c["L"]
    row_len = (I + L + 1) * 9
    obs_rec = np.zeros((c["steps"] + 1, N, row_len), np.float32)
    rew = np.zeros((c["steps"], N), np.float64)
    done = np.zeros((c["steps"], N), np.uint8)
    counter = np.zeros((c["steps"], N), np.int32)
    ratio = np.zeros((c["steps"], N), np.float64)
    item_set = [(i, j, k) for i in range(1, 6) for j in range(1, 6) for k in range(1, 6)]
    for e in range(N):
        env = PD(setting=c["setting"], container_size=list(c["container"]), item_set=item_set,
                 internal_node_holder=I, leaf_node_holder=L, shuffle=False, LNES="EMS")
        env.box_creator = gg.scripted_creator(items[e])
        obs = env.reset()
        for t in range(c["steps"]):
            obs_rec[t, e] = obs.astype(np.float32)
            obs, r, d, info = env.step([int(v) for v in acts[t, e]])  # len 3: (flag, lx, ly), D/bin3D.py:151-153 (integers: they index the heightmap)
            rew[t, e], done[t, e], counter[t, e], ratio[t, e] = r, d, info["counter"], info.get("ratio", 0.0)
            if d:
                obs = env.reset()
        obs_rec[c["steps"], e] = obs.astype(np.float32)
    return dict(obs=obs_rec, reward=rew, done=done, counter=counter, ratio=ratio)


def run_oracle(items, acts):
    from oracle.oracle_lib import OracleVecEnv
    c = CASE
    item_set = [(i, j, k) for i in range(1, 6) for j in range(1, 6) for k in range(1, 6)]
    env = OracleVecEnv(c["N"], setting=c["setting"], container_size=c["container"], item_set=item_set,
                       internal_node_holder=c["I"], leaf_node_holder=c["L"], env_id_base=c["base"])
    env.set_item_stream(items)
    N, I, L = c["N"], c["I"], c["L"]
    row_len = (I + L + 1) * 9
    obs_rec = np.zeros((c["steps"] + 1, N, row_len), np.float32)
    rew = np.zeros((c["steps"], N), np.float64)
    done = np.zeros((c["steps"], N), np.uint8)
    counter = np.zeros((c["steps"], N), np.int32)
    ratio = np.zeros((c["steps"], N), np.float64)
    env.reset()
    for t in range(c["steps"]):
        obs_rec[t] = env.obs.astype(np.float32)
        env.step_rows(acts[t].astype(np.float64))
        rew[t], done[t], counter[t], ratio[t] = env.reward, env.done, env.counter, env.ratio
    obs_rec[c["steps"]] = env.obs.astype(np.float32)
    assert not env.flags.any(), env.flags
    env.close()
    return dict(obs=obs_rec, reward=rew, done=done, counter=counter, ratio=ratio)


def main():
    items, acts = scripts()
    # widths of the least-squares systems the reference solves in this run
    widths = {}
    inner = gg._np_lstsq

    def counting(a, b, rcond=None):
        widths[a.shape[1]] = widths.get(a.shape[1], 0) + 1
        return inner(a, b, rcond=rcond)

    np.linalg.lstsq = counting
    ref = run_reference(items, acts)
    np.linalg.lstsq = gg._counting_lstsq
    ora = run_oracle(items, acts)
    for key in ("obs", "reward", "done", "counter", "ratio"):
        a, b = ref[key], ora[key]
        if key == "ratio":
            a = a * (ref["done"] != 0)
            b = b * (ora["done"] != 0)
        if not np.array_equal(a, b):
            raise SystemExit("MISMATCH plate/%s first at %s" % (key, np.argwhere(a != b)[0]))
    print("plate fixtures: %d envs x %d steps, episodes ended %d, np.linalg.lstsq calls by unknowns %s -- oracle == reference" % (
        CASE["N"], CASE["steps"], int(ref["done"].sum()), dict(sorted(widths.items()))))
    meta = dict(CASE, envs=list(ENVS), lstsq_widths=dict(sorted(widths.items())))
    np.savez_compressed(os.path.join(HERE, "plate_discrete_s1.npz"), meta=np.array(repr(meta)), stream=items, actions=acts,
                        obs=ref["obs"], reward=ref["reward"], done=ref["done"], counter=ref["counter"],
                        ratio=ref["ratio"] * (ref["done"] != 0))


if __name__ == "__main__":
    main()

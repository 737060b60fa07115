"""debug helper: runs a stability fixture on the GPU, prints the step at which outputs or flags first deviate"""
import importlib
import sys

import numpy as np

sys.path.insert(0, ".")
from tests.common import load_case, case_items, case_density  # noqa: E402

pkg = importlib.import_module("online-3d-bpp-pct_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "continuous_s1_flat_lstsq"
c, z = load_case(name)
kw = dict(setting=c["setting"], container_size=c["container"], internal_node_holder=c["I"], leaf_node_holder=c["L"],
          env_id_base=c["base"], item_stream=z["stream"], device="cuda:0", strict=False)
if name.startswith("continuous"):
    env = pkg.PctVecEnv(c["N"], continuous=True, sample_left_bound=c["lo"], sample_right_bound=c["hi"], **kw)
else:
    env = pkg.PctVecEnv(c["N"], item_set=case_items(c), **kw)
if case_density(z) is not None:
    env.set_density_stream(case_density(z))
obs = env.reset()
for t in range(c["steps"]):
    o = obs.cpu().numpy()
    want = z["obs"][t].astype(np.float32)
    bad = np.argwhere((o != want).any(1)).ravel()
    raw = env._flags.cpu().numpy().view(np.uint32)
    if len(bad) or raw.any():
        print("step", t, "obs mismatch envs", bad.tolist(), "flags", [hex(int(x)) for x in raw], "counter", env._counter.cpu().numpy().tolist())
        if len(bad):
            break
    env.step_hash_policy(1)
    obs, rew, done, infos = env.step_wait()
else:
    print("all", c["steps"], "steps identical; flags", [hex(int(x)) for x in env._flags.cpu().numpy().view(np.uint32)])

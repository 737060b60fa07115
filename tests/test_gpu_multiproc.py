"""GPU: the multi-process layout on real device handles (no 8-GPU node is needed for correctness):
  * RCCL: a world-size-1 "nccl" process group on cuda:0 -- init, barrier, all-reduce and the rollout
    all-gather of `RolloutSlots.gather()` on DEVICE tensors filled by the transition kernel;
  * two ranks, each with its own HIP handle on the one GPU, sharded by global env id (env_id_base), their
    shards exchanged over gloo: together they must reproduce one big HIP batch bit for bit -- item picks,
    shuffle priorities, densities and the stand-in policy are all keyed by the GLOBAL env id."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _items():
    return [(i, j, k) for i in range(1, 6) for j in range(1, 6) for k in range(1, 6)]


def _nccl_worker(rank, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    pkg = importlib.import_module("online-3d-bpp-pct_amd")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    t = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    N, T = 96, 4
    env = pkg.PctVecEnv(N, item_set=_items(), seed=5, device=dev)
    env.reset()
    ro = pkg.RolloutSlots(T, N, (131, 9), 1.0, dev)
    ro.begin(env)

    def policy(nodes):
        k = (nodes[:, 80:130, 8] != 0).sum(1)
        return torch.zeros(N, 1, device=dev), torch.clamp(k - 1, min=0).unsqueeze(1)

    pkg.collect(env, policy, ro)
    g = ro.gather()  # RCCL all_gather_into_tensor on device tensors
    ok = all(torch.equal(g[n], getattr(ro, n)) for n in ("obs", "rewards", "masks", "actions"))
    ok = ok and float(t.item()) == 1.5 and g["obs"].is_cuda
    open(os.path.join(out_dir, "nccl.txt"), "w").write("ok" if ok else "mismatch")
    env.close()
    dist.destroy_process_group()


def test_rccl_world1_rollout_gather_on_device(tmp_path):
    mp.spawn(_nccl_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    assert open(tmp_path / "nccl.txt").read() == "ok"


def _make(pkg, n, base, variant):
    if variant == "setting3_shuffle":
        return pkg.PctVecEnv(n, setting=3, item_set=_items(), env_id_base=base, shuffle=True, seed=77, device="cuda:0")
    if variant == "continuous":
        return pkg.PctVecEnv(n, continuous=True, sample_left_bound=1.0, sample_right_bound=5.0, env_id_base=base, seed=77,
                             device="cuda:0")
    return pkg.PctVecEnv(n, item_set=_items(), env_id_base=base, seed=77, device="cuda:0")


def _shard_worker(rank, world, port, total, steps, variant, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("online-3d-bpp-pct_amd")
    base, n = pkg.shard_envs(total, rank, world)
    env = _make(pkg, n, base, variant)
    env.reset()
    env.step_hash_policy(steps)
    obs, reward, done, _ = env.step_wait()
    full = pkg.gather_rollout(obs.cpu())          # unequal shards: 2 ranks share 1001 envs
    rew = pkg.gather_rollout(reward[:, 0].clone())
    if rank == 0:
        np.save(os.path.join(out_dir, "obs.npy"), full.numpy())
        np.save(os.path.join(out_dir, "rew.npy"), rew.numpy())
    dist.barrier()
    env.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("variant", ["plain", "setting3_shuffle", "continuous"])
def test_two_hip_ranks_on_one_gpu_equal_one_batch(tmp_path, variant):
    total, steps, world = 1001, 40, 2
    mp.spawn(_shard_worker, args=(world, _free_port(), total, steps, variant, str(tmp_path)), nprocs=world, join=True)
    pkg = importlib.import_module("online-3d-bpp-pct_amd")
    env = _make(pkg, total, 0, variant)
    env.reset()
    env.step_hash_policy(steps)
    obs, reward, done, _ = env.step_wait()
    assert np.array_equal(np.load(tmp_path / "obs.npy"), obs.cpu().numpy())
    assert np.array_equal(np.load(tmp_path / "rew.npy"), reward[:, 0].numpy())
    env.close()


def test_bench_two_rank_code_path_on_one_gpu(tmp_path):
    """the N > 1 line of bench.py -- torch.distributed.run launch, barrier, MAX-reduce of the elapsed
    time over ranks, per-rank kernel-time gather, ONE JSON line from rank 0 -- executed end to end with two ranks that
    share cuda:0 over gloo (an 8-GPU node is not available to the builder; RCCL itself is covered at world size 1
    above).  Not a measurement: only the contract of the line is checked."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "10",
           "--envs-per-gpu", "512", "--dist-backend", "gloo", "--share-gpu", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 30 and d["warmup"] == 10 and d["scaling"] == "weak"
    assert d["config"]["global_envs"] == 1024 and d["config"]["envs_per_gpu"] == 512
    assert len(d["roofline"]["kernel_avg_us_per_rank"]) == 2 and all(x > 0 for x in d["roofline"]["kernel_avg_us_per_rank"])
    assert abs(d["value"] - 1024 * 30 / (d["ms_per_step"] * 30 / 1e3)) < 1e-6 * d["value"]

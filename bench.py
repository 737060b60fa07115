#!/usr/bin/env python
"""bench.py -- env-steps/sec of the batched PCT env hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE batched transition of the BASELINE.json configs[1] workload on every GPU:
PctDiscrete0 setting 2 (EMS leaves), bin 10x10x10, 80 internal / 50 leaf nodes, 4096 envs
per GPU, items drawn uniformly from (1..5)^3 by the on-device counter-based sampler.  Per
step the stand-in policy kernel reads the leaf mask from the observation and writes one
float32 leaf row per env ([N,9], what train_tools.py:66-67 hands the env), then
pct_step_rows runs the transition kernel, which regenerates the full [131,9] float32
observation, reward, done and info for every env (auto-reset included).  Everything stays
in HBM; the host only enqueues.  Envs shard across GPUs by global env id with no collective
on the step path ("scaling": "weak", per-GPU work fixed).

Printed JSON (one line, rank 0): the driver contract plus
  roofline     -- the transition kernel against the HBM roof: algorithmic bytes per launch
                  (4757 B per env-step x envs per launch) / its average
                  duration measured with HIP events recorded by the library on the launch
                  stream during the timed region;
  cpu_baseline -- the CPU oracle (C restatement of the reference env, oracle/) timed on
                  this box's host cores on a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

I_NODES, L_NODES = 80, 50
ALG_BYTES_PER_STEP = 4 * 9 * (I_NODES + L_NODES + 1) + 36 + 4 + 1  # 4757, SURVEY.md 8(d)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def item_set():
    return [(i, j, k) for i in range(1, 6) for j in range(1, 6) for k in range(1, 6)]


def cpu_baseline(envs, budget_s, workload="c2"):
    """The oracle on the host cores: same workload, bounded sample."""
    from oracle.oracle_lib import OracleVecEnv
    threads = max(1, min(os.cpu_count() or 1, 64))
    if workload == "c3":
        env = OracleVecEnv(envs, setting=2, container_size=(10, 10, 10), env_kind=1, sample_bounds=(1.0, 5.0),
                           internal_node_holder=I_NODES, leaf_node_holder=L_NODES, threads=threads)
    else:
        env = OracleVecEnv(envs, setting=2, container_size=(10, 10, 10), item_set=item_set(),
                           internal_node_holder=I_NODES, leaf_node_holder=L_NODES, threads=threads)
    env.set_sampler(4)
    env.reset()
    env.step_hash_policy(50)  # de-synchronise the episodes
    t0 = time.perf_counter()
    steps = 0
    chunk = 20
    while time.perf_counter() - t0 < budget_s:
        env.step_hash_policy(chunk)
        steps += chunk
    dt = time.perf_counter() - t0
    env.close()
    return {"value": envs * steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": "oracle/pct_oracle.c (C restatement of the reference env, OpenMP over envs), %d envs x %d "
                      "batched steps after 50 warm-up steps, same config/sampler/policy, %.1f s" % (envs, steps, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--mode", choices=["rows", "fused", "host"], default="rows",
                    help="rows: policy kernel + pct_step_rows per step (default); fused: pct_step_hash_policy(1); "
                         "host: the reference trainer's hand-over -- leaf rows to the host as numpy "
                         "(train_tools.py:66-67), VecEnv.step(numpy), reward / done back on the host every step "
                         "(PCIe and a stream sync inside the timed region; never the headline value)")
    ap.add_argument("--pipelines", type=int, default=1,
                    help="split each GPU's envs into this many independently stepped groups, one HIP stream each "
                         "(1 = one batch per step, the headline configuration; 2 overlaps one group's tail with "
                         "the other group's kernels)")
    ap.add_argument("--workload", choices=["c2", "c3", "c5"], default="c2",
                    help="c2: BASELINE configs[1] (discrete, the headline metric); c3: configs[2] (continuous setting 2)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    pkg = importlib.import_module("online-3d-bpp-pct_amd")
    n_local = args.envs_per_gpu
    P = max(1, args.pipelines)
    assert n_local % P == 0, "--envs-per-gpu must be a multiple of --pipelines"
    n_grp = n_local // P

    def make_env(g):
        base = rank * n_local + g * n_grp
        if args.workload == "c5":  # BASELINE.json configs[4]: 100^3, 200/200, items U(5,25)
            return pkg.PctVecEnv(n_grp, setting=2, container_size=(100, 100, 100), continuous=True, sample_left_bound=5.0,
                                 sample_right_bound=25.0, internal_node_holder=I_NODES, leaf_node_holder=L_NODES, seed=4,
                                 env_id_base=base, device=dev, monitor=False, ems_capacity=768,
                                 candidate_capacity=32768)
        if args.workload == "c3":
            return pkg.PctVecEnv(n_grp, setting=2, container_size=(10, 10, 10), continuous=True, sample_left_bound=1.0,
                                 sample_right_bound=5.0, internal_node_holder=I_NODES, leaf_node_holder=L_NODES, seed=4,
                                 env_id_base=base, device=dev, monitor=False)
        return pkg.PctVecEnv(n_grp, setting=2, container_size=(10, 10, 10), item_set=item_set(),
                             internal_node_holder=I_NODES, leaf_node_holder=L_NODES, seed=4,
                             env_id_base=base, device=dev, monitor=False)

    if args.workload == "c5":
        global I_NODES, L_NODES, ALG_BYTES_PER_STEP
        I_NODES, L_NODES = 200, 200
        ALG_BYTES_PER_STEP = 4 * 9 * (I_NODES + L_NODES + 1) + 36 + 4 + 1
    envs = [make_env(g) for g in range(P)]
    env = envs[0]
    streams = [torch.cuda.current_stream(dev)] if P == 1 else [torch.cuda.Stream(dev) for _ in range(P)]
    rows = [torch.empty(n_grp, 9, dtype=torch.float32, device=dev) for _ in range(P)]
    for ev in envs:
        ev.reset()
    torch.cuda.synchronize(dev)

    def one_step():
        for g in range(P):
            with torch.cuda.stream(streams[g]):
                if args.mode == "rows":
                    envs[g].policy_hash_rows(rows[g])
                    envs[g].step_rows_device(rows[g])
                elif args.mode == "host":
                    envs[g].policy_hash_rows(rows[g])
                    envs[g].step(rows[g].cpu().numpy())  # obs (device), reward (CPU), done (numpy), infos
                else:
                    envs[g].step_hash_policy(1)

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize(dev)
    for ev in envs:
        ev.profile_enable(True)
        ev.profile_read()

    def barrier():
        if dist is not None:
            dist.barrier()

    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    n_launch, kern_ms = 0, 0.0
    for ev in envs:
        nl, km = ev.profile_read()
        n_launch += nl
        kern_ms += km
        ev.profile_enable(False)
        flags = ev.error_flags
        assert not flags.any(), "env error flags raised during the bench: %s" % flags[flags != 0][:8]

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        k = torch.tensor([kern_ms / max(n_launch, 1)], dtype=torch.float64, device=dev)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        kern_avg_ms = float(k.item())
    else:
        kern_avg_ms = kern_ms / max(n_launch, 1)

    total_steps = world * n_local * args.steps
    value = total_steps / elapsed
    achieved_gbs = ALG_BYTES_PER_STEP * n_grp / (kern_avg_ms * 1e-3) / 1e9  # per launch (n_grp envs)
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and args.workload == "c2" and n_grp == 4096:
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "env-steps/sec (whole node), discrete setting 2, 80 internal/50 leaf" if args.workload == "c2"
                  else "env-steps/sec (whole node), continuous setting 2, 80 internal/50 leaf",
        "value": value,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "i32" if args.workload == "c2" else "f64",
        "data": "synthetic",
        "config": {
            "workload": ("PctDiscrete0 setting 2 (EMS leaves), bin 10x10x10, 80 internal / 50 leaf, %d batched envs per "
                         "MI355X (BASELINE.json configs[1]); items ~ U{(1..5)^3} from the on-device counter sampler; "
                         "per step: policy kernel -> float32 [N,9] leaf rows -> transition kernel (observation rows "
                         "rewritten, auto-reset)" % n_local) if args.workload == "c2" else
                        ("PctContinuous0 setting 2, bin 10x10x10, 80 internal / 50 leaf, %d batched envs per MI355X "
                         "(BASELINE.json configs[2]); item sizes round(U(1,5),3) from the on-device counter sampler; "
                         "per step: policy kernel -> float32 [N,9] leaf rows -> float64 transition kernel" % n_local),
            "envs_per_gpu": n_local,
            "global_envs": world * n_local,
            "mode": args.mode,
            "pipelines": P,
            "parallelism": "envs sharded by global id x%d, no collective on the step path" % world,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS,
            "traffic": traffic,
            "kernel": ("pct_discrete_kernel<u32,5," if args.workload == "c2" else "pct_continuous_kernel<") +
                      ("ACT_HASH>" if args.mode == "fused" else "ACT_ROWS>"),
            "kernel_avg_us": kern_avg_ms * 1e3,
            "launches_timed": n_launch,
            "alg_bytes_per_env_step": ALG_BYTES_PER_STEP,
        },
    }
    if args.workload == "c5":
        out["metric"] = "env-steps/sec (whole node), continuous setting 2, 100^3 bin, 200 internal/200 leaf"
        out["config"]["workload"] = ("PctContinuous0 setting 2, bin 100^3, 200 internal / 200 leaf, %d batched envs per MI355X "
                                     "(BASELINE.json configs[4]); item sizes round(U(5,25),3)" % n_local)
        args.no_cpu_baseline = True
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(256, args.cpu_seconds, args.workload)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    for ev in envs:
        ev.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

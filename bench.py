#!/usr/bin/env python
"""bench.py -- env-steps/sec of the batched PCT env hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c4|c3|c5|c1|c3s1]
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE batched transition of the workload on every GPU.  The default workload (c2) is
BASELINE.json configs[1], the configuration the headline metric is quoted on: PctDiscrete0 setting 2
(EMS leaves), bin 10x10x10, 80 internal / 50 leaf nodes, 4096 envs per GPU, items drawn uniformly from
(1..5)^3 by the on-device counter-based sampler.  Per step pct_step_rows runs the transition kernel on one
float32 leaf row per env ([N,9] in HBM, what train_tools.py:66-67 hands the env): it regenerates the
[I+L+1,9] float32 observation, reward, done and info for every env (auto-reset included).  The rows come
from the stand-in policy (leaf = pct_mix32(env, t) % k over the k valid leaves of the observation): by
default written by the PREVIOUS launch's policy epilogue (pct_bind_policy_rows: `--mode epilogue`, one
transition dispatch per step), or by the stand-in policy as its own kernel between two steps (`--mode rows`,
the rounds 1-3 default).  Before --warmup come 200 untimed de-synchronisation steps (`--desync`), so that a
short run measures episodes of every length in flight, not 4096 envs that all start empty.  Everything
stays in HBM; the host only enqueues.  Envs shard across GPUs by global env id with no collective on the
step path ("scaling": "weak", per-GPU work fixed).

Other workloads (parity-test configurations of BASELINE.json, measurable with the same contract):
  c4    the per-GPU slice of configs[3] (65 536 envs over 8 GPUs): c2 with 8192 envs per GPU --
        `bench.py --gpus 8 --workload c4` under torch.distributed.run IS the configs[3] line (the default
        `--gpus N` line shards configs[1]'s 4096 envs per GPU)
  c3    configs[2]: PctContinuous0 setting 2, 10^3, 80/50, 4096 envs per GPU (float64 kernel)
  c5    configs[4]: PctContinuous0 setting 2, 100^3, 200/200, items U(5,25), 2048 envs per GPU
  c1    configs[0]'s geometry on the GPU: PctDiscrete0 setting 1 (stability check), 10^3, 4096 envs
  c3s1  PctContinuous0 setting 1 (stability) in the unit bin, items 2 x U(0.1,0.5) + z in {0.1..0.5}

Printed JSON (one line, rank 0): the driver contract plus
  roofline        the transition kernel against the HBM roof: algorithmic bytes per launch
                  (B(I,L) = 36 (I+L+1) + 41 per env-step x envs per launch) / its average
                  duration measured with HIP events over the timed region (the library hands an event pair
                  to hipExtLaunchKernel as the step kernel's start / stop events: exactly that dispatch, on
                  the launch stream -- on every `timed_every`-th launch: a dispatch that carries events costs
                  ~7 us of its own, so the average is sampled); `kernel` = the name rocprofv3 prints for it; `traffic` = HBM bytes
                  per launch from rocprofv3 PMC passes of the same command (a separate profiled run;
                  `traffic_source` names the committed file);
  roofline_issue  the same kernel against the instruction-issue rate of the chip (VALU + SALU wave
                  instructions per launch from the committed PMC pass / the measured duration);
  cpu_baseline    the CPU oracle (C restatement of the reference env, oracle/) timed on this box's host
                  cores on a bounded sample of the same workload;
  cpu_baseline_reference  the REFERENCE's own Python path (envs.make_vec_envs -> ShmemVecEnv(fork)),
                  timed in the build container by scripts/time_reference_cpu.py (the reference tree does
                  not travel to the GPU box) -- carried from profiles/cpu_reference_baseline.json.
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

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# issue roof: 256 CUs x 2.4 GHz x (4 SIMD-32s x one wave64 VALU instruction per 2 cycles + one SALU
# instruction per cycle on the CU's scalar unit) wave-instructions per second (MI355X_MICROARCH.md:
# "v_fma_f32 (wave64) 2 cyc (SIMD-32)"; the scalar rate is the usual one-per-cycle-per-CU assumption)
ISSUE_PEAK_GINST = 256 * 2.4 * (4 * 0.5 + 1.0)


def item_set():
    return [(i, j, k) for i in range(1, 6) for j in range(1, 6) for k in range(1, 6)]


WORKLOADS = {
    # name: (env kind, setting, container, I, L, envs/GPU, sample bounds, description, reference-baseline key)
    "c2": dict(cont=False, setting=2, container=(10, 10, 10), I=80, L=50, envs=4096, bounds=None, ref="discrete_s2_shmem",
               metric="env-steps/sec (whole node), discrete setting 2, 80 internal/50 leaf",
               what="PctDiscrete0 setting 2 (EMS leaves), bin 10x10x10, 80 internal / 50 leaf, %d batched envs per MI355X "
                    "(BASELINE.json configs[1]); items ~ U{(1..5)^3} from the on-device counter sampler"),
    "c4": dict(cont=False, setting=2, container=(10, 10, 10), I=80, L=50, envs=8192, bounds=None, ref="discrete_s2_shmem",
               metric="env-steps/sec (whole node), discrete setting 2, 80 internal/50 leaf, 8192 envs per GPU",
               what="PctDiscrete0 setting 2, bin 10x10x10, 80/50, %d batched envs per MI355X (the per-GPU slice of "
                    "BASELINE.json configs[3]: 65 536 envs over 8 GPUs)"),
    "c3": dict(cont=True, setting=2, container=(10, 10, 10), I=80, L=50, envs=4096, bounds=(1.0, 5.0), ref="continuous_s2_shmem",
               metric="env-steps/sec (whole node), continuous setting 2, 80 internal/50 leaf",
               what="PctContinuous0 setting 2, bin 10x10x10, 80 internal / 50 leaf, %d batched envs per MI355X "
                    "(BASELINE.json configs[2]); item sizes round(U(1,5),3) from the on-device counter sampler; float64 kernel"),
    "c5": dict(cont=True, setting=2, container=(100, 100, 100), I=200, L=200, envs=2048, bounds=(5.0, 25.0), ref="continuous_c5_shmem",
               metric="env-steps/sec (whole node), continuous setting 2, 100^3 bin, 200 internal/200 leaf",
               what="PctContinuous0 setting 2, bin 100^3, 200 internal / 200 leaf, %d batched envs per MI355X (the per-GPU "
                    "slice of BASELINE.json configs[4]); item sizes round(U(5,25),3) (SURVEY.md 8(d))"),
    "c1": dict(cont=False, setting=1, container=(10, 10, 10), I=80, L=50, envs=4096, bounds=None, ref="discrete_s1_dummy_1env",
               metric="env-steps/sec (whole node), discrete setting 1 (stability check), 80 internal/50 leaf",
               what="PctDiscrete0 setting 1 (stability check, two orientations), bin 10x10x10, 80/50, %d batched envs per "
                    "MI355X (the geometry of BASELINE.json configs[0], batched on the GPU)"),
    "c3s1": dict(cont=True, setting=1, container=(1, 1, 1), I=80, L=50, envs=4096, bounds=(0.1, 0.5), ref=None,
                 metric="env-steps/sec (whole node), continuous setting 1 (stability check), unit bin, 80 internal/50 leaf",
                 what="PctContinuous0 setting 1 (stability check), unit bin, 80/50, %d batched envs per MI355X; items "
                      "2 x round(U(0.1,0.5),3) + z in {0.1..0.5} (C/bin3D.py:110-112)"),
}


def alg_bytes(w):
    return 4 * 9 * (w["I"] + w["L"] + 1) + 36 + 4 + 1  # 4757 at 80/50, 14477 at 200/200


def make_oracle(w, envs, threads):
    from oracle.oracle_lib import OracleVecEnv
    kw = dict(setting=w["setting"], container_size=w["container"], internal_node_holder=w["I"], leaf_node_holder=w["L"],
              threads=threads)
    if w["cont"]:
        return OracleVecEnv(envs, env_kind=1, sample_bounds=w["bounds"], **kw)
    return OracleVecEnv(envs, item_set=item_set(), **kw)


def cpu_baseline(w, budget_s, lstsq="gelsd", envs=0):
    """The oracle on the host cores: same workload (config, sampler, stand-in policy) at the workload's OWN env count, the
    median of three samples of budget_s / 3 seconds each (round 4 timed 256 envs once for 12 s: with 64 OpenMP threads that is
    four envs per thread and a +-40 % number -- VERDICT r4 weak point 9a)."""
    from oracle import oracle_lib
    oracle_lib.set_lstsq_mode({"jacobi": oracle_lib.LSTSQ_JACOBI, "gelsd": oracle_lib.LSTSQ_GELSD, "gelsd_avx2": oracle_lib.LSTSQ_GELSD_AVX2}[lstsq])
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    threads = max(1, min(affinity, 64))
    envs = envs or w["envs"]
    env = make_oracle(w, envs, threads)
    env.set_sampler(4)
    env.reset()
    env.step_hash_policy(20 if w["I"] <= 80 else 4)  # de-synchronise the episodes (bounded: the 200-node oracle steps at ~10 k env-steps/s)
    rates, nsteps = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < budget_s / 3.0:
            env.step_hash_policy(1)
            steps += 1
        dt = time.perf_counter() - t0
        rates.append(envs * steps / dt)
        nsteps.append(steps)
    env.close()
    rates.sort()
    return {"value": rates[1], "unit": "env-steps/s", "cores": threads, "kind": "port", "samples": rates,
            "affinity_cpus": affinity, "os_cpu_count": os.cpu_count(),
            "sample": "oracle/ (C restatement of the reference env, OpenMP over envs, %d threads = min(64, the %d CPUs this process may run on)), "
                      "%d envs (the workload's own size), median of three samples of %.1f s (%s batched steps), same config / sampler / "
                      "stand-in policy / lstsq mode" % (threads, affinity, envs, budget_s / 3.0, "/".join(str(n) for n in nsteps))}


def reference_baseline(w):
    """The reference's own Python VecEnv path, timed in the build container (see scripts/time_reference_cpu.py)."""
    path = os.path.join(ROOT, "profiles", "cpu_reference_baseline.json")
    if w["ref"] is None or not os.path.exists(path):
        return None
    try:
        z = json.load(open(path))
        c = z["configs"][w["ref"]]
    except Exception:
        return None
    return {"value": c["value"], "unit": "env-steps/s", "cores": c["workers"], "kind": "reference",
            "measured_on": "build container (%d host cores; /root/reference does not travel to the GPU box), not in this run"
                           % z["host"]["cores"],
            "path": c["path"], "sample": "%d iterations over %.0f s; %s" % (c["iterations"], c["seconds"], z["recipe"]),
            "source": "profiles/cpu_reference_baseline.json"}


def pmc_profile(name, envs):
    """Counter values per launch from the committed rocprofv3 PMC passes of this workload, or None."""
    for tag in ("r06", "r05", "r04", "r03", "r02"):  # the newest committed profile of this workload
        rel = "profiles/%s_pmc_%s.json" % (tag, name)
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        try:
            z = json.load(open(path))
            if int(z.get("envs_per_launch", -1)) != envs:
                continue
            return z, rel
        except Exception:
            continue
    return None, None


def traced_kernel_name(name, needle):
    """The step kernel as rocprofv3 printed it: the row of the newest committed raw kernel-stats CSV of this workload
    (profiles/r0x_trace_<w>_kernel_stats.csv, rocprofv3 --kernel-trace --stats) whose name holds `needle`; (None, None) without one."""
    import csv
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_trace_%s_kernel_stats.csv" % name)), reverse=True):
        try:
            for row in csv.DictReader(open(path)):
                if needle in row.get("Name", ""):
                    return row["Name"], os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="default: the workload's own size (4096 for c2)")
    ap.add_argument("--mode", choices=["epilogue", "rows", "fused", "host", "host_overlap", "slot"], default="epilogue",
                    help="epilogue (default): pct_step_rows per step, the float32 [N,9] leaf rows it reads having been "
                         "written to HBM by the previous launch's stand-in policy epilogue (pct_bind_policy_rows) -- one "
                         "transition dispatch per step; rows: the stand-in policy as its own kernel + pct_step_rows (the "
                         "rounds 1-3 default); fused: pct_step_hash_policy(1), no rows at all; "
                         "host: the reference trainer's hand-over -- leaf rows to the host as numpy "
                         "(train_tools.py:66-67), VecEnv.step(numpy), reward / done back on the host every step "
                         "(PCIe and a stream sync inside the timed region; never the headline value); "
                         "host_overlap: the same hand-over for a caller whose next action needs only the device-resident observation "
                         "(PctVecEnv.step_outputs_async: step t + 1 is launched before step t's reward / done / infos are consumed on the host); "
                         "slot: the trainer-shaped device path -- a rollout slot is bound (pct_bind_rollout_slot: the kernel writes every "
                         "observation row, reward and mask of step t straight into storage.py's [T + 1, N, ...] tensors, T = 5 as "
                         "arguments.py), the leaf index comes from a torch stand-in policy on the slot's observation, pct_step_index steps")
    ap.add_argument("--no-rows-line", action="store_true",
                    help="epilogue mode: do not also time the `rows` flavour (stand-in policy as its own kernel in front of every "
                         "transition) in the same process")
    ap.add_argument("--desync", type=int, default=200,
                    help="untimed transitions right after the synchronous reset, before --warmup, so that a short run "
                         "(the driver's --warmup 5 --steps 20) measures the steady state -- episodes of every length in "
                         "flight -- and not the first 25 steps of 4096 envs that all start empty")
    ap.add_argument("--pipelines", type=int, default=1,
                    help="split each GPU's envs into this many independently stepped groups, one HIP stream each "
                         "(1 = one batch per step, the headline configuration)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--no-overflow-retry", action="store_true",
                    help="discrete env: do not enqueue the large-capacity retry pass behind every transition (the product "
                         "default enqueues it: an env that outgrows its LDS lists is re-run instead of terminated; costs one "
                         "more, usually empty, kernel launch per step)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend of the N > 1 line (nccl = RCCL over xGMI; gloo only for the two-rank smoke "
                         "test that shares one GPU, tests/test_gpu_multiproc.py)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="smoke test only: every rank uses cuda:0 (the N > 1 code path -- barrier, MAX-reduce of the "
                         "elapsed time, per-rank gather -- on a one-GPU box; not a measurement)")
    ap.add_argument("--lstsq", choices=["jacobi", "gelsd", "gelsd_avx2"], default="gelsd",
                    help="stability workloads (c1, c3s1): the solver behind np.linalg.lstsq -- LAPACK dgelsd as the reference's NumPy executes "
                         "it (the library's default since round 5) or the Jacobi stand-in of rounds 1-4 (pct_set_lstsq_mode, "
                         "include/pct_env.h); the cpu_baseline oracle follows")
    ap.add_argument("--ems-capacity", type=int, default=0, help="experiments: pct_config.ems_capacity (0 = the library's default)")
    ap.add_argument("--candidate-capacity", type=int, default=0, help="experiments: pct_config.candidate_capacity (0 = default)")
    ap.add_argument("--time-every", type=int, default=-1,
                    help="hand the step kernel its HIP event pair on every K-th launch of the timed region (default: 8, or 4 when "
                         "--steps < 64; 1 = every launch; 0 = none: the step's time stands in for the kernel's).  A dispatch that "
                         "carries events costs ~7 us of its own on this runtime (C2: 64.6 M env-steps/s with a pair on every launch, "
                         "72.6 M with none), so the average kernel time is SAMPLED over the timed region")
    ap.add_argument("--graph", type=int, default=0, help="experiment: capture this many steps into one hipGraph and replay it")
    ap.add_argument("--torch-policy", action="store_true",
                    help="--mode slot: the stand-in policy as torch elementwise kernels (rounds 1-5) instead of pct_policy_hash_index")
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed regions of --steps steps each, median reported (0 = 7 below 64 steps, 3 below 512, else 1)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    w = WORKLOADS[args.workload]

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    red_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")  # where the reduction tensors live

    pkg = importlib.import_module("online-3d-bpp-pct_amd")
    n_local = args.envs_per_gpu or w["envs"]
    P = max(1, args.pipelines)
    assert n_local % P == 0, "--envs-per-gpu must be a multiple of --pipelines"
    n_grp = n_local // P
    B = alg_bytes(w)

    def make_env(g):
        base = rank * n_local + g * n_grp
        kw = dict(setting=w["setting"], container_size=w["container"], internal_node_holder=w["I"], leaf_node_holder=w["L"],
                  seed=4, env_id_base=base, device=dev, monitor=False, overflow_retry=not args.no_overflow_retry,
                  ems_capacity=args.ems_capacity, candidate_capacity=args.candidate_capacity, lstsq=args.lstsq)
        if w["cont"]:
            return pkg.PctVecEnv(n_grp, continuous=True, sample_left_bound=w["bounds"][0], sample_right_bound=w["bounds"][1], **kw)
        return pkg.PctVecEnv(n_grp, item_set=item_set(), **kw)

    envs = [make_env(g) for g in range(P)]
    streams = [torch.cuda.current_stream(dev)] if P == 1 else [torch.cuda.Stream(dev) for _ in range(P)]
    rows = [torch.empty(n_grp, 9, dtype=torch.float32, device=dev) for _ in range(P)]
    if args.mode == "epilogue":
        for g in range(P):
            envs[g].bind_policy_rows(rows[g])  # reset() and every transition write the next step's rows
    for g, ev in enumerate(envs):
        with torch.cuda.stream(streams[g]):
            ev.reset()
    torch.cuda.synchronize(dev)

    slots, tickets, hbase, tcount = None, [None] * P, None, [0]
    if args.mode == "slot":
        rollout = importlib.import_module("online-3d-bpp-pct_amd.rollout")
        slots = [rollout.RolloutSlots(5, n_grp, (envs[g].row_len,), 1.0, dev) for g in range(P)]
        for g in range(P):
            with torch.cuda.stream(streams[g]):
                slots[g].begin(envs[g])
        hbase = torch.arange(n_grp, device=dev, dtype=torch.int64) * 2654435761
    mode_now = [args.mode]

    def one_step():
        for g in range(P):
            with torch.cuda.stream(streams[g]):
                if mode_now[0] == "epilogue":
                    envs[g].step_rows_device(rows[g])
                elif mode_now[0] == "slot":
                    # a torch stand-in for the policy's leaf choice, on the slot the kernel wrote: k = valid leaf rows, index = hash % k
                    s = slots[g]
                    if args.torch_policy:
                        # rounds 1-5: the stand-in as torch elementwise / reduce kernels (six launches + a copy per step)
                        ob = s.obs[s.step].view(n_grp, w["I"] + w["L"] + 1, 9)
                        k = (ob[:, w["I"]:w["I"] + w["L"], 8] != 0).sum(1).clamp_(min=1)
                        idx = (hbase + tcount[0] * 40503) % k
                    else:
                        # a policy hands back an index tensor in ONE launch: the stand-in does too, straight into actions[t]
                        idx = envs[g].policy_hash_index(s.actions[s.step].view(n_grp))
                    s.step_env(envs[g], idx)
                    if s.step == 0:
                        s.after_update()  # storage.py:41-43, once per T steps
                elif mode_now[0] == "host_overlap":
                    envs[g].policy_hash_rows(rows[g])
                    envs[g].step_rows_device(rows[g])
                    prev, tickets[g] = tickets[g], envs[g].step_outputs_async()
                    if prev is not None:
                        prev.wait()  # reward (CPU), done (numpy), infos of the PREVIOUS step, while this one runs
                elif mode_now[0] == "rows":
                    envs[g].policy_hash_rows(rows[g])
                    envs[g].step_rows_device(rows[g])
                elif mode_now[0] == "host":
                    envs[g].policy_hash_rows(rows[g])
                    envs[g].step(rows[g].cpu().numpy())  # obs (device), reward (CPU), done (numpy), infos
                else:
                    envs[g].step_hash_policy(1)
        tcount[0] += 1

    for _ in range(max(0, args.desync) + args.warmup):
        one_step()
    torch.cuda.synchronize(dev)
    graph = None
    if args.graph > 0:
        # experiment: --graph K steps captured into ONE hipGraph and replayed -- does a graph launch shorten
        # the dependent-dispatch gaps between the transition kernel and the retry pass?  K even (the retry queue's counters ping-pong).
        assert args.graph % 2 == 0 and args.steps % args.graph == 0 and P == 1 and args.mode in ("epilogue", "fused", "rows")
        graph = torch.cuda.CUDAGraph()
        cs = torch.cuda.Stream(dev)
        cs.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(cs):
            streams[0] = cs
            with torch.cuda.graph(graph, stream=cs):
                for _ in range(args.graph):
                    one_step()
        torch.cuda.current_stream(dev).wait_stream(cs)
        torch.cuda.synchronize(dev)
        tcount[0] = 0
        args.no_rows_line = True  # (the second timed loop would replay the same graph)

        def one_step():  # (one replay = args.graph steps; the loops below count steps)
            tcount[0] += 1
            if tcount[0] % args.graph == 0:
                graph.replay()
    time_every = args.time_every if args.time_every >= 0 else (8 if args.steps >= 64 else 4)
    if graph is not None:
        time_every = 0  # (event pairs are not captured: the kernel average falls back to the step time)
    for ev in envs:
        ev.profile_enable(time_every)
        ev.profile_read()

    def barrier():
        if dist is not None:
            dist.barrier()

    # The timed region: EXACTLY --steps steps between barrier + synchronize on both sides.  A short region (the driver's 20 steps
    # are 1.2 ms on c2) is within +-3 % of launch jitter, so it is REPEATED and the median region reported:
    # every repeat is the contract's region, back to back, nothing untimed in between but the barrier.
    repeats = args.repeats if args.repeats > 0 else (7 if args.steps < 64 else (3 if args.steps < 512 else 1))
    regions = []
    for _ in range(repeats):
        barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        torch.cuda.synchronize(dev)
        regions.append(time.perf_counter() - t0)  # (this rank's; the MAX over ranks below covers the stragglers -- the barrier follows the stamp)
        barrier()
    n_launch, kern_ms = 0, 0.0
    for ev in envs:
        nl, km = ev.profile_read()
        n_launch += nl
        kern_ms += km
        ev.profile_enable(False)
        flags = ev.error_flags
        assert not flags.any(), "env error flags raised during the bench: %s" % flags[flags != 0][:8]

    for g in range(P):
        if tickets[g] is not None:
            tickets[g].wait()
    rows_line = None
    if args.mode == "epilogue" and not args.no_rows_line:
        # the same loop with the stand-in policy as its own kernel in front of every transition (the rounds 1-3 headline flavour),
        # timed in the same process right after the headline region (both values on the line)
        mode_now[0] = "rows"
        for _ in range(max(2, args.warmup)):
            one_step()
        torch.cuda.synchronize(dev)
        barrier()
        tr = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        torch.cuda.synchronize(dev)
        el_rows = time.perf_counter() - tr
        barrier()
        mode_now[0] = "epilogue"
        if dist is not None:
            t = torch.tensor([el_rows], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_rows = float(t.item())
        rows_line = {"value": world * n_local * args.steps / el_rows, "unit": "env-steps/s", "ms_per_step": el_rows / args.steps * 1e3,
                     "what": "policy kernel -> float32 [N,9] leaf rows -> transition kernel per step (two dispatches; the default of rounds 1-3), "
                             "same process, timed right after the headline region"}
        for ev in envs:
            ev.profile_read()  # (its launches do not belong to the headline kernel average)

    if dist is not None:
        t = torch.tensor(regions, dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # (per region: the slowest rank's)
        regions = [float(x) for x in t.tolist()]
    elapsed = sorted(regions)[len(regions) // 2]
    kern_avg_ms = kern_ms / n_launch if n_launch else elapsed / args.steps * 1e3  # (--time-every 0: the step's time)
    per_rank_us = [kern_avg_ms * 1e3]
    if dist is not None:
        mine = torch.tensor([kern_avg_ms * 1e3], dtype=torch.float64, device=red_dev)
        allk = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allk, mine)
        per_rank_us = [float(x.item()) for x in allk]
        kern_avg_ms = max(per_rank_us) / 1e3  # the slowest rank's kernel bounds the job

    total_steps = world * n_local * args.steps
    value = total_steps / elapsed
    achieved_gbs = B * n_grp / (kern_avg_ms * 1e-3) / 1e9  # per launch (n_grp envs)
    pmc, pmc_src = pmc_profile(args.workload, n_grp)
    traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
    # the transition kernel as rocprofv3 names it (profiles/r0x_trace_<workload>.txt); ACT: 0 = rows, 2 = stand-in policy
    act = 2 if args.mode == "fused" else 0
    stab = "true" if w["setting"] != 2 else "false"
    if w["cont"]:
        # candidate table in HBM: beyond 8192 slots (pct_create: an explicit capacity, or the default of bins beyond 12 units under the
        # stability settings; setting 2 defaults to an 8192-slot LDS table there since round 5)
        cc = args.candidate_capacity or ((8192 if w["setting"] == 2 else 32768) if max(w["container"]) > 12 else 2048)
        gt = "true" if cc > 8192 else "false"
        kernel_name = ("void pct::pct_continuous_kernel<%d, false, %s, %s, false>(pct::ContinuousParams, void const*, int, int, "
                       "int const*, int)" % (act, gt, stab))
        if w["setting"] == 2 and cc == 8192:
            # round 6: the two-wave candidate pipeline (csrc/pct_continuous_pipe.hip) is the normal pass wherever the 8192-slot LDS table is
            kernel_name = ("void pct::pct_continuous_kernel<%d, false, false, false, false, true>(pct::ContinuousParams, void const*, int, "
                           "int, int const*, int)" % act)
    else:
        kernel_name = ("void pct::pct_discrete_kernel<unsigned int, 5, %d, false, %s, 0, 0>(pct::DiscreteParams, void const*, "
                       "int, int, int const*, int)" % (act, stab))
        if w["setting"] == 2 and not args.no_overflow_retry and n_grp <= 4096:
            # round 6: where every env is resident at once the launch carries the retry pass as its own tail workgroups
            # (pct_discrete_tail_kernel: one dispatch per step); its duration is the normal pass + the tail's hand-over
            kernel_name = ("void pct::pct_discrete_tail_kernel<unsigned int, 5, %d>(pct::DiscreteParams, void const*, int, int, "
                           "pct::DiscreteParams)" % (1 if args.mode == "slot" else act))

    kernel_src = "assembled from the template arguments (no committed kernel-stats CSV of this workload names it)"
    traced, traced_path = traced_kernel_name(args.workload, kernel_name.split("(")[0].replace("void ", ""))
    if traced:
        kernel_name, kernel_src = traced, traced_path + " (rocprofv3 --kernel-trace --stats of this command)"

    out = {
        "metric": w["metric"],
        "value": value,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "timed_regions": {"repeats": repeats, "reported": "median", "steps_each": args.steps,
                          "ms_per_step_each": [r / args.steps * 1e3 for r in regions]},
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64" if w["cont"] else ("i32+f64" if w["setting"] != 2 else "i32"),
        "data": "synthetic",
        "config": {
            "workload": (w["what"] % n_local) + "; per step: " + {
                "epilogue": "float32 [N,9] leaf rows (HBM, written by the previous launch's stand-in policy epilogue) -> "
                            "transition kernel (observation rows rewritten, auto-reset, next rows)",
                "rows": "policy kernel -> float32 [N,9] leaf rows -> transition kernel (observation rows rewritten, auto-reset)",
                "fused": "transition kernel with the stand-in policy inside (no rows)",
                "host": "policy kernel -> rows to the host -> VecEnv.step(numpy) -> reward / done / infos on the host",
                "host_overlap": "policy kernel -> transition kernel (rows stay on the device) -> packed reward / done / infos to the host, "
                                "consumed one step late (step_outputs_async)",
                "slot": "stand-in policy kernel on the rollout slot's observation (one launch) -> int64 leaf index in actions[t] -> pct_step_index with the slot bound: "
                        "every observation row (the B = 36 (I+L+1) + 41 algorithmic bytes per env), reward and mask written into "
                        "storage.py-shaped [T+1,N,...] tensors by the transition kernel"}[args.mode],
            "name": args.workload,
            "envs_per_gpu": n_local,
            "global_envs": world * n_local,
            "mode": args.mode,
            "desync_steps": max(0, args.desync),
            "pipelines": P,
            "overflow_retry_pass": not args.no_overflow_retry,
            "lstsq": args.lstsq,
            "ems_capacity": args.ems_capacity or "default", "candidate_capacity": args.candidate_capacity or "default",
            "parallelism": "envs sharded by global id x%d, no collective on the step path" % world,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": (pmc_src + " (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of this command in a "
                               "separate profiled run; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes)") if traffic else None,
            "kernel": kernel_name,
            "kernel_name_source": kernel_src,
            "kernel_avg_us": kern_avg_ms * 1e3,
            "kernel_avg_us_per_rank": per_rank_us,
            "launches_timed": n_launch,
            "timed_every": time_every,
            "alg_bytes_per_env_step": B,
            "envs_per_launch": n_grp,
        },
    }
    if pmc and pmc.get("valu_salu_insts_per_launch"):
        ginst = pmc["valu_salu_insts_per_launch"] / (kern_avg_ms * 1e-3) / 1e9
        out["roofline_issue"] = {"bound": "issue", "achieved": ginst, "peak": ISSUE_PEAK_GINST, "unit": "G wave-instructions/s",
                                 "frac": ginst / ISSUE_PEAK_GINST, "insts_per_launch": pmc["valu_salu_insts_per_launch"],
                                 "insts_source": pmc_src + " (SQ_INSTS_VALU + SQ_INSTS_SALU, separate profiled run)",
                                 "peak_model": "256 CUs x 2.4 GHz x (4 SIMD-32 x 1 wave64 VALU / 2 cycles + 1 SALU / cycle)"}
    else:
        out["roofline_issue"] = None
    if rows_line is not None:
        out["rows_mode"] = rows_line
        out["config"]["headline_note"] = ("`value` is the transition kernel fed by the previous launch's policy epilogue: the env hot path "
                                          "alone, an UPPER BOUND for a loop that dispatches a policy between two steps; `rows_mode` is the "
                                          "same loop with the stand-in policy as its own kernel (the flavour rounds 1-3 reported as `value`)")
    if rank == 0 and not args.no_cpu_baseline:
        # (N > 1: timed on rank 0 after the final barrier, while the other ranks wait in destroy_process_group)
        out["cpu_baseline"] = cpu_baseline(w, args.cpu_seconds, args.lstsq, n_local)
    elif rank == 0:
        out["cpu_baseline"] = None
    out["cpu_baseline_reference"] = reference_baseline(w)
    if rank == 0:
        print(json.dumps(out))
    for ev in envs:
        ev.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

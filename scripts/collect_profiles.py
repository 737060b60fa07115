"""Turns the rocprofv3 rocpd databases under gpurun_out/prof_<tag>_<workload>/ into the small text / JSON
summaries committed under profiles/:  <tag>_trace_<workload>.txt (kernel-trace stats) and
<tag>_pmc_<workload>.json (counters per launch; bench.py reads hbm_bytes_per_launch and
valu_salu_insts_per_launch from it for roofline.traffic / roofline_issue).
    python scripts/collect_profiles.py <tag> <workload> <envs-per-launch> <alg-bytes-per-env-step>"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, wl = sys.argv[1], sys.argv[2]
envs = int(sys.argv[3])
alg = int(sys.argv[4])
tsteps = int(sys.argv[5]) if len(sys.argv) > 5 else 2000  # timed steps of the traced run
psteps = int(sys.argv[6]) if len(sys.argv) > 6 else 100   # timed (= warm-up) steps of each PMC run
suf = os.environ.get("PROFILE_SUFFIX", "")
src = os.path.join(ROOT, "gpurun_out", "prof_%s_%s%s" % (tag, wl, suf))
dst = os.environ.get("PCT_PROFILE_DST") or os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
# the step kernel's name: pct_discrete_kernel / pct_continuous_kernel, or -- round 6 -- pct_discrete_tail_kernel where the launch
# carries the retry pass as its own tail workgroups (the plain setting-2 steps of a batch that is resident at once: c2)
KERNEL = "pct_continuous_kernel" if wl in ("c3", "c5", "c3s1") else "pct_discrete_%kernel"

out = {"tag": tag, "workload": wl, "envs_per_launch": envs,
       "command": "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --workload %s --steps %d --warmup 200; "
                  "--desync 0 --steps %d --warmup 100 for each --pmc pass (scripts/profile_gpu.sh)" % (wl, tsteps, psteps)}
db = os.path.join(src, "trace", "trace_results.db")
lines = []
if os.path.exists(db):
    cur = sqlite3.connect(db).cursor()
    # one row per (kernel, grid): the small-grid retry pass is the same kernel template as the normal pass
    rows = list(cur.execute("select name || ' [grid ' || (grid_x / workgroup_x) || ']', count(*), sum(duration), avg(duration), "
                            "min(duration), max(duration), max(lds_size), max(scratch_size), max(vgpr_count), max(sgpr_count), "
                            "max(grid_x), max(workgroup_x) from kernels group by name, grid_x order by sum(duration) desc"))
    tot = sum(r[2] for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary (%s, workload %s): python bench.py --workload %s --steps %d --warmup 200" % (tag, wl, wl, tsteps),
             "%-100s %8s %12s %10s %10s %10s %6s %7s %5s %5s %9s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "lds_B", "vgpr", "sgpr", "scratch_B")]
    for r in rows[:8]:
        nm = r[0] if len(r[0]) <= 100 else r[0][:84] + ".." + r[0][-14:]
        lines.append("%-100s %8d %12.1f %10.2f %10.2f %10.2f %6.2f %7d %5d %5d %9d" % (
            nm, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100 * r[2] / tot, r[6], r[8], r[9], r[7]))
    # steady-state average of the step kernel: the timed 2000 launches are the last 2000 of the main (non-retry) grid
    d = [x[0] for x in cur.execute("select duration from kernels where name like ? and grid_x >= ? order by start",
                                   ("%" + KERNEL + "%", envs * 64))]
    steady = d[-tsteps:]
    out["step_kernel_avg_us_timed_region"] = sum(steady) / max(len(steady), 1) / 1e3
    lines += ["", "step kernel (grid = %d envs), last %d launches (= bench timed region): avg %.2f us" % (
        envs, len(steady), out["step_kernel_avg_us_timed_region"])]
    bj = os.path.join(src, "trace_bench.json")
    if os.path.exists(bj):
        for ln in open(bj):
            if ln.startswith("{"):
                lines += ["", "bench line of the traced run:", ln.strip()]
    open(os.path.join(dst, "%s_trace_%s%s.txt" % (tag, wl, suf)), "w").write("\n".join(lines) + "\n")

pmc = {}
for sub in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    p = os.path.join(src, sub, "pmc_results.db")
    if not os.path.exists(p):
        continue
    c = sqlite3.connect(p).cursor()
    rows = list(c.execute("select counter_name, dispatch_id, value from counters_collection where kernel_name like ?",
                          ("%" + KERNEL + "%",)))
    if not rows:
        continue
    # steady state: the last 50 steps; with a retry pass every step has two dispatches of this kernel, so the
    # counters are summed per step = per pair of consecutive dispatch ids of the policy/transition sequence
    by_d = {}
    for n, did, v in rows:
        by_d.setdefault(n, {})[did] = v
    for n, dv in by_d.items():
        ids = sorted(dv)
        per_step = len(ids) / (2.0 * psteps)  # warm-up + timed steps (+ reset); 1 where the step is one dispatch, 2 with a retry dispatch
        k = max(1, int(round(per_step)))
        m = max(1, psteps // 2)
        last = ids[-m * k:]
        pmc[n] = sum(dv[i] for i in last) / float(m)
out["pmc_per_launch"] = pmc
if "SQ_INSTS_VALU" in pmc and "SQ_INSTS_SALU" in pmc:
    out["valu_salu_insts_per_launch"] = pmc["SQ_INSTS_VALU"] + pmc["SQ_INSTS_SALU"]
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide
    # coalesced read stream -> doubled (conservative for our narrow reads); WRITE_SIZE is taken as reported.
    fetch, write = pmc["FETCH_SIZE"] * 1024, pmc["WRITE_SIZE"] * 1024
    out.update({"fetch_bytes_raw": fetch, "fetch_bytes_corrected_x2": 2 * fetch, "write_bytes": write,
                "hbm_bytes_per_launch": 2 * fetch + write, "algorithmic_bytes_per_launch": alg * envs})
json.dump(out, open(os.path.join(dst, "%s_pmc_%s%s.json" % (tag, wl, suf)), "w"), indent=1)
print("\n".join(lines[:5]))
print(json.dumps({k: v for k, v in out.items() if k != "pmc_per_launch"}, indent=1))

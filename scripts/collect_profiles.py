"""Turns the rocprofv3 rocpd databases under gpurun_out/prof_<tag>/ into the small text/JSON
summaries committed under profiles/ (and profiles/pmc_traffic.json, which bench.py reads
for roofline.traffic)."""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
KERNEL = "pct_discrete_kernel<unsigned int, 5, 0, false, false, 0"

out = {"tag": tag, "command": "python bench.py --no-cpu-baseline --steps 2000 --warmup 200 (kernel trace); "
                              "--steps 200 --warmup 50 for each --pmc pass"}
db = os.path.join(src, "trace", "trace_results.db")
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                        "max(lds_size), max(scratch_size), max(vgpr_count), max(sgpr_count), max(grid_x), max(workgroup_x) "
                        "from kernels group by name order by sum(duration) desc"))
tot = sum(r[2] for r in rows)
lines = ["# rocprofv3 --kernel-trace --stats summary (%s): python bench.py --steps 2000 --warmup 200" % tag,
         "%-100s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
ks = []
for r in rows:
    lines.append("%-100s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (r[0][:100], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                                 r[5] / 1e3, 100 * r[2] / tot))
    ks.append(dict(name=r[0], calls=r[1], total_us=r[2] / 1e3, avg_us=r[3] / 1e3, min_us=r[4] / 1e3, max_us=r[5] / 1e3,
                   lds_bytes=r[6], scratch=r[7], vgpr=r[8], sgpr=r[9], grid=r[10], block=r[11]))
# steady-state average of the step kernel: the timed 2000 launches are the last 2000
d = [x[0] for x in cur.execute("select duration from kernels where name like ? order by start", ("%" + KERNEL + "%",))]
steady = d[-2000:]
out["step_kernel_avg_us_timed_region"] = sum(steady) / len(steady) / 1e3
lines.append("")
lines.append("step kernel, last 2000 launches (= bench timed region): avg %.2f us" % out["step_kernel_avg_us_timed_region"])
out["kernels"] = ks
open(os.path.join(dst, "%s_kernel_trace_stats.txt" % tag), "w").write("\n".join(lines) + "\n")

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
    maxd = max(r[1] for r in rows)
    agg = {}
    for n, did, v in rows:
        if did > maxd - 200:  # steady state: last 100 steps (2 kernels per step)
            agg.setdefault(n, []).append(v)
    for n, v in agg.items():
        pmc[n] = sum(v) / len(v)
out["pmc_per_launch_4096_envs"] = pmc
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE reports half the
    # bytes of a wide coalesced read stream -> doubled (conservative for our narrow reads);
    # WRITE_SIZE is taken as reported.
    fetch = pmc["FETCH_SIZE"] * 1024
    write = pmc["WRITE_SIZE"] * 1024
    traffic = {"fetch_bytes_raw": fetch, "fetch_bytes_corrected_x2": 2 * fetch, "write_bytes": write,
               "hbm_bytes_per_launch": 2 * fetch + write, "envs_per_launch": 4096,
               "algorithmic_bytes_per_launch": 4757 * 4096, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, " + tag}
    json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    out["traffic"] = traffic
json.dump(out, open(os.path.join(dst, "%s_profile_summary.json" % tag), "w"), indent=1)
print("\n".join(lines[:6]))
print(json.dumps(out.get("traffic"), indent=1))

import sqlite3, sys, glob, os
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "pct_discrete_kernel<unsigned int, 5, 0"
for db in sorted(glob.glob(os.path.join(d, "*", "*.db"))):
    cur = sqlite3.connect(db).cursor()
    # skip warm-up dispatches: keep the last 100 dispatches of the kernel
    q = ("select counter_name, count(*), avg(value) from (select counter_name, value, dispatch_id from counters_collection "
         "where kernel_name like ? order by dispatch_id desc) group by counter_name")
    rows = list(cur.execute("select counter_name, dispatch_id, value from counters_collection where kernel_name like ?", ("%" + pat + "%",)))
    if not rows:
        continue
    maxd = max(r[1] for r in rows)
    agg = {}
    for n, did, v in rows:
        if did > maxd - 200:  # 2 kernels per step -> last 100 steps
            agg.setdefault(n, []).append(v)
    for n in sorted(agg):
        print("%-28s n=%4d avg=%14.1f per-wave(4096)=%10.1f" % (n, len(agg[n]), sum(agg[n]) / len(agg[n]), sum(agg[n]) / len(agg[n]) / 4096))

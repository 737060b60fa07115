"""one-line summary of a bench JSON line: python scripts/r04/bline.py <label> <file>"""
import json, sys
label, path = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(path) if l.startswith("{")][-1])
    r = d["roofline"]
    print("%-34s %8.3f M/s  ms/step %.4f  kernel_us %7.1f  frac %.4f  launches %d" % (
        label, d["value"] / 1e6, d["ms_per_step"], r["kernel_avg_us"], r["frac"], r["launches_timed"]))
except Exception as e:
    print(label, "ERR", e, open(path.replace(".json", ".err")).read()[-400:] if path.endswith(".json") else "")

"""usage: tools/pmc_traffic.py <raw dir with pmc_fetch/ and pmc_write/> <output prefix> [steps]
Per kernel: dispatches, mean FETCH_SIZE / WRITE_SIZE (KB) and the HBM bytes per launch they stand for
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section: separate --pmc passes; FETCH_SIZE doubled on gfx950, WRITE_SIZE as
reported).  With `steps` also the bytes per step.  JSON on stdout."""
import collections
import csv
import glob
import json
import re
import sys

raw, prefix = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else None


def avg(sub, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{raw}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
                agg[m.group(1) if m else r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


fetch, write = avg("pmc_fetch", "FETCH_SIZE"), avg("pmc_write", "WRITE_SIZE")
out = {}
for k in sorted(fetch, key=lambda k: -(2 * fetch[k][0] + write.get(k, (0, 0))[0]) * fetch[k][1]):
    f_kb, n = fetch[k]
    w_kb, _ = write.get(k, (0.0, 0))
    out[k] = {"dispatches": n, "FETCH_SIZE_KB_avg": round(f_kb, 1), "WRITE_SIZE_KB_avg": round(w_kb, 1),
              "hbm_read_bytes_per_launch": round(2.0 * f_kb * 1024.0), "hbm_write_bytes_per_launch": round(w_kb * 1024.0),
              "hbm_bytes_per_launch": round((2.0 * f_kb + w_kb) * 1024.0)}
    if steps:
        out[k]["launches_per_step"] = round(n / steps, 2)
        out[k]["hbm_bytes_per_step"] = round((2.0 * f_kb + w_kb) * 1024.0 * n / steps)
if steps:
    out["_total_hbm_bytes_per_step"] = sum(v["hbm_bytes_per_step"] for v in out.values())
print(json.dumps(out, indent=1))

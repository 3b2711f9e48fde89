"""Turn the raw rocprofv3 output of a gpurun call (gpurun_out/<round>/...) into the small summaries committed under
profiles/: the per-kernel stats CSV, and the PMC-derived HBM traffic of the two dominant kernels.

    python tools/summarize_profiles.py gpurun_out/r01 r01 [destination, default profiles/]

(The raw traces exceed what gpurun copies back: run this ON the GPU box with a destination under gpurun_out/ and copy the
summaries into profiles/ afterwards.)

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE (KB) collected in
SEPARATE --pmc passes; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads, so it is doubled;
WRITE_SIZE is used as reported (it matches the algorithmic write volume of the dense layer to 2 %)."""
import collections
import csv
import json
import re
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", f"{tag}_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))


def avg_counter(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
            agg[m.group(1) if m else r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


fetch = avg_counter(os.path.join(src, "pmc_fetch", f"{tag}_counter_collection.csv"), "FETCH_SIZE")
write = avg_counter(os.path.join(src, "pmc_write", f"{tag}_counter_collection.csv"), "WRITE_SIZE")
out = {}
for k in fetch:
    f_kb, n = fetch[k]
    w_kb, _ = write.get(k, (0.0, 0))
    out[k] = {"dispatches": n, "FETCH_SIZE_KB_avg": f_kb, "WRITE_SIZE_KB_avg": w_kb,
              "hbm_read_bytes_per_launch": 2.0 * f_kb * 1024.0, "hbm_write_bytes_per_launch": w_kb * 1024.0,
              "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"), "w"), indent=1)
# the roofline kernel of bench.py = exactly the launch set its event timing uses (EventProfiler.summary: work["n"] > 64 on the
# 16-bit kernels): k_linear_dma<TN, ...> with TN * 32 > 64, i.e. TN >= 3 -- TN = 2 (N = 64: the last conv layer, the embedding
# tail) is NOT part of it -- and no instance that only ran in warm-up (fewer dispatches than steps: the fp32 fallback launches
# before the bounds exist).  r03's file mixed the TN = 2 launches and six warm-up launches in.
STEPS = int(os.environ.get("RGNN_PROFILE_STEPS", "0")) or None


def _wide(k):
    m = re.match(r"k_linear_dma<(\d+), (true|false), (\d+)", k)
    if m is not None:
        return int(m.group(1)) * 32 > 64 and int(m.group(3)) >= 1        # (FMT 0 = the bf16x3 warm-up form)
    m = re.match(r"k_linear_x3<\d+, (\d+)", k)
    return m is not None and int(m.group(1)) > 64


lin = [(k, v) for k, v in out.items() if _wide(k)]
if lin:
    n = sum(v["dispatches"] for _, v in lin)
    json.dump({"kernel": "k_linear_dma<TN >= 3, .., FMT >= 1> (the N > 64 launches of the 16-bit dense kernel: the set bench.py times), "
                         "dispatch-weighted mean",
               "instances": {k: {"dispatches": v["dispatches"], "hbm_bytes_per_launch": v["hbm_bytes_per_launch"],
                                 "hbm_read_bytes_per_launch": v["hbm_read_bytes_per_launch"],
                                 "hbm_write_bytes_per_launch": v["hbm_write_bytes_per_launch"]} for k, v in lin},
               "hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] * v["dispatches"] for _, v in lin) / n,
               "hbm_read_bytes_per_launch": sum(v["hbm_read_bytes_per_launch"] * v["dispatches"] for _, v in lin) / n,
               "hbm_write_bytes_per_launch": sum(v["hbm_write_bytes_per_launch"] * v["dispatches"] for _, v in lin) / n,
               "source": f"profiles/{tag}_pmc_hbm_traffic.json"},
              open(os.path.join(dst, "pmc_linear_summary.json"), "w"), indent=1)
# the edge kernel of the step: the window form where the layers take it (r04: also on the r = 1 m graphs of C2), else the per-edge one
mp = [(k, v) for k, v in out.items() if k.startswith("k_mpnn_win")]
edge_kernel = "k_mpnn_win"
if not mp:
    mp = [(k, v) for k, v in out.items() if k.startswith("k_mpnn_max") or k.startswith("k_mpnn_fast")]
    edge_kernel = "k_mpnn_max"
if mp:
    n = sum(v["dispatches"] for _, v in mp)
    json.dump({"kernel": ("k_mpnn_win (window form of the edge kernel)" if edge_kernel == "k_mpnn_win" else
                          "k_mpnn_max / k_mpnn_fast (edge kernel)") + ", dispatch-weighted mean over the layers of the step",
               "edge_kernel": edge_kernel,
               "instances": {k: v["dispatches"] for k, v in mp},
               "hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] * v["dispatches"] for _, v in mp) / n,
               "hbm_read_bytes_per_launch": sum(v["hbm_read_bytes_per_launch"] * v["dispatches"] for _, v in mp) / n,
               "hbm_write_bytes_per_launch": sum(v["hbm_write_bytes_per_launch"] * v["dispatches"] for _, v in mp) / n,
               "source": f"profiles/{tag}_pmc_hbm_traffic.json"},
              open(os.path.join(dst, "pmc_mpnn_summary.json"), "w"), indent=1)
log = os.path.join(src, "bench_under_rocprof.log")
if os.path.exists(log):
    for line in open(log):
        if line.startswith("{"):
            open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w").write(line)
print(json.dumps(out, indent=1))

"""Kernels of the LAST step in a rocprofv3 kernel-trace CSV with their start offsets (gaps between launches become visible):
    python tools/attic/list_step_timeline.py <kernel_trace.csv>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_grid_frame" in r["Kernel_Name"] or "k_frame_grid" in r["Kernel_Name"]]
i0, i1 = starts[-2], starts[-1]
tot, t0 = 0.0, int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    m = re.search(r"(k_\w+(<[^>]*>)?|__amd\w+|at::native::\w+)", r["Kernel_Name"])
    name = (m.group(1) if m else r["Kernel_Name"])[:50]
    at = (int(r["Start_Timestamp"]) - t0) / 1e3
    print(f"{name:50s} {d:7.1f} us  at {at:8.1f}")
print("kernel total", round(tot, 1), "us; step span", (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, "us;", i1 - i0, "launches")

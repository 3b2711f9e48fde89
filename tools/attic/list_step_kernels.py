"""Print the kernels of the LAST step found in a rocprofv3 kernel-trace CSV (from the last k_frame_grid on)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_frame_grid" in r["Kernel_Name"] or "k_grid_frame" in r["Kernel_Name"]]
i0 = starts[-2] if len(starts) > 1 else starts[-1]
i1 = starts[-1] if len(starts) > 1 else len(rows)
tot = 0
for r in rows[i0:i1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    m = re.search(r"(k_\w+(<[^>]*>)?|__amd\w+|at::native::\w+)", r["Kernel_Name"])
    print(f"{(m.group(1) if m else r['Kernel_Name'])[:60]:60s} {d:9.1f} us  grid={r.get('Grid_Size','')}")
print("step kernel total us", tot, "span us", (int(rows[i1-1]["End_Timestamp"]) - int(rows[i0]["Start_Timestamp"]))/1e3)

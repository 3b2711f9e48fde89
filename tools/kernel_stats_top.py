"""Print the heaviest kernels of a rocprofv3 --kernel-trace --stats run:  python tools/kernel_stats_top.py <kernel_stats.csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms")
for r in rows[:n]:
    print(f"{float(r['TotalDurationNs']) / 1e6:8.2f} ms {int(r['Calls']):5d} calls {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:120]}")

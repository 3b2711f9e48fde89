"""Device busy / idle time from a rocprofv3 kernel trace (csv): union of the kernels' [start, end] intervals over the span of the trace's
last `frac` part (the steady state), the gaps by length, and the kernels that FOLLOW the longest gaps.
    python tools/trace_idle.py <..._kernel_trace.csv> [frac=0.6]"""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t1 - (t1 - t0) * frac
ev = [e for e in ev if e[0] >= cut]
busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
for s, e, name in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, name))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(e[1] for e in ev) - ev[0][0]
print(f"span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), {len(ev)} kernels, sum of durations {sum(e[1] - e[0] for e in ev) / 1e6:.2f} ms")
hist = Counter()
for g, _ in gaps:
    hist["<2us" if g < 2000 else "2-5us" if g < 5000 else "5-10us" if g < 10000 else "10-30us" if g < 30000 else "30-100us" if g < 100000 else ">100us"] += g
print("idle time by gap length (ms):", {k: round(v / 1e6, 2) for k, v in hist.items()})
after = Counter()
for g, name in gaps:
    after[name[:60]] += g
print("idle time by the kernel that follows the gap (ms):")
for k, v in after.most_common(14):
    print(f"  {v / 1e6:7.2f}  {k}")

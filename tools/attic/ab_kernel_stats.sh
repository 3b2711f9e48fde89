#!/bin/bash
# usage (GPU box, repo root): tools/attic/ab_kernel_stats.sh <other tree> <outdir under gpurun_out>  -- per-kernel averages of the default
# bench of this tree and of another built copy (rocprofv3 --kernel-trace --stats each), side by side.
OTHER=${1:?tree}; TAG=${2:-ab}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/$TAG; cd /tmp; export TMPDIR=/tmp
for t in new other; do
  d=$ROOT; [ $t = other ] && d=$ROOT/$OTHER
  (cd $d && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abks_$t -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > /tmp/abks_$t.log 2>&1)
  cp $(find /tmp/abks_$t -name "r_kernel_stats.csv" | head -1) $ROOT/gpurun_out/$TAG/${t}_kernel_stats.csv
done
python3 - $ROOT/gpurun_out/$TAG <<'P'
import csv, sys, re
d = sys.argv[1]
def load(f):
    out = {}
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"\(.*", "", n).replace("void ", "")
        out[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6)
    return out
a, b = load(d + "/new_kernel_stats.csv"), load(d + "/other_kernel_stats.csv")
print("%-52s %6s %9s %9s %8s" % ("kernel", "calls", "new us", "other us", "d ms"))
tot = 0.0
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0, 0))[2] + b.get(k, (0, 0, 0))[2])):
    x, y = a.get(k, (0, 0.0, 0.0)), b.get(k, (0, 0.0, 0.0))
    tot += x[2] - y[2]
    if x[2] + y[2] > 0.05: print("%-52s %6d %9.1f %9.1f %+8.3f" % (k[:52], x[0] or y[0], x[1], y[1], x[2] - y[2]))
print("total delta ms over the run: %+.3f" % tot)
P

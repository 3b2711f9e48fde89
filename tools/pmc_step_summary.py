"""usage: tools/pmc_step_summary.py <dir with p*/ passes> <kernel name substring> [min duration share]
Averages every collected counter over the dispatches of the kernels whose name contains the substring."""
import collections, csv, glob, sys
root, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"{k:28s} mean={acc[k][0] / acc[k][1]:16.1f} n={acc[k][1]}")
g = lambda k: (acc[k][0] / acc[k][1] or float("nan")) if k in acc else float("nan")
print("MFMA busy / SQ busy cycles      :", g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES"))
print("wave cycles waiting (any)       :", g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"))
print("wave cycles waiting on an inst  :", g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"))
print("LDS bank conflict / LDS active  :", g("SQ_LDS_BANK_CONFLICT") / g("SQ_ACTIVE_INST_LDS"))

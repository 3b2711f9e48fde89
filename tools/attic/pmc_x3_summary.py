"""Mean of every counter per kernel over the dispatches of a tools/attic/pmc_x3.sh run (reads the rocprofv3 counter CSVs)."""
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        short = "dma" if "k_linear_dma" in k else ("x3" if "k_linear_x3" in k else None)
        if short is None:
            continue
        acc[short + " " + k.split("<")[1].split(">")[0] if "<" in k else short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print("==", k)
    m = {c: sum(v) / len(v) for c, v in d.items()}
    for c in sorted(m):
        print(f"  {c:28s} mean={m[c]:18.1f} n={len(d[c])}")
    g = m.get
    if g("SQ_INSTS_MFMA") and g("SQ_BUSY_CYCLES"):
        print(f"  matrix pipe busy / (4 SIMD x SQ_BUSY_CYCLES/..)  : MFMA_BUSY/INSTS_MFMA = {g('SQ_VALU_MFMA_BUSY_CYCLES', 0) / g('SQ_INSTS_MFMA'):.1f} cycles per MFMA")
    if g("SQ_WAVE_CYCLES"):
        print(f"  waiting (s_waitcnt / barrier) share of wave cycles : {g('SQ_WAIT_ANY', 0) / g('SQ_WAVE_CYCLES'):.3f}")
        print(f"  issue-stalled share of wave cycles                 : {g('SQ_WAIT_INST_ANY', 0) / g('SQ_WAVE_CYCLES'):.3f}")
        print(f"  active-issuing share of wave cycles                : {g('SQ_ACTIVE_INST_ANY', 0) / g('SQ_WAVE_CYCLES'):.3f}")
    if g("SQ_LDS_IDX_ACTIVE"):
        print(f"  LDS bank conflict cycles / LDS active cycles       : {g('SQ_LDS_BANK_CONFLICT', 0) / g('SQ_LDS_IDX_ACTIVE'):.3f}")
    if g("FETCH_SIZE") is not None:
        print(f"  HBM read  MB (FETCH_SIZE KB x2 on gfx950)          : {g('FETCH_SIZE', 0) * 2 / 1024:.1f}")
    if g("WRITE_SIZE") is not None:
        print(f"  HBM write MB (WRITE_SIZE KB)                       : {g('WRITE_SIZE', 0) / 1024:.1f}")

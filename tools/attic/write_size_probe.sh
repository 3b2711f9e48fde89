#!/bin/bash
# usage (GPU box, repo root): tools/attic/write_size_probe.sh  -> gpurun_out/write_size_calibration.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $ROOT/tools/attic/write_size_probe.hip -o /tmp/write_size_probe || exit 1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/wsp_w -o w -- /tmp/write_size_probe > /tmp/wsp_w.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/wsp_f -o f -- /tmp/write_size_probe > /tmp/wsp_f.log 2>&1
python3 - <<'PY' > $ROOT/gpurun_out/write_size_calibration.txt
import csv, glob, collections
BYTES = 512 << 20
rows = BYTES // (464 * 4)
known = {"k_store16": BYTES, "k_store4": BYTES, "k_store4_rows": rows * 464 * 4, "k_store16_nt": BYTES, "k_store_rows464": rows * 464 * 4, "k_load16": BYTES}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/wsp_w", "/tmp/wsp_f"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("rocprofv3 WRITE_SIZE / FETCH_SIZE (KB, mean of 3 dispatches) against the bytes each kernel moves exactly once (512 MiB buffers)")
print(f"{'kernel':18s} {'known MB':>10s} {'WRITE_SIZE MB':>14s} {'ratio':>7s} {'FETCH_SIZE MB':>14s} {'ratio':>7s}")
for k, nb in known.items():
    w = acc[k].get("WRITE_SIZE", [float('nan')]); f = acc[k].get("FETCH_SIZE", [float('nan')])
    wm, fm = sum(w) / len(w) * 1024, sum(f) / len(f) * 1024
    print(f"{k:18s} {nb / 1e6:10.1f} {wm / 1e6:14.1f} {wm / nb:7.3f} {fm / 1e6:14.1f} {fm / nb:7.3f}")
PY
cat $ROOT/gpurun_out/write_size_calibration.txt

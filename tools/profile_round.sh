#!/bin/bash
# usage (on the GPU box, from the repo root):  tools/profile_round.sh r01
# Produces gpurun_out/<tag>/{trace,pmc_fetch,pmc_write} for tools/summarize_profiles.py:
#   1. rocprofv3 --kernel-trace --stats of the default bench command (per-kernel time),
#   2. FETCH_SIZE and 3. WRITE_SIZE in SEPARATE --pmc passes (MI355X_MICROARCH.md, HBM section), kernel trace only.
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${RGNN_PROFILE_RAW:-/tmp/rgnn_prof}/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export RGNN_NO_PLAN_SIDE=1   # every kernel alone on the device: per-kernel durations and counters mean what they say
CMD="python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-pcie --no-live-traffic"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- $CMD > $OUT/bench_under_rocprof.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o $TAG -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o $TAG -- $CMD > $OUT/pmc_write.log 2>&1
mkdir -p $ROOT/gpurun_out/${TAG}_summary
python3 $ROOT/tools/summarize_profiles.py $OUT $TAG $ROOT/gpurun_out/${TAG}_summary > /dev/null
ls $ROOT/gpurun_out/${TAG}_summary

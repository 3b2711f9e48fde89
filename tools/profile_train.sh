#!/bin/bash
# usage (GPU box, repo root): tools/profile_train.sh <tag> [steps]
# rocprofv3 --kernel-trace --stats of tools/train_step_bench.py -> gpurun_out/<tag>_train_step_kernel_stats.csv (+ the step time it printed)
TAG=${1:?tag}; STEPS=${2:-12}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
RAW=${RGNN_PROFILE_RAW:-/tmp/rgnn_prof}/$TAG/train
mkdir -p $RAW $ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o train -- python $ROOT/tools/train_step_bench.py $STEPS > $RAW/trace.log 2>&1
cp $(find $RAW/trace -name "train_kernel_stats.csv" | head -1) $ROOT/gpurun_out/${TAG}_train_step_kernel_stats.csv
grep -v Warning $RAW/trace.log | tail -3
python3 $ROOT/tools/trace_idle.py $(find $RAW/trace -name "train_kernel_trace.csv" | head -1) 0.5
python3 $ROOT/tools/kernel_stats_top.py $ROOT/gpurun_out/${TAG}_train_step_kernel_stats.csv 40

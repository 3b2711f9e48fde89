#!/bin/bash
# usage (GPU box, repo root): tools/profile_config.sh <tag> <c3|c4|c5> [sq-kernel-substring ...]
# For one of the other BASELINE configurations (tools/c{3,4,5}_profile.py: eager steps of the whole hot path):
#   1. rocprofv3 --kernel-trace --stats  -> gpurun_out/<tag>/<cfg>_kernel_stats.csv
#   2. FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (MI355X_MICROARCH.md, HBM section; kernel trace only)
#        -> gpurun_out/<tag>/<cfg>_pmc_hbm_traffic.json (per kernel: mean bytes per launch, FETCH doubled on gfx950)
#   3. SQ counter passes (<= 6 counters each) -> gpurun_out/<tag>/<cfg>_pmc_sq_<kernel>.txt for every substring given
TAG=${1:?tag}; CFG=${2:?c3|c4|c5}; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
RAW=${RGNN_PROFILE_RAW:-/tmp/rgnn_prof}/$TAG/$CFG
DST=$ROOT/gpurun_out/$TAG
mkdir -p $RAW $DST
cd /tmp; export TMPDIR=/tmp
export RGNN_NO_PLAN_SIDE=1   # every kernel alone on the device: per-kernel durations and counters mean what they say
STEPS=${RGNN_PROFILE_STEPS:-10}
CMD="python $ROOT/tools/${CFG}_profile.py $STEPS"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o $CFG -- $CMD > $RAW/trace.log 2>&1
cp $(find $RAW/trace -name "${CFG}_kernel_stats.csv" | head -1) $DST/${TAG}_${CFG}_kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/pmc_fetch -o $CFG -- $CMD > $RAW/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/pmc_write -o $CFG -- $CMD > $RAW/pmc_write.log 2>&1
python3 $ROOT/tools/pmc_traffic.py $RAW $CFG $STEPS > $DST/${TAG}_${CFG}_pmc_hbm_traffic.json
if [ $# -gt 0 ]; then
  SQCMD="python $ROOT/tools/${CFG}_profile.py 3"
  pass() { n=$1; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $RAW/p$n -o r -- $SQCMD > $RAW/p$n.log 2>&1; }
  pass 1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
  pass 2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
  pass 3 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  pass 4 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
  pass 5 GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
  for k in "$@"; do python3 $ROOT/tools/pmc_step_summary.py $RAW $k > $DST/${TAG}_${CFG}_pmc_sq_$k.txt 2>&1; done
fi
python3 $ROOT/tools/kernel_stats_top.py $DST/${TAG}_${CFG}_kernel_stats.csv 14

#!/bin/bash
# usage (GPU box, repo root): tools/pmc_step.sh <outdir-under-gpurun_out>  -- SQ counter passes (<= 6 per pass, kernel trace
# only) over a short default bench run; tools/pmc_step_summary.py averages them per kernel.
TAG=${1:-pmc_step}; OUT=${RGNN_PROFILE_RAW:-/tmp/rgnn_prof}/$TAG; mkdir -p $OUT $GRAFT_REPO_ROOT/gpurun_out/$TAG
cd /tmp; export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs --no-pcie --no-live-traffic --launch-mode eager"
pass() { n=$1; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/p$n -o r -- $CMD > $OUT/p$n.log 2>&1; }
pass 1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
pass 2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pass 3 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass 4 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
pass 5 GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
for k in k_linear_dma k_mpnn_max k_mpnn_win; do python3 $GRAFT_REPO_ROOT/tools/pmc_step_summary.py $OUT $k > $GRAFT_REPO_ROOT/gpurun_out/$TAG/sq_$k.txt 2>&1; done
ls $GRAFT_REPO_ROOT/gpurun_out/$TAG

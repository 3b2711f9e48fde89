#!/bin/bash
# usage (GPU box, repo root): tools/attic/pmc_cmd.sh <tag> <kernel-substring>[,<kernel-substring>...] <command ...>
# SQ / TCC counter passes (<= 6 counters each, kernel trace only) of an arbitrary command, summarised per kernel substring by
# tools/pmc_step_summary.py into gpurun_out/<tag>_pmc_sq_<kernel>.txt
TAG=${1:?tag}; KERNELS=${2:?kernels}; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
RAW=${RGNN_PROFILE_RAW:-/tmp/rgnn_prof}/$TAG
mkdir -p $RAW $ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
pass() { n=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $RAW/p$n -o r -- $CMD > $RAW/p$n.log 2>&1; }
CMD="$*"
pass 1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
pass 2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pass 3 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass 4 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
pass 5 GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
pass 6 FETCH_SIZE
pass 7 WRITE_SIZE
pass 8 TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum
for k in ${KERNELS//,/ }; do python3 $ROOT/tools/pmc_step_summary.py $RAW $k > $ROOT/gpurun_out/${TAG}_pmc_sq_$k.txt 2>&1; done

#!/bin/bash
# usage (GPU box, repo root): tools/attic/pmc_x3.sh <outdir-under-gpurun_out> <shape index> <variant spec>
# SQ / GRBM counter passes (kernel trace only) over tools/x3_bench.bin for ONE shape and ONE library variant.
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_x3}; SHAPE=${2:-7}; LIB=${3:-radargnn_amd/librgnn.so}; mkdir -p $OUT
ROOT=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
pass() { n=$1; shift; (cd $ROOT && timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/p$n -o r -- ./tools/x3_bench.bin -r 1 -s $SHAPE $LIB > $OUT/p$n.log 2>&1); }
pass 1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
pass 2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pass 3 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass 4 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE
pass 5 GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
pass 6 FETCH_SIZE
pass 7 WRITE_SIZE
python3 $ROOT/tools/attic/pmc_x3_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

#!/bin/bash
# usage: tools/attic/pmc_gemm.sh <binary> <outdir>   -- SQ counter passes over the GEMM micro-benchmark (<= 6 counters per pass)
BIN=$1; OUT=$2; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail > $GRAFT_REPO_ROOT/$OUT/avail.txt 2>&1
WANT="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_VMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_FLAT"
HAVE=""
for c in $WANT; do if grep -qw "$c" $GRAFT_REPO_ROOT/$OUT/avail.txt; then HAVE="$HAVE $c"; fi; done
echo "available: $HAVE" > $GRAFT_REPO_ROOT/$OUT/have.txt
set -- $HAVE
i=0
while [ $# -gt 0 ]; do
  grp=""; n=0
  while [ $# -gt 0 ] && [ $n -lt 6 ]; do grp="$grp $1"; shift; n=$((n+1)); done
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/p$i -o r -- $GRAFT_REPO_ROOT/$BIN s > $GRAFT_REPO_ROOT/$OUT/p$i.log 2>&1
done

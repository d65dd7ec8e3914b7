#!/bin/bash
# Instruction-mix / stall counters of the big kernels (tuning aid): separate rocprofv3 --pmc passes (kernel trace only),
# summarised by profiles/pmc_dump.py into gpurun_out/<tag>/pmc_sq.txt.   usage: bash profiles/pmc_sq.sh r02_sq
TAG=${1:-sq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --no-cpu --no-configs --steps 6 --warmup 2"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT" "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_IFETCH SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p -- $BENCH > $OUT/p$i.log 2>&1
  DB=$(find $OUT/p$i -name '*_results.db' | head -1)
  python profiles/pmc_dump.py $DB >> $OUT/pmc_sq.txt 2>&1
  rm -rf $OUT/p$i
done
grep -E "k_flatten_build|k_fill|k_stroke|^#|counter" $OUT/pmc_sq.txt | head -100

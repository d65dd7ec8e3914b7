#!/bin/bash
# Tuning helper: memory-pipeline counters of the emit kernels (k_fill*, k_stroke, k_cache_copy*) for one run of
# profiles/stage_times.py. Usage (on the GPU box): profiles/pmc_emit.sh <tag> [stage_times workload]   (env: VGX_FILL, VGX_LIB ...)
TAG=$1; WL=${2:-tiger}
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for set in "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU" \
           "TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_WRITE_WAVEFRONTS" \
           "TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ TCP_TOTAL_WRITE" \
           "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_TCR_TCP_STALL_CYCLES" \
           "TCP_WRITE_TAGCONFLICT_STALL_CYCLES TCP_TCC_WRITE_REQ_LATENCY TCP_TCP_TA_DATA_STALL_CYCLES TCP_LFIFO_STALL_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p -- python profiles/stage_times.py $WL > $OUT/p$i.log 2>&1
  DB=$(find $OUT/p$i -name '*_results.db' | head -1)
  python profiles/pmc_dump.py $DB k_ >> $OUT/pmc.txt 2>&1
  rm -rf $OUT/p$i
done

#!/bin/bash
# What does k_stroke wait for on BASELINE configs[3] (10k polylines x 1k segments, Round joins + caps)? Separate rocprofv3 --pmc passes
# (kernel trace only), summarised by profiles/pmc_dump.py.   usage (on the GPU box, repo root): bash profiles/pmc_stroke_round10k.sh r06_sq_round10k [config]
TAG=${1:-r06_sq_round10k}; CFG=${2:-round10k}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="python bench.py --config $CFG --no-cpu --no-configs --steps 6 --warmup 2"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT" \
           "TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_WRITE_WAVEFRONTS" \
           "TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ TCP_TOTAL_WRITE" \
           "TCP_WRITE_TAGCONFLICT_STALL_CYCLES TCP_TCC_WRITE_REQ_LATENCY TCP_TCP_TA_DATA_STALL_CYCLES TCP_LFIFO_STALL_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p -- $BENCH > $OUT/p$i.log 2>&1
  DB=$(find $OUT/p$i -name '*_results.db' | head -1)
  python profiles/pmc_dump.py $DB k_ >> $OUT/pmc_sq.txt 2>&1
  rm -rf $OUT/p$i
done
grep -A5 "k_stroke(\|k_emit_tiles\|k_tmpl_emit\|k_flatten_build" $OUT/pmc_sq.txt | head -150

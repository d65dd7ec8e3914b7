OUT=gpurun_out/r04_sq_round; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="python bench.py --config round10k --no-cpu --no-configs --steps 6 --warmup 2"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p -- $BENCH > $OUT/p$i.log 2>&1
  DB=$(find $OUT/p$i -name '*_results.db' | head -1)
  python profiles/pmc_dump.py $DB k_ >> $OUT/pmc_sq.txt 2>&1
  rm -rf $OUT/p$i
done
cat $OUT/pmc_sq.txt | head -150

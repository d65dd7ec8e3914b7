OUT=gpurun_out/r06_tcp_build; mkdir -p $OUT; export TMPDIR=/tmp
CFG=tiger10k_command_parallel
BENCH="python bench.py --config $CFG --no-cpu --no-configs --steps 6 --warmup 2"
i=0
for set in "TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ TCP_TOTAL_WRITE" "TCP_WRITE_TAGCONFLICT_STALL_CYCLES TCP_TCC_WRITE_REQ_LATENCY TCP_TCP_TA_DATA_STALL_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p -- $BENCH > $OUT/p$i.log 2>&1
  DB=$(find $OUT/p$i -name '*_results.db' | head -1)
  python profiles/pmc_dump.py $DB k_ >> $OUT/pmc_$CFG.txt 2>&1
  rm -rf $OUT/p$i
done
grep -A5 "k_flatten_build<false>" $OUT/pmc_$CFG.txt | head -40

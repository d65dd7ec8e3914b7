# SQ instruction / wait counters of the template emit kernels, Bevel against Round joins (same box): bash profiles/pmc_sq_tmpl.sh [tag]
TAG=${1:-r05_sq_tmpl}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for J in 2 1; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_IFETCH SQ_WAIT_ANY"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p -- python profiles/tmpl_phases.py 10000 $J > $OUT/p$i.log 2>&1
    DB=$(find $OUT/p$i -name '*_results.db' | head -1)
    python profiles/pmc_dump.py $DB k_tmpl_emit >> $OUT/pmc_sq_join$J.txt 2>&1
    rm -rf $OUT/p$i
  done
done
cat $OUT/pmc_sq_join2.txt $OUT/pmc_sq_join1.txt

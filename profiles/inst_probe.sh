#!/bin/bash
# Tuning run for k_flatten_inst (one gpurun call): parity of the default build, stage times of the variant builds
# (profiles/ab_variants.sh), of the command-parallel kernel (VGX_INST=0) and of several grid sizes, then SQ counters.
OUT=gpurun_out/inst_probe
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_inst.py -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
for v in "$@"; do
  VGX_LIB=vg-renderer_amd/dbg/libvgx_$v.so timeout 120 python profiles/stage_times.py 2>/dev/null | tail -1
done
echo "--- VGX_INST=0"; VGX_INST=0 timeout 120 python profiles/stage_times.py 2>/dev/null | tail -1
for w in $WAVES; do echo "--- waves $w"; VGX_INST_WAVES=$w timeout 120 python profiles/stage_times.py 2>/dev/null | tail -1; done
if [ -n "$PMC" ]; then
BENCH="python bench.py --no-cpu --no-configs --steps 6 --warmup 2 --placements 1"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p -- $BENCH > $OUT/p$i.log 2>&1
  DB=$(find $OUT/p$i -name '*_results.db' | head -1)
  python profiles/pmc_dump.py $DB k_flatten >> $OUT/pmc_sq.txt 2>&1
  rm -rf $OUT/p$i
done
cat $OUT/pmc_sq.txt
fi

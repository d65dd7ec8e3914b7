TAG=r05_f; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for CFG in round10k; do
  SFX="_$CFG"
  BENCH="python bench.py --config $CFG --no-cpu --no-configs --steps 8 --warmup 2 --placements 1"
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace$SFX.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $BENCH > $OUT/fetch$SFX.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- $BENCH > $OUT/write$SFX.log 2>&1
  grep '^{' $OUT/trace$SFX.log > $OUT/bench_profiled$SFX.json
  T=$(find $OUT/trace -name '*_results.db' | head -1); F=$(find $OUT/fetch -name '*_results.db' | head -1); W=$(find $OUT/write -name '*_results.db' | head -1)
  python profiles/summarize.py trace $T > $OUT/kernel_stats$SFX.txt 2>&1
  python profiles/summarize.py pmc $F FETCH_SIZE > $OUT/pmc_fetch$SFX.txt 2>&1
  python profiles/summarize.py pmc $W WRITE_SIZE > $OUT/pmc_write$SFX.txt 2>&1
  python profiles/summarize.py traffic $F $W "$TAG" > $OUT/traffic$SFX.json 2>&1
  rm -rf $OUT/trace $OUT/fetch $OUT/write
  head -8 $OUT/kernel_stats$SFX.txt
done

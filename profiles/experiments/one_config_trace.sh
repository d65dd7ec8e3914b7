# kernel trace of ONE bench config (no counter passes): TAG=r05_g CFG=round10k bash profiles/experiments/one_config_trace.sh
TAG=${TAG:-r05_g}; CFG=${CFG:-round10k}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="python bench.py --config $CFG --no-cpu --no-configs --steps 8 --warmup 2 --placements 1"
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace_$CFG.log 2>&1
grep '^{' $OUT/trace_$CFG.log > $OUT/bench_profiled_$CFG.json
T=$(find $OUT/trace -name '*_results.db' | head -1)
python profiles/summarize.py trace $T > $OUT/kernel_stats_$CFG.txt 2>&1
rm -rf $OUT/trace
head -12 $OUT/kernel_stats_$CFG.txt

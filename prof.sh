set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01_trace -o trace -- python $R/bench.py --instances 10000 --steps 5 --warmup 1 --no-cpu > $R/gpurun_out/prof_r01_bench.json 2> $R/gpurun_out/prof_r01_trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_r01_fetch -o fetch -- python $R/bench.py --instances 10000 --steps 2 --warmup 1 --no-cpu > /dev/null 2> $R/gpurun_out/prof_r01_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_r01_write -o write -- python $R/bench.py --instances 10000 --steps 2 --warmup 1 --no-cpu > /dev/null 2> $R/gpurun_out/prof_r01_write.log
cd $R/gpurun_out && find . -name "*.csv" | head -30; du -sh .
# keep only the small summaries
find . -name "*kernel_trace.csv" -size +20M -delete

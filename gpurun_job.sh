cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
( time timeout 300 python bench.py > gpurun_out/r01_b_bench.json 2> gpurun_out/r01_b_bench.err ) 2>&1 | grep real
cat gpurun_out/r01_b_bench.json | cut -c1-300
export TMPDIR=/tmp
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b_trace -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu > $R/gpurun_out/prof_b_trace.log 2>&1; echo trace rc=$?
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_b_fetch -o fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/prof_b_fetch.log 2>&1; echo fetch rc=$?
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_b_write -o write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/prof_b_write.log 2>&1; echo write rc=$?
ls $R/gpurun_out/prof_b_trace $R/gpurun_out/prof_b_fetch $R/gpurun_out/prof_b_write

#!/bin/bash
# One gpurun call that produces everything a round's profiles/ entry needs, from the SAME box and the SAME build:
#   bench line, rocprofv3 kernel trace (+stats), and the two separate --pmc passes (FETCH_SIZE, WRITE_SIZE).
# usage (on the GPU box, from the repo root): bash profiles/run_round.sh r02_a [--skip-tests]
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$2" != "--skip-tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
  echo "pytest rc=$?" >> $OUT/pytest.log
  tail -3 $OUT/pytest.log
fi
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
BENCH="python bench.py --no-cpu --no-configs --steps 16 --warmup 2"  # headline workload only: per-kernel averages must not mix batch sizes
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $BENCH > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- $BENCH > $OUT/write.log 2>&1
T=$(find $OUT/trace -name '*_results.db' | head -1); F=$(find $OUT/fetch -name '*_results.db' | head -1); W=$(find $OUT/write -name '*_results.db' | head -1)
python profiles/summarize.py trace $T > $OUT/kernel_stats.txt 2>&1
python profiles/summarize.py pmc $F FETCH_SIZE > $OUT/pmc_fetch.txt 2>&1
python profiles/summarize.py pmc $W WRITE_SIZE > $OUT/pmc_write.txt 2>&1
python profiles/summarize.py traffic $F $W "$TAG" > $OUT/traffic.json 2>&1
rm -rf $OUT/trace $OUT/fetch $OUT/write
head -12 $OUT/kernel_stats.txt

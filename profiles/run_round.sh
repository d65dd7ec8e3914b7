#!/bin/bash
# One gpurun call that produces everything a round's profiles/ entry needs, from the SAME box and the SAME build:
#   bench line, and per config a rocprofv3 kernel trace (+stats) and the two separate --pmc passes (FETCH_SIZE, WRITE_SIZE).
# usage (on the GPU box, from the repo root): bash profiles/run_round.sh r03_a [--tests] [config ...]   (default: tiger10k)
TAG=${1:-rXX}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$1" == "--tests" ]; then
  shift
  timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 > $OUT/pytest.log 2>&1
  echo "pytest rc=$?" >> $OUT/pytest.log
  tail -3 $OUT/pytest.log
fi
CONFIGS=${@:-tiger10k}
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cp bench_details.json $OUT/bench_details.json 2>/dev/null
tail -c 400 $OUT/bench.json
PARTS=""
for CFG in $CONFIGS; do
  SFX=""; [ "$CFG" != "tiger10k" ] && SFX="_$CFG"
  # one workload per profile: per-kernel averages must not mix batch sizes
  BENCH="python bench.py --config $CFG --no-cpu --no-configs --steps 8 --warmup 2 --placements 1"
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace$SFX.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $BENCH > $OUT/fetch$SFX.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- $BENCH > $OUT/write$SFX.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $OUT/valu -o valu -- $BENCH > $OUT/valu$SFX.log 2>&1
  grep '^{' $OUT/trace$SFX.log > $OUT/bench_profiled$SFX.json  # the bench line of the traced run itself (HIP-event times to compare the trace with)
  T=$(find $OUT/trace -name '*_results.db' | head -1); F=$(find $OUT/fetch -name '*_results.db' | head -1); W=$(find $OUT/write -name '*_results.db' | head -1); V=$(find $OUT/valu -name '*_results.db' | head -1)
  python profiles/summarize.py trace $T > $OUT/kernel_stats$SFX.txt 2>&1
  python profiles/summarize.py pmc $F FETCH_SIZE > $OUT/pmc_fetch$SFX.txt 2>&1
  python profiles/summarize.py pmc $W WRITE_SIZE > $OUT/pmc_write$SFX.txt 2>&1
  python profiles/summarize.py pmc $V SQ_INSTS_VALU > $OUT/pmc_valu$SFX.txt 2>&1
  python profiles/summarize.py pmc $V SQ_ACTIVE_INST_VALU >> $OUT/pmc_valu$SFX.txt 2>&1
  python profiles/summarize.py traffic $F $W "$TAG" $V > $OUT/traffic$SFX.json 2>&1
  PARTS="$PARTS $CFG=$OUT/traffic$SFX.json"
  rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/valu
  head -8 $OUT/kernel_stats$SFX.txt
done
python profiles/summarize.py merge "$TAG" $PARTS > $OUT/traffic_all.json

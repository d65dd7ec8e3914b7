cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 250 python bench.py --steps 10 --warmup 3 --no-cpu 2>&1 | tail -1 > gpurun_out/bench_d.json; cat gpurun_out/bench_d.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES -d $R/gpurun_out/pmc_e1 -o e1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $R/gpurun_out/pmc_e1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_WR -d $R/gpurun_out/pmc_e2 -o e2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $R/gpurun_out/pmc_e2.log 2>&1
ls $R/gpurun_out/pmc_e1 $R/gpurun_out/pmc_e2

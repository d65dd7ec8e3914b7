R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace -d $R/gpurun_out/pmc_sq1 -o sq1 -- python $R/bench.py --instances 10000 --steps 2 --warmup 1 --no-cpu > /dev/null 2> $R/gpurun_out/pmc_sq1.log
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD --kernel-trace -d $R/gpurun_out/pmc_sq2 -o sq2 -- python $R/bench.py --instances 10000 --steps 2 --warmup 1 --no-cpu > /dev/null 2> $R/gpurun_out/pmc_sq2.log
ls -la $R/gpurun_out/pmc_sq1 $R/gpurun_out/pmc_sq2; tail -3 $R/gpurun_out/pmc_sq2.log

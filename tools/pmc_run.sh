export TMPDIR=/tmp
cd /root/repo
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d gpurun_out/pmc2 -o p -- python tools/bench_kernels.py gemm > gpurun_out/pmc2.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA --kernel-trace -d gpurun_out/pmc3 -o p -- python tools/bench_kernels.py gemm > gpurun_out/pmc3.log 2>&1
ls gpurun_out/pmc2 gpurun_out/pmc3; tail -3 gpurun_out/pmc2.log

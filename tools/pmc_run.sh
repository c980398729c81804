export TMPDIR=/tmp
cd /root/repo
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace -d gpurun_out/pmc1 -o p -- python tools/bench_kernels.py gemm > gpurun_out/pmc1.log 2>&1
ls gpurun_out/pmc1

#!/bin/bash
# HBM traffic per kernel from the memory-side L2 counters: two separate PMC passes (FETCH_SIZE and WRITE_SIZE do
# not fit one pass; no other trace domains - see MI355X_MICROARCH.md "rocprofv3 PMC slots").  Run through gpurun;
# raw traces under /tmp (the 64 MiB copy-back limit), the summary gpurun_out/pmc_${R}_traffic.{txt,json}.
export TMPDIR=/tmp
R=${1:-r01}
cd /root/repo
CMD="python bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-train-mode --no-decode ${ST_BENCH_ARGS:-}"      # ST_BENCH_ARGS="--config 3" profiles config 3
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_${R}_fetch -o p -- $CMD > gpurun_out/pmc_${R}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_${R}_write -o p -- $CMD > gpurun_out/pmc_${R}_write.log 2>&1
python tools/summarize_pmc.py /tmp/pmc_${R}_fetch/p_results.db /tmp/pmc_${R}_write/p_results.db gpurun_out/pmc_${R}_traffic
head -20 gpurun_out/pmc_${R}_traffic.txt

#!/bin/bash
# Produce the per-round profile artefacts on the GPU box (run through gpurun); outputs under gpurun_out/.
export TMPDIR=/tmp
R=${1:-r01}
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${R}.json 2> gpurun_out/bench_${R}.err
tail -c 400 gpurun_out/bench_${R}.json
rocprofv3 --kernel-trace --stats -d /tmp/prof_${R} -o trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-train-mode --no-decode > gpurun_out/prof_${R}.log 2>&1
python tools/summarize_rocprof.py /tmp/prof_${R}/trace_results.db > gpurun_out/rocprof_${R}_kernel_stats.txt
head -30 gpurun_out/rocprof_${R}_kernel_stats.txt

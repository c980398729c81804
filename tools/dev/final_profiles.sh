# Dev: the round's committed artefacts in one GPU call (profiles/r05_*): PMC traffic first (bench.py credits the committed summary,
# stamped with the SHA below), then the bench lines, the kernel trace, the step sequence, the issue mix; config 3 likewise.
export TMPDIR=/tmp GIT_SHA=7081455
cd /root/repo
bash tools/pmc_traffic.sh r05 > gpurun_out/pmc_r05.log 2>&1
cp gpurun_out/pmc_r05_traffic.json profiles/r05_pmc_traffic.json; cp gpurun_out/pmc_r05_traffic.txt profiles/r05_pmc_traffic.txt
ST_BENCH_ARGS="--config 3" bash tools/pmc_traffic.sh r05_c3 > gpurun_out/pmc_r05_c3.log 2>&1
cp gpurun_out/pmc_r05_c3_traffic.json profiles/r05_pmc_traffic_c3.json; cp gpurun_out/pmc_r05_c3_traffic.txt profiles/r05_pmc_traffic_c3.txt
bash tools/profile_round.sh r05 > gpurun_out/profile_round_r05.log 2>&1
rocprofv3 --kernel-trace -d /tmp/seg -o t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-mode --no-decode --no-dp-probe > /tmp/g.log 2>&1; python tools/dev/step_segment.py /tmp/seg/t_results.db 8 > gpurun_out/r05_step_sequence.txt 2>&1
ST_PMC_OUT=r05_pmc_issue_mix bash tools/pmc_issue_mix.sh > gpurun_out/pmc_issue_r05.log 2>&1
python bench.py --config 3 --steps 10 --warmup 3 --no-decode > gpurun_out/bench_r05_c3.json 2> gpurun_out/bench_r05_c3.err
rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o trace -- python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-train-mode --no-decode --no-dp-probe > gpurun_out/prof_c3.log 2>&1; python tools/summarize_rocprof.py /tmp/prof_c3/trace_results.db > gpurun_out/rocprof_r05_c3_kernel_stats.txt
ls -la gpurun_out | grep r05

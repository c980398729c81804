# Dev: round-6 baseline artefacts in one GPU call: GPU tests, bench line (with --pmc measured traffic), kernel trace, step sequence, config 3.
export TMPDIR=/tmp GIT_SHA=29a3209
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests.log 2>&1; echo "gpu tests rc $?" >> gpurun_out/r06_gputests.log
tail -3 gpurun_out/r06_gputests.log
python bench.py --steps 20 --warmup 5 --pmc > gpurun_out/bench_r06.json 2> gpurun_out/bench_r06.err
tail -c 600 gpurun_out/bench_r06.json
rocprofv3 --kernel-trace --stats -d /tmp/prof_r06 -o trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-train-mode --no-decode --no-dp-probe > gpurun_out/prof_r06.log 2>&1
python tools/summarize_rocprof.py /tmp/prof_r06/trace_results.db > gpurun_out/rocprof_r06_kernel_stats.txt
head -30 gpurun_out/rocprof_r06_kernel_stats.txt
rocprofv3 --kernel-trace -d /tmp/seg -o t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-mode --no-decode --no-dp-probe > /tmp/g.log 2>&1; python tools/dev/step_segment.py /tmp/seg/t_results.db 8 > gpurun_out/r06_step_sequence.txt 2>&1
python bench.py --config 3 --steps 10 --warmup 3 --no-decode > gpurun_out/bench_r06_c3.json 2> gpurun_out/bench_r06_c3.err
rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o trace -- python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-train-mode --no-decode --no-dp-probe > gpurun_out/prof_c3.log 2>&1; python tools/summarize_rocprof.py /tmp/prof_c3/trace_results.db > gpurun_out/rocprof_r06_c3_kernel_stats.txt

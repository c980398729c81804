cd /root/repo; export TMPDIR=/tmp
python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err
tail -c 600 gpurun_out/bench_r03.json
python bench.py --config 3 > gpurun_out/bench_r03_c3.json 2> gpurun_out/bench_r03_c3.err
rocprofv3 --kernel-trace --stats -d /tmp/prof_r03 -o trace -- python bench.py --steps 25 --warmup 3 --no-cpu-baseline --no-train-mode --no-decode > gpurun_out/prof_r03.log 2>&1
python tools/summarize_rocprof.py /tmp/prof_r03/trace_results.db > gpurun_out/rocprof_r03_kernel_stats.txt
python tools/bench_kernels.py > gpurun_out/bench_kernels_r03.txt 2>&1

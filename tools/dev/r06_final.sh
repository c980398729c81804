# Dev: the round's committed artefacts in one GPU call (profiles/r06_*), stamped with the product tree they were taken at.
export TMPDIR=/tmp GIT_SHA=5c20b16
cd /root/repo; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests.log 2>&1; grep -E "passed|failed" gpurun_out/r06_gputests.log | tail -1
python bench.py --steps 20 --warmup 5 --pmc > gpurun_out/bench_r06.json 2> gpurun_out/bench_r06.err
tail -c 300 gpurun_out/bench_r06.json
rocprofv3 --kernel-trace --stats -d /tmp/prof_r06 -o trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-train-mode --no-decode --no-dp-probe > gpurun_out/prof_r06.log 2>&1
python tools/summarize_rocprof.py /tmp/prof_r06/trace_results.db > gpurun_out/rocprof_r06_kernel_stats.txt
rocprofv3 --kernel-trace -d /tmp/seg -o t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-mode --no-decode --no-dp-probe > /tmp/g.log 2>&1; python tools/dev/step_segment.py /tmp/seg/t_results.db 8 > gpurun_out/r06_step_sequence.txt 2>&1
ST_PMC_OUT=r06_pmc_issue_mix bash tools/pmc_issue_mix.sh > gpurun_out/pmc_issue_r06.log 2>&1
ST_HIP_LIB=tools/dev/_ab/libst_trace.so python tools/dev/chain_bwd_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_chain_bwd_phases.txt
ST_HIP_LIB=tools/dev/_ab/libst_trace.so python tools/dev/chain_pipe_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_chain_fwd_phases.txt
python tools/dev/shard_times.py 2>&1 | grep "N = " > gpurun_out/r06_shard_times.txt
python bench.py --config 3 --steps 10 --warmup 3 --no-decode > gpurun_out/bench_r06_c3.json 2> gpurun_out/bench_r06_c3.err
rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o trace -- python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-train-mode --no-decode --no-dp-probe > gpurun_out/prof_c3.log 2>&1; python tools/summarize_rocprof.py /tmp/prof_c3/trace_results.db > gpurun_out/rocprof_r06_c3_kernel_stats.txt
ls gpurun_out | grep r06

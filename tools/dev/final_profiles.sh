export TMPDIR=/tmp GIT_SHA=d3ab82e
cd /root/repo
bash tools/profile_round.sh r05 > gpurun_out/profile_round_r05.log 2>&1
rocprofv3 --kernel-trace -d /tmp/seg -o t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-mode --no-decode --no-dp-probe > /tmp/g.log 2>&1; python tools/dev/step_segment.py /tmp/seg/t_results.db 8 > gpurun_out/r05_step_sequence.txt 2>&1
bash tools/pmc_traffic.sh r05 > gpurun_out/pmc_r05.log 2>&1
ST_PMC_OUT=r05_pmc_issue_mix bash tools/pmc_issue_mix.sh > gpurun_out/pmc_issue_r05.log 2>&1
python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-decode > gpurun_out/bench_r05_c3.json 2> gpurun_out/bench_r05_c3.err
ST_BENCH_ARGS="--config 3" bash tools/pmc_traffic.sh r05_c3 > gpurun_out/pmc_r05_c3.log 2>&1
ls -la gpurun_out | grep r05

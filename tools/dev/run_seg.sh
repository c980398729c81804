cd /root/repo; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_seg -o trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-mode --no-decode > gpurun_out/seg.log 2>&1
python tools/dev/step_segment.py /tmp/prof_seg/trace_results.db 8 gpurun_out/seg_sequence.txt > gpurun_out/seg_census.txt 2>&1

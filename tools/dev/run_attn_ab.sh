cd /root/repo
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or dropout" 2>&1 | tail -3
python tools/bench_kernels.py attn 2>&1 | grep "attn"

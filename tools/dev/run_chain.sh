cd /root/repo
python -m pytest tests/test_kernels_gpu.py -x -q -k "chain" 2>&1 | tail -3
for i in 1 2; do python tools/bench_kernels.py chain 2>&1 | grep "row_chain"; done

cd /root/repo
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or dropout" 2>&1 | tail -3
for impl in 0 50 51; do echo "== ST_ATTN_IMPL=$impl"; ST_ATTN_IMPL=$impl python tools/bench_kernels.py attn 2>&1 | grep "attn"; done

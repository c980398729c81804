cd /root/repo
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or dropout" 2>&1 | tail -3
for v in 0 61 0 61; do echo "== ST_ATTN_IMPL=$v"; ST_ATTN_IMPL=$v python tools/bench_kernels.py attn 2>&1 | grep "fwd  enc self"; done

cd /root/repo
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -6
for s in 1 2 4; do echo "== ST_ATTN_KEY_SPLITS=$s"; ST_ATTN_KEY_SPLITS=$s python tools/bench_kernels.py attn 2>&1 | grep "cross"; done

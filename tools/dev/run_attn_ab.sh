cd /root/repo
for impl in 40; do echo "== tests ST_ATTN_IMPL=$impl"; ST_ATTN_IMPL=$impl python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -3; done
for impl in 1 40 1 40; do echo "== ST_ATTN_IMPL=$impl"; ST_ATTN_IMPL=$impl python tools/bench_kernels.py attn 2>&1 | grep "fwd  enc self"; done
bash tools/dev/pmc_attn.sh 40 30 2>&1 | grep -v raw

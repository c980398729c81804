cd /root/repo
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -2
for v in "" 1 "" 1; do echo "== ST_NO_XCD_AFFINITY=$v"; ST_NO_XCD_AFFINITY=$v python tools/bench_kernels.py attn 2>&1 | grep "fwd \|bwd all"; done

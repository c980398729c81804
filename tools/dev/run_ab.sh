cd /root/repo
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or dropout" 2>&1 | tail -6
for i in 1 2; do python tools/bench_kernels.py attn 2>&1 | grep "attn"; done
python bench.py --steps 30 --warmup 5 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline'])"

cd /root/repo
python -m pytest tests/test_kernels_gpu.py -x -q -k "grad_norm or cross_entropy or adam" 2>&1 | tail -2
python -m pytest tests/test_modules_gpu.py -x -q 2>&1 | grep -E "passed|failed"
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-mode --no-decode 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['loss'], d['grad_norm'])"; done

cd /root/repo
python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_ln or lnbwd or gemm_lnbwd" 2>&1 | tail -2
for rep in 1 2; do
echo "== 64-row tiles"; ST_GEMM_LN_MB1=1 python tools/bench_kernels.py gemm 2>&1 | grep "gemm_ln.* d512"
echo "== 128-row tiles"; python tools/bench_kernels.py gemm 2>&1 | grep "gemm_ln.* d512"
done
for v in 1 0; do echo "== ST_GEMM_LN_MB1=$v (0 = unset)"; if [ $v = 1 ]; then export ST_GEMM_LN_MB1=1; else unset ST_GEMM_LN_MB1; fi
python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-train-mode --no-decode 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k: round(v['ms_per_step'],3) for k,v in d['kernels'].items() if k.startswith('gemm_ln')})"; done

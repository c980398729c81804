cd /root/repo
python -m pytest tests/test_kernels_gpu.py -x -q -k "grad_norm" 2>&1 | tail -3
python tools/bench_kernels.py misc 2>&1 | grep "grad_norm"
python tools/bench_kernels.py misc 2>&1 | grep "grad_norm"

cd /root/repo
L=/root/repo/speech-tranformer-pytorch_amd/lib
for rep in 1 2 3; do for v in _base ""; do echo "== lib$v"; ST_HIP_LIB=$L/libst_hip$v.so python tools/bench_kernels.py chain 2>&1 | grep "row_chain fwd"; done; done

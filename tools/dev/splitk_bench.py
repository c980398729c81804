import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/speech-tranformer-pytorch_amd")
import torch
from st_amd import native as nv
M, N, K = 1206, 256, 4344
x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(K, N, device="cuda") * K ** -0.5).bfloat16()
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
def t(fn):
    for _ in range(5): fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 200
print("st_gemm         %.1f us" % t(lambda: nv.gemm(x, w, out, y_cmajor=True)))
for s in (2, 3, 4, 6, 8, 12):
    print("splitk %2d       %.1f us" % (s, t(lambda: nv.gemm_splitk(x, w, out, s, y_cmajor=True))))

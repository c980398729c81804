import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/speech-tranformer-pytorch_amd')
import torch
from st_amd import native as nv
from tools.tile_rounds_test import rnd, timed
F32, BF16 = torch.float32, torch.bfloat16
N = 256
for M, K in ((1206, 1024), (1206, 768), (1206, 504), (1206, 256), (8192, 1024), (16384, 1024), (24060, 1024), (24060, 256), (32768, 1024)):
    X, W, res = rnd(M, K), rnd(N, K), rnd(M, N)
    b, ga, be = rnd(N, dtype=F32), rnd(N, dtype=F32), rnd(N, dtype=F32)
    out, xh = torch.empty(M, N, dtype=BF16, device='cuda'), torch.empty(M, N, dtype=BF16, device='cuda')
    rstd = torch.empty(M, dtype=F32, device='cuda')
    t1 = timed(lambda: nv.gemm_ln(X, W, b, res, ga, be, out, xh, rstd))
    Wd = rnd(K, N); acc = [torch.zeros(N, dtype=F32, device='cuda') for _ in range(3)]
    t2 = timed(lambda: nv.gemm_lnbwd(X, Wd, res, xh, rstd, ga, out, acc[0], acc[1], acc[2]))
    print("M=%d K=%d: gemm_ln %.1f us  gemm_lnbwd %.1f us" % (M, K, t1, t2))

#!/usr/bin/env python3
"""Dev micro-benchmark: st_gemm_lnbwd against the two launches it replaces (dgrad + aux, then st_ln_bwd)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import torch  # noqa: E402
from st_amd import native as nv  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32


def t(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for M, K in ((24060, 1024), (24060, 768), (24060, 256), (1206, 1024), (1206, 256)):
    d = 256
    dY = (torch.randn(M, K, device="cuda") * 0.5).to(BF16)
    W = (torch.randn(K, d, device="cuda") * K ** -0.5).to(BF16)
    aux, xhat = (torch.randn(M, d, device="cuda").to(BF16) for _ in range(2))
    rstd, gamma = torch.rand(M, device="cuda") + 0.5, torch.randn(d, device="cuda")
    tmp, dx = torch.empty(M, d, dtype=BF16, device="cuda"), torch.empty(M, d, dtype=BF16, device="cuda")
    a, b, c = (torch.zeros(d, device="cuda") for _ in range(3))
    t_g = t(lambda: nv.gemm(dY, W, tmp, epi=nv.EPI_BF16_ADD, aux=aux, y_cmajor=True))
    t_l = t(lambda: nv.ln_bwd(tmp, xhat, rstd, gamma, dx, a, b, c))
    t_f = t(lambda: nv.gemm_lnbwd(dY, W, aux, xhat, rstd, gamma, dx, a, b, c))
    print("M=%5d K=%4d: dgrad %.1f + ln_bwd %.1f = %.1f us   fused %.1f us" % (M, K, t_g, t_l, t_g + t_l, t_f))

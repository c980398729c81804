#!/usr/bin/env python3
"""Dev experiment: is the GEMM bound by where its operands come from?  Same launch with the X rows aliased onto one
row (stride 0: every load hits L1/L2), and with the output rows aliased too."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import torch  # noqa: E402
from st_amd import native as nv  # noqa: E402

M, K, N = 24060, 256, 1024
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.06).bfloat16()
b = torch.randn(N, device="cuda")
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")


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


xa = x[:1].expand(M, K)
print("normal              %.2f us" % t(lambda: nv.gemm(x, w, out, bias=b, epi=nv.EPI_BF16_RELU)))
print("X rows aliased      %.2f us" % t(lambda: nv.gemm(xa, w, out, bias=b, epi=nv.EPI_BF16_RELU)))
for kk in (512, 1024, 2048):
    xk = torch.randn(M, kk, device="cuda").bfloat16()
    wk = (torch.randn(256, kk, device="cuda") * 0.03).bfloat16()
    ok = torch.empty(M, 256, dtype=torch.bfloat16, device="cuda")
    us = t(lambda: nv.gemm(xk, wk, ok))
    print("K=%4d N=256: %.2f us  %.0f TFLOP/s  loads %.1f TB/s (CU-side)" % (kk, us, 2.0 * M * 256 * kk / us / 1e6,
          (188 * 2) * (kk / 64) * 32768 / us / 1e6))

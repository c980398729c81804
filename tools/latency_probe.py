#!/usr/bin/env python3
"""Dev tool: is the st_gemm_ln k-loop bound by memory latency?  Same launch with X read normally, with every X row
aliased onto one row (stride 0: L1 hits), and with tiny K-invariant W (all k-tiles alias one 32-column slab)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from st_amd import native as nv  # noqa: E402
from tools.tile_rounds_test import rnd, timed  # noqa: E402

dev, BF16, F32 = "cuda", torch.bfloat16, torch.float32

N = 256
for K in (1024, 256):
    for M in (16384, 24060):
        X, W, res = rnd(M, K), rnd(N, K), rnd(M, N)
        b, ga, be = rnd(N, dtype=F32), rnd(N, dtype=F32), rnd(N, dtype=F32)
        out, xh = torch.empty(M, N, dtype=BF16, device=dev), torch.empty(M, N, dtype=BF16, device=dev)
        rstd = torch.empty(M, dtype=F32, device=dev)
        X1 = X[:1].expand(M, K)                       # every row = row 0
        t0 = timed(lambda: nv.gemm_ln(X, W, b, res, ga, be, out, xh, rstd))
        t1 = timed(lambda: nv.gemm_ln(X1, W, b, res, ga, be, out, xh, rstd))
        t2 = timed(lambda: nv.gemm_ln(X1, W, b, None, ga, be, out, None, None))
        print("K=%4d M=%5d: normal %5.1f us | X aliased to one row %5.1f us | + no residual / xhat / rstd %5.1f us" % (K, M, t0, t1, t2))

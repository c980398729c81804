#!/usr/bin/env python3
"""Dev tool: how st_gemm_ln / st_gemm_lnbwd / st_gemm time depends on the number of 64-row tiles per CU
(256 tiles = one workgroup per CU, 512 = two co-resident ones).  Graph-replayed chains, us per launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from st_amd import native as nv  # noqa: E402

dev, BF16, F32 = "cuda", torch.bfloat16, torch.float32


def rnd(*s, dtype=BF16):
    return (torch.randn(*s, device=dev) * 0.5).to(dtype)


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (5 * n)


def main():
  for K in (1024, 256):
    for M in (8192, 16384, 20480, 24060, 24576, 32768, 49152):
          N = 256
          X, W, res = rnd(M, K), rnd(N, K), rnd(M, N)
          b, ga, be = rnd(N, dtype=F32), rnd(N, dtype=F32), rnd(N, dtype=F32)
          out, xh = torch.empty(M, N, dtype=BF16, device=dev), torch.empty(M, N, dtype=BF16, device=dev)
          rstd = torch.empty(M, dtype=F32, device=dev)
          t1 = timed(lambda: nv.gemm_ln(X, W, b, res, ga, be, out, xh, rstd))
          Wd = rnd(K, N)
          acc = [torch.zeros(N, dtype=F32, device=dev) for _ in range(3)]
          t2 = timed(lambda: nv.gemm_lnbwd(X, Wd, res, xh, rstd, ga, out, acc[0], acc[1], acc[2]))
          print("K=%4d M=%5d (%.2f tiles/CU): gemm_ln %6.1f us  gemm_lnbwd %6.1f us   per 64 rows/CU: %5.1f / %5.1f"
                % (K, M, M / 64 / 256, t1, t2, t1 / (M / 64 / 256), t2 / (M / 64 / 256)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Dev: timings of the hand-scheduled attention backward (csrc/st_attn_bwd64.hip) that separate the loop from the per-item
costs: the config-2 encoder batch, a uniform batch of LONG utterances (32 x 1024: 16 tiles per item) and one of SHORT ones
(128 x 256: 4 tiles per item) with the same number of work items - per body and merged.  ST_ATTN_BWD64=0 rows = general kernels."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from st_amd import native as nv  # noqa: E402
from st_amd import synthetic  # noqa: E402
from st_amd.functional import Rows, attn_work  # noqa: E402

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32
dev = "cuda"
H, dk = 4, 64
d = H * dk
scale = 1 / math.sqrt(dk)
N_IT = int(os.environ.get("N_IT", "30"))


def bench(lens, label, modes=("1",)):
    lens_t = torch.tensor(lens)
    M = int(lens_t.sum())
    g = (torch.randn(M, 3 * d, device=dev) * 0.7).to(BF16)
    dO = (torch.randn(M, d, device=dev) * 0.5).to(BF16)
    Q, K, V = g[:, :d], g[:, d:2 * d], g[:, 2 * d:]
    rows = Rows.packed(lens_t, dev)
    wf, wq, wk = attn_work(rows, rows, False, dk, H)
    off = torch.zeros_like(lens_t)
    off[1:] = torch.cumsum(lens_t, 0)[:-1]
    q_off, q_len = off.to(dev, I32), lens_t.to(dev, I32)
    O = torch.empty(M, d, dtype=BF16, device=dev)
    lse = torch.empty(H * M, dtype=F32, device=dev)
    nv.attn_fwd(Q, K, V, O, lse, q_off, q_len, q_off, q_len, H, max(lens), False, scale, work=wf, max_k=max(lens))
    delta = (dO.float() * O.float()).view(M, H, dk).sum(-1).t().contiguous().view(-1)
    dQ, dK, dV = (torch.empty(M, d, dtype=BF16, device=dev) for _ in range(3))
    pairs = float((lens_t.double() ** 2).sum()) * H
    blocks = pairs / 1024
    out = []
    for mode in modes:
        os.environ["ST_ATTN_BWD64"] = mode
        nv.env_refresh()
        for parts, nm, mf in ((3, "all", 28), (1, "dq", 12), (2, "dkv", 16)):
            f = lambda: nv.attn_bwd(Q, K, V, None, dO, lse, delta, dQ, dK, dV, q_off, q_len, q_off, q_len, H, max(lens), max(lens), False,
                                    scale, parts=parts, work_q=wq, work_k=wk)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(N_IT):
                f()
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) / N_IT * 1e3
            # matrix-pipe time of the executed MFMAs at 2.4 GHz over 1024 SIMDs
            ideal = blocks * mf * 32 / 1024 / 2400
            out.append("%s%s %6.1f us (mfma %4.1f us = %2.0f%%)" % (nm, "" if mode == "1" else "[gen]", us, ideal, 100 * ideal / us))
    print("%-22s %s" % (label, " | ".join(out)))


if __name__ == "__main__":
    _, _, in_len, _, _ = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
    bench(in_len.tolist(), "config 2 encoder", modes=("0", "1") if os.environ.get("WITH_OLD") else ("1",))
    bench([1024] * 32, "uniform 32 x 1024")
    bench([256] * 128, "uniform 128 x 256")
    bench([2048] * 16, "uniform 16 x 2048")

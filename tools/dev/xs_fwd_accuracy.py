#!/usr/bin/env python3
"""Dev: accuracy of the few-queries attention forward (csrc/st_attn_xs.hip) and of the general kernel (ST_ATTN_XS=0) against an
fp64 reference at the decoder-encoder shape of config 2: the context O, the pair O + Ores the backward's delta uses, the LSE."""
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

BF16, F32 = torch.bfloat16, torch.float32
dev, H, dk = "cuda", 4, 64
d = H * dk
scale = 1 / math.sqrt(dk)
_, _, in_len, tgt_len, _ = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
torch.manual_seed(1)
Mq, Mk = int(tgt_len.sum()), int(in_len.sum())
for qs, label in ((0.7, "random q, k (|s| ~ 3)"), (0.15, "near-uniform attention (|s| ~ 0.2)")):
    Q = (torch.randn(Mq, d, device=dev) * qs).to(BF16)
    kv = (torch.randn(Mk, 2 * d, device=dev) * 0.7).to(BF16)
    K, V = kv[:, :d], kv[:, d:]
    qr, kr = Rows.packed(tgt_len, dev), Rows.packed(in_len, dev)
    wf = attn_work(qr, kr, False, dk, H)[0]
    # fp64 reference
    O64 = torch.zeros(Mq, d, dtype=torch.float64, device=dev)
    L64 = torch.zeros(H, Mq, dtype=torch.float64, device=dev)
    oq = ok = 0
    for a, b in zip(tgt_len.tolist(), in_len.tolist()):
        for h in range(H):
            sl = slice(h * dk, (h + 1) * dk)
            s = Q[oq:oq + a, sl].double() @ K[ok:ok + b, sl].double().T * scale
            O64[oq:oq + a, sl] = torch.softmax(s, -1) @ V[ok:ok + b, sl].double()
            L64[h, oq:oq + a] = torch.logsumexp(s, -1) * 1.4426950408889634
        oq += a
        ok += b
    print(label)
    for mode in ("1", "0"):
        os.environ["ST_ATTN_XS"] = mode
        nv.env_refresh()
        O = torch.empty(Mq, d, dtype=BF16, device=dev)
        Ores = torch.empty(Mq, d, dtype=BF16, device=dev)
        lse = torch.empty(H * Mq, dtype=F32, device=dev)
        nv.attn_fwd(Q, K, V, O, lse, qr.off, qr.len, kr.off, kr.len, H, int(tgt_len.max()), False, scale, work=wf if mode == "1" else None,
                    max_k=int(in_len.max()), ores=Ores)
        torch.cuda.synchronize()
        rel = lambda x, y: float((x.double() - y).norm() / y.norm())
        print("   ST_ATTN_XS=%s: O rel %.3e   O + Ores rel %.3e   lse abs max %.3e rms %.3e" %
              (mode, rel(O, O64), float(((O.double() + Ores.double()) - O64).norm() / O64.norm()),
               float((lse.view(H, Mq).double() - L64).abs().max()), float((lse.view(H, Mq).double() - L64).pow(2).mean().sqrt())))

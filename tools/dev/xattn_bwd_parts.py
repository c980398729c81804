#!/usr/bin/env python3
"""Dev: the decoder-encoder attention backward at config 2's shape (32 utterances, 25-50 queries against 500-1000 keys, 4 heads of 64):
the merged launch, its dQ items alone (parts = 1) and its dK/dV items alone (parts = 2) - which body is the launch's critical path."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch
from st_amd import native as nv, synthetic
from st_amd.functional import Rows, attn_work
BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32
dev, H, dk = "cuda", 4, 64
d = H * dk
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
_, _, in_len, tgt_len, _ = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
in_len, tgt_len = in_len[:nb], tgt_len[:nb]
Mq, Mk = int(tgt_len.sum()), int(in_len.sum())
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF16)
Q, K, V, dO = rnd(Mq, d), rnd(Mk, d), rnd(Mk, d), rnd(Mq, d)
qr, kr = Rows.packed(tgt_len, dev), Rows.packed(in_len, dev)
wf, wq, wk = attn_work(qr, kr, False, dk, H)
def offs(lens):
    o = torch.zeros_like(lens); o[1:] = torch.cumsum(lens, 0)[:-1]
    return o.to(dev, I32), lens.to(dev, I32)
qo, ql = offs(tgt_len); ko, kl = offs(in_len)
scale = 1 / math.sqrt(dk)
O, lse = torch.empty(Mq, d, dtype=BF16, device=dev), torch.empty(H * Mq, dtype=F32, device=dev)
nv.attn_fwd(Q, K, V, O, lse, qo, ql, ko, kl, H, int(tgt_len.max()), False, scale, work=wf, max_k=int(in_len.max()))
delta = (dO.float() * O.float()).view(Mq, H, dk).sum(-1).t().contiguous().view(-1)
dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
def run(parts):
    nv.attn_bwd(Q, K, V, None, dO, lse, delta, dQ, dK, dV, qo, ql, ko, kl, H, int(tgt_len.max()), int(in_len.max()), False, scale,
                parts=parts, work_q=wq, work_k=wk)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    s.record()
    for _ in range(n // 10): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (n // 10 * 10) * 1e3
print("%d utterances: %d query rows, %d key rows" % (nb, Mq, Mk))
for parts, name in ((3, "merged"), (1, "dQ items only"), (2, "dK/dV items only")):
    print("  %-18s %6.2f us per launch (graph replay, back to back)" % (name, timeit(lambda: run(parts))))

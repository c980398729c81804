#!/usr/bin/env python3
"""Dev: the decoder's grouped weight-gradient launch (st_wgrad_group, 128 x 128 tiles) at config 2's target side (1,206 rows; a
4-utterance shard: 161) by token splits - graph replay, back to back."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch
from st_amd import native as nv
dev, BF16, F32 = "cuda", torch.bfloat16, torch.float32
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF16)


def timeit(fn, n=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(5): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n // 5): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (n // 5 * 5) * 1e3


for M in (1206, 161):
    shapes = [(768, 256), (256, 256), (256, 256), (256, 256), (1024, 256), (256, 1024)] * 6 + [(4344, 256)]
    probs = [(rnd(M, k), rnd(M, n), torch.zeros(n, k, dtype=F32, device=dev), torch.zeros(n, dtype=F32, device=dev), 1, n) for (n, k) in shapes]
    out = []
    for sp in (1, 2, 3, 4, 6):
        if M // sp < 32: continue
        pl = [p[:4] + (sp, p[5]) for p in probs]
        out.append("%d: %.1f" % (sp, timeit(lambda: nv.wgrad_group(pl))))
    print("M = %4d rows, %d problems: grouped kernel by token splits (us) %s" % (M, len(probs), "  ".join(out)))

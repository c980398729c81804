#!/usr/bin/env python3
"""Dev: the encoder-sized weight gradients at a DP rank's shard (3,120 / 6,536 / 12,657 tokens: 4 / 8 / 16 utterances): st_wgrad_wide
at 1 .. 8 token splits, what functional._wide_plan picks, and the 128 x 128 grouped kernel - graph replay, back to back."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch
from st_amd import native as nv
from st_amd.functional import _wide_plan
dev, BF16, F32 = "cuda", torch.bfloat16, torch.float32
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF16)


def timeit(fn, n=30):
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


for M in (3120, 6536, 12657):
    probs = []
    for _ in range(6):
        for (n, k) in ((768, 256), (256, 256), (1024, 256), (256, 1024)):
            probs.append((rnd(M, k), rnd(M, n), torch.zeros(n, k, dtype=F32, device=dev), torch.zeros(n, dtype=F32, device=dev), 1, n))
    enc = rnd(M, 256)
    for _ in range(6):
        probs.append((enc, rnd(M, 512), torch.zeros(512, 256, dtype=F32, device=dev), torch.zeros(512, dtype=F32, device=dev), 1, 512))
    out = []
    for sp in (1, 2, 3, 4, 6, 8):
        pl = [p[:4] + (sp, p[5]) for p in probs]
        out.append("%d: %.1f" % (sp, timeit(lambda: nv.wgrad_group(pl, wide=True))))
    plan = _wide_plan(probs)
    t_plan = timeit(lambda: [nv.wgrad_group(l, wide=True) for l in plan])
    t_group = timeit(lambda: nv.wgrad_group([p[:4] + (max(1, M // 1024), p[5]) for p in probs]))
    print("M = %5d tokens, 30 problems (85 tiles): wide by splits (us) %s | _wide_plan picks %s: %.1f | grouped 128 x 128 kernel: %.1f"
          % (M, "  ".join(out), [l[0][4] for l in plan], t_plan, t_group))

# one encoder layer's four problems (12 tiles: a DP step flushes per finished layer)
for M in (3120, 24060):
    probs = [(rnd(M, k), rnd(M, n), torch.zeros(n, k, dtype=F32, device=dev), torch.zeros(n, dtype=F32, device=dev), 1, n)
             for (n, k) in ((768, 256), (256, 256), (1024, 256), (256, 1024))]
    out = []
    for sp in (1, 2, 4, 6, 8, 12, 16, 21):
        if M // sp < 256: continue
        pl = [p[:4] + (sp, p[5]) for p in probs]
        out.append("%d: %.1f" % (sp, timeit(lambda: nv.wgrad_group(pl, wide=True))))
    plan = _wide_plan(probs)
    print("M = %5d tokens, ONE layer (12 tiles): wide by splits (us) %s | _wide_plan picks %s" % (M, "  ".join(out), [l[0][4] for l in plan]))

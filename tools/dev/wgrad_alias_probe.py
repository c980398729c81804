#!/usr/bin/env python3
"""Dev: is st_wgrad_wide bound by HBM?  Config 2's 24 encoder problems (6 layers x qkv / wo / w1 / w2, 24,060 tokens) as the step
issues them (every operand its own tensor: 1.2 GB), then with all problems reading the SAME operand tensors (61 MB: Infinity-Cache
resident), then with 2,048-token operands repeated (L2 resident) - same launch geometry, same MFMA / LDS / atomic work."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch
from st_amd import native as nv
from st_amd.functional import _wide_plan
dev, BF16, F32 = "cuda", torch.bfloat16, torch.float32
M = 24060
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF16)
shapes = ((768, 256), (256, 256), (1024, 256), (256, 1024))


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def problems(mode):
    out = []
    x0, y0 = rnd(M, 1024), rnd(M, 1024)
    xs, ys = rnd(2048, 1024), rnd(2048, 1024)
    for _ in range(6):
        for (n, k) in shapes:
            if mode == "own":
                X, dY = rnd(M, k), rnd(M, n)
            elif mode == "shared":
                X, dY = x0[:, :k], y0[:, :n]
            else:      # every 2,048-token stretch aliases the same rows (stride trick: a [M, k] view cannot repeat rows, so the token count is cut
                X, dY = xs[:, :k], ys[:, :n]      # and the launch repeated: 12 launches of 2,048 tokens ~ 24,576 tokens)
            out.append((X, dY, torch.zeros(n, k, dtype=F32, device=dev), None if os.environ.get("NOBIAS") else torch.zeros(n, dtype=F32, device=dev), 1, n))
    return out


for mode in ("own", "shared"):
    pr = problems(mode)
    for sp in (3, 7):
        pl = [[p[:4] + (sp, p[5]) for p in pr]]
        us = timeit(lambda: [nv.wgrad_group(w, wide=True) for w in pl])
        print("%-8s %d launch(es), %d splits: %.1f us" % (mode, len(pl), pl[0][0][4], us))
pr = problems("small")
pl = [[p[:4] + (1, p[5]) for p in pr]]      # 72 tiles x 1 split of 2,048 tokens
us = timeit(lambda: [nv.wgrad_group(w, wide=True) for w in pl])
print("small (2,048 tokens, 1 split, 72 of 256 CUs busy): %.1f us per launch -> x (8020 / 2048) = %.1f us for a split of the full problem" % (us, us * 8020 / 2048))

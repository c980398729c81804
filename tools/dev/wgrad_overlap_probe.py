#!/usr/bin/env python3
"""Dev (verdict r5 item 4): what is there to win by running the HBM-bound weight-gradient launch BESIDE the issue-bound attention
backward?  The upper bound, free of graph-fork costs: the six encoder attention-backward launches of a config-2 step on one stream,
st_wgrad_wide (24 encoder problems, 3 splits, every operand its own tensor) on another, each alone and both at once, wall time
from a common start event to the later end (eager, two streams, events)."""
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
_, _, in_len, _, _ = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
lens_t = torch.tensor(in_len.tolist())
M = int(lens_t.sum())
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF16)
g, dO = rnd(M, 3 * d), rnd(M, d)
Q, K, V = g[:, :d], g[:, d:2 * d], g[:, 2 * d:]
rows = Rows.packed(lens_t, dev)
wf, wq, wk = attn_work(rows, rows, False, dk, H)
off = torch.zeros_like(lens_t); off[1:] = torch.cumsum(lens_t, 0)[:-1]
q_off, q_len = off.to(dev, I32), lens_t.to(dev, I32)
O = torch.empty(M, d, dtype=BF16, device=dev); lse = torch.empty(H * M, dtype=F32, device=dev)
mx = int(lens_t.max())
nv.attn_fwd(Q, K, V, O, lse, q_off, q_len, q_off, q_len, H, mx, False, 1 / math.sqrt(dk), work=wf, max_k=mx)
delta = (dO.float() * O.float()).view(M, H, dk).sum(-1).t().contiguous().view(-1)
dQ, dK, dV = (torch.empty(M, d, dtype=BF16, device=dev) for _ in range(3))
attn = lambda: nv.attn_bwd(Q, K, V, None, dO, lse, delta, dQ, dK, dV, q_off, q_len, q_off, q_len, H, mx, mx, False, 1 / math.sqrt(dk), parts=3, work_q=wq, work_k=wk)
probs = []
for _ in range(6):
    for (n, k) in ((768, 256), (256, 256), (1024, 256), (256, 1024)):
        probs.append((rnd(M, k), rnd(M, n), torch.zeros(n, k, dtype=F32, device=dev), torch.zeros(n, dtype=F32, device=dev), 3, n))
wgrad = lambda: nv.wgrad_group(probs, wide=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(do_attn, do_wgrad, n_attn=6):
    torch.cuda.synchronize()
    t0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0.record()
    s1.wait_event(t0); s2.wait_event(t0)
    with torch.cuda.stream(s1):
        if do_attn:
            for _ in range(n_attn): attn()
        e1.record()
    with torch.cuda.stream(s2):
        if do_wgrad: wgrad()
        e2.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(e1) * 1e3, t0.elapsed_time(e2) * 1e3


for _ in range(3): run(True, True)
for rep in range(3):
    a = run(True, False)[0]; w = run(False, True)[1]; b = run(True, True)
    print("6 x attention backward alone %.1f us | st_wgrad_wide alone %.1f us | together: attention done at %.1f, weight gradients at %.1f "
          "(serial %.1f -> %.1f: %.1f us hidden)" % (a, w, b[0], b[1], a + w, max(b), a + w - max(b)))

#!/usr/bin/env python3
"""Dev: per-workgroup clock stamps of the hand-scheduled attention backward (merged launch, config-2 encoder batch):
effective shader clock, item durations by body, slot occupancy over time, the drain.
Needs a DEVELOPMENT build of the library (the shipped one has no trace hook):
    ST_DEV_TRACE=1 python -m st_amd.build   (or `ST_DEV_TRACE=1 python __graft_entry__.py`) before running this script."""
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
_, _, in_len, _, _ = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
lens = in_len.tolist() if len(sys.argv) < 2 else [int(sys.argv[2])] * int(sys.argv[1])
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
n_k, n_q = wk.numel() * H, wq.numel() * H
trace = torch.zeros(n_k + n_q, 6, dtype=torch.int64, device=dev)
f = lambda: nv.attn_bwd(Q, K, V, None, dO, lse, delta, dQ, dK, dV, q_off, q_len, q_off, q_len, H, max(lens), max(lens), False, scale,
                        work_q=wq, work_k=wk)
for _ in range(3):
    f()
os.environ["ST_ATTN_TRACE_PTR"] = hex(trace.data_ptr())
for _ in range(2):
    f()
torch.cuda.synchronize()
del os.environ["ST_ATTN_TRACE_PTR"]
t = trace.cpu().double()
clk = ((t[:, 1] - t[:, 0]).sum() / (t[:, 3] - t[:, 2]).sum()).item() * 100e6      # shader clocks per 100 MHz tick
t0 = t[:, 2].min()
start, end = (t[:, 2] - t0) / 100.0, (t[:, 3] - t0) / 100.0     # us
dur = end - start
span = end.max().item()
print("workgroups %d (dK/dV %d, dQ %d); effective shader clock %.2f GHz; span %.1f us; slot time %.1f us (of 512 slots: %.0f%% occupied)"
      % (len(dur), n_k, n_q, clk / 1e9, span, dur.sum().item() / 512, 100 * dur.sum().item() / 512 / span))
for nm, sl in (("dK/dV", slice(0, n_k)), ("dQ", slice(n_k, None))):
    x = dur[sl]
    print("  %-5s items: duration min %.1f median %.1f max %.1f us; first start %.1f last start %.1f last end %.1f"
          % (nm, x.min(), x.median(), x.max(), start[sl].min(), start[sl].max(), end[sl].max()))
pro, epi = (t[:, 4] - t[:, 2]) / 100.0, (t[:, 3] - t[:, 5]) / 100.0
for nm, sl in (("dK/dV", slice(0, n_k)), ("dQ", slice(n_k, None))):
    print("  %-5s prologue (C++ part) median %.2f us (p90 %.2f), epilogue median %.2f us (p90 %.2f)" % (nm, pro[sl].median(), pro[sl].quantile(0.9), epi[sl].median(), epi[sl].quantile(0.9)))
# occupancy over time
import numpy as np
grid = np.linspace(0, span, 41)
s_, e_ = start.numpy(), end.numpy()
print("  resident workgroups at t (us):", " ".join("%d" % int(((s_ <= x) & (e_ > x)).sum()) for x in grid))
# gaps: time between an item's end and the next start on... (approximation: sort starts, compare with ends)
order = np.argsort(s_)
late = s_[order][512:]
ends_sorted = np.sort(e_)[:len(late)]
print("  relaunch latency (k-th start past the first 512 minus k-th end): median %.2f us, p90 %.2f us" % (np.median(late - ends_sorted), np.percentile(late - ends_sorted, 90)))
# per-step cost from the items' durations vs their step counts
wk_c, wq_c = wk.cpu().numpy(), wq.cpu().numpy()
def steps(w, other_lens):
    b = w >> 16
    return np.ceil(np.array(other_lens)[b] / 64.0) * 2
for nm, w, sl in (("dK/dV", wk_c, slice(0, n_k)), ("dQ", wq_c, slice(n_k, None))):
    st = np.repeat(steps(w, lens), H)
    x = dur[sl].numpy()
    A = np.vstack([st, np.ones_like(st)]).T
    coef, *_ = np.linalg.lstsq(A, x, rcond=None)
    print("  %-5s item duration ~ %.3f us per step + %.2f us fixed (fit over %d items, steps %d..%d)" % (nm, coef[0], coef[1], len(x), st.min(), st.max()))

"""Dev: per-workgroup timeline of the merged attention backward at the encoder shape (needs the temporary trace hook in
attn_bwd_kernel: st_dev_bwd_trace)."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import torch
from st_amd import native as nv, synthetic
from st_amd.functional import Rows, attn_work
dev = "cuda"
_, _, in_len, _, _ = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
rows = Rows.packed(in_len, dev)
M, H, d = int(in_len.sum()), 4, 256
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
qkv, dO = rnd(M, 3 * d), rnd(M, d)
O = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
lse, delta = torch.empty(H * M, dtype=torch.float32, device=dev), torch.randn(H * M, device=dev) * 0.01
wf, wq, wk = attn_work(rows, rows, False, 64, H)
Q, K, V = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
nv.attn_fwd(Q, K, V, O, lse, rows.off, rows.len, rows.off, rows.len, H, rows.max_len, False, 0.125, work=wf, max_k=rows.max_len)
dqkv = torch.empty(M, 3 * d, dtype=torch.bfloat16, device=dev)
run = lambda: nv.attn_bwd(Q, K, V, None, dO, lse, delta, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], rows.off, rows.len, rows.off, rows.len, H,
                          rows.max_len, rows.max_len, False, 0.125, work_q=wq, work_k=wk)
for _ in range(3): run()
torch.cuda.synchronize()
nk, nq = wk.numel() * H, wq.numel() * H
n = nk + nq
trace = torch.zeros(n * 4, dtype=torch.int64, device=dev)
lib = nv.load()._cdll
lib.st_dev_bwd_trace.argtypes = [ctypes.c_void_p]
assert lib.st_dev_bwd_trace(trace.data_ptr()) == 0
run(); torch.cuda.synchronize()
lib.st_dev_bwd_trace(None)
t = trace.view(n, 4).cpu()
live = t[:, 1] > 0
t0 = t[live, 0].min()
st, en = (t[:, 0] - t0).double() / 100, (t[:, 1] - t0).double() / 100
kinds = torch.zeros(n, dtype=torch.bool); kinds[:nk] = True      # dK/dV items first
print("workgroups %d (dK/dV %d, dQ %d; live %d); span %.1f us" % (n, nk, nq, int(live.sum()), en[live].max()))
for nm, m in (("dK/dV", kinds & live), ("dQ", ~kinds & live)):
    dur = (en - st)[m]
    print("  %-6s items: duration avg %.1f us (min %.1f max %.1f); start avg %.1f (max %.1f); end max %.1f" % (
        nm, dur.mean(), dur.min(), dur.max(), st[m].mean(), st[m].max(), en[m].max()))
for lo in range(0, int(math.ceil(en[live].max())), 5):
    act = live & (st <= lo) & (en > lo)
    print("  t = %3d us: %4d resident (%4d dK/dV, %4d dQ)" % (lo, int(act.sum()), int((act & kinds).sum()), int((act & ~kinds).sum())))
busy = (en - st)[live].sum()
print("sum of workgroup durations %.0f us = %.1f us x 512 slots" % (busy, busy / 512))

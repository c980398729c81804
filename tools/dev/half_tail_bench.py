"""Dev: the merged attention backward (encoder self-attention shape of config 2) with the cheapest FRAC of its dQ items handed over
as HALF tiles (work-list entry bit 15: 64 query rows through the key-split body) against the plain list: results and time.
NEEDS the experimental kernel (not in the tree: attn_bwd_kernel<64, false, 1, HALF = true> - attn_bwd_dq_body decodes the item and
sends entries with bit 15 to attn_bwd_dq_item<.., 2>; DESIGN.md section 4, "half tiles"): with the shipped library the flagged
entries are no-ops and dQ comes out incomplete."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")): sys.path.insert(0, p)
from st_amd import native as nv, synthetic
from st_amd.functional import Rows, attn_work
BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32
dev = "cuda"
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF16)
_, _, in_len, _, _ = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
M, d, H = int(in_len.sum()), 256, 4
rows = Rows.packed(in_len, dev)
wf, wq, wk = attn_work(rows, rows, False, 64, H)
qkv = rnd(M, 3 * d)
Q, K, V = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
O, dO = torch.empty(M, d, dtype=BF16, device=dev), rnd(M, d)
lse, delta = torch.empty(H * M, dtype=F32, device=dev), torch.empty(H * M, dtype=F32, device=dev)
scale = 0.125
nv.attn_fwd(Q, K, V, O, lse, rows.off, rows.len, rows.off, rows.len, H, int(in_len.max()), False, scale, work=wf, max_k=int(in_len.max()))
delta.copy_((dO.float() * O.float()).view(M, H, 64).sum(-1).t().reshape(-1))
lens = in_len.tolist()


def half_list(frac):
    ent = wq.cpu().tolist()
    real = [i for i, e in enumerate(ent) if (e & 0xffff) != 0xffff]
    n = int(len(real) * frac)
    tail = set(real[len(real) - n:]) if n else set()
    out = []
    for i, e in enumerate(ent):
        if i in tail:
            b, t = e >> 16, e & 0xffff
            out.append((b << 16) | 0x8000 | (2 * t))
            if 128 * t + 64 < lens[b]:
                out.append((b << 16) | 0x8000 | (2 * t + 1))
        else:
            out.append(e)
    return torch.tensor(out, dtype=I32, device=dev)


def run(work_q, n=30):
    dQ, dK, dV = (torch.zeros(M, d, dtype=BF16, device=dev) for _ in range(3))
    f = lambda: nv.attn_bwd(Q, K, V, None, dO, lse, delta, dQ, dK, dV, rows.off, rows.len, rows.off, rows.len, H, int(in_len.max()),
                            int(in_len.max()), False, scale, parts=3, work_q=work_q, work_k=wk)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3, dQ, dK, dV


base = run(wq)
for frac in (0.0, 0.15, 0.25, 0.35, 0.5):
    us, dQ, dK, dV = run(half_list(frac))
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    print("half tiles for the cheapest %3d %% of the dQ items: %6.1f us   dQ rel %.2e dK rel %.2e dV rel %.2e (vs the plain list)"
          % (frac * 100, us, rel(dQ, base[1]), rel(dK, base[2]), rel(dV, base[3])))

#!/usr/bin/env python3
"""Dev: the hand-scheduled attention backward (csrc/st_attn_bwd64.hip) against the general kernels (ST_ATTN_BWD64=0, same
process) and an fp64 reference, on small ragged shapes and at the encoder shape of config 2; timings of both."""
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


def reference(Q, K, V, dO, lens):
    """fp64 softmax attention backward per utterance / head; returns O, lse (log2), delta, dQ, dK, dV"""
    M = Q.shape[0]
    O = torch.zeros(M, d, dtype=torch.float64)
    dQ, dK, dV = torch.zeros_like(O), torch.zeros_like(O), torch.zeros_like(O)
    lse = torch.zeros(H, M, dtype=torch.float64)
    delta = torch.zeros(H, M, dtype=torch.float64)
    off = 0
    for L in lens:
        for h in range(H):
            sl = slice(h * dk, (h + 1) * dk)
            q, k, v, do = (t[off:off + L, sl].double() for t in (Q, K, V, dO))
            s = q @ k.T * scale
            p = torch.softmax(s, -1)
            o = p @ v
            O[off:off + L, sl] = o
            lse[h, off:off + L] = torch.logsumexp(s, -1) * 1.4426950408889634
            dl = (do * o).sum(-1)
            delta[h, off:off + L] = dl
            dp = do @ v.T
            ds = p * (dp - dl[:, None])
            dQ[off:off + L, sl] = ds @ k * scale
            dK[off:off + L, sl] = ds.T @ q * scale
            dV[off:off + L, sl] = p.T @ do
        off += L
    return O, lse, delta, dQ, dK, dV


def run_case(lens, seed, check_ref=True, time_it=False, label=""):
    torch.manual_seed(seed)
    lens_t = torch.tensor(lens)
    M = int(lens_t.sum())
    qkv = (torch.randn(M, 3 * d) * 0.7).to(BF16)
    dO = (torch.randn(M, d) * 0.5).to(BF16)
    Qc, Kc, Vc = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    rows = Rows.packed(lens_t, dev)
    wf, wq, wk = attn_work(rows, rows, False, dk, H)
    off = torch.zeros_like(lens_t)
    off[1:] = torch.cumsum(lens_t, 0)[:-1]
    q_off, q_len = off.to(dev, I32), lens_t.to(dev, I32)
    g = qkv.to(dev)
    Q, K, V = g[:, :d], g[:, d:2 * d], g[:, 2 * d:]
    dOg = dO.to(dev)
    O = torch.empty(M, d, dtype=BF16, device=dev)
    lse = torch.empty(H * M, dtype=F32, device=dev)
    nv.attn_fwd(Q, K, V, O, lse, q_off, q_len, q_off, q_len, H, max(lens), False, scale, work=wf, max_k=max(lens))
    # delta = rowsum(dO * O) per head, as the producer of dO supplies it
    delta = (dOg.float() * O.float()).view(M, H, dk).sum(-1).t().contiguous().view(-1)
    outs = {}
    for mode in ("0", "1"):
        os.environ["ST_ATTN_BWD64"] = mode
        nv.env_refresh()
        dQ = torch.full((M, d), float("nan"), dtype=BF16, device=dev)
        dK, dV = torch.full_like(dQ, float("nan")), torch.full_like(dQ, float("nan"))
        nv.attn_bwd(Q, K, V, None, dOg, lse, delta, dQ, dK, dV, q_off, q_len, q_off, q_len, H, max(lens), max(lens), False, scale,
                    work_q=wq, work_k=wk)
        torch.cuda.synchronize()
        outs[mode] = (dQ.float().cpu(), dK.float().cpu(), dV.float().cpu())
    ok = True
    ref = reference(Qc, Kc, Vc, dO, lens)[3:] if check_ref else None
    for i, nm in enumerate(("dQ", "dK", "dV")):
        a, b = outs["0"][i], outs["1"][i]
        fin = bool(torch.isfinite(b).all())
        rel = ((a - b).norm() / a.norm()).item() if fin else float("nan")
        line = "%-18s %s new-vs-old rel-L2 %.3e finite=%s" % (label, nm, rel, fin)
        if ref is not None:
            r = ref[i].float()
            line += "   vs fp64: old %.3e new %.3e" % (((a - r).norm() / r.norm()).item(), ((b - r).norm() / r.norm()).item() if fin else float("nan"))
        print(line)
        if not fin or rel > 2e-2:
            ok = False
            bad = (~torch.isfinite(b)) | ((a - b).abs() > 0.05 * a.abs().max())
            rows_bad = bad.any(1).nonzero().flatten()
            print("   bad rows: %d of %d; first %s ... last %s" % (rows_bad.numel(), M, rows_bad[:8].tolist(), rows_bad[-4:].tolist()))
            cols_bad = bad.any(0).nonzero().flatten()
            print("   bad cols: %d; first %s" % (cols_bad.numel(), cols_bad[:16].tolist()))
    if time_it:
        for mode in ("0", "1"):
            os.environ["ST_ATTN_BWD64"] = mode
            nv.env_refresh()
            dQ, dK, dV = (torch.empty(M, d, dtype=BF16, device=dev) for _ in range(3))
            f = lambda: nv.attn_bwd(Q, K, V, None, dOg, lse, delta, dQ, dK, dV, q_off, q_len, q_off, q_len, H, max(lens), max(lens), False,
                                    scale, work_q=wq, work_k=wk)
            for parts, nm in ((3, "all"), (1, "dq"), (2, "dkv")):
                g_ = lambda: nv.attn_bwd(Q, K, V, None, dOg, lse, delta, dQ, dK, dV, q_off, q_len, q_off, q_len, H, max(lens), max(lens),
                                         False, scale, parts=parts, work_q=wq, work_k=wk)
                for _ in range(3):
                    g_()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(30):
                    g_()
                e.record()
                torch.cuda.synchronize()
                print("   time %-4s ST_ATTN_BWD64=%s: %.1f us" % (nm, mode, s.elapsed_time(e) / 30 * 1e3))
    return ok


if __name__ == "__main__":
    ok = True
    for i, lens in enumerate(([200, 131], [64], [129, 130, 257], [300, 520, 191], [1000, 640], [65, 63, 64, 128, 192])):
        ok &= run_case(lens, 10 + i, label=str(lens)[:18])
    _, _, in_len, _, _ = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
    ok &= run_case(in_len.tolist(), 99, check_ref=False, time_it=True, label="config 2 encoder")
    print("ALL OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)

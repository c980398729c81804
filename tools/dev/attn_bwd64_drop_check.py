#!/usr/bin/env python3
"""Dev: the dropout variant of the hand-scheduled attention backward (csrc/st_attn_bwd64.hip, *_drop.inc) against the general
kernels with the same Drop (ST_ATTN_BWD64=e: streams in eval mode only) - the masks must be the forward's, so the two agree
to rounding - and timings of both at the encoder shape of config 2."""
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


def run_case(lens, seed, p=0.1, time_it=False, label=""):
    torch.manual_seed(seed)
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
    drop = nv.Drop(torch.tensor([1234 + seed], dtype=I32, device=dev), 77, p)
    O = torch.empty(M, d, dtype=BF16, device=dev)
    lse = torch.empty(H * M, dtype=F32, device=dev)
    nv.attn_fwd(Q, K, V, O, lse, q_off, q_len, q_off, q_len, H, max(lens), False, scale, work=wf, max_k=max(lens), drop=drop)
    delta = (dO.float() * O.float()).view(M, H, dk).sum(-1).t().contiguous().view(-1)
    outs = {}
    for mode in ("e", "1"):
        os.environ["ST_ATTN_BWD64"] = mode
        nv.env_refresh()
        dQ, dK, dV = (torch.full((M, d), float("nan"), dtype=BF16, device=dev) for _ in range(3))
        nv.attn_bwd(Q, K, V, None, dO, lse, delta, dQ, dK, dV, q_off, q_len, q_off, q_len, H, max(lens), max(lens), False, scale,
                    work_q=wq, work_k=wk, drop=drop)
        torch.cuda.synchronize()
        outs[mode] = (dQ.float().cpu(), dK.float().cpu(), dV.float().cpu())
    ok = True
    for i, nm in enumerate(("dQ", "dK", "dV")):
        a, b = outs["e"][i], outs["1"][i]
        fin = bool(torch.isfinite(b).all())
        rel = ((a - b).norm() / a.norm()).item() if fin else float("nan")
        print("%-18s %s stream-vs-general rel-L2 %.3e finite=%s" % (label, nm, rel, fin))
        if not fin or rel > 1e-2:
            ok = False
            bad = (~torch.isfinite(b)) | ((a - b).abs() > 0.05 * a.abs().max())
            rows_bad = bad.any(1).nonzero().flatten()
            print("   bad rows: %d of %d; first %s ... last %s" % (rows_bad.numel(), M, rows_bad[:8].tolist(), rows_bad[-4:].tolist()))
    if time_it:
        for mode in ("e", "1"):
            os.environ["ST_ATTN_BWD64"] = mode
            nv.env_refresh()
            dQ, dK, dV = (torch.empty(M, d, dtype=BF16, device=dev) for _ in range(3))
            for parts, nm in ((3, "all"), (1, "dq"), (2, "dkv")):
                f = lambda: nv.attn_bwd(Q, K, V, None, dO, lse, delta, dQ, dK, dV, q_off, q_len, q_off, q_len, H, max(lens), max(lens),
                                        False, scale, parts=parts, work_q=wq, work_k=wk, drop=drop)
                for _ in range(3):
                    f()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(30):
                    f()
                e.record()
                torch.cuda.synchronize()
                print("   time %-4s ST_ATTN_BWD64=%s: %.1f us" % (nm, mode, s.elapsed_time(e) / 30 * 1e3))
    return ok


if __name__ == "__main__":
    ok = True
    for i, lens in enumerate(([200, 131], [129, 130, 257], [300, 520, 191], [1000, 640], [65, 63, 64, 128, 192])):
        ok &= run_case(lens, 10 + i, label=str(lens)[:18])
    ok &= run_case([300, 257], 31, p=0.5, label="[300, 257] p=0.5")
    _, _, in_len, _, _ = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
    ok &= run_case(in_len.tolist(), 99, time_it=True, label="config 2 encoder")
    print("ALL OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)

#!/usr/bin/env python3
"""Dev: kernel-level accuracy of the encoder's attention backward at the config-2 encoder shape (32 utterances of 500-1000
frames, 4 heads of 64): dQ / dK / dV of the generated instruction streams (csrc/st_attn_bwd64.hip: the register-resident
operand pre-multiplied by scale * log2 e and re-rounded to bf16) and of the general kernels (ST_ATTN_BWD64=0: scores scaled
in fp32) against an fp64 reference on the same bf16 inputs - eval mode and with attention dropout (the forward's own masks,
re-derived with the host implementation of the counter hash, tests/_emul.py).  Third column (round 5, what the product runs):
the streams on PRE-SCALED keys K~ = bf16(scale log2e k) from an fp32 key projection (one rounding, as st_row_chain's epilogue
does it; k_prescaled = 1) against the fp64 reference on those keys.  `sharp` = the same with 3x larger q / k
(score spread 9x: peaky attention, where a perturbed score matters most).  VERDICT r4 "weak 1" asked for this table."""
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
from tests import _emul as em  # noqa: E402

BF16, F32, I32, F64 = torch.bfloat16, torch.float32, torch.int32, torch.float64
dev = "cuda"
H, dk = 4, 64
d = H * dk
scale = 1 / math.sqrt(dk)


def keep_mask(drop, bh, lq, lk):
    """the forward's keep decisions of (utterance, head) bh as a [lq, lk] 0/1 fp64 matrix x 1 / (1 - p)"""
    e = em.Drop(drop.seed.detach().cpu(), drop.salt, drop.thresh / 256.0)
    key = em._key(e)
    q = torch.arange(lq, dtype=torch.int64, device=dev).view(-1, 1)
    k = torch.arange(lk, dtype=torch.int64, device=dev).view(1, -1)
    cnt = ((((q >> 1) << 15) | (k >> 1)) + bh * 0x85ebca6b) & em._M32
    bits = em._hash32(cnt ^ key)
    keep = ((bits >> (8 * (2 * (q & 1) + (k & 1)))) & 0xFF) >= e.thresh
    return keep.double() * e.scale


def reference(Q, K, V, dO, lens, drop):
    M = Q.shape[0]
    dQ, dK, dV = (torch.zeros(M, d, dtype=F64, device=dev) for _ in range(3))
    off = 0
    for b, L in enumerate(lens):
        for h in range(H):
            sl = slice(h * dk, (h + 1) * dk)
            q, k, v, do = (t[off:off + L, sl].double() for t in (Q, K, V, dO))      # (K may be an fp32 matrix: the pre-scaled keys divided back)
            p = torch.softmax(q @ k.T * scale, -1)
            m = keep_mask(drop, b * H + h, L, L) if drop is not None else None
            pd = p * m if m is not None else p
            o = pd @ v
            dl = (do * o).sum(-1, keepdim=True)
            dp = do @ v.T
            if m is not None:
                dp = dp * m
            ds = p * (dp - dl)
            dQ[off:off + L, sl] = ds @ k * scale
            dK[off:off + L, sl] = ds.T @ q * scale
            dV[off:off + L, sl] = pd.T @ do
        off += L
    return dQ, dK, dV


def rel(a, b):
    return ((a.double() - b).norm() / b.norm()).item()


def run(lens, gain, p, label):
    torch.manual_seed(5)
    lens_t = torch.tensor(lens)
    M = int(lens_t.sum())
    g = (torch.randn(M, 3 * d, device=dev) * 0.7)
    g[:, :2 * d] *= gain
    k32 = g[:, d:2 * d].clone()
    g = g.to(BF16)
    dO = (torch.randn(M, d, device=dev) * 0.5).to(BF16)
    Q, K, V = g[:, :d], g[:, d:2 * d], g[:, 2 * d:]
    rows = Rows.packed(lens_t, dev)
    wf, wq, wk = attn_work(rows, rows, False, dk, H)
    off = torch.zeros_like(lens_t)
    off[1:] = torch.cumsum(lens_t, 0)[:-1]
    q_off, q_len = off.to(dev, I32), lens_t.to(dev, I32)
    drop = nv.Drop(torch.tensor([4242], dtype=I32, device=dev), 77, p) if p else None
    O, Ores = torch.empty(M, d, dtype=BF16, device=dev), torch.empty(M, d, dtype=BF16, device=dev)
    lse = torch.empty(H * M, dtype=F32, device=dev)
    nv.attn_fwd(Q, K, V, O, lse, q_off, q_len, q_off, q_len, H, max(lens), False, scale, work=wf, max_k=max(lens), drop=drop, ores=Ores)
    delta = (dO.float() * (O.float() + Ores.float())).view(M, H, dk).sum(-1).t().contiguous().view(-1)      # as the chain's epilogue forms it
    ref = reference(Q, K, V, dO, lens, drop)
    out = {}
    for mode, name in (("1", "streams"), ("0", "general")):
        os.environ["ST_ATTN_BWD64"] = mode
        nv.env_refresh()
        got = [torch.full((M, d), float("nan"), dtype=BF16, device=dev) for _ in range(3)]
        nv.attn_bwd(Q, K, V, None, dO, lse, delta, *got, q_off, q_len, q_off, q_len, H, max(lens), max(lens), False, scale,
                    work_q=wq, work_k=wk, drop=drop)
        torch.cuda.synchronize()
        out[name] = [rel(a, r) for a, r in zip(got, ref)]
    # the product's form: keys pre-scaled once from their fp32 values
    c2 = scale * nv.K_LOG2_SCALE
    Kt = (k32 * c2).to(BF16)
    os.environ["ST_ATTN_BWD64"] = "1"
    nv.env_refresh()
    nv.attn_fwd(Q, Kt, V, O, lse, q_off, q_len, q_off, q_len, H, max(lens), False, scale, work=wf, max_k=max(lens), drop=drop, ores=Ores,
                k_prescaled=True)
    delta = (dO.float() * (O.float() + Ores.float())).view(M, H, dk).sum(-1).t().contiguous().view(-1)
    ref_t = reference(Q, (Kt.float() / c2), V, dO, lens, drop)
    got = [torch.full((M, d), float("nan"), dtype=BF16, device=dev) for _ in range(3)]
    nv.attn_bwd(Q, Kt, V, None, dO, lse, delta, *got, q_off, q_len, q_off, q_len, H, max(lens), max(lens), False, scale,
                work_q=wq, work_k=wk, drop=drop, k_prescaled=True)
    torch.cuda.synchronize()
    out["kpre"] = [rel(a, r) for a, r in zip(got, ref_t)]
    print("%-34s streams, re-rounded operand dQ %.3e dK %.3e dV %.3e | general dQ %.3e dK %.3e dV %.3e | streams, pre-scaled keys dQ %.3e dK %.3e dV %.3e"
          % (label, *out["streams"], *out["general"], *out["kpre"]), flush=True)


if __name__ == "__main__":
    _, _, in_len, _, _ = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
    lens = in_len.tolist()
    print("# rel-L2 against fp64 (same bf16 inputs; delta from O + Ores as in the step), config-2 encoder shape: %d rows" % sum(lens))
    run(lens, 1.0, 0.0, "eval")
    run(lens, 1.0, 0.1, "dropout 0.1")
    run(lens, 3.0, 0.0, "eval, sharp (q, k x 3)")
    run(lens, 3.0, 0.1, "dropout 0.1, sharp (q, k x 3)")
    run(lens[:8], 6.0, 0.0, "eval, very sharp (q, k x 6), 8 utt")

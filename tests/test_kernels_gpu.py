"""-m gpu: every C-ABI kernel, called through st_amd.native on a real MI355X,
against a plain PyTorch fp32 reference of the same op (tests/_emul.py - the
same functions the CPU composition test uses) on identical seeded inputs.

Tolerances: operands are bf16 on both sides, accumulation is fp32 on both
sides, so differences come from accumulation order and one bf16 rounding of the
result: rel-L2 <= 1e-2 for bf16 outputs, <= 2e-3 for fp32 outputs / reductions.
"""
import math
import os

import pytest
import torch

from st_amd import native as nv
from tests import _emul as em

pytestmark = pytest.mark.gpu
BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def check(got, ref, tol, what):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert torch.isfinite(got).all(), "%s: non-finite output" % what
    r = rel(got, ref)
    if r > tol:
        err = (got - ref).abs()
        idx = torch.nonzero(err == err.max())[0].tolist()
        raise AssertionError("%s: rel-L2 %.3e > %.1e; max |err| %.4g at %s (got %.5g, ref %.5g)"
                             % (what, r, tol, err.max().item(), idx, got[tuple(idx)].item(), ref[tuple(idx)].item()))


def g(*shape, seed=0, scale=1.0, dtype=BF16):
    gen = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=gen) * scale).to(dtype)


def cu(t):
    return None if t is None else t.cuda()


# ---- hardware layout probes -----------------------------------------------------------------
def test_probe_tr16_layout():
    """ds_read_b64_tr_b16 fragment: lane l gets c = hi*8..hi*8+7 of row (l & 31) from a [c][row] tile.
    Bit patterns (c*64 + row) travel through the read untouched, so the check is exact."""
    tile = (torch.arange(16).view(16, 1) * 64 + torch.arange(64).view(1, 64)).to(torch.int16)
    out = torch.zeros(64, 8, dtype=BF16, device="cuda")
    nv.probe_tr16(tile.cuda().contiguous().view(BF16), out)
    torch.cuda.synchronize()
    exp = torch.zeros(64, 8, dtype=torch.int16)
    for l in range(64):
        for j in range(8):
            exp[l, j] = ((l >> 5) * 8 + j) * 64 + (l & 31)
    got = out.view(torch.int16).cpu()
    assert torch.equal(got, exp), "tr16 fragment layout differs: got (c,row) per lane\n%s" % [
        [(int(v) // 64, int(v) % 64) for v in got[l]] for l in (0, 1, 4, 16, 17, 32, 48)]


def test_probe_mfma_layout():
    A = g(32, 16, seed=1)
    Bt = g(32, 16, seed=2)
    D = torch.zeros(32, 32, dtype=F32, device="cuda")
    nv.probe_mfma(A.cuda(), Bt.cuda(), D)
    torch.cuda.synchronize()
    check(D, A.float() @ Bt.float().t(), 1e-5, "mfma 32x32x16 layout (asymmetric operands)")


# ---- GEMM family ---------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 256, 80), (1000, 768, 256), (257, 1024, 256), (77, 256, 1024),
                                     (200, 256, 2304), (1206, 256, 4344)])     # long K, few tiles: the 8-wave split-K variant
@pytest.mark.parametrize("epi", [nv.EPI_BF16, nv.EPI_BF16_RELU, nv.EPI_F32])
def test_gemm_forward(M, N, K, epi):
    X, W, b = g(M, K, seed=1), g(N, K, seed=2, scale=K ** -0.5), g(N, seed=3, dtype=F32)
    odt = F32 if epi == nv.EPI_F32 else BF16
    ref = em.gemm(X, W, torch.zeros(M, N, dtype=odt), bias=b, epi=epi)
    out = nv.gemm(cu(X), cu(W), torch.full((M, N), float("nan"), dtype=odt, device="cuda"), bias=cu(b), epi=epi)
    check(out, ref, 2e-3 if epi == nv.EPI_F32 else 1e-2, "gemm fwd %s epi %d" % ((M, N, K), epi))


def test_gemm_strided_operands():
    """Operands / outputs that are column slices of wider matrices (the fused qkv buffer)."""
    M, d = 500, 256
    big = g(M, 3 * d, seed=4)
    W = g(d, d, seed=5, scale=d ** -0.5)
    outbig = torch.zeros(M, 2 * d, dtype=BF16)
    ref = em.gemm(big[:, d:2 * d], W, outbig.clone()[:, d:], bias=None)
    ob = cu(outbig)
    nv.gemm(cu(big)[:, d:2 * d], cu(W), ob[:, d:])
    check(ob[:, d:], ref, 1e-2, "gemm strided")
    assert ob[:, :d].abs().max().item() == 0


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (1000, 256, 1024), (333, 768, 256), (64, 4344, 256), (1206, 4344, 256)])
@pytest.mark.parametrize("epi", [nv.EPI_BF16, nv.EPI_BF16_MASK, nv.EPI_BF16_ADD])
def test_gemm_dgrad(M, N, K, epi):
    """dx[M,K] = dy[M,N] W[N,K] (W read contraction-major through the transposing LDS read)."""
    dy, W = g(M, N, seed=1), g(N, K, seed=2, scale=N ** -0.5)
    aux = g(M, K, seed=3) if epi != nv.EPI_BF16 else None
    ref = em.gemm(dy, W, torch.zeros(M, K, dtype=BF16), aux=aux, epi=epi, y_cmajor=True)
    out = nv.gemm(cu(dy), cu(W), torch.full((M, K), float("nan"), dtype=BF16, device="cuda"), aux=cu(aux), epi=epi,
                  y_cmajor=True)
    check(out, ref, 1e-2, "gemm dgrad %s epi %d" % ((M, N, K), epi))


@pytest.mark.parametrize("M,N,K,splits", [(256, 128, 128, 1), (1000, 256, 256, 4), (5000, 768, 256, 16),
                                          (999, 1024, 256, 3), (2000, 256, 1024, 8), (1206, 4344, 256, 5),
                                          (777, 256, 80, 2)])
def test_gemm_wgrad(M, N, K, splits):
    """dW[N,K] += dy[M,N]^T x[M,K]: both operands contraction-major, split-K with fp32 atomics."""
    dy, x = g(M, N, seed=1), g(M, K, seed=2)
    init = g(N, K, seed=3, dtype=F32)
    ref = em.gemm(dy, x, init.clone(), epi=nv.EPI_F32_ATOMIC, x_cmajor=True, y_cmajor=True, m=N)
    out = nv.gemm(cu(dy), cu(x), cu(init.clone()), epi=nv.EPI_F32_ATOMIC, x_cmajor=True, y_cmajor=True, splits=splits,
                  m=N)
    check(out, ref, 2e-3, "gemm wgrad %s" % ((M, N, K, splits),))
    # the production form: lane axis = k, transposed (coalesced) atomic store
    out_t = nv.gemm(cu(x), cu(dy), cu(init.clone()), epi=nv.EPI_F32_ATOMIC_T, x_cmajor=True, y_cmajor=True,
                    splits=splits, n=N)
    check(out_t, ref, 2e-3, "gemm wgrad (transposed store) %s" % ((M, N, K, splits),))
    # ... which can also accumulate the bias gradient (column sums of dy)
    b0 = g(1, N, seed=4, dtype=F32).view(-1)
    db = cu(b0.clone())
    out_b = nv.gemm(cu(x), cu(dy), cu(init.clone()), bias=db, epi=nv.EPI_F32_ATOMIC_T, x_cmajor=True, y_cmajor=True,
                    splits=splits, n=N)
    check(out_b, ref, 2e-3, "gemm wgrad + bias grad %s" % ((M, N, K, splits),))
    check(db, b0 + dy.float().sum(0), 2e-3, "fused bias grad %s" % ((M, N, K, splits),))


@pytest.mark.parametrize("M,N,K", [(300, 128, 128), (1000, 256, 256), (130, 256, 1024), (70, 512, 512), (500, 256, 80),
                                     (1206, 256, 1024), (1206, 256, 544), (700, 128, 1000), (9000, 256, 1024),   # K >= 512, M <= 8192: 8-wave K split
                                     (16500, 256, 544),                                                         # 8-wave 128-row tiles
                                     (8250, 512, 512), (8230, 512, 1024)])                                      # d_model 512, M > 8192: 8-wave 64-row tiles
@pytest.mark.parametrize("variant", ["res", "relu_pe"])
def test_gemm_ln(M, N, K, variant):
    X, W = g(M, K, seed=1), g(N, K, seed=2, scale=K ** -0.5)
    b, gamma, beta = g(N, seed=3, dtype=F32), 1 + 0.2 * g(N, seed=4, dtype=F32), 0.2 * g(N, seed=5, dtype=F32)
    res = g(M, N, seed=6) if variant == "res" else None
    pe = g(64, N, seed=7, dtype=F32) if variant == "relu_pe" else None
    pos = (torch.arange(M) % 64).to(I32) if variant == "relu_pe" else None
    relu = variant == "relu_pe"

    def run(fn, dev):
        mv = (lambda t: None if t is None else t.to(dev))
        out, xhat, pre = (torch.zeros(M, N, dtype=BF16, device=dev) for _ in range(3))
        rstd = torch.zeros(M, dtype=F32, device=dev)
        fn(mv(X), mv(W), mv(b), mv(res), mv(gamma), mv(beta), out, xhat, rstd, eps=1e-6, relu=relu, pe=mv(pe),
           pos=mv(pos), pre=pre)
        return out, xhat, rstd, pre

    r = run(em.gemm_ln, "cpu")
    o = run(nv.gemm_ln, "cuda")
    for got, ref, nm, tol in zip(o, r, ("out", "xhat", "rstd", "pre"), (1e-2, 1e-2, 2e-3, 1e-2)):
        check(got, ref, tol, "gemm_ln %s %s %s" % ((M, N, K), variant, nm))


@pytest.mark.parametrize("M,N", [(100, 128), (1000, 256), (333, 512), (5000, 256)])
@pytest.mark.parametrize("masked", [False, True])
def test_ln_bwd(M, N, masked):
    dy, xhat = g(M, N, seed=1), g(M, N, seed=2)
    rstd, gamma = g(M, seed=3, dtype=F32).abs() + 0.5, 1 + 0.2 * g(N, seed=4, dtype=F32)
    mask = g(M, N, seed=5) if masked else None

    def run(fn, dev):
        mv = (lambda t: None if t is None else t.to(dev))
        dx = torch.zeros(M, N, dtype=BF16, device=dev)
        acc = [torch.ones(N, dtype=F32, device=dev) for _ in range(3)]   # accumulate semantics
        fn(mv(dy), mv(xhat), mv(rstd), mv(gamma), dx, acc[0], acc[1], acc[2], mask=mv(mask))
        return [dx] + acc

    r, o = run(em.ln_bwd, "cpu"), run(nv.ln_bwd, "cuda")
    for got, ref, nm, tol in zip(o, r, ("dx", "dgamma", "dbeta", "dbias"), (1e-2, 3e-3, 3e-3, 5e-3)):
        check(got, ref, tol, "ln_bwd %s masked=%s %s" % ((M, N), masked, nm))


# ---- attention ---------------------------------------------------------------------------
def _attn_case(B, H, dk, q_lens, k_lens, causal, packed, seed):
    d = H * dk
    self_attn = q_lens is None
    ql = k_lens if self_attn else q_lens
    if packed:
        q_off = [sum(ql[:i]) for i in range(B)]
        k_off = [sum(k_lens[:i]) for i in range(B)]
        Mq, Mk = sum(ql), sum(k_lens)
    else:
        Tq, Tk = max(ql), max(k_lens)
        q_off, k_off = [i * Tq for i in range(B)], [i * Tk for i in range(B)]
        Mq, Mk = B * Tq, B * Tk
    ti = lambda v: torch.tensor(v, dtype=I32)
    if self_attn:
        qkv = g(Mq, 3 * d, seed=seed)
        Q, K, V = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    else:
        Q = g(Mq, d, seed=seed)
        kv = g(Mk, 2 * d, seed=seed + 1)
        K, V = kv[:, :d], kv[:, d:]
    dO = g(Mq, d, seed=seed + 2)
    return dict(Q=Q, K=K, V=V, dO=dO, q_off=ti(q_off), q_len=ti(ql), k_off=ti(k_off), k_len=ti(k_lens), H=H,
                max_q=max(ql), max_k=max(k_lens), causal=causal, scale=1 / math.sqrt(dk), Mq=Mq, Mk=Mk, d=d)


ATTN_CASES = [
    # B, H, dk, q_lens (None = self), k_lens, causal, packed
    (2, 2, 32, None, [7, 4], False, True),
    (2, 2, 32, None, [7, 4], True, True),
    (3, 4, 64, None, [200, 131, 64], False, True),
    (3, 4, 64, None, [200, 131, 64], False, False),
    (2, 4, 64, None, [300, 257], True, True),
    (2, 4, 32, None, [129, 1], True, True),
    (2, 4, 64, [50, 33], [1000, 517], False, True),
    (3, 4, 32, [10, 5, 8], [60, 31, 45], False, True),
    (2, 4, 64, None, [1000, 640], False, True),
    (3, 4, 64, [64, 1, 40], [700, 256, 65], False, True),      # <= 64 queries, >= 256 keys: key-split kernels
    (2, 4, 32, [20, 33], [300, 129], False, False),
    (3, 2, 128, None, [200, 131, 64], False, True),            # d_k = 128 (config/character.yaml: d_model 512, 4 heads)
    (2, 2, 128, None, [300, 257], True, True),
    (3, 2, 128, [64, 1, 40], [700, 256, 65], False, True),     # ... and its key-split kernels
    (2, 2, 128, [20, 33], [300, 129], False, False),
]


@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention_fwd_bwd(case):
    c = _attn_case(*case, seed=11)

    def run(fwd, bwd, dev):
        mv = lambda t: t.to(dev)
        Q, K, V, dO = mv(c["Q"]), mv(c["K"]), mv(c["V"]), mv(c["dO"])
        meta = [mv(c[k]) for k in ("q_off", "q_len", "k_off", "k_len")]
        O = torch.zeros(c["Mq"], c["d"], dtype=BF16, device=dev)
        lse = torch.zeros(c["H"] * c["Mq"], dtype=F32, device=dev)
        fwd(Q, K, V, O, lse, *meta, c["H"], c["max_q"], c["causal"], c["scale"], max_k=c["max_k"])
        delta = torch.zeros_like(lse)
        dQ = torch.zeros(c["Mq"], c["d"], dtype=BF16, device=dev)
        dK, dV = (torch.zeros(c["Mk"], c["d"], dtype=BF16, device=dev) for _ in range(2))
        bwd(Q, K, V, O, dO, lse, delta, dQ, dK, dV, *meta, c["H"], c["max_q"], c["max_k"], c["causal"], c["scale"])
        return O, lse, dQ, dK, dV

    r = run(em.attn_fwd, em.attn_bwd, "cpu")
    o = run(nv.attn_fwd, nv.attn_bwd, "cuda")
    # rows outside every utterance (padded layout) are untouched zeros on both sides
    for got, ref, nm, tol in zip(o, r, ("O", "lse", "dQ", "dK", "dV"), (1.5e-2, 2e-3, 2.5e-2, 2.5e-2, 2e-2)):
        check(got, ref, tol, "attention %s %s" % (case, nm))


# delta supplied by the producer of dO (O = None): ONE launch for dQ and dK/dV - the form the training step uses.  Long non-causal
# problems with 64-wide heads take the hand-scheduled streams of csrc/st_attn_bwd64.hip (tile counts 1..16, utterances whose last
# 64-row tile holds 1 / 31 / 32 / 33 / 63 / 64 rows, a batch whose last 128-row tile has idle waves, padded layout); the others
# (causal, 32-wide heads, few queries) the general kernels' merged launch.
DELTA_CASES = [
    (3, 4, 64, None, [200, 131, 64], False, True),
    (3, 4, 64, None, [129, 130, 257], False, True),
    (3, 4, 64, None, [300, 520, 191], False, True),
    (6, 4, 64, None, [65, 63, 64, 128, 192, 161], False, True),
    (2, 4, 64, None, [1000, 640], False, True),
    (4, 4, 64, None, [417, 96, 33, 160], False, False),
    (2, 2, 64, None, [223, 352], False, True),
    (2, 4, 64, None, [300, 257], True, True),
    (2, 4, 32, None, [300, 131], False, True),
    (3, 4, 64, [64, 1, 40], [700, 256, 65], False, True),      # few queries, many keys: the dQ items' keys over up to 4 workgroups (round 6)
    (2, 4, 64, [50, 33], [1000, 517], False, True),
    (5, 4, 64, [38, 20, 64, 7, 45], [900, 640, 511, 129, 384], False, True),
    (3, 2, 128, [64, 1, 40], [700, 256, 65], False, True),
    (3, 4, 32, [33, 64, 2], [513, 300, 256], False, False),
]


def test_attention_backward_key_split_across_workgroups_is_reproducible():
    """The merged backward launch of a few-queries / many-keys problem cuts each dQ item's key tiles over up to four workgroups and the
    last arriver adds the fp32 partials in part order: the same bits every time (no atomics), tickets left zero, and within bf16
    rounding of the two-launch form (parts 1, then 2), which does not split."""
    c = _attn_case(6, 4, 64, [38, 20, 64, 7, 45, 33], [900, 640, 511, 129, 384, 1000], False, True, seed=77)
    H, Mq, dk = c["H"], c["Mq"], c["d"] // c["H"]
    assert nv.load()._cdll.st_attn_bwd_split_kib(6, H, dk, c["max_q"], c["max_k"], 0) > 0
    assert nv.load()._cdll.st_attn_bwd_split_kib(6, H, dk, c["max_k"], c["max_k"], 0) == 0      # (self-attention shapes do not split)
    meta = [cu(c[k]) for k in ("q_off", "q_len", "k_off", "k_len")]
    Q, K, V, dO = cu(c["Q"]), cu(c["K"]), cu(c["V"]), cu(c["dO"])
    O, lse = torch.zeros(Mq, c["d"], dtype=BF16, device="cuda"), torch.zeros(H * Mq, dtype=F32, device="cuda")
    nv.attn_fwd(Q, K, V, O, lse, *meta, H, c["max_q"], False, c["scale"], max_k=c["max_k"])
    delta = (dO.float() * O.float()).view(Mq, H, dk).sum(-1).t().contiguous().view(-1)

    def run(parts_seq):
        out = [torch.full((Mq, c["d"]), float("nan"), dtype=BF16, device="cuda")] + \
              [torch.full((c["Mk"], c["d"]), float("nan"), dtype=BF16, device="cuda") for _ in range(2)]
        for parts in parts_seq:
            nv.attn_bwd(Q, K, V, None, dO, lse, delta, *out, *meta, H, c["max_q"], c["max_k"], False, c["scale"], parts=parts)
        torch.cuda.synchronize()
        return out

    a, b, two = run((3,)), run((3,)), run((1, 2))
    for x, y, z, nm in zip(a, b, two, ("dQ", "dK", "dV")):
        assert torch.isfinite(x.float()).all(), nm
        assert torch.equal(x, y), "merged backward launch not reproducible: %s" % nm
        check(x.cpu(), z.cpu(), 8e-3, "key split across workgroups vs the two-launch form: %s" % nm)
    work = nv._ATTN_SPLIT_WORK[torch.device("cuda", torch.cuda.current_device()).index]
    assert int(work[:4096].abs().sum()) == 0, "tickets not reset"


@pytest.mark.parametrize("case", DELTA_CASES)
@pytest.mark.parametrize("use_work", [False, True])
def test_attention_bwd_delta_supplied(case, use_work):
    from st_amd.functional import Rows, attn_work
    c = _attn_case(*case, seed=23)
    H, Mq, dk = c["H"], c["Mq"], c["d"] // c["H"]
    meta_c = [c[k] for k in ("q_off", "q_len", "k_off", "k_len")]
    O, lse = torch.zeros(Mq, c["d"], dtype=BF16), torch.zeros(H * Mq, dtype=F32)
    em.attn_fwd(c["Q"], c["K"], c["V"], O, lse, *meta_c, H, c["max_q"], c["causal"], c["scale"], max_k=c["max_k"])
    delta = (c["dO"].float() * O.float()).view(Mq, H, dk).sum(-1).t().contiguous().view(-1)
    ref = [torch.zeros(Mq, c["d"], dtype=BF16), torch.zeros(c["Mk"], c["d"], dtype=BF16), torch.zeros(c["Mk"], c["d"], dtype=BF16)]
    em.attn_bwd(c["Q"], c["K"], c["V"], None, c["dO"], lse, delta, *ref, *meta_c, H, c["max_q"], c["max_k"], c["causal"], c["scale"])
    wq = wk = None
    if use_work:
        if not case[6]:
            pytest.skip("work lists are built for packed layouts")
        q_rows = Rows.packed(c["q_len"].long(), "cuda")
        k_rows = q_rows if case[3] is None else Rows.packed(c["k_len"].long(), "cuda")
        _, wq, wk = attn_work(q_rows, k_rows, c["causal"], dk, H)
    got = [torch.full((Mq, c["d"]), float("nan"), dtype=BF16, device="cuda")] + \
          [torch.full((c["Mk"], c["d"]), float("nan"), dtype=BF16, device="cuda") for _ in range(2)]
    nv.attn_bwd(cu(c["Q"]), cu(c["K"]), cu(c["V"]), None, cu(c["dO"]), cu(lse), cu(delta), *got, *[cu(m) for m in meta_c], H, c["max_q"],
                c["max_k"], c["causal"], c["scale"], work_q=wq, work_k=wk)
    for g_, r_, nm, tol, off, ln in zip(got, ref, ("dQ", "dK", "dV"), (2.5e-2, 2.5e-2, 2e-2), ("q_off", "k_off", "k_off"),
                                        ("q_len", "k_len", "k_len")):
        rows = torch.cat([torch.arange(int(o), int(o) + int(n)) for o, n in zip(c[off], c[ln])])      # utterance rows (padded layout: the rest is untouched)
        assert torch.isfinite(g_.float().cpu()[rows]).all(), "non-finite %s %s" % (nm, case)
        check(g_.cpu()[rows], r_[rows], tol, "attention (delta supplied) %s %s" % (case, nm))


# Pre-scaled keys (round 5): K~ = bf16(scale * log2(e) * k), scaled in the fp32 epilogue of the projection that produced the keys
# (st_row_chain's post_kscale), handed over with k_prescaled = True - no kernel multiplies per score, forward and both backward
# bodies exponentiate bit-identical scores.  Against the emulation on the SAME K~ (it divides the scale out in fp32): every kernel
# family - the long-sequence forward and the backward streams, the general kernels (short, causal, 32- and 128-wide heads), the
# padded layout - forward, backward with delta supplied (one launch) and st_attn_probs.
PRESCALED_CASES = [
    (3, 4, 64, None, [200, 131, 64], False, True),
    (2, 4, 64, None, [1000, 640], False, True),
    (4, 4, 64, None, [417, 96, 33, 160], False, False),
    (2, 4, 64, None, [300, 257], True, True),
    (2, 4, 32, None, [300, 131], False, True),
    (3, 2, 128, None, [200, 131, 64], False, True),
]


@pytest.mark.parametrize("case", PRESCALED_CASES)
def test_attention_prescaled_keys(case):
    c = _attn_case(*case, seed=31)
    H, Mq, Mk, d = c["H"], c["Mq"], c["Mk"], c["d"]
    dk = d // H
    kt = (c["K"].float() * (c["scale"] * nv.K_LOG2_SCALE)).to(BF16)      # what the projection's epilogue hands over
    meta_c = [c[k] for k in ("q_off", "q_len", "k_off", "k_len")]
    O, lse = torch.zeros(Mq, d, dtype=BF16), torch.zeros(H * Mq, dtype=F32)
    em.attn_fwd(c["Q"], kt, c["V"], O, lse, *meta_c, H, c["max_q"], c["causal"], c["scale"], max_k=c["max_k"], k_prescaled=True)
    delta = (c["dO"].float() * O.float()).view(Mq, H, dk).sum(-1).t().contiguous().view(-1)
    ref = [torch.zeros(Mq, d, dtype=BF16), torch.zeros(Mk, d, dtype=BF16), torch.zeros(Mk, d, dtype=BF16)]
    em.attn_bwd(c["Q"], kt, c["V"], None, c["dO"], lse, delta, *ref, *meta_c, H, c["max_q"], c["max_k"], c["causal"], c["scale"],
                k_prescaled=True)
    meta = [cu(m) for m in meta_c]
    Og, lseg = torch.zeros(Mq, d, dtype=BF16, device="cuda"), torch.zeros(H * Mq, dtype=F32, device="cuda")
    nv.attn_fwd(cu(c["Q"]), cu(kt), cu(c["V"]), Og, lseg, *meta, H, c["max_q"], c["causal"], c["scale"], max_k=c["max_k"], k_prescaled=True)
    check(Og, O, 1.5e-2, "prescaled keys %s O" % (case,))
    check(lseg, lse, 2e-3, "prescaled keys %s lse" % (case,))
    got = [torch.full((Mq, d), float("nan"), dtype=BF16, device="cuda")] + \
          [torch.full((Mk, d), float("nan"), dtype=BF16, device="cuda") for _ in range(2)]
    nv.attn_bwd(cu(c["Q"]), cu(kt), cu(c["V"]), None, cu(c["dO"]), cu(lse), cu(delta), *got, *meta, H, c["max_q"], c["max_k"], c["causal"],
                c["scale"], k_prescaled=True)
    for g_, r_, nm, tol, off, ln in zip(got, ref, ("dQ", "dK", "dV"), (2.5e-2, 2.5e-2, 2e-2), ("q_off", "k_off", "k_off"),
                                        ("q_len", "k_len", "k_len")):
        rows = torch.cat([torch.arange(int(o), int(o) + int(n)) for o, n in zip(c[off], c[ln])])
        assert torch.isfinite(g_.float().cpu()[rows]).all(), "non-finite %s %s" % (nm, case)
        check(g_.cpu()[rows], r_[rows], tol, "prescaled keys %s %s" % (case, nm))
    # ... and the result is that of the plain call on the unscaled keys, up to the keys' one extra rounding realisation
    plain = [torch.zeros(Mq, d, dtype=BF16, device="cuda")] + [torch.zeros(Mk, d, dtype=BF16, device="cuda") for _ in range(2)]
    nv.attn_bwd(cu(c["Q"]), cu(c["K"]), cu(c["V"]), None, cu(c["dO"]), cu(lse), cu(delta), *plain, *meta, H, c["max_q"], c["max_k"],
                c["causal"], c["scale"])
    for g_, p_, nm, off, ln in zip(got, plain, ("dQ", "dK", "dV"), ("q_off", "k_off", "k_off"), ("q_len", "k_len", "k_len")):
        rows = torch.cat([torch.arange(int(o), int(o) + int(n)) for o, n in zip(c[off], c[ln])])      # (padded layout: the rest is untouched)
        check(g_.cpu()[rows], p_.cpu()[rows], 3e-2, "prescaled vs plain keys %s %s" % (case, nm))
    Lq, Lk = c["max_q"], c["max_k"]
    P = nv.attn_probs(cu(c["Q"]), cu(kt), *meta, H, Lq, Lk, c["causal"], c["scale"], k_prescaled=True)
    check(P, em.attn_probs(c["Q"], kt, *meta_c, H, Lq, Lk, c["causal"], c["scale"], k_prescaled=True), 1e-4, "attn_probs, prescaled keys %s" % (case,))


def test_attention_backward_streams_stay_accurate_when_attention_is_sharp():
    """What the pre-scaled keys are for (profiles/r05_bwd64_accuracy_*.txt): with peaky attention (q, k three times larger: the
    regime of a TRAINED model) the backward streams of round 4 - register-resident operand pre-multiplied by scale * log2 e and
    re-rounded to bf16, a different perturbation of every score in each body - were 3.3x less accurate than the general kernels
    (14x at six times larger q, k).  With K~ from the producer they must be as accurate: each of dQ / dK / dV within 1.25x of
    the general kernels' error against an fp64 reference."""
    torch.manual_seed(5)
    H, dk, lens = 4, 64, [417, 300, 520, 191]
    d, M, scale = H * dk, sum(lens), 1 / math.sqrt(dk)
    qkv = torch.randn(M, 3 * d) * 0.7
    qkv[:, :2 * d] *= 3.0
    k32 = qkv[:, d:2 * d].clone()
    qkv = qkv.to(BF16)
    Q, K, V = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    kt = (k32 * (scale * nv.K_LOG2_SCALE)).to(BF16)          # one rounding, as the projection's epilogue does it
    dO = (torch.randn(M, d) * 0.5).to(BF16)
    off = [sum(lens[:i]) for i in range(len(lens))]
    ti = lambda v: torch.tensor(v, dtype=I32, device="cuda")
    meta = [ti(off), ti(lens), ti(off), ti(lens)]

    def reference(Kx, prescaled):
        out = [torch.zeros(M, d, dtype=torch.float64) for _ in range(3)]
        for o, L in zip(off, lens):
            for h in range(H):
                sl = slice(h * dk, (h + 1) * dk)
                q, k, v, do = (t[o:o + L, sl].double() for t in (Q, Kx, V, dO))
                if prescaled:
                    k = k / (scale * nv.K_LOG2_SCALE)
                p = torch.softmax(q @ k.T * scale, -1)
                dl = (do * (p @ v)).sum(-1, keepdim=True)
                ds = p * (do @ v.T - dl)
                out[0][o:o + L, sl], out[1][o:o + L, sl], out[2][o:o + L, sl] = ds @ k * scale, ds.T @ q * scale, p.T @ do
        return out

    def run(Kx, prescaled, streams):
        os.environ["ST_ATTN_BWD64"] = "1" if streams else "0"
        nv.env_refresh()
        O, Ores = (torch.empty(M, d, dtype=BF16, device="cuda") for _ in range(2))
        lse = torch.empty(H * M, dtype=F32, device="cuda")
        nv.attn_fwd(cu(Q), cu(Kx), cu(V), O, lse, *meta, H, max(lens), False, scale, max_k=max(lens), ores=Ores, k_prescaled=prescaled)
        delta = (cu(dO).float() * (O.float() + Ores.float())).view(M, H, dk).sum(-1).t().contiguous().view(-1)
        got = [torch.full((M, d), float("nan"), dtype=BF16, device="cuda") for _ in range(3)]
        nv.attn_bwd(cu(Q), cu(Kx), cu(V), None, cu(dO), lse, delta, *got, *meta, H, max(lens), max(lens), False, scale, k_prescaled=prescaled)
        ref = reference(Kx, prescaled)
        return [float((a.double().cpu() - r).norm() / r.norm()) for a, r in zip(got, ref)]

    try:
        general = run(K, False, False)
        streams = run(kt, True, True)
    finally:
        os.environ.pop("ST_ATTN_BWD64", None)
        nv.env_refresh()
    for e_s, e_g, nm in zip(streams, general, ("dQ", "dK", "dV")):
        assert e_s < 1.25 * e_g and e_s < 6e-3, "sharp attention, %s: streams %.3e vs general kernels %.3e" % (nm, e_s, e_g)


@pytest.mark.parametrize("name", ["mha_self_small", "mha_self_small_causal", "mha_cross_small"])
def test_attn_probs_kernel_vs_reference_maps(name, golden_dir):
    """st_attn_probs against the reference's own `attns` (Attention.py:89,96; fixtures generated by importing the reference:
    d_model 16, two heads of width 8 - a head width only this diagnostic kernel serves): projections in fp32 on the host,
    rounded to bf16 as the kernels keep them."""
    import numpy as np
    fx = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    H = int(fx["n_head"])
    q = torch.from_numpy(fx["q"]).float()
    kv = torch.from_numpy(fx["kv"]).float() if "kv" in fx else q
    B, Lq, d = q.shape
    Lk = kv.shape[1]
    w = lambda n: torch.from_numpy(fx["w/" + n]).float()
    Q = (q.reshape(-1, d) @ w("linear_q.weight").t() + w("linear_q.bias")).to(BF16)
    K = (kv.reshape(-1, d) @ w("linear_k.weight").t() + w("linear_k.bias")).to(BF16)
    ti = lambda v: torch.tensor(v, dtype=I32)
    q_off, k_off = ti([b * Lq for b in range(B)]), ti([b * Lk for b in range(B)])
    q_len, k_len = ti([Lq] * B), torch.from_numpy(fx["k_len"]).to(I32)       # the reference fills every query row (padding_info_mask masks keys)
    causal = bool(fx["causal"])
    P = nv.attn_probs(cu(Q), cu(K), cu(q_off), cu(q_len), cu(k_off), cu(k_len), H, Lq, Lk, causal, 1 / math.sqrt(d // H))
    ref = torch.from_numpy(fx["f64/attn"])
    assert tuple(P.shape) == tuple(ref.shape)
    check(P, ref, 1.5e-2, "attn_probs %s" % name)
    check(P, em.attn_probs(Q, K, q_off, q_len, k_off, k_len, H, Lq, Lk, causal, 1 / math.sqrt(d // H)), 1e-5, "attn_probs vs emulation %s" % name)


def test_attention_work_lists_match_plain_enumeration():
    """The longest-first work lists only reorder workgroups: results are bit-identical to the plain grid."""
    from st_amd.functional import Rows, attn_work
    lens_q, lens_k = torch.tensor([50, 33, 1, 47]), torch.tensor([1000, 517, 130, 64])
    for q_lens, k_lens, causal in ((lens_k, lens_k, False), (lens_q, lens_q, True), (lens_q, lens_k, False)):
        self_attn = q_lens is k_lens
        c = _attn_case(4, 4, 64, None if self_attn else q_lens.tolist(), k_lens.tolist(), causal, True, seed=3)
        q_rows = Rows.packed(q_lens, "cuda")
        k_rows = q_rows if self_attn else Rows.packed(k_lens, "cuda")
        wf, wq, wk = attn_work(q_rows, k_rows, causal, 64)
        outs = []
        for use in (False, True):
            Q, K, V, dO = cu(c["Q"]), cu(c["K"]), cu(c["V"]), cu(c["dO"])
            meta = [cu(c[k]) for k in ("q_off", "q_len", "k_off", "k_len")]
            O = torch.zeros(c["Mq"], c["d"], dtype=BF16, device="cuda")
            lse = torch.zeros(c["H"] * c["Mq"], dtype=F32, device="cuda")
            nv.attn_fwd(Q, K, V, O, lse, *meta, c["H"], c["max_q"], causal, c["scale"], work=wf if use else None,
                        max_k=c["max_k"])
            delta = torch.zeros_like(lse)
            dQ = torch.zeros(c["Mq"], c["d"], dtype=BF16, device="cuda")
            dK, dV = (torch.zeros(c["Mk"], c["d"], dtype=BF16, device="cuda") for _ in range(2))
            nv.attn_bwd(Q, K, V, O, dO, lse, delta, dQ, dK, dV, *meta, c["H"], c["max_q"], c["max_k"], causal,
                        c["scale"], work_q=wq if use else None, work_k=wk if use else None)
            outs.append((O, lse, dQ, dK, dV))
        for a, b, nm in zip(outs[0], outs[1], ("O", "lse", "dQ", "dK", "dV")):
            assert torch.equal(a, b), "work list changed %s (causal=%s)" % (nm, causal)


def test_attention_softmax_rescale_branch():
    """One key spiked against one query so the running max jumps in a late tile."""
    c = _attn_case(1, 1, 64, None, [300], False, True, seed=5)
    Q, K = c["Q"].clone(), c["K"].clone()
    K[250] = Q[17] * 4.0
    O = torch.zeros(300, 64, dtype=BF16)
    lse = torch.zeros(300, dtype=F32)
    meta = [c[k] for k in ("q_off", "q_len", "k_off", "k_len")]
    em.attn_fwd(Q, K, c["V"], O, lse, *meta, 1, 300, False, c["scale"])
    Og, lg = torch.zeros(300, 64, dtype=BF16, device="cuda"), torch.zeros(300, dtype=F32, device="cuda")
    nv.attn_fwd(cu(Q), cu(K), cu(c["V"]), Og, lg, *[cu(m) for m in meta], 1, 300, False, c["scale"])
    check(Og, O, 1.5e-2, "attention rescale O")
    check(lg, lse, 2e-3, "attention rescale lse")


def test_attention_long64_range_fallback():
    """The 64-rows-per-wave forward subtracts no maximum and repeats an item with the running-maximum loop when a row sum
    leaves fp32's comfort zone: force both ends - one score at ~ +260 (log2 units; exp2 overflows), one query whose
    scores are all ~ -300 (every exp2 underflows) - and a workgroup that needs no repeat next to them."""
    c = _attn_case(2, 2, 64, None, [300, 520], False, True, seed=9)
    Q, K = c["Q"].clone(), c["K"].clone()
    K[250, :64] = Q[17, :64] * 90.0            # head 0, utterance 0: one huge positive score for query 17
    K[300:820, 64] += 30.0                     # head 1, utterance 1: query 400 far below every key
    Q[300 + 400, 64] = -60.0
    Mq = c["Mq"]
    O = torch.zeros(Mq, c["d"], dtype=BF16)
    lse = torch.zeros(c["H"] * Mq, dtype=F32)
    meta = [c[k] for k in ("q_off", "q_len", "k_off", "k_len")]
    em.attn_fwd(Q, K, c["V"], O, lse, *meta, c["H"], c["max_q"], False, c["scale"])
    s17 = (Q[17, :64].float() @ K[250, :64].float()) * c["scale"] * 1.4426950408889634
    assert s17 > 150, s17
    Og, lg = torch.zeros(Mq, c["d"], dtype=BF16, device="cuda"), torch.zeros(c["H"] * Mq, dtype=F32, device="cuda")
    nv.attn_fwd(cu(Q), cu(K), cu(c["V"]), Og, lg, *[cu(m) for m in meta], c["H"], c["max_q"], False, c["scale"], max_k=c["max_k"])
    assert torch.isfinite(Og.float()).all() and torch.isfinite(lg).all()
    check(Og, O, 1.5e-2, "attention long64 fallback O")
    check(lg, lse, 2e-3, "attention long64 fallback lse")
    assert lse.min() < -150 and lse.max() > 150        # the case really leaves the plain-exponential range


# ---- streaming kernels ----------------------------------------------------------------------
def test_misc_kernels():
    B, T, Fd, D, L, V = 3, 50, 80, 128, 12, 30
    lens, tl = torch.tensor([50, 20, 33]), torch.tensor([12, 5, 9])
    off = torch.tensor([0, 50, 70], dtype=I32)
    toff = torch.tensor([0, 12, 17], dtype=I32)
    x = g(B, T, Fd, seed=1, dtype=F32)
    rows = int(lens.sum())
    for dev, mod in (("cpu", em), ("cuda", nv)):
        mv = lambda t: t.to(dev)
        out = torch.zeros(rows, Fd, dtype=BF16, device=dev)
        mod.pack_rows(mv(x), mv(off), mv(lens.to(I32)), out)
        back = torch.full((B, T, Fd), 7.0, dtype=F32, device=dev)
        mod.unpack_rows(out, mv(off), mv(lens.to(I32)), back)
        pg = torch.zeros(rows, Fd, dtype=BF16, device=dev)
        mod.pack_grad(mv(x), mv(off), mv(lens.to(I32)), pg)
        pos = torch.zeros(rows, dtype=I32, device=dev)
        mod.row_index(mv(off), mv(lens.to(I32)), 50, pos)
        tok = torch.randint(0, V, (B, L), generator=torch.Generator().manual_seed(3))
        emb, pe = g(V, D, seed=4, dtype=F32), g(L, D, seed=5, dtype=F32)
        eo = torch.zeros(int(tl.sum()), D, dtype=BF16, device=dev)
        mod.embed_pe_fwd(mv(tok), mv(emb), mv(pe), mv(toff), mv(tl.to(I32)), eo)
        demb = torch.ones(V, D, dtype=F32, device=dev)
        mod.embed_bwd(mv(tok), eo, mv(toff), mv(tl.to(I32)), 0, demb)
        xw = g(B, T, 1040, seed=2, dtype=F32)        # more float4 chunks per frame than threads in a workgroup
        outw = torch.zeros(rows, 1040, dtype=BF16, device=dev)
        mod.pack_rows(mv(xw), mv(off), mv(lens.to(I32)), outw)
        sh = torch.zeros(V * D, dtype=BF16, device=dev)
        mod.cast_bf16(mv(emb).view(-1), sh)
        res = dict(pack=out, pack_wide=outw, unpack=back, pack_grad=pg, pos=pos, embed=eo, demb=demb, cast=sh)
        if dev == "cpu":
            ref = res
    for k in ref:
        check(res[k], ref[k], 3e-3 if k == "demb" else 1e-6, "misc %s" % k)


# ---- training-mode dropout: counter-based masks, bit-identical to the emulation's hash -----------------
def _drops(salt, p, seed=1234567):
    """The same dropout call site for the kernels (device seed) and the emulation (host seed)."""
    return (nv.Drop(torch.tensor([seed], dtype=I32, device="cuda"), salt, p),
            em.Drop(torch.tensor([seed], dtype=I32), salt, p))


def _zero_pattern_equal(got, ref, what):
    """Dropped elements are exact zeros on both sides: the MASKS must agree element for element."""
    zg, zr = (got.detach().float().cpu() == 0), (ref.detach().float().cpu() == 0)
    bad = (zg != zr).sum().item()
    # an un-dropped value can round to an exact bf16 zero on one side only - vanishingly rare, never systematic
    assert bad <= 1e-5 * zg.numel(), "%s: dropout masks differ in %d of %d elements" % (what, bad, zg.numel())


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_gemm_relu_dropout_and_mask_scale(p):
    M, N, K = 777, 1024, 256
    x, W, b = g(M, K, seed=1), g(N, K, seed=2, scale=K ** -0.5), g(N, seed=3, dtype=F32)
    dn, de = _drops(11, p)
    ref = em.gemm(x, W, torch.zeros(M, N, dtype=BF16), bias=b, epi=nv.EPI_BF16_RELU, drop=de)
    out = nv.gemm(cu(x), cu(W), torch.zeros(M, N, dtype=BF16, device="cuda"), bias=cu(b), epi=nv.EPI_BF16_RELU, drop=dn)
    check(out, ref, 1e-2, "gemm relu+dropout p=%g" % p)
    _zero_pattern_equal(out, ref, "gemm relu+dropout p=%g" % p)
    kept = (out.float() != 0).float().mean().item() / (ref.float().relu() >= 0).float().mean().item()
    # backward of the same site: dgrad masked by the dropped activation, survivors scaled by 1/(1-p)
    dy, W2 = g(M, 256, seed=4), g(256, N, seed=5, scale=N ** -0.5)
    ref_d = em.gemm(dy, W2, torch.zeros(M, N, dtype=BF16), aux=ref, epi=nv.EPI_BF16_MASK, y_cmajor=True, drop=de)
    out_d = nv.gemm(cu(dy), cu(W2), torch.zeros(M, N, dtype=BF16, device="cuda"), aux=out, epi=nv.EPI_BF16_MASK,
                    y_cmajor=True, drop=dn)
    check(out_d, ref_d, 1e-2, "gemm mask+scale p=%g" % p)
    assert kept > 0


@pytest.mark.parametrize("N", [128, 256])
@pytest.mark.parametrize("where", [1, 2])
def test_gemm_ln_dropout(N, where):
    M, K, p = 500, 256 if where == 2 else 80, 0.2 if where == 2 else 0.5
    X, W = g(M, K, seed=1), g(N, K, seed=2, scale=K ** -0.5)
    b, gamma, beta = g(N, seed=3, dtype=F32), 1 + 0.2 * g(N, seed=4, dtype=F32), 0.2 * g(N, seed=5, dtype=F32)
    res = g(M, N, seed=6) if where == 2 else None
    dn, de = _drops(5 + where, p)

    def run(fn, dev, d):
        mv = (lambda t: None if t is None else t.to(dev))
        out, xhat, pre = (torch.zeros(M, N, dtype=BF16, device=dev) for _ in range(3))
        rstd = torch.zeros(M, dtype=F32, device=dev)
        fn(mv(X), mv(W), mv(b), mv(res), mv(gamma), mv(beta), out, xhat, rstd, eps=1e-6, relu=where == 1, pre=pre,
           drop=d, drop_where=where)
        return out, xhat, rstd, pre

    r, o = run(em.gemm_ln, "cpu", de), run(nv.gemm_ln, "cuda", dn)
    for got, ref, nm, tol in zip(o, r, ("out", "xhat", "rstd", "pre"), (1e-2, 1e-2, 2e-3, 1e-2)):
        check(got, ref, tol, "gemm_ln dropout where=%d N=%d %s" % (where, N, nm))
    _zero_pattern_equal(o[0] if where == 2 else o[3], r[0] if where == 2 else r[3], "gemm_ln dropout where=%d" % where)


def test_ln_bwd_dropout_and_mask_scale():
    M, N = 1000, 256
    dy, xhat = g(M, N, seed=1), g(M, N, seed=2)
    rstd, gamma = g(M, seed=3, dtype=F32).abs() + 0.5, 1 + 0.2 * g(N, seed=4, dtype=F32)
    mask = g(M, N, seed=5)
    dn, de = _drops(9, 0.2)

    def run(fn, dev, d, mk, ms):
        mv = (lambda t: None if t is None else t.to(dev))
        dx = torch.zeros(M, N, dtype=BF16, device=dev)
        acc = [torch.ones(N, dtype=F32, device=dev) for _ in range(3)]
        fn(mv(dy), mv(xhat), mv(rstd), mv(gamma), dx, acc[0], acc[1], acc[2], mask=mv(mk), drop=d, mask_scale=ms)
        return [dx] + acc

    for mk, ms, d_n, d_e, what in ((None, 1.0, dn, de, "dropout on dy"), (mask, 2.0, None, None, "mask scale")):
        r, o = run(em.ln_bwd, "cpu", d_e, mk, ms), run(nv.ln_bwd, "cuda", d_n, mk, ms)
        for got, ref, nm, tol in zip(o, r, ("dx", "dgamma", "dbeta", "dbias"), (1e-2, 3e-3, 3e-3, 5e-3)):
            check(got, ref, tol, "ln_bwd %s %s" % (what, nm))


@pytest.mark.parametrize("case", [(2, 2, 32, None, [7, 4], True, True), (3, 4, 64, None, [200, 131, 64], False, True),
                                  (2, 4, 64, [50, 33], [300, 257], False, True),
                                  (2, 4, 32, None, [129, 70], True, False),
                                  (2, 2, 128, None, [200, 131], False, True), (2, 2, 128, [50, 33], [300, 257], False, True)])
def test_attention_dropout(case):
    c = _attn_case(*case, seed=21)
    dn, de = _drops(3, 0.2)

    def run(fwd, bwd, dev, d):
        mv = lambda t: t.to(dev)
        Q, K, V, dO = mv(c["Q"]), mv(c["K"]), mv(c["V"]), mv(c["dO"])
        meta = [mv(c[k]) for k in ("q_off", "q_len", "k_off", "k_len")]
        O = torch.zeros(c["Mq"], c["d"], dtype=BF16, device=dev)
        lse = torch.zeros(c["H"] * c["Mq"], dtype=F32, device=dev)
        fwd(Q, K, V, O, lse, *meta, c["H"], c["max_q"], c["causal"], c["scale"], drop=d, max_k=c["max_k"])
        delta = torch.zeros_like(lse)
        dQ = torch.zeros(c["Mq"], c["d"], dtype=BF16, device=dev)
        dK, dV = (torch.zeros(c["Mk"], c["d"], dtype=BF16, device=dev) for _ in range(2))
        bwd(Q, K, V, O, dO, lse, delta, dQ, dK, dV, *meta, c["H"], c["max_q"], c["max_k"], c["causal"], c["scale"],
            drop=d)
        return O, lse, dQ, dK, dV

    r = run(em.attn_fwd, em.attn_bwd, "cpu", de)
    o = run(nv.attn_fwd, nv.attn_bwd, "cuda", dn)
    for got, ref, nm, tol in zip(o, r, ("O", "lse", "dQ", "dK", "dV"), (2e-2, 2e-3, 3e-2, 3e-2, 2.5e-2)):
        check(got, ref, tol, "attention dropout %s %s" % (case, nm))
    # and the masks really are drawn: the dropout-free output differs
    o0 = run(nv.attn_fwd, nv.attn_bwd, "cuda", None)
    assert rel(o[0], o0[0]) > 5e-2


def test_wgrad_group_matches_individual_launches():
    """st_wgrad_group: many weight-gradient problems (different shapes, split counts, with / without a bias
    gradient, more than one launch's worth) in one call == the same problems launched one by one."""
    shapes = [(1206, 768, 256, 4, True), (1206, 256, 256, 2, True), (1206, 1024, 256, 4, True),
              (1206, 256, 1024, 1, False), (333, 128, 80, 3, True), (50, 4344, 256, 1, False)] * 9      # 54 > GROUP_MAX
    probs, ref = [], []
    for q, (m, n, k, sp, with_b) in enumerate(shapes):
        dy, x = cu(g(m, n, seed=10 + q)), cu(g(m, k, seed=60 + q))
        init, b0 = g(n, k, seed=5, dtype=F32), g(1, n, seed=6, dtype=F32).view(-1)
        gw, gb = cu(init.clone()), (cu(b0.clone()) if with_b else None)
        probs.append((x, dy, gw, gb, sp, n))
        rw, rb = cu(init.clone()), (cu(b0.clone()) if with_b else None)
        nv.gemm(x, dy, rw, bias=rb, epi=nv.EPI_F32_ATOMIC_T, x_cmajor=True, y_cmajor=True, splits=sp, n=n)
        ref.append((rw, rb))
    nv.wgrad_group(probs)
    for q, ((_, _, gw, gb, _, _), (rw, rb)) in enumerate(zip(probs, ref)):
        check(gw, rw, 1e-5, "wgrad_group dW problem %d" % q)       # same kernel body; only the atomic order differs
        if gb is not None:
            check(gb, rb, 1e-5, "wgrad_group db problem %d" % q)


def test_wgrad_wide_matches_fp32():
    """st_wgrad_wide (256 x 256 tiles, encoder-sized token counts): dW += dY^T X and db += colsum(dY) against fp32 matmuls
    of the same bf16 operands - shapes with partial tiles on every axis, 1 .. 7 token splits (also more splits than
    k-tiles), with / without a bias gradient, more than one launch's worth of problems, accumulation on top of existing
    gradients."""
    shapes = [(9000, 768, 256, 3, True), (8200, 256, 256, 7, True), (8300, 1024, 256, 2, True),
              (8192, 256, 1024, 1, False), (333, 296, 200, 3, True), (50, 520, 256, 5, False),
              (1000, 264, 72, 4, True)] * 7      # 49 > WIDE_MAX
    probs, ref = [], []
    for q, (m, n, k, sp, with_b) in enumerate(shapes):
        dy, x = g(m, n, seed=10 + q), g(m, k, seed=60 + q)
        init, b0 = g(n, k, seed=5, dtype=F32), g(1, n, seed=6, dtype=F32).view(-1)
        probs.append((cu(x), cu(dy), cu(init.clone()), cu(b0.clone()) if with_b else None, sp, n))
        ref.append((init + dy.float().t() @ x.float(), b0 + dy.float().sum(0)))
    nv.wgrad_group(probs, wide=True)
    for q, ((_, _, gw, gb, _, _), (rw, rb)) in enumerate(zip(probs, ref)):
        check(gw, rw, 2e-5, "wgrad_wide dW problem %d" % q)
        if gb is not None:
            check(gb, rb, 2e-5, "wgrad_wide db problem %d" % q)


def test_wgrad_wide_at_decoder_side_shapes_through_the_plan():
    """Since round 5 problems from 2,048 token rows up take st_wgrad_wide (_Deferred.WIDE_ROWS): the decoder side's ragged vocabulary
    projection (N = 4,344 rows of the weight: 16 full tiles + a 248-row one) and a wide contraction (K_in = 2,048) at 2-4 k token
    rows, planned by functional._wide_plan as the step plans them (the wide kernel was only tested at encoder shapes)."""
    from st_amd.functional import _wide_plan
    shapes = [(2304, 4344, 256, True), (3100, 256, 2048, True), (4000, 4344, 256, False), (2048, 1032, 2048, True)]
    probs, ref = [], []
    for q, (m, n, k, with_b) in enumerate(shapes):
        dy, x = g(m, n, seed=110 + q), g(m, k, seed=160 + q)
        init, b0 = g(n, k, seed=7, dtype=F32), g(1, n, seed=8, dtype=F32).view(-1)
        probs.append((cu(x), cu(dy), cu(init.clone()), cu(b0.clone()) if with_b else None, 1, n))
        ref.append((init + dy.float().t() @ x.float(), b0 + dy.float().sum(0)))
    launches = _wide_plan(probs)
    assert sum(len(l) for l in launches) == len(probs) and all(p[4] >= 1 for l in launches for p in l)
    for l in launches:
        nv.wgrad_group(l, wide=True)
    for q, ((_, _, gw, gb, _, _), (rw, rb)) in enumerate(zip(probs, ref)):
        check(gw, rw, 2e-5, "wgrad_wide (planned) dW problem %d" % q)
        if gb is not None:
            check(gb, rb, 2e-5, "wgrad_wide (planned) db problem %d" % q)


def test_feat_stack_kernel():
    """st_feat_stack (CMVN + frame stacking + subsampling + ragged pack) against the oracle restatement of Dataset.py."""
    from tests import test_features_cpu as tf
    tf.run_stack_frames("cuda")


@pytest.mark.parametrize("M,N,K,hd", [(333, 256, 256, 64), (1000, 128, 128, 32), (130, 512, 512, 64), (260, 512, 512, 128)])
def test_gemm_dgrad_with_delta_epilogue(M, N, K, hd):
    """ST_EPI_BF16_DELTA: the dgrad that produces d(context) also emits delta[h][i] = rowsum_h(d(context) * context)."""
    dy, W, O = g(M, K, seed=1), g(K, N, seed=2, scale=K ** -0.5), g(M, N, seed=3)
    H = N // hd
    ref_d = torch.zeros(H * M, dtype=F32)
    ref = em.gemm(dy, W, torch.zeros(M, N, dtype=BF16), aux=O, epi=nv.EPI_BF16_DELTA, y_cmajor=True, delta=ref_d,
                  head_dim=hd)
    got_d = torch.full((H * M,), float("nan"), dtype=F32, device="cuda")
    got = nv.gemm(cu(dy), cu(W), torch.zeros(M, N, dtype=BF16, device="cuda"), aux=cu(O), epi=nv.EPI_BF16_DELTA,
                  y_cmajor=True, delta=got_d, head_dim=hd)
    check(got, ref, 1e-2, "dgrad+delta out")
    # delta is a sum of products of bf16 values: compare against the kernel's OWN bf16 output to isolate the reduction
    own = (got.float().cpu() * O.float()).view(M, H, hd).sum(-1).t().reshape(-1)
    check(got_d, own, 1e-5, "dgrad+delta delta")


@pytest.mark.parametrize("case", [(3, 4, 64, None, [200, 131, 64], False, True), (2, 4, 64, [50, 33], [1000, 517], False, True),
                                  (2, 4, 32, None, [129, 70], True, True), (2, 2, 128, None, [300, 131], False, True),
                                  (2, 2, 128, [50, 33], [700, 517], False, True)])
def test_attention_backward_single_launch(case):
    """O = None: delta comes in precomputed and dQ + dK/dV run as one launch - identical results to the two-kernel path
    where the general kernels serve both; long non-causal problems with 64-wide heads take the hand-scheduled streams of
    csrc/st_attn_bwd64.hip in the one-launch form (K or Q pre-multiplied by scale * log2 e and rounded to bf16 once more:
    the same mathematics within a bf16 rounding of the scores, not bit-identical)."""
    c = _attn_case(*case, seed=31)
    Q, K, V, dO = cu(c["Q"]), cu(c["K"]), cu(c["V"]), cu(c["dO"])
    meta = [cu(c[k]) for k in ("q_off", "q_len", "k_off", "k_len")]
    O = torch.zeros(c["Mq"], c["d"], dtype=BF16, device="cuda")
    lse = torch.zeros(c["H"] * c["Mq"], dtype=F32, device="cuda")
    nv.attn_fwd(Q, K, V, O, lse, *meta, c["H"], c["max_q"], c["causal"], c["scale"], max_k=c["max_k"])
    outs = []
    for single in (False, True):
        delta = torch.zeros_like(lse)
        dQ = torch.zeros(c["Mq"], c["d"], dtype=BF16, device="cuda")
        dK, dV = (torch.zeros(c["Mk"], c["d"], dtype=BF16, device="cuda") for _ in range(2))
        if single:
            delta = outs[0][3].clone()
        nv.attn_bwd(Q, K, V, None if single else O, dO, lse, delta, dQ, dK, dV, *meta, c["H"], c["max_q"], c["max_k"],
                    c["causal"], c["scale"])
        outs.append((dQ, dK, dV, delta))
    streams = case[2] == 64 and not case[5] and min(c["max_q"], c["max_k"]) > 128
    # (few queries against many keys: the one-launch form cuts the dQ items' keys over workgroups - other fp32 partial sums, round 6)
    split = nv.load()._cdll.st_attn_bwd_split_kib(case[0], c["H"], case[2], c["max_q"], c["max_k"], int(case[5])) > 0
    for a, b, nm in zip(outs[0][:3], outs[1][:3], ("dQ", "dK", "dV")):
        if streams:
            check(b, a, 6e-3, "single-launch backward (hand-scheduled streams) %s" % nm)
        elif split and nm == "dQ":
            check(b, a, 6e-3, "single-launch backward (keys split across workgroups) %s" % nm)
        else:
            assert torch.equal(a, b), "single-launch backward changed %s" % nm


@pytest.mark.parametrize("lens,p", [([200, 131], 0.1), ([129, 130, 257], 0.1), ([1000, 640], 0.1), ([65, 63, 64, 128, 192], 0.1),
                                    ([300, 257], 0.5), ([520, 191, 333], 0.25)])
def test_attention_backward_streams_with_dropout(lens, p, monkeypatch):
    """Training mode (Attention.py:89): the dropout variant of the hand-scheduled backward streams (csrc/st_attn_bwd64.hip,
    *_drop.inc) regenerates the forward's keep decisions inside the instruction stream (SDWA byte compares on the lowbias32
    words of st_attn_common.cuh keep16, one DPP exchange per 2 x 2 block pair).  Against the general kernels with the SAME
    Drop (ST_ATTN_BWD64=e keeps the streams for eval mode only): a single wrong keep decision is an O(1 / sqrt(n)) error, the
    two agree to the bf16 rounding of the pre-scaled operand (measured 2.8e-3)."""
    from st_amd.functional import Rows, attn_work
    H, dk = 4, 64
    d, scale = H * dk, 1 / math.sqrt(dk)
    lens_t = torch.tensor(lens)
    M = int(lens_t.sum())
    qkv, dO = cu(g(M, 3 * d, seed=5, scale=0.7)), cu(g(M, d, seed=6, scale=0.5))
    Q, K, V = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    rows = Rows.packed(lens_t, "cuda")
    wf, wq, wk = attn_work(rows, rows, False, dk, H)
    off = torch.zeros_like(lens_t)
    off[1:] = torch.cumsum(lens_t, 0)[:-1]
    q_off, q_len = off.to("cuda", torch.int32), lens_t.to("cuda", torch.int32)
    drop = nv.Drop(torch.tensor([4321], dtype=torch.int32, device="cuda"), 55, p)
    O = torch.empty(M, d, dtype=BF16, device="cuda")
    lse = torch.empty(H * M, dtype=F32, device="cuda")
    nv.attn_fwd(Q, K, V, O, lse, q_off, q_len, q_off, q_len, H, max(lens), False, scale, work=wf, max_k=max(lens), drop=drop)
    delta = (dO.float() * O.float()).view(M, H, dk).sum(-1).t().contiguous().view(-1)
    outs = {}
    for mode in ("e", "1"):
        monkeypatch.setenv("ST_ATTN_BWD64", mode)
        nv.env_refresh()          # (the library caches its development switches)
        got = [torch.full((M, d), float("nan"), dtype=BF16, device="cuda") for _ in range(3)]
        nv.attn_bwd(Q, K, V, None, dO, lse, delta, *got, q_off, q_len, q_off, q_len, H, max(lens), max(lens), False, scale,
                    work_q=wq, work_k=wk, drop=drop)
        torch.cuda.synchronize()
        outs[mode] = got
    for a, b, nm in zip(outs["e"], outs["1"], ("dQ", "dK", "dV")):
        assert torch.isfinite(b.float()).all(), nm
        check(b, a, 6e-3, "backward streams with dropout p = %s: %s" % (p, nm))
    # and the masks matter: without them the result is far away (guards against a variant that silently ignores the Drop)
    monkeypatch.delenv("ST_ATTN_BWD64")
    nv.env_refresh()
    plain = [torch.zeros(M, d, dtype=BF16, device="cuda") for _ in range(3)]
    nv.attn_bwd(Q, K, V, None, dO, lse, delta, *plain, q_off, q_len, q_off, q_len, H, max(lens), max(lens), False, scale, work_q=wq, work_k=wk)
    assert float((plain[2].float() - outs["1"][2].float()).norm() / outs["1"][2].float().norm()) > 0.1


@pytest.mark.parametrize("M,N,K,with_aux,p", [(300, 128, 128, True, 0), (1000, 256, 1024, True, 0), (130, 256, 768, False, 0),
                                               (70, 512, 512, True, 0), (999, 256, 256, True, 0),
                                               (999, 256, 768, True, 0.1), (130, 128, 256, True, 0.3), (200, 512, 512, False, 0.2),
                                               (16500, 256, 768, True, 0), (16450, 256, 512, True, 0.1),       # 8-wave 128-row tiles
                                               (8250, 512, 512, True, 0), (8230, 512, 1024, True, 0.1)])        # d_model 512: 8-wave 64-row tiles
def test_gemm_lnbwd(M, N, K, with_aux, p):
    """st_gemm_lnbwd == st_gemm (dgrad, + aux) followed by st_ln_bwd, in one launch; p > 0: the LayerNorm output was
    dropped in the forward (the mask is regenerated from the same counters as st_ln_bwd's)."""
    dn, de = _drops(11, p) if p else (None, None)
    dY, W = g(M, K, seed=1), g(K, N, seed=2, scale=K ** -0.5)
    aux = g(M, N, seed=3) if with_aux else None
    xhat, rstd, gamma = g(M, N, seed=4), g(M, seed=5, dtype=F32).abs() + 0.5, 1 + 0.2 * g(N, seed=6, dtype=F32)

    def run(fn, dev, d):
        mv = (lambda t: None if t is None else t.to(dev))
        dx = torch.zeros(M, N, dtype=BF16, device=dev)
        acc = [torch.ones(N, dtype=F32, device=dev) for _ in range(3)]
        fn(mv(dY), mv(W), mv(aux), mv(xhat), mv(rstd), mv(gamma), dx, acc[0], acc[1], acc[2], drop=d)
        return [dx] + acc

    r, o = run(em.gemm_lnbwd, "cpu", de), run(nv.gemm_lnbwd, "cuda", dn)
    for got, ref, nm, tol in zip(o, r, ("dx", "dgamma", "dbeta", "dbias"), (1.5e-2, 5e-3, 5e-3, 8e-3)):
        check(got, ref, tol, "gemm_lnbwd %s %s" % ((M, N, K, with_aux, p), nm))


@pytest.mark.parametrize("M,blocks,rows,K", [(700, 3, 256, 128), (1206, 6, 512, 256), (77, 2, 128, 80)])
def test_gemm_stacked_weights(M, blocks, rows, K):
    """st_gemm_stacked: the Y operand (and bias) is `blocks` equally spaced [rows, K] blocks inside a larger buffer
    (how the parameter arena holds the same weight of consecutive layers) == the same GEMMs on the gathered stack,
    forward (more output columns) and dgrad (longer contraction)."""
    ldy = (K + 7) // 8 * 8
    w_stride, b_stride = rows * ldy + 4096 + 64, rows + 192
    arena_w = cu(g(1, blocks * w_stride + 128, seed=1, scale=K ** -0.5).view(-1))
    arena_b = cu(g(1, blocks * b_stride + 64, seed=2, dtype=F32).view(-1))
    W0 = torch.as_strided(arena_w, (rows, K), (ldy, 1), 64)                 # block 0 starts 64 elements in
    b0 = arena_b[32:32 + rows]
    Wcat = torch.cat([torch.as_strided(arena_w, (rows, K), (ldy, 1), 64 + l * w_stride) for l in range(blocks)]).contiguous()
    bcat = torch.cat([arena_b[32 + l * b_stride:32 + l * b_stride + rows] for l in range(blocks)]).contiguous()
    x = cu(g(M, K, seed=3))
    Kp = ldy
    xp = torch.zeros(M, Kp, dtype=BF16, device="cuda")
    xp[:, :K] = x
    out, ref = torch.empty(M, blocks * rows, dtype=BF16, device="cuda"), torch.empty(M, blocks * rows, dtype=BF16, device="cuda")
    nv.gemm(xp[:, :K], W0, out, bias=b0, stack=(blocks, w_stride, b_stride))
    wc = torch.zeros(blocks * rows, Kp, dtype=BF16, device="cuda")
    wc[:, :K] = Wcat
    nv.gemm(xp[:, :K], wc[:, :K], ref, bias=bcat)
    assert torch.equal(out, ref), "stacked forward differs from the gathered GEMM"
    # dgrad: dx[M, K] = dy[M, blocks * rows] @ stack (+ aux)
    if K % 8 == 0:
        dy, aux = cu(g(M, blocks * rows, seed=4)), cu(g(M, K, seed=5))
        dx, dref = torch.empty(M, K, dtype=BF16, device="cuda"), torch.empty(M, K, dtype=BF16, device="cuda")
        nv.gemm(dy, W0, dx, y_cmajor=True, stack=(blocks, w_stride, 0), epi=nv.EPI_BF16_ADD, aux=aux)
        nv.gemm(dy, Wcat, dref, y_cmajor=True, epi=nv.EPI_BF16_ADD, aux=aux)
        assert torch.equal(dx, dref), "stacked dgrad differs from the gathered GEMM"


@pytest.mark.parametrize("M,d,K", [(300, 128, 128), (24060, 512, 512), (777, 256, 256)])
def test_gemm_kscale_scales_the_key_block_before_its_rounding(M, d, K):
    """st_gemm_kscale: a q | k | v projection whose KEY block leaves multiplied by scale * log2(e) in fp32, rounded once (the
    per-GEMM path's form of st_row_chain's post_kscale); the other two blocks are st_gemm's, bit for bit."""
    x, W, b = g(M, K, seed=1), g(3 * d, K, seed=2, scale=K ** -0.5), g(3 * d, seed=3, dtype=F32)
    ks = 0.125 * nv.K_LOG2_SCALE
    got = nv.gemm_kscale(cu(x), cu(W), torch.full((M, 3 * d), float("nan"), dtype=BF16, device="cuda"), cu(b), d, 2 * d, ks)
    plain = nv.gemm(cu(x), cu(W), torch.zeros(M, 3 * d, dtype=BF16, device="cuda"), bias=cu(b))
    assert torch.equal(got[:, :d], plain[:, :d]) and torch.equal(got[:, 2 * d:], plain[:, 2 * d:])
    ref = em.gemm_kscale(x, W, torch.zeros(M, 3 * d, dtype=BF16), b, d, 2 * d, ks)
    check(got, ref, 1e-2, "gemm_kscale")
    acc = (x.float() @ W.float().t() + b)[:, d:2 * d] * ks
    err_once = (got[:, d:2 * d].float().cpu() - acc).norm() / acc.norm()
    err_twice = ((plain[:, d:2 * d].float().cpu() * ks).to(BF16).float() - acc).norm() / acc.norm()
    assert err_once < 0.8 * err_twice, (float(err_once), float(err_twice))      # one rounding, not two


def test_adam_clip_matches_torch_clip_and_fused_adam():
    """st_adam_clip == clip_grad_norm_ (global norm over the flat buffer) followed by torch.optim.Adam(fused, capturable)
    with a device learning-rate tensor, over several steps with a changing rate (the Noam schedule) and gradients both
    above and below the clipping threshold."""
    n, max_norm = 40000, 5.0
    torch.manual_seed(3)
    p0 = torch.randn(n, device="cuda")
    pa, pb = p0.clone(), torch.nn.Parameter(p0.clone())
    lr_t = torch.zeros((), device="cuda")
    opt = torch.optim.Adam([pb], lr=lr_t, betas=(0.9, 0.98), eps=1e-9, fused=True, capturable=True)
    m, v, step = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros((), device="cuda")
    for it in range(6):
        g = torch.randn(n, device="cuda") * (0.2 if it % 2 else 0.002)        # norm 40 (clipped) / 0.4 (not clipped)
        lr_t.fill_(1e-3 * (it + 1))
        ga = g.clone()
        gnorm = torch.linalg.vector_norm(ga)
        step.add_(1)
        nv.adam_clip(pa, ga, m, v, lr_t, step, gnorm, max_norm, 0.9, 0.98, 1e-9)
        pb.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([pb], max_norm)
        opt.step()
        check(ga, pb.grad, 1e-6, "adam_clip clipped gradient, step %d" % it)
        check(pa, pb.detach(), 2e-6, "adam_clip parameters, step %d" % it)
    st = opt.state[pb]
    check(m, st["exp_avg"], 1e-6, "adam_clip exp_avg")
    check(v, st["exp_avg_sq"], 1e-6, "adam_clip exp_avg_sq")
    assert float(step) == float(st["step"])


def test_grad_scale_folds_the_rank_average_into_norm_clip_and_adam():
    """grad_scale (st_grad_norm, st_adam_clip): a buffer that holds world x the gradient (what a SUMMING all-reduce leaves,
    st_amd.dp.GradReducer.synchronize(divide=False)) updated with grad_scale = 1 / world == the divided buffer updated with
    grad_scale 1: the norm, the clipped gradient left in the buffer, the parameters and both moments."""
    n, max_norm, world = 40000, 5.0, 8
    torch.manual_seed(5)
    p0 = torch.randn(n, device="cuda")
    lr_t = torch.full((), 2e-3, device="cuda")
    for gs in (0.2, 0.002):            # clipped / not clipped
        g = torch.randn(n, device="cuda") * gs
        res = []
        for buf, scale in ((g * world, 1.0 / world), (g.clone(), 1.0)):
            p, m, v, step = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros((), device="cuda")
            gn = nv.grad_norm(buf, nv.grad_norm_scratch("cuda"), torch.empty((), device="cuda"), step=step, grad_scale=scale)
            nv.adam_clip(p, buf, m, v, lr_t, step, gn, max_norm, 0.9, 0.98, 1e-9, grad_scale=scale)
            res.append((gn.clone(), buf, p, m, v))
        assert abs(float(res[0][0]) - float(torch.linalg.vector_norm(g.double()))) < 1e-5 * float(res[0][0])
        for a, b, what in zip(res[0], res[1], ("norm", "gradient left in the buffer", "parameters", "exp_avg", "exp_avg_sq")):
            check(a, b, 1e-6, "grad_scale: " + what)


def test_zero_tails_zeroes_exactly_the_unassigned_rows():
    """st_zero_tails: rows [valid, capacity) of every listed matrix become zero, nothing else is touched; the valid-row counts
    are read from device memory at launch time (a captured launch follows the batch); entries with a null address are
    skipped."""
    dev = "cuda"
    valid_a, valid_b = torch.tensor([37], dtype=I32, device=dev), torch.tensor([0], dtype=I32, device=dev)
    bufs = [(torch.full((100, 256), 3.0, dtype=BF16, device=dev), valid_a), (torch.full((64, 768), 5.0, dtype=BF16, device=dev), valid_b),
            (torch.full((100, 8), 7.0, dtype=BF16, device=dev), valid_a)]
    rows = []
    for i, (t, v) in enumerate(bufs):
        if i == 2:
            rows += [0, 0, 0, 0]                        # an unused slot in the middle of the table
        rows += [t.data_ptr(), t.shape[1] * 2, t.shape[0], v.data_ptr()]
    table = torch.tensor(rows + [0] * (4 * 8 - len(rows)), dtype=torch.int64, device=dev)
    nv.zero_tails(table, 8)
    for (t, v), fill in zip(bufs, (3.0, 5.0, 7.0)):
        k = int(v)
        assert bool((t[:k].float() == fill).all()), "a valid row was touched"
        assert bool((t[k:].float() == 0).all()), "an unassigned row survived"
    valid_a.fill_(100)                      # nothing unassigned: untouched
    bufs[0][0].fill_(2.0)
    nv.zero_tails(table, 8)
    assert bool((bufs[0][0].float() == 2.0).all())


@pytest.mark.parametrize("n", [4, 1000, 262144 + 8, 13_000_004])
def test_grad_norm_matches_torch_and_advances_the_step(n):
    """st_grad_norm == torch.linalg.vector_norm over the flat gradient buffer (fp64 reference), repeated launches on the same
    scratch (the ticket resets itself), the optional step counter advanced by one per launch; bit-identical from launch to
    launch (the partials are added in index order, not in arrival order)."""
    torch.manual_seed(n % 1000)
    gbuf = torch.randn(n, device="cuda") * 0.3
    scratch = nv.grad_norm_scratch("cuda")
    out, step = torch.zeros((), device="cuda"), torch.full((), 41.0, device="cuda")
    seen = []
    for it in range(3):
        nv.grad_norm(gbuf, scratch, out, step=step if it else None)
        seen.append(float(out))
        ref = float(torch.linalg.vector_norm(gbuf.double()))
        assert abs(seen[-1] - ref) <= 2e-6 * ref, (n, it, seen[-1], ref)
    assert seen[0] == seen[1] == seen[2] and float(step) == 43.0
    assert float(scratch[-1].view(torch.int32)) == 0
    gbuf.zero_()
    assert float(nv.grad_norm(gbuf, scratch, out)) == 0.0


# ---- st_gemm_ws: the weight-stationary streaming GEMM -------------------------------------------------------
@pytest.mark.parametrize("M,N", [(32, 256), (33, 256), (1206, 768), (4097, 1024), (24060, 768), (13000, 256)])
@pytest.mark.parametrize("relu", [False, True])
def test_gemm_ws_forward(M, N, relu):
    """K = 256, N a multiple of 256; ragged last chunk (M % 32 != 0), one chunk, many chunks per workgroup, strided
    input / output views (column slices of wider buffers)."""
    K = 256
    x, W, b = g(M, K, seed=1), g(N, K, seed=2, scale=K ** -0.5), g(N, seed=3, dtype=F32)
    ref = em.gemm(x, W, torch.zeros(M, N, dtype=BF16), bias=b, epi=nv.EPI_BF16_RELU if relu else nv.EPI_BF16)
    xw = torch.zeros(M, K + 64, dtype=BF16, device="cuda")
    xw[:, 8:8 + K] = cu(x)
    ow = torch.full((M + 3, N + 128), float("nan"), dtype=BF16, device="cuda")
    out = nv.gemm_ws(xw[:, 8:8 + K], cu(W), ow[:M, 64:64 + N], bias=cu(b), relu=relu)
    check(out, ref, 1e-2, "gemm_ws %dx%d relu=%d" % (M, N, relu))
    assert torch.isnan(ow[M:].float()).all() and torch.isnan(ow[:, :64].float()).all() and torch.isnan(ow[:, 64 + N:].float()).all(), \
        "gemm_ws wrote outside its output view"
    nb = nv.gemm_ws(cu(x), cu(W), torch.empty(M, N, dtype=BF16, device="cuda"), relu=relu)       # no bias
    check(nb, em.gemm(x, W, torch.zeros(M, N, dtype=BF16), epi=nv.EPI_BF16_RELU if relu else nv.EPI_BF16), 1e-2, "gemm_ws no bias")


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_gemm_ws_relu_dropout_mask(p):
    """The ReLU + dropout epilogue draws the SAME counter-based mask as st_gemm's (the backward regenerates it there)."""
    M, N, K = 13001, 1024, 256
    x, W, b = g(M, K, seed=1), g(N, K, seed=2, scale=K ** -0.5), g(N, seed=3, dtype=F32)
    dn, de = _drops(11, p)
    ref = em.gemm(x, W, torch.zeros(M, N, dtype=BF16), bias=b, epi=nv.EPI_BF16_RELU, drop=de)
    out = nv.gemm_ws(cu(x), cu(W), torch.zeros(M, N, dtype=BF16, device="cuda"), bias=cu(b), relu=True, drop=dn)
    check(out, ref, 1e-2, "gemm_ws relu+dropout p=%g" % p)
    _zero_pattern_equal(out, ref, "gemm_ws relu+dropout p=%g" % p)
    tiled = nv.gemm(cu(x), cu(W), torch.zeros(M, N, dtype=BF16, device="cuda"), bias=cu(b), epi=nv.EPI_BF16_RELU, drop=dn)
    _zero_pattern_equal(out, tiled, "gemm_ws vs st_gemm dropout mask")


def test_last_arriver_merges_under_uneven_load():
    """The in-launch merges (st_gemm_splitk: fp32 partial tiles of the K splits; st_grad_norm: per-workgroup partial sums) publish
    with write-through stores + a drained vmcnt + a device-scope ticket and read with device-scope loads (the `sc1` store / `sc1`
    load form of MI355X_MICROARCH.md, "Valid forms") - no L2 write-back fence.  A stale partial would be SILENT, and idle chips
    hide such races: run both 300 times while a second stream streams 1 GiB copies through every XCD's L2, with inputs that
    change every iteration (a merge that picked up the previous iteration's partial is then wrong, not accidentally right),
    and require every result to equal the reference computed WITHOUT an in-launch merge (st_gemm's own product for the
    split-K GEMM, bit for bit reproducible between two runs; a float64 norm for st_grad_norm)."""
    M, N, K, splits = 1206, 256, 4344, 6
    side = torch.cuda.Stream()
    a, b = torch.empty(1 << 28, dtype=torch.float32, device="cuda"), torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    stop = torch.zeros(1, device="cuda")
    scratch, out = nv.grad_norm_scratch("cuda"), torch.zeros((), device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(5)
    bad = []
    for it in range(300):
        if it % 4 == 0:                                    # keep the memory system busy and the load uneven
            with torch.cuda.stream(side):
                b.copy_(a)
        x = (torch.randn(M, K, device="cuda", generator=gen) * 0.5).to(BF16)
        w = (torch.randn(K, N, device="cuda", generator=gen) * K ** -0.5).to(BF16)
        o1 = nv.gemm_splitk(x, w, torch.full((M, N), float("nan"), dtype=BF16, device="cuda"), splits, y_cmajor=True)
        o2 = nv.gemm_splitk(x, w, torch.full((M, N), float("nan"), dtype=BF16, device="cuda"), splits, y_cmajor=True)
        ref = (x.float() @ w.float())
        gbuf = torch.randn(3_000_000 + 4 * it, device="cuda", generator=gen)
        nrm = float(nv.grad_norm(gbuf, scratch, out))
        if not torch.equal(o1, o2) or rel(o1, ref) > 1e-2 or abs(nrm - float(gbuf.double().norm())) > 1e-5 * nrm:
            bad.append(it)
    torch.cuda.synchronize()
    assert not bad, "stale or torn partials in iterations %s" % bad[:10]
    for wk in nv._splitk_work.values():
        assert int(wk[:1024].abs().sum()) == 0, "split-K tickets not reset"


def test_attention_key_split_merge_under_uneven_load():
    """The decoder-encoder attention backward's dQ key split (round 6) publishes fp32 partials the same way (write-through stores,
    drained vmcnt, device-scope ticket, device-scope loads): 200 launches on inputs that change every iteration, a second stream
    pushing copies through every XCD's L2 - the merged launch must equal the two-launch form (which does not split) to bf16
    rounding EVERY time, and two merged launches must agree bit for bit."""
    c = _attn_case(4, 4, 64, [38, 20, 64, 45], [900, 640, 511, 384], False, True, seed=3)
    H, Mq, dk = c["H"], c["Mq"], c["d"] // c["H"]
    meta = [cu(c[k]) for k in ("q_off", "q_len", "k_off", "k_len")]
    side = torch.cuda.Stream()
    a, b = torch.empty(1 << 27, dtype=torch.float32, device="cuda"), torch.empty(1 << 27, dtype=torch.float32, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(11)
    O, lse = torch.zeros(Mq, c["d"], dtype=BF16, device="cuda"), torch.zeros(H * Mq, dtype=F32, device="cuda")
    bad = []
    for it in range(200):
        if it % 4 == 0:
            with torch.cuda.stream(side):
                b.copy_(a)
        Q = (torch.randn(Mq, c["d"], device="cuda", generator=gen) * 0.7).to(BF16)
        KV = (torch.randn(c["Mk"], 2 * c["d"], device="cuda", generator=gen) * 0.7).to(BF16)
        K, V = KV[:, :c["d"]], KV[:, c["d"]:]
        dO = (torch.randn(Mq, c["d"], device="cuda", generator=gen) * 0.5).to(BF16)
        nv.attn_fwd(Q, K, V, O, lse, *meta, H, c["max_q"], False, c["scale"], max_k=c["max_k"])
        delta = (dO.float() * O.float()).view(Mq, H, dk).sum(-1).t().contiguous().view(-1)
        outs = []
        for parts_seq in ((3,), (3,), (1, 2)):
            o = [torch.full((Mq, c["d"]), float("nan"), dtype=BF16, device="cuda")] + \
                [torch.full((c["Mk"], c["d"]), float("nan"), dtype=BF16, device="cuda") for _ in range(2)]
            for parts in parts_seq:
                nv.attn_bwd(Q, K, V, None, dO, lse, delta, *o, *meta, H, c["max_q"], c["max_k"], False, c["scale"], parts=parts)
            outs.append(o)
        if not torch.equal(outs[0][0], outs[1][0]) or not bool(torch.isfinite(outs[0][0].float()).all()) or rel(outs[0][0], outs[2][0].float()) > 8e-3:
            bad.append(it)
    torch.cuda.synchronize()
    assert not bad, "stale or torn dQ partials in iterations %s" % bad[:10]
    work = nv._ATTN_SPLIT_WORK[torch.device("cuda", torch.cuda.current_device()).index]
    assert int(work[:4096].abs().sum()) == 0, "tickets not reset"


def test_gemm_ws_stacked_weights():
    """Stacked weights (the decoder-encoder K/V projections of all layers, read in place from the arena) == the
    gathered GEMM, and == st_gemm_stacked."""
    M, blocks, rows, K = 13000, 6, 512, 256
    w_stride, b_stride = rows * K + 4096 + 64, rows + 192
    arena_w = cu(g(1, blocks * w_stride + 128, seed=1, scale=K ** -0.5).view(-1))
    arena_b = cu(g(1, blocks * b_stride + 64, seed=2, dtype=F32).view(-1))
    W0 = torch.as_strided(arena_w, (rows, K), (K, 1), 64)
    b0 = arena_b[32:32 + rows]
    Wcat = torch.cat([torch.as_strided(arena_w, (rows, K), (K, 1), 64 + l * w_stride) for l in range(blocks)]).contiguous()
    bcat = torch.cat([arena_b[32 + l * b_stride:32 + l * b_stride + rows] for l in range(blocks)]).contiguous()
    x = cu(g(M, K, seed=3))
    out = nv.gemm_ws(x, W0, torch.empty(M, blocks * rows, dtype=BF16, device="cuda"), bias=b0, stack=(blocks, w_stride, b_stride))
    ref = nv.gemm_ws(x, Wcat, torch.empty(M, blocks * rows, dtype=BF16, device="cuda"), bias=bcat)
    assert torch.equal(out, ref), "stacked gemm_ws differs from the gathered one"
    tiled = nv.gemm(x, W0, torch.empty(M, blocks * rows, dtype=BF16, device="cuda"), bias=b0, stack=(blocks, w_stride, b_stride))
    check(out, tiled, 5e-3, "gemm_ws vs st_gemm_stacked")


@pytest.mark.parametrize("M,N,K,splits,ycm", [(1206, 256, 4344, 8, True), (5, 256, 2104, 3, True), (300, 384, 4096, 5, False),
                                              (129, 128, 96, 2, True)])
def test_gemm_splitk_matches_gemm(M, N, K, splits, ycm):
    """st_gemm_splitk (contraction cut over workgroups, fp32 partial tiles merged by the last arriver of each output tile, in
    split order) == st_gemm's product (fp32 summation order differs: bf16 results within one rounding of each other and of
    the fp32 reference), bit-reproducible from launch to launch, tickets back at zero; ragged M, K not a multiple of the
    k-tile, more splits than k-tiles allow."""
    x = g(M, K, seed=1)
    w = g(K, N, seed=2, scale=K ** -0.5) if ycm else g(N, K, seed=2, scale=K ** -0.5)
    ref = em.gemm(x, w, torch.zeros(M, N, dtype=BF16), y_cmajor=ycm)
    plain = nv.gemm(cu(x), cu(w), torch.zeros(M, N, dtype=BF16, device="cuda"), y_cmajor=ycm)
    a = nv.gemm_splitk(cu(x), cu(w), torch.full((M, N), float("nan"), dtype=BF16, device="cuda"), splits, y_cmajor=ycm)
    b = nv.gemm_splitk(cu(x), cu(w), torch.full((M, N), float("nan"), dtype=BF16, device="cuda"), splits, y_cmajor=ycm)
    check(a, ref, 1e-2, "gemm_splitk vs fp32 reference")
    check(a, plain, 1e-2, "gemm_splitk vs st_gemm")
    assert torch.equal(a, b), "gemm_splitk not reproducible"
    for wk in nv._splitk_work.values():
        assert int(wk[:1024].abs().sum()) == 0, "split-K tickets not reset"


# ---- row chains (csrc/st_rowchain.hip) -----------------------------------------------------------------------------
@pytest.mark.parametrize("M", [5, 320, 1206, 3120, 9000, 17000, 24700])      # 3120: 98 row blocks - a split chain takes two chunks per part; 24700: past one round of 96-row tiles -> two rounds of 64-row ones
@pytest.mark.parametrize("variant", ["pre+post1", "pre+ffn+post3", "pre+ffn", "ffn", "ffn+post1", "pre+ffn+post3+drop",
                                     "pre+ffn+post3+split", "pre+ffn+split", "ffn+split", "ffn+post1+split", "pre+ffn+post3+drop+split",
                                     # ks: the key block of the q | k | v projection leaves pre-scaled (post_kscale); post3 alone =
                                     # the encoder's layer-0 projection as a chain of its own
                                     "post3+ks", "pre+ffn+post3+ks", "pre+ffn+post3+ks+split"])
def test_row_chain_matches_the_separate_kernels(M, variant):
    """One st_row_chain launch == output_linear + residual + LayerNorm, feed-forward sublayer and the next projection as
    separate kernels (the emulation composes their emulations): every tensor the backward reads, ragged last row block,
    strided operand views; with dropout the masks must be the ones st_gemm / st_gemm_ln draw (the backward kernels
    regenerate them)."""
    from st_amd import chains
    d, dff = 256, 1024
    parts = variant.split("+")
    has_pre, has_ffn, drop = "pre" in parts, "ffn" in parts, "drop" in parts
    nb = 3 if "post3" in parts else 1 if "post1" in parts else 0
    wo, w1, w2 = g(d, d, seed=1, scale=d ** -0.5), g(dff, d, seed=2, scale=d ** -0.5), g(d, dff, seed=3, scale=dff ** -0.5)
    wp = g(256 * max(nb, 1), d, seed=4, scale=d ** -0.5)
    bo, b1, b2, bp = g(d, seed=5, dtype=F32), g(dff, seed=6, dtype=F32), g(d, seed=7, dtype=F32), g(256 * max(nb, 1), seed=8, dtype=F32)
    g0, be0, g1, be1 = g(d, seed=9, dtype=F32) * 0.2 + 1, g(d, seed=10, dtype=F32) * 0.1, g(d, seed=11, dtype=F32) * 0.2 + 1, g(d, seed=12, dtype=F32) * 0.1
    A, R = g(M, d, seed=13), g(M, d, seed=14)
    dn1, de1 = _drops(21, 0.1) if drop else (None, None)
    dn2, de2 = (nv.Drop(dn1.seed, 22, 0.1), em.Drop(de1.seed, 22, 0.1)) if drop else (None, None)   # one device seed, two sites

    def blocks(dev):
        f = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t)
        b = []
        if has_pre:
            b += chains.blocks_of(f(wo))
        if has_ffn:
            b += chains.ffn_blocks(f(w1), f(w2))
        if nb:
            b += chains.blocks_of(f(wp))
        return b

    def run(dev, rc, d1, d2):
        f = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t)
        E = lambda *s, dt=BF16: torch.zeros(*s, dtype=dt, device=dev)
        o = dict(out0=E(M, d), xhat0=E(M, d), rstd0=E(M, dt=F32), H=E(M, dff), out1=E(M, d), xhat1=E(M, d), rstd1=E(M, dt=F32),
                 P=E(M, 256 * max(nb, 1)))
        if dev == "cuda":
            cs = chains.ChainSet("cuda")
            cid = cs.add(blocks("cuda"))
            cs.finalize().rebuild()
            ch = cs.chain(cid)
            if "split" in parts:        # the feed-forward's hidden dimension over 4 workgroups per row block (M <= 2048; else ignored)
                ch.split_work = split_work
            Aw = torch.zeros(M, d + 64, dtype=BF16, device="cuda")      # strided operand views
            Aw[:, 32:32 + d] = f(A)
            Rw = torch.zeros(M + 2, d + 8, dtype=BF16, device="cuda")
            Rw[:M, :d] = f(R)
            a_in, r_in = Aw[:, 32:32 + d], Rw[:M, :d]
        else:
            ch = chains.Chain(None, len(blocks("cpu")), blocks("cpu"))
            a_in, r_in = A, R
        if has_ffn and dev == "cuda":
            o["bits"] = torch.zeros(nv.chain_mask_words(M, dff), dtype=torch.int64, device=dev)
        rc(a_in, ch,
           pre=(r_in, f(bo), f(g0), f(be0), o["out0"], o["xhat0"], o["rstd0"]) if has_pre else None,
           ffn=(dff, f(b1), f(b2), f(g1), f(be1), o["H"], o["out1"], o["xhat1"], o["rstd1"], d1, d2, o.get("bits")) if has_ffn else None,
           post=(nb, f(bp), o["P"]) if nb else None, **({"post_kscale": 0.125 * nv.K_LOG2_SCALE} if "ks" in parts else {}))
        return o

    split_work = torch.zeros(nv.split_work_words(), dtype=torch.int32, device="cuda") if "split" in parts else None
    got, ref = run("cuda", nv.row_chain, dn1, dn2), run("cpu", em.row_chain, de1, de2)
    if split_work is not None:
        # the tickets are back at zero, and a second launch on the same scratch gives bit-identical results (the partials are
        # added in chunk order, whoever arrives last)
        assert int(split_work[:256].abs().sum()) == 0, "split tickets not reset"
        again = run("cuda", nv.row_chain, dn1, dn2)
        for n in got:
            assert torch.equal(got[n], again[n]), "split row chain not reproducible: %s" % n
        if M == 5:      # the same scratch after a launch with more row blocks (whose partials lie where nothing else may)
            M_big = 320
            chains_mod = chains
            cs = chains_mod.ChainSet("cuda")
            cid = cs.add(blocks("cuda"))
            cs.finalize().rebuild()
            chb = cs.chain(cid)
            chb.split_work = split_work
            Ab, Rb = cu(g(M_big, d, seed=31)), cu(g(M_big, d, seed=32))
            Eb = lambda *s, dt=BF16: torch.zeros(*s, dtype=dt, device="cuda")
            nv.row_chain(Ab, chb,
                         pre=(Rb, cu(bo), cu(g0), cu(be0), Eb(M_big, d), Eb(M_big, d), Eb(M_big, dt=F32)) if has_pre else None,
                         ffn=(dff, cu(b1), cu(b2), cu(g1), cu(be1), Eb(M_big, dff), Eb(M_big, d), Eb(M_big, d), Eb(M_big, dt=F32), dn1, dn2),
                         post=(nb, cu(bp), Eb(M_big, 256 * max(nb, 1))) if nb else None)
            again = run("cuda", nv.row_chain, dn1, dn2)
            for n in got:
                assert torch.equal(got[n], again[n]), "split row chain after a larger launch on the same scratch: %s" % n
    names = (["out0", "xhat0", "rstd0"] if has_pre else []) + (["H", "out1", "xhat1", "rstd1"] if has_ffn else []) + (["P"] if nb else [])
    for n in names:
        check(got[n], ref[n], 2e-3 if n.startswith("rstd") else 1e-2, "row_chain %s M=%d: %s" % (variant, M, n))
    if has_ffn:      # the ReLU-mask bits the backward chain will read == the bits of the H the same launch wrote
        want = nv.relu_bits_from(got["H"])
        diff = got["bits"] ^ want
        nc, n_wg = dff // 256, want.numel() // (dff // 256 * 512)
        mt = -(-M // (32 * n_wg))
        sh = torch.arange(mt * 16, device="cuda", dtype=torch.int64)
        bad = ((diff.unsqueeze(1) >> sh) & 1).view(n_wg, nc, 8, 2, 32, mt, 4, 4).permute(0, 5, 4, 1, 2, 6, 3, 7).reshape(n_wg * mt * 32, dff)
        assert int(bad[:M].sum()) == 0, "row_chain %s M=%d: relu_bits differ from H > 0 inside the valid rows" % (variant, M)
    if drop:
        _zero_pattern_equal(got["H"], ref["H"], "row_chain dropout1")
        _zero_pattern_equal(got["out1"], ref["out1"], "row_chain dropout2")
        # and against the kernels the backward pairs with: st_gemm (ReLU + dropout) / st_gemm_ln (drop_where = 2)
        h = nv.gemm(got["out0"], cu(w1), torch.zeros(M, dff, dtype=BF16, device="cuda"), bias=cu(b1), epi=nv.EPI_BF16_RELU, drop=dn1)
        _zero_pattern_equal(got["H"], h, "row_chain dropout1 vs st_gemm")


@pytest.mark.parametrize("M", [5, 64, 1000, 9000, 24060])
@pytest.mark.parametrize("variant", ["post6", "plain", "post6+ks", "post6+drop", "drop"])
def test_row_chain512_matches_the_separate_kernels(M, variant):
    """st_row_chain512 (d_model 512: BASELINE config 3's encoder layer between two attention kernels as ONE launch) == output_linear +
    residual + LayerNorm, the feed-forward sublayer and the next q | k | v projection as separate kernels: every tensor the backward
    reads, the ReLU bits, ragged last row block, strided operand views, the dropout masks of st_gemm / st_gemm_ln."""
    from st_amd import chains
    d, dff = 512, 1024
    parts = variant.split("+")
    post, drop = "post6" in parts, "drop" in parts
    wo, w1, w2 = g(d, d, seed=1, scale=d ** -0.5), g(dff, d, seed=2, scale=d ** -0.5), g(d, dff, seed=3, scale=dff ** -0.5)
    wp = g(3 * d, d, seed=4, scale=d ** -0.5)
    bo, b1, b2, bp = g(d, seed=5, dtype=F32), g(dff, seed=6, dtype=F32), g(d, seed=7, dtype=F32), g(3 * d, seed=8, dtype=F32)
    g0, be0, g1, be1 = g(d, seed=9, dtype=F32) * 0.2 + 1, g(d, seed=10, dtype=F32) * 0.1, g(d, seed=11, dtype=F32) * 0.2 + 1, g(d, seed=12, dtype=F32) * 0.1
    A, R = g(M, d, seed=13), g(M, d, seed=14)
    dn1, de1 = _drops(21, 0.1) if drop else (None, None)
    dn2, de2 = (nv.Drop(dn1.seed, 22, 0.1), em.Drop(de1.seed, 22, 0.1)) if drop else (None, None)

    def run(dev, rc, d1, d2):
        f = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t)
        E = lambda *s, dt=BF16: torch.zeros(*s, dtype=dt, device=dev)
        o = dict(out0=E(M, d), xhat0=E(M, d), rstd0=E(M, dt=F32), H=E(M, dff), out1=E(M, d), xhat1=E(M, d), rstd1=E(M, dt=F32), P=E(M, 3 * d))
        blocks = chains.encoder512_blocks(f(wo), f(w1), f(w2), f(wp) if post else None)
        if dev == "cuda":
            cs = chains.ChainSet("cuda")
            cid = cs.add(blocks)
            cs.finalize().rebuild()
            ch = cs.chain(cid)
            Aw = torch.zeros(M, d + 64, dtype=BF16, device="cuda")      # strided operand views
            Aw[:, 32:32 + d] = f(A)
            Rw = torch.zeros(M + 2, d + 8, dtype=BF16, device="cuda")
            Rw[:M, :d] = f(R)
            a_in, r_in = Aw[:, 32:32 + d], Rw[:M, :d]
            o["bits"] = torch.zeros(nv.chain_mask_words(M, dff, d), dtype=torch.int64, device=dev)
        else:
            ch = chains.Chain(None, len(blocks), blocks)
            a_in, r_in = A, R
        rc(a_in, ch, pre=(r_in, f(bo), f(g0), f(be0), o["out0"], o["xhat0"], o["rstd0"]),
           ffn=(dff, f(b1), f(b2), f(g1), f(be1), o["H"], o["out1"], o["xhat1"], o["rstd1"], d1, d2, o.get("bits")),
           post=(6, f(bp), o["P"]) if post else None, **({"post_kscale": 0.125 * nv.K_LOG2_SCALE} if "ks" in parts else {}))
        return o

    got, ref = run("cuda", nv.row_chain, dn1, dn2), run("cpu", em.row_chain, de1, de2)
    for n in ["out0", "xhat0", "rstd0", "H", "out1", "xhat1", "rstd1"] + (["P"] if post else []):
        check(got[n], ref[n], 2e-3 if n.startswith("rstd") else 1e-2, "row_chain512 %s M=%d: %s" % (variant, M, n))
    want = nv.relu_bits_from(got["H"], d)
    diff = got["bits"] ^ want
    nc, n_wg = dff // 256, want.numel() // (dff // 256 * 512)
    sh = torch.arange(32, device="cuda", dtype=torch.int64)
    bad = ((diff.unsqueeze(1) >> sh) & 1).view(n_wg, nc, 8, 2, 32, 2, 4, 4).permute(0, 5, 4, 1, 2, 6, 3, 7).reshape(n_wg * 64, dff)
    assert int(bad[:M].sum()) == 0, "row_chain512 %s M=%d: relu_bits differ from H > 0 inside the valid rows" % (variant, M)
    if drop:
        _zero_pattern_equal(got["H"], ref["H"], "row_chain512 dropout1")
        _zero_pattern_equal(got["out1"], ref["out1"], "row_chain512 dropout2")


@pytest.mark.parametrize("case", ["train", "train+drop", "decode", "short-keys"])
def test_attn_f1_fwd_equals_row_chain_plus_attn_fwd(case):
    """st_attn_f1_fwd (the decoder-encoder attention with its chain stage - output_linear + LayerNorm + q projection - in the
    prologue of the few-queries kernel) against the two launches it replaces: every tensor bit for bit (same arithmetic in
    the same order; the attention then sees the same q).  'short-keys' does not qualify for the fused kernel: the wrapper must
    fall back to the two launches."""
    from st_amd import chains
    from st_amd.functional import Rows, attn_work
    d, H = 256, 4
    gen = torch.Generator().manual_seed(11)
    if case == "decode":
        B, beam = 7, 10
        q_len = torch.full((B,), beam, dtype=torch.int64)
        k_len = torch.randint(300, 900, (B,), generator=gen)
    elif case == "short-keys":
        B = 5
        q_len, k_len = torch.randint(3, 50, (B,), generator=gen), torch.randint(40, 200, (B,), generator=gen)
    else:
        B = 9
        q_len, k_len = torch.randint(1, 64, (B,), generator=gen), torch.randint(260, 1000, (B,), generator=gen)
        q_len[0], q_len[1], k_len[0] = 64, 33, 999
    q_rows, k_rows = Rows.packed(q_len, "cuda"), Rows.packed(k_len, "cuda")
    M, Mk = int(q_len.sum()), int(k_len.sum())
    wo, wq = cu(g(d, d, seed=1, scale=d ** -0.5)), cu(g(d, d, seed=2, scale=d ** -0.5))
    bo, bq = cu(g(d, seed=3, dtype=F32)), cu(g(d, seed=4, dtype=F32))
    g0, be0 = cu(g(d, seed=5, dtype=F32) * 0.2 + 1), cu(g(d, seed=6, dtype=F32) * 0.1)
    A, R, kv = cu(g(M, d, seed=7)), cu(g(M, d, seed=8)), cu(g(Mk, 2 * d, seed=9))
    cs = chains.ChainSet("cuda")
    cid = cs.add(chains.blocks_of(wo) + chains.blocks_of(wq))
    cs.finalize().rebuild()
    ch = cs.chain(cid)
    drop = _drops(31, 0.1)[0] if "drop" in case else None
    work = attn_work(q_rows, k_rows, False, d // H, H)[0]
    scale = (d // H) ** -0.5

    def bufs():
        E = lambda *s, dt=BF16: torch.zeros(*s, dtype=dt, device="cuda")
        return dict(out=E(M, d), xhat=E(M, d), rstd=E(M, dt=F32), q=E(M, d), O=E(M, d), ores=E(M, d), lse=E(H * M, dt=F32))

    a, b = bufs(), bufs()
    nv.row_chain(A, ch, pre=(R, bo, g0, be0, a["out"], a["xhat"], a["rstd"]), post=(1, bq, a["q"]))
    nv.attn_fwd(a["q"], kv[:, :d], kv[:, d:], a["O"], a["lse"], q_rows.off, q_rows.len, k_rows.off, k_rows.len, H, int(q_len.max()),
                False, scale, work=work, drop=drop, max_k=int(k_len.max()), ores=a["ores"])
    nv.attn_f1_fwd(A, ch, (R, bo, g0, be0, b["out"], b["xhat"], b["rstd"]), (1, bq, b["q"]), kv[:, :d], kv[:, d:], b["O"], b["lse"],
                   q_rows.off, q_rows.len, k_rows.off, k_rows.len, H, int(q_len.max()), scale, work=work, drop=drop,
                   max_k=int(k_len.max()), ores=b["ores"])
    torch.cuda.synchronize()
    for n in a:
        assert torch.equal(a[n], b[n]), "attn_f1_fwd %s: %s differs (max |d| %.3e)" % (case, n, (a[n].float() - b[n].float()).abs().max().item())
    assert float(b["O"].float().abs().sum()) > 0


@pytest.mark.parametrize("case", ["train", "train+drop", "short-keys"])
def test_attn_sf1_fwd_equals_the_three_launches(case):
    """st_attn_sf1_fwd (a decoder layer's causal self-attention, the chain stage behind it and the decoder-encoder attention as one
    launch) against st_attn_fwd(causal) + st_row_chain + st_attn_fwd: every tensor bit for bit (the fused self-attention repeats
    attn_fwd_kernel's arithmetic in its order) except the self-attention's bf16 residual, which agrees to its own precision.  'short-keys' does not qualify: the wrapper falls back to the three launches."""
    from st_amd import chains
    from st_amd.functional import Rows, attn_work
    d, H = 256, 4
    gen = torch.Generator().manual_seed(12)
    B = 9
    if case == "short-keys":
        q_len, k_len = torch.randint(3, 50, (B,), generator=gen), torch.randint(40, 200, (B,), generator=gen)
    else:
        q_len, k_len = torch.randint(1, 64, (B,), generator=gen), torch.randint(260, 1000, (B,), generator=gen)
        q_len[0], q_len[1], q_len[2], k_len[0] = 64, 33, 32, 999
    q_rows, k_rows = Rows.packed(q_len, "cuda"), Rows.packed(k_len, "cuda")
    M, Mk = int(q_len.sum()), int(k_len.sum())
    wo, wq = cu(g(d, d, seed=1, scale=d ** -0.5)), cu(g(d, d, seed=2, scale=d ** -0.5))
    bo, bq = cu(g(d, seed=3, dtype=F32)), cu(g(d, seed=4, dtype=F32))
    g0, be0 = cu(g(d, seed=5, dtype=F32) * 0.2 + 1), cu(g(d, seed=6, dtype=F32) * 0.1)
    qkv, R, kv = cu(g(M, 3 * d, seed=7)), cu(g(M, d, seed=8)), cu(g(Mk, 2 * d, seed=9))
    cs = chains.ChainSet("cuda")
    cid = cs.add(chains.blocks_of(wo) + chains.blocks_of(wq))
    cs.finalize().rebuild()
    ch = cs.chain(cid)
    d_self = _drops(41, 0.1)[0] if "drop" in case else None
    d_cross = nv.Drop(d_self.seed, 42, 0.1) if "drop" in case else None
    w_self = attn_work(q_rows, q_rows, True, d // H, H)[0]
    w_cross = attn_work(q_rows, k_rows, False, d // H, H)[0]
    scale, mq, mk = (d // H) ** -0.5, int(q_len.max()), int(k_len.max())

    def bufs():
        E = lambda *s, dt=BF16: torch.zeros(*s, dtype=dt, device="cuda")
        return dict(ctx=E(M, d), ores_s=E(M, d), lse_s=E(H * M, dt=F32), out=E(M, d), xhat=E(M, d), rstd=E(M, dt=F32), q=E(M, d),
                    O=E(M, d), ores=E(M, d), lse=E(H * M, dt=F32))

    a, b = bufs(), bufs()
    nv.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], a["ctx"], a["lse_s"], q_rows.off, q_rows.len, q_rows.off, q_rows.len, H, mq,
                True, scale, work=w_self, drop=d_self, max_k=mq, ores=a["ores_s"])
    nv.row_chain(a["ctx"], ch, pre=(R, bo, g0, be0, a["out"], a["xhat"], a["rstd"]), post=(1, bq, a["q"]))
    nv.attn_fwd(a["q"], kv[:, :d], kv[:, d:], a["O"], a["lse"], q_rows.off, q_rows.len, k_rows.off, k_rows.len, H, mq, False, scale,
                work=w_cross, drop=d_cross, max_k=mk, ores=a["ores"])
    nv.attn_sf1_fwd(qkv, b["ctx"], b["lse_s"], (R, bo, g0, be0, b["out"], b["xhat"], b["rstd"]), (1, bq, b["q"]), ch, kv[:, :d], kv[:, d:],
                    b["O"], b["lse"], q_rows.off, q_rows.len, k_rows.off, k_rows.len, H, mq, scale, work_self=w_self, work=w_cross,
                    drop_self=d_self, drop=d_cross, max_k=mk, ores_self=b["ores_s"], ores=b["ores"])
    torch.cuda.synchronize()
    for n in a:
        if n == "ores_s":
            # the self-attention's bf16 RESIDUAL of the context (what bf16 rounding dropped: O + Ores = the fp32 context to ~16 bits):
            # the fused stage reproduces the context and the LSE bit for bit, its fp32 context agrees to a few 1e-6 relative (it shows
            # only here) - well inside the 2^-16 the pair promises
            err = (a[n].float() - b[n].float()).abs()
            assert bool((err <= 2.0 ** -15 * a["ctx"].float().abs() + 1e-9).all()), "attn_sf1_fwd %s: ores_s off by %.3e" % (case, float(err.max()))
            continue
        assert torch.equal(a[n], b[n]), "attn_sf1_fwd %s: %s differs (max |d| %.3e)" % (case, n, (a[n].float() - b[n].float()).abs().max().item())
    assert float(b["O"].float().abs().sum()) > 0 and float(b["ctx"].float().abs().sum()) > 0


@pytest.mark.parametrize("M", [5, 320, 1206, 3120, 9000, 17000, 24700])      # 3120: 98 row blocks - a split chain takes two chunks per part; 24700: past one round of 96-row tiles -> two rounds of 64-row ones
@pytest.mark.parametrize("variant", ["head1+tail", "head3+ffn+tail", "ffn+tail", "head3+ffn+tail+drop", "head3+ffn", "tail",
                                     "head0+ffn+tail+drop", "head3+ffn+tail+split", "ffn+tail+split", "head3+ffn+tail+drop+split",
                                     "head3+ffn+split", "head0+ffn+tail+drop+split", "ffn+split"])
def test_row_chain_bwd_matches_the_separate_kernels(M, variant):
    """One st_row_chain_bwd launch == st_gemm_lnbwd, st_gemm (mask epilogue), st_gemm_lnbwd and st_gemm (delta epilogue) as
    separate kernels (the emulation composes their emulations): every gradient tensor, the atomically accumulated
    LayerNorm / bias gradients, delta; ragged last row block; both row-tile geometries (M = 9000: 96-row workgroups)."""
    from st_amd import chains
    d, dff = 256, 1024
    parts = variant.split("+")
    nb = 3 if "head3" in parts else 1 if "head1" in parts else 0
    has_head = nb > 0 or "head0" in parts          # head0: the bare LayerNorm backward of G (no GEMM in front)
    has_ffn, has_tail, drop = "ffn" in parts, "tail" in parts, "drop" in parts
    wp, w1, w2, wo = g(256 * max(nb, 1), d, seed=1, scale=d ** -0.5), g(dff, d, seed=2, scale=d ** -0.5), \
        g(d, dff, seed=3, scale=dff ** -0.5), g(d, d, seed=4, scale=d ** -0.5)
    dP, G, DS = g(M, 256 * max(nb, 1), seed=5, scale=0.3), g(M, d, seed=6, scale=0.3), g(M, d, seed=7, scale=0.3)
    xa, xb = g(M, d, seed=8), g(M, d, seed=9)
    ra, rb = g(M, seed=10, dtype=F32).abs() + 0.5, g(M, seed=11, dtype=F32).abs() + 0.5
    ga, gb = g(d, seed=12, dtype=F32) * 0.2 + 1, g(d, seed=13, dtype=F32) * 0.2 + 1
    H = torch.relu(g(M, dff, seed=14))
    O, Ores = g(M, d, seed=15), g(M, d, seed=16, scale=2.0 ** -9)
    dn, de = _drops(31, 0.1) if drop else (None, None)

    def blocks(f):
        b = []
        if nb:
            b += chains.t_blocks(chains.blocks_of(f(wp)))
        if has_ffn:
            b += chains.ffn_blocks_bwd(f(w1), f(w2))
        if has_tail:
            b += chains.t_blocks(chains.blocks_of(f(wo)))
        return b

    def run(dev, fn, dr):
        f = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t)
        Z = lambda *s, dt=BF16: torch.zeros(*s, dtype=dt, device=dev)
        o = dict(ds_a=Z(M, d), dga=Z(d, dt=F32) + 1, dba=Z(d, dt=F32) + 2, dbia=Z(d, dt=F32) + 3, dH=Z(M, dff), ds_b=Z(M, d),
                 dgb=Z(d, dt=F32) - 1, dbb=Z(d, dt=F32) - 2, dbib=Z(d, dt=F32) - 3, dctx=Z(M, d), delta=Z(4 * M, dt=F32))
        if dev == "cuda":
            cs = chains.ChainSet("cuda")
            cid = cs.add(blocks(f))
            cs.finalize().rebuild()
            ch = cs.chain(cid)
            if "split" in parts:        # the hidden dimension over 4 workgroups per row block (M <= 2048; else ignored)
                ch.split_work = split_work
        else:
            ch = chains.Chain(None, len(blocks(f)), blocks(f))
        fn(ch, M,
           head=(nb, f(dP) if nb else None, f(G), f(xa), f(ra), f(ga), dr, o["ds_a"], o["dga"], o["dba"], o["dbia"]) if has_head else None,
           ds_in=None if has_head else f(DS),
           ffn=(dff, (nv if dev == "cuda" else em).relu_bits_from(f(H)), 1.0 / 0.9 if drop else 1.0, o["dH"], f(xb), f(rb), f(gb),
                o["ds_b"], o["dgb"], o["dbb"], o["dbib"]) if has_ffn else None,
           tail=(f(O), f(Ores), o["dctx"], o["delta"]) if has_tail else None)
        return o

    split_work = torch.zeros(nv.split_work_words(), dtype=torch.int32, device="cuda") if "split" in parts else None
    got, ref = run("cuda", nv.row_chain_bwd, dn), run("cpu", em.row_chain_bwd, de)
    if split_work is not None:
        assert int(split_work[:256].abs().sum()) == 0, "split tickets not reset"
        again = run("cuda", nv.row_chain_bwd, dn)
        for n in ("ds_a", "dH", "ds_b", "dctx", "delta"):       # (the column sums are atomic adds: order-dependent rounding)
            assert torch.equal(got[n], again[n]), "split backward row chain not reproducible: %s" % n
    names = (["ds_a", "dga", "dba", "dbia"] if has_head else []) + (["dH", "ds_b", "dgb", "dbb", "dbib"] if has_ffn else []) + \
        (["dctx", "delta"] if has_tail else [])
    for n in names:
        tol = 1e-2 if got[n].dtype == BF16 else 5e-3
        check(got[n], ref[n], tol, "row_chain_bwd %s M=%d: %s" % (variant, M, n))


@pytest.mark.parametrize("n", [4, 1000, 13_300_004, 4 * 1024 * 2048 * 4 + 12])
def test_zero_buffer(n):
    """st_zero (zero_grad of the flat gradient buffer): every byte of the range, nothing around it; odd sizes go to torch."""
    buf = torch.full((n + 8,), 3.0, device="cuda")
    nv.zero_(buf[4:4 + n])
    torch.cuda.synchronize()
    assert float(buf[4:4 + n].abs().sum()) == 0.0
    assert float(buf[:4].sum()) == 12.0 and float(buf[4 + n:].sum()) == 12.0
    odd = torch.full((7,), 2.0, device="cuda")
    nv.zero_(odd[1:])
    assert odd.tolist() == [2.0] + [0.0] * 6


@pytest.mark.parametrize("M", [9000, 24060])
def test_row_chain_bwd_column_sums_through_the_workspace(M):
    """Encoder-sized backward chains leave their LayerNorm column sums in a per-workgroup workspace and st_colsum_fold adds it to the
    gradients (ABI 4): nothing reaches the gradient vectors while the fold is deferred, the deferred fold gives what the immediate
    one gives, and - no atomics between workgroups any more - twice the same launch gives the same bits."""
    from st_amd import chains
    d, dff = 256, 1024
    assert nv.load()._cdll.st_row_chain_bwd_colsum_rows(M, 1, dff, 1) == -(-M // (96 if M > 64 * 256 else 64))
    assert nv.load()._cdll.st_row_chain_bwd_colsum_rows(1206, 1, dff, 1) == 0
    wp, w1, w2, wo = g(768, d, seed=1, scale=d ** -0.5), g(dff, d, seed=2, scale=d ** -0.5), g(d, dff, seed=3, scale=dff ** -0.5), g(d, d, seed=4, scale=d ** -0.5)
    c = lambda t: t.cuda()
    cs = chains.ChainSet("cuda")
    cid = cs.add(chains.t_blocks(chains.blocks_of(c(wp))) + chains.ffn_blocks_bwd(c(w1), c(w2)) + chains.t_blocks(chains.blocks_of(c(wo))))
    cs.finalize().rebuild()
    ch = cs.chain(cid)
    dP, G, xa, xb = c(g(M, 768, seed=5, scale=0.3)), c(g(M, d, seed=6, scale=0.3)), c(g(M, d, seed=8)), c(g(M, d, seed=9))
    ra, rb = c(g(M, seed=10, dtype=F32).abs() + 0.5), c(g(M, seed=11, dtype=F32).abs() + 0.5)
    ga, gb = c(g(d, seed=12, dtype=F32) * 0.2 + 1), c(g(d, seed=13, dtype=F32) * 0.2 + 1)
    bits = nv.relu_bits_from(torch.relu(c(g(M, dff, seed=14))))
    O, Ores = c(g(M, d, seed=15)), c(g(M, d, seed=16, scale=2.0 ** -9))
    E = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device="cuda")

    def run(deferred):
        acc = [torch.full((d,), float(i), device="cuda") for i in range(6)]
        nv.fold_deferred = deferred
        try:
            nv.row_chain_bwd(ch, M, head=(3, dP, G, xa, ra, ga, None, E(M, d), acc[0], acc[1], acc[2]),
                             ffn=(dff, bits, 1.0, E(M, dff), xb, rb, gb, E(M, d), acc[3], acc[4], acc[5]), tail=(O, Ores, E(M, d), E(4 * M, dt=F32)))
            if deferred:
                torch.cuda.synchronize()
                for i in range(6):
                    assert torch.equal(acc[i], torch.full((d,), float(i), device="cuda")), "a deferred fold reached gradient %d" % i
                assert len(nv._fold_pending) == 1
                nv.flush_colsum_folds()
                assert not nv._fold_pending
        finally:
            nv.fold_deferred = False
        torch.cuda.synchronize()
        return acc

    a, b, c2 = run(False), run(True), run(False)
    for i in range(6):
        assert torch.equal(a[i], b[i]) and torch.equal(a[i], c2[i]), "column sums not reproducible: vector %d" % i
        assert float((a[i] - i).abs().max()) > 0


@pytest.mark.parametrize("M", [5, 64, 1000, 9000, 24060])
@pytest.mark.parametrize("variant", ["head6", "head0", "head6+drop", "head6+nores"])
def test_row_chain512_bwd_matches_the_separate_kernels(M, variant):
    """st_row_chain512_bwd (d_model 512: HEAD + FFN + TAIL of BASELINE config 3's encoder layer as ONE launch) == st_gemm_lnbwd, st_gemm
    (mask epilogue), st_gemm_lnbwd and st_gemm (delta epilogue, 8 heads) as separate kernels: every gradient tensor, the atomically
    accumulated LayerNorm / bias gradients, delta; ragged last row block; the bare LayerNorm backward as head; without Ores."""
    from st_amd import chains
    d, dff = 512, 1024
    parts = variant.split("+")
    nb, drop, nores = (6 if "head6" in parts else 0), "drop" in parts, "nores" in parts
    wp, w1, w2, wo = g(3 * d, d, seed=1, scale=d ** -0.5), g(dff, d, seed=2, scale=d ** -0.5), g(d, dff, seed=3, scale=dff ** -0.5), \
        g(d, d, seed=4, scale=d ** -0.5)
    dP, G = g(M, 3 * d, seed=5, scale=0.3), g(M, d, seed=6, scale=0.3)
    xa, xb = g(M, d, seed=8), g(M, d, seed=9)
    ra, rb = g(M, seed=10, dtype=F32).abs() + 0.5, g(M, seed=11, dtype=F32).abs() + 0.5
    ga, gb = g(d, seed=12, dtype=F32) * 0.2 + 1, g(d, seed=13, dtype=F32) * 0.2 + 1
    H = torch.relu(g(M, dff, seed=14))
    O, Ores = g(M, d, seed=15), g(M, d, seed=16, scale=2.0 ** -9)
    dn, de = _drops(31, 0.1) if drop else (None, None)

    def run(dev, fn, dr):
        f = (lambda t: t.cuda()) if dev == "cuda" else (lambda t: t)
        Z = lambda *s, dt=BF16: torch.zeros(*s, dtype=dt, device=dev)
        o = dict(ds_a=Z(M, d), dga=Z(d, dt=F32) + 1, dba=Z(d, dt=F32) + 2, dbia=Z(d, dt=F32) + 3, dH=Z(M, dff), ds_b=Z(M, d),
                 dgb=Z(d, dt=F32) - 1, dbb=Z(d, dt=F32) - 2, dbib=Z(d, dt=F32) - 3, dctx=Z(M, d), delta=Z(8 * M, dt=F32))
        blocks = chains.encoder512_blocks_bwd(f(wo), f(w1), f(w2), f(wp) if nb else None)
        if dev == "cuda":
            cs = chains.ChainSet("cuda")
            cid = cs.add(blocks)
            cs.finalize().rebuild()
            ch = cs.chain(cid)
            bits = nv.relu_bits_from(f(H), d)
        else:
            ch = chains.Chain(None, len(blocks), blocks)
            bits = em.relu_bits_from(H, d)
        fn(ch, M, head=(nb, f(dP) if nb else None, f(G), f(xa), f(ra), f(ga), dr, o["ds_a"], o["dga"], o["dba"], o["dbia"]),
           ffn=(dff, bits, 1.0 / 0.9 if drop else 1.0, o["dH"], f(xb), f(rb), f(gb), o["ds_b"], o["dgb"], o["dbb"], o["dbib"]),
           tail=(f(O), None if nores else f(Ores), o["dctx"], o["delta"]))
        return o

    got, ref = run("cuda", nv.row_chain_bwd, dn), run("cpu", em.row_chain_bwd, de)
    for n in ["ds_a", "dga", "dba", "dbia", "dH", "ds_b", "dgb", "dbb", "dbib", "dctx", "delta"]:
        tol = 1e-2 if got[n].dtype == BF16 else 5e-3
        check(got[n], ref[n], tol, "row_chain512_bwd %s M=%d: %s" % (variant, M, n))


@pytest.mark.parametrize("two_launch", [False, True])
@pytest.mark.parametrize("B,beam,V", [(5, 4, 30), (32, 10, 4337), (3, 16, 1000), (7, 10, 5120), (2, 1, 100), (4, 3, 257)])
def test_beam_advance_matches_torch_formulation(B, beam, V, two_launch):
    """st_beam_advance (log-softmax + top-k over beam x V + Beam.py's bookkeeping; one launch, or - given scratch - the
    row-best + merge pair) against the torch formulation (tests/_emul.py) over several steps, with finished utterances
    (frozen), -inf slots (step 0) and EOS."""
    work = torch.zeros(nv.beam_work_words(B, beam), dtype=torch.long, device="cuda") if two_launch else None
    anc_a = torch.arange(B * beam, dtype=torch.int32).unsqueeze(1).repeat(1, 9).contiguous() if two_launch else None   # lineage table
    anc_b = anc_a.clone() if two_launch else None
    if two_launch:
        anc_a = anc_a.cuda()
    gen = torch.Generator().manual_seed(7)
    S, eos, ld = 6, 2, (V + 7) // 8 * 8

    def state(dev):
        sc = torch.full((B, beam), float("-inf"))
        sc[:, 0] = 0.0
        return dict(scores=sc.to(dev), tokens=torch.ones(B * beam, dtype=torch.long, device=dev), done=torch.zeros(B, dtype=torch.bool, device=dev),
                    lengths=torch.zeros(B, dtype=torch.long, device=dev), hist=torch.zeros(S, B, beam, device=dev),
                    back=torch.zeros(S, B, beam, dtype=torch.long, device=dev), toks=torch.zeros(S, B, beam, dtype=torch.long, device=dev),
                    order=torch.zeros(B * beam, dtype=torch.long, device=dev), step=torch.zeros(1, dtype=torch.long, device=dev))

    a, b = state("cuda"), state("cpu")
    emb, pe = torch.randn(V, 64, generator=gen), torch.randn(S + 1, 64, generator=gen)
    a["x"], b["x"] = torch.zeros(B * beam, 64, dtype=BF16, device="cuda"), torch.zeros(B * beam, 64, dtype=BF16)
    for t in range(S):
        logits = torch.randn(B * beam, ld, generator=gen) * 4
        if t >= 2:
            logits[0, eos] += 30.0          # utterance 0 finishes: its best hypothesis emits EOS
        for st, fn, dev in ((a, nv.beam_advance, "cuda"), (b, em.beam_advance, "cpu")):
            fn(logits.to(dev), V, beam, st["step"], eos, st["scores"], st["tokens"], st["done"], st["lengths"], st["hist"],
               st["back"], st["toks"], st["order"], work=work if dev == "cuda" else None, anc=anc_a if dev == "cuda" else anc_b,
               advance_step=two_launch, embed=(emb.to(dev), pe.to(dev), st["x"]) if two_launch else None)
            if not two_launch:
                st["step"] += 1
        assert int(a["step"]) == t + 1 and int(b["step"]) == t + 1
        assert torch.equal(a["done"].cpu(), b["done"]) and torch.equal(a["lengths"].cpu(), b["lengths"]), t
        live = ~b["done"] | (b["lengths"] == t + 1)        # rows of utterances that advanced in this step
        assert torch.equal(a["back"].cpu()[t], b["back"][t]) and torch.equal(a["order"].cpu(), b["order"]), t
        assert torch.equal(a["toks"].cpu()[t][live], b["toks"][t][live]) and torch.equal(a["tokens"].cpu(), b["tokens"]), t
        if two_launch:
            assert torch.equal(anc_a.cpu()[:, :t + 1], anc_b[:, :t + 1]), t
            assert torch.equal(a["x"].cpu(), b["x"]), t          # the next step's decoder input (embedding + PE, one rounding)
        assert torch.allclose(a["scores"].cpu(), b["scores"], atol=2e-5, rtol=1e-6) and \
            torch.allclose(a["hist"].cpu()[t], b["hist"][t], atol=2e-5, rtol=1e-6), t
    assert bool(b["done"][0]) and not bool(b["done"][1:].all())


@pytest.mark.parametrize("two_launch", [False, True])
def test_beam_advance_vs_reference_trellis_golden(two_launch):
    """st_beam_advance against tests/golden/beam_trellis.npz - the trellis the reference's own Beam class produced
    (tools/make_beam_goldens.py, repair R5): back-pointers and tokens bit-exact, scores to fp32 rounding."""
    from tests.test_decode_cpu import run_beam_advance_vs_trellis
    run_beam_advance_vs_trellis(nv, "cuda", two_launch=two_launch)


@pytest.mark.parametrize("B,beam,V", [(3, 5, 3), (6, 10, 4337), (4, 16, 700), (2, 10, 4096)])
def test_beam_advance_two_launches_equal_one(B, beam, V):
    """The row-best + merge pair of st_beam_advance against its one-workgroup-per-utterance kernel: the same winners in the
    same order - including ties (equal logits in every row: lowest flat index first) and fewer finite candidates than beam
    slots (V = 3 < beam at step 0: -inf candidates fill the beam) - and the same scores to fp32 rounding."""
    gen = torch.Generator().manual_seed(11)
    S, eos, ld = 4, 2, (V + 7) // 8 * 8

    def state():
        sc = torch.full((B, beam), float("-inf"))
        sc[:, 0] = 0.0
        return dict(scores=sc.cuda(), tokens=torch.ones(B * beam, dtype=torch.long, device="cuda"), done=torch.zeros(B, dtype=torch.bool, device="cuda"),
                    lengths=torch.zeros(B, dtype=torch.long, device="cuda"), hist=torch.zeros(S, B, beam, device="cuda"),
                    back=torch.zeros(S, B, beam, dtype=torch.long, device="cuda"), toks=torch.zeros(S, B, beam, dtype=torch.long, device="cuda"),
                    order=torch.zeros(B * beam, dtype=torch.long, device="cuda"), step=torch.zeros(1, dtype=torch.long, device="cuda"))

    a, b = state(), state()
    work = torch.zeros(nv.beam_work_words(B, beam), dtype=torch.long, device="cuda")
    for t in range(S):
        logits = (torch.randn(B * beam, ld, generator=gen) * 3).cuda()
        logits[: beam] = 0.0                       # utterance 0: every candidate of a row ties
        if t == 2:
            logits[beam: 2 * beam] = logits[beam].clone()      # utterance 1: its rows are copies of each other
        for st, w in ((a, None), (b, work)):
            nv.beam_advance(logits, V, beam, st["step"], eos, st["scores"], st["tokens"], st["done"], st["lengths"], st["hist"],
                            st["back"], st["toks"], st["order"], work=w)
            st["step"] += 1
        for k in ("done", "lengths", "back", "toks", "order", "tokens"):
            assert torch.equal(a[k], b[k]), (k, t)
        for k in ("scores", "hist"):
            assert torch.allclose(a[k], b[k], atol=2e-5, rtol=1e-6, equal_nan=True), (k, t)


@pytest.mark.parametrize("R,V", [(7, 30), (1206, 4337)])
def test_cross_entropy_rows_matches_torch(R, V):
    """st_ce_fwd / st_ce_bwd (functional.cross_entropy_rows) == nn.CrossEntropyLoss(ignore_index=0) on fp32 logits rows whose
    padding columns hold -1e30 (what VocabFn hands the loss): value and gradient (bf16), ignored rows, a scaled upstream
    gradient."""
    from st_amd.functional import cross_entropy_rows
    vp = (V + 7) // 8 * 8
    gen = torch.Generator().manual_seed(3)
    logits = torch.full((R, vp), -1e30)
    logits[:, :V] = torch.randn(R, V, generator=gen) * 3
    target = torch.randint(1, V, (R,), generator=gen)
    target[::5] = 0                                        # ignored rows
    ref_in = logits[:, :V].clone().requires_grad_(True)
    ref = torch.nn.CrossEntropyLoss(ignore_index=0)(ref_in, target)
    (ref * 0.7).backward()
    x = logits.cuda().requires_grad_(True)
    loss = cross_entropy_rows(x, target.cuda(), 0)
    (loss * 0.7).backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item())
    if vp > V:      # (autograd casts the bf16 gradient to the leaf's fp32; functional.VocabCeFn consumes it as bf16)
        assert float(x.grad[:, V:].abs().max()) == 0.0
    check(x.grad[:, :V], ref_in.grad, 6e-3, "cross-entropy gradient")
    assert float(x.grad[::5].abs().max()) == 0.0
    # the indexed form (row r's target = truth[index[r]]: the padded ground truth read through the ragged rows' positions)
    # against the gathered form: same kernels, same bits
    idx = torch.randperm(2 * R, generator=gen)[:R]
    truth = torch.zeros(2 * R, dtype=torch.long)
    truth[idx] = target
    lg = logits.cuda()
    outs = []
    for tgt, index in ((target.cuda(), None), (truth.cuda(), idx.cuda())):
        lse, sums = torch.empty(R, device="cuda"), torch.empty(3, device="cuda")
        nv.ce_fwd(lg, tgt, 0, lse, sums, index=index)
        dl = torch.empty(R, vp, dtype=BF16, device="cuda")
        nv.ce_bwd(lg, tgt, 0, lse, sums, torch.ones(1, device="cuda"), dl, index=index)
        outs.append((lse, sums, dl))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert abs(float(outs[0][1][2]) - ref.item()) < 1e-5 * abs(ref.item())


@pytest.mark.parametrize("lineage", [False, True])
@pytest.mark.parametrize("t", [0, 1, 15, 16, 37, 63, 64, 99])
def test_decode_self_attention_matches_reference(t, lineage):
    """st_decode_self_attn (one query per hypothesis over the KV cache, appending this step's K | V) == the fp32 softmax
    attention over cache positions 0 .. t, for positions inside and beyond the first 64-key pass; with a lineage table
    the earlier positions come from the cache rows it names (random rows here), the step's own from the hypothesis' row."""
    n, H, d, S = 37, 4, 256, 100
    qkv = g(n, 3 * d, seed=1)
    cache = g(n, S, 2 * d, seed=2)
    step = torch.tensor([t], dtype=torch.long)
    anc = torch.randint(0, n, (n, S), generator=torch.Generator().manual_seed(5), dtype=torch.int32) if lineage else None
    cg, ce = cache.clone().cuda(), cache.clone()
    out_g, out_e = torch.zeros(n, d, dtype=BF16, device="cuda"), torch.zeros(n, d, dtype=BF16)
    nv.decode_self_attn(cu(qkv), cg, step.cuda(), out_g, H, 0.125, anc=anc.cuda() if lineage else None)
    em.decode_self_attn(qkv, ce, step, out_e, H, 0.125, anc=anc)
    check(out_g, out_e, 1e-2, "decode self-attention t=%d" % t)
    assert torch.equal(cg.cpu()[:, t], qkv[:, d:]) and torch.equal(cg.cpu()[:, t + 1:], cache[:, t + 1:]) and \
        torch.equal(cg.cpu()[:, :t], cache[:, :t])


@pytest.mark.parametrize("V,lens,C", [(23, [30, 17, 25], 8), (4337, [300, 211], 41), (1000, [64, 1, 33, 128], 12)])
def test_ctc_gather_and_dlogits_kernels(V, lens, C):
    """st_ctc_gather / st_ctc_dlogits (BASELINE config 4's CTC branch around torch's ctc_loss) against their emulation: the
    row log-sum-exp, the gathered log-probabilities of the columns the loss reads, and the dense bf16 logits gradient with
    the scattered label columns (duplicates and -1 entries in the scatter list, rows of a padded layout that belong to no
    utterance)."""
    from st_amd.functional import Rows
    gen = torch.Generator().manual_seed(V)
    lens_t = torch.tensor(lens)
    B, T, R = len(lens), int(max(lens)), int(sum(lens))
    v_pad = (V + 1 + 7) // 8 * 8
    logits = torch.randn(R, v_pad, generator=gen) * 3
    logits[:, V:] = -1e30
    cols = torch.randint(0, V, (B, C), generator=gen, dtype=torch.int32)
    cols[:, 0] = 0
    cols[0, 3] = cols[0, 1]                                   # a duplicate column: only its first occurrence scatters
    scat = cols.clone()
    scat[0, 3] = -1
    scat[:, C - 1] = -1                                       # a padded target position
    rowmap = Rows.packed(lens_t, "cpu").scatter_index(T)
    roww = torch.rand(B, generator=gen) * 0.1
    gsmall = torch.randn(B, T, C, generator=gen) * 0.05
    gout = torch.tensor([0.7])
    outs = []
    for dev, mod in (("cpu", em), ("cuda", nv)):
        mv = lambda t: t.to(dev)
        lse = torch.zeros(R, dtype=F32, device=dev)
        lp = torch.zeros(B, T, C, dtype=F32, device=dev)
        mod.ctc_gather(mv(logits), mv(rowmap), T, mv(cols), lse, lp, V=V)
        dl = torch.full((R, v_pad), 7.0, dtype=BF16, device=dev)
        mod.ctc_dlogits(mv(logits), lse, mv(rowmap), T, mv(roww), mv(scat), mv(gsmall), mv(gout), dl, V=V)
        outs.append((lse.cpu(), lp.cpu(), dl.float().cpu()))
    (l0, p0, d0), (l1, p1, d1) = outs
    assert torch.allclose(l0, l1, rtol=0, atol=2e-5), float((l0 - l1).abs().max())
    assert torch.allclose(p0, p1, rtol=0, atol=3e-5), float((p0 - p1).abs().max())
    assert float((d0[:, :V] - d1[:, :V]).abs().max()) <= 1e-2 * float(d0[:, :V].abs().max()) + 1e-6
    assert float(d1[:, V:].abs().max()) == 0.0                # the padding columns carry no gradient

#!/usr/bin/env python3
"""Micro-benchmark of the individual C-ABI kernels at the config-2 shapes (development aid).

Each kernel is launched N times back-to-back on one stream and timed with one event pair, so
launch gaps are amortised; reports us/launch, algorithmic TFLOP/s and GB/s."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from st_amd import native as nv  # noqa: E402
from st_amd import synthetic  # noqa: E402

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32
dev = "cuda"
N_IT = int(os.environ.get("N_IT", "30"))


COLD = bool(os.environ.get("ST_COLD"))
_flush = torch.empty(256 << 20, dtype=torch.uint8, device=dev) if COLD else None


def timeit(fn, n=N_IT):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if COLD:   # every launch starts with cold caches: 256 MiB memset in between, only the kernel is timed
        tot = 0.0
        for _ in range(n):
            _flush.fill_(1)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            tot += s.elapsed_time(e)
        return tot / n * 1e3
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def rnd(*shape, dtype=BF16):
    return (torch.randn(*shape, device=dev) * 0.5).to(dtype)


def report(name, us, flops=0.0, nbytes=0.0):
    print("%-44s %9.2f us  %8.1f TFLOP/s  %8.1f GB/s" % (name, us, flops / us / 1e6, nbytes / us / 1e3))


_, _, in_len, tgt_len, _ = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
M, Md = int(in_len.sum()), int(tgt_len.sum())
d, dff, H = 256, 1024, 4
which = sys.argv[1:] or ["gemm", "attn", "misc"]

if "gemm" in which:
    for (m, n, k, tag) in [(M, 768, 256, "qkv"), (M, 1024, 256, "ffn1"), (M, 512, 256, "kv"), (Md, 768, 256, "dec qkv"),
                           (Md, 1024, 256, "dec ffn1")]:
        X, W, b, out = rnd(m, k), rnd(n, k), rnd(n, dtype=F32), torch.empty(m, n, dtype=BF16, device=dev)
        us = timeit(lambda: nv.gemm(X, W, out, bias=b, epi=nv.EPI_BF16_RELU))
        report("fwd   %-8s [%d,%d]x[%d,%d]^T" % (tag, m, k, n, k), us, 2.0 * m * n * k, 2.0 * (m * k + n * k + m * n))
    for (m, n, k, tag) in [(M, 1024, 256, "dh=ds*W2"), (M, 256, 1024, "dx=dh*W1"), (M, 768, 256, "dx=dqkv*Wqkv"),
                           (M, 256, 256, "dctx=ds*Wo"), (Md, 256, 1024, "dec dx=dh*W1")]:
        # out[m, n] = dY[m, k] W[k, n]
        dY, W, out, aux = rnd(m, k), rnd(k, n), torch.empty(m, n, dtype=BF16, device=dev), rnd(m, n)
        us = timeit(lambda: nv.gemm(dY, W, out, aux=aux, epi=nv.EPI_BF16_ADD, y_cmajor=True))
        report("dgrad %-14s [%d,%d]x[%d,%d]" % (tag, m, k, k, n), us, 2.0 * m * n * k, 2.0 * (m * k + n * k + 2 * m * n))
    from st_amd.functional import _splits
    for (m, n, k, tag) in [(M, 768, 256, "qkv"), (M, 1024, 256, "w1"), (M, 256, 1024, "w2"), (M, 256, 256, "wo"),
                           (Md, 1024, 256, "dec w1")]:
        dY, X, g = rnd(m, n), rnd(m, k), torch.zeros(n, k, dtype=F32, device=dev)
        sp = int(os.environ.get("ST_SPLITS", _splits(m, n, k)))
        us = timeit(lambda: nv.gemm(X, dY, g, epi=nv.EPI_F32_ATOMIC_T, x_cmajor=True, y_cmajor=True, splits=sp, n=n))
        report("wgrad %-8s dW[%d,%d] m=%d s=%d" % (tag, n, k, m, sp), us, 2.0 * m * n * k, 2.0 * (m * n + m * k))
    for (m, k, tag) in [(M, 256, "wo"), (M, 1024, "ffn2"), (Md, 256, "dec wo"), (Md, 1024, "dec ffn2")]:
        X, W = rnd(m, k), rnd(d, k)
        b, gm, bt = rnd(d, dtype=F32), rnd(d, dtype=F32), rnd(d, dtype=F32)
        res, out, xh = rnd(m, d), torch.empty(m, d, dtype=BF16, device=dev), torch.empty(m, d, dtype=BF16, device=dev)
        rstd = torch.empty(m, dtype=F32, device=dev)
        us = timeit(lambda: nv.gemm_ln(X, W, b, res, gm, bt, out, xh, rstd))
        report("gemm_ln %-8s [%d,%d]" % (tag, m, k), us, 2.0 * m * d * k, 2.0 * (m * k + 3 * m * d))
    for (m, k, tag) in [(M, 1024, "dh*W1"), (M, 768, "dqkv*Wqkv"), (Md, 1024, "dec dh*W1"), (Md, 256, "dec dq*Wq")]:
        dY, W, aux, xh = rnd(m, k), rnd(k, d), rnd(m, d), rnd(m, d)
        rstd, gm = rnd(m, dtype=F32).abs() + 0.5, rnd(d, dtype=F32)
        dx, acc = torch.empty(m, d, dtype=BF16, device=dev), [torch.zeros(d, dtype=F32, device=dev) for _ in range(3)]
        us = timeit(lambda: nv.gemm_lnbwd(dY, W, aux, xh, rstd, gm, dx, acc[0], acc[1], acc[2]))
        report("gemm_lnbwd %-10s [%d,%d]" % (tag, m, k), us, 2.0 * m * d * k, 2.0 * (m * k + 3 * m * d))
    # config 3 (d_model 512, d_ff 2048): the LayerNorm-fused GEMMs it runs instead of row chains
    d5 = 512
    for (m, k, tag) in [(M, 512, "wo"), (M, 2048, "ffn2")]:
        X, W = rnd(m, k), rnd(d5, k)
        b, gm, bt = rnd(d5, dtype=F32), rnd(d5, dtype=F32), rnd(d5, dtype=F32)
        res, out, xh = rnd(m, d5), torch.empty(m, d5, dtype=BF16, device=dev), torch.empty(m, d5, dtype=BF16, device=dev)
        rstd = torch.empty(m, dtype=F32, device=dev)
        us = timeit(lambda: nv.gemm_ln(X, W, b, res, gm, bt, out, xh, rstd))
        report("gemm_ln d512 %-6s [%d,%d]" % (tag, m, k), us, 2.0 * m * d5 * k, 2.0 * (m * k + 3 * m * d5))
    for (m, k, tag) in [(M, 2048, "dh*W1"), (M, 1536, "dqkv*Wqkv")]:
        dY, W, aux, xh = rnd(m, k), rnd(k, d5), rnd(m, d5), rnd(m, d5)
        rstd, gm = rnd(m, dtype=F32).abs() + 0.5, rnd(d5, dtype=F32)
        dx, acc = torch.empty(m, d5, dtype=BF16, device=dev), [torch.zeros(d5, dtype=F32, device=dev) for _ in range(3)]
        us = timeit(lambda: nv.gemm_lnbwd(dY, W, aux, xh, rstd, gm, dx, acc[0], acc[1], acc[2]))
        report("gemm_lnbwd d512 %-10s [%d,%d]" % (tag, m, k), us, 2.0 * m * d5 * k, 2.0 * (m * k + 3 * m * d5))
    for (m, n, k, tag) in [(M, 2048, 512, "h=x*W1^T"), (M, 1536, 512, "qkv"), (M, 512, 2048, "y=h*W2^T (no LN)")]:
        X, W, out = rnd(m, k), rnd(n, k), torch.empty(m, n, dtype=BF16, device=dev)
        us = timeit(lambda: nv.gemm(X, W, out, bias=rnd(n, dtype=F32)))
        report("gemm d512 %-18s [%d,%d]x[%d,%d]" % (tag, m, k, n, k), us, 2.0 * m * n * k, 2.0 * (m * k + n * k + m * n))
    # the encoder's weight gradients as the step issues them: 6 layers x (qkv, wo, w1, w2) in ONE grouped launch
    from st_amd.functional import _Deferred
    probs, fl = [], 0.0
    for _ in range(6):
        for (n, k) in ((768, 256), (256, 256), (1024, 256), (256, 1024)):
            probs.append((rnd(M, k), rnd(M, n), torch.zeros(n, k, dtype=F32, device=dev), torch.zeros(n, dtype=F32, device=dev),
                          _Deferred.splits(M), n))
            fl += 2.0 * M * n * k
    us = timeit(lambda: nv.wgrad_group(probs), n=5)
    report("wgrad_group encoder (24 problems)", us, fl, 0.0)
    from st_amd.functional import _wide_plan
    wide = _wide_plan(probs)
    us = timeit(lambda: [nv.wgrad_group(w, wide=True) for w in wide], n=5)
    report("wgrad_wide  encoder (24 problems, %d token splits)" % wide[0][0][4], us, fl, 0.0)

if "attn" in which:
    def offs(lens):
        o = torch.zeros_like(lens)
        o[1:] = torch.cumsum(lens, 0)[:-1]
        return o.to(dev, I32), lens.to(dev, I32)

    qo, ql = offs(in_len)
    to, tl = offs(tgt_len)
    scale = 1 / math.sqrt(64)
    from st_amd.functional import Rows, attn_work
    in_rows, t_rows = Rows.packed(in_len, dev), Rows.packed(tgt_len, dev)
    use_work = not os.environ.get("ST_NO_WORK")
    cases = [("enc self", M, M, qo, ql, qo, ql, int(in_len.max()), int(in_len.max()), False, True, in_rows, in_rows),
             ("dec self causal", Md, Md, to, tl, to, tl, 50, 50, True, True, t_rows, t_rows),
             ("cross", Md, M, to, tl, qo, ql, 50, int(in_len.max()), False, False, t_rows, in_rows)]
    for name, mq, mk, q_off, q_len, k_off, k_len, maxq, maxk, causal, self_attn, qr, kr in cases:
        wf, wq, wk = attn_work(qr, kr, causal, 64, H) if use_work else (None, None, None)
        if self_attn:
            qkv = rnd(mq, 3 * d)
            Q, K, V = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        else:
            Q, kv = rnd(mq, d), rnd(mk, 2 * d)
            K, V = kv[:, :d], kv[:, d:]
        O, dO = torch.empty(mq, d, dtype=BF16, device=dev), rnd(mq, d)
        lse, delta = torch.empty(H * mq, dtype=F32, device=dev), torch.empty(H * mq, dtype=F32, device=dev)
        dQ = torch.empty(mq, d, dtype=BF16, device=dev)
        dK, dV = torch.empty(mk, d, dtype=BF16, device=dev), torch.empty(mk, d, dtype=BF16, device=dev)
        pairs = float((q_len.double() * k_len.double()).sum()) * (0.5 if causal else 1.0)
        fl = 4.0 * pairs * d
        us = timeit(lambda: nv.attn_fwd(Q, K, V, O, lse, q_off, q_len, k_off, k_len, H, maxq, causal, scale, work=wf, max_k=maxk))
        report("attn fwd  " + name, us, fl)
        for part, nm in ((1, "dq "), (2, "dkv"), (3, "all")):
            us = timeit(lambda: nv.attn_bwd(Q, K, V, None if part == 3 else O, dO, lse, delta, dQ, dK, dV, q_off, q_len, k_off, k_len, H, maxq,
                                            maxk, causal, scale, parts=part, work_q=wq, work_k=wk))
            report("attn bwd %s " % nm + name, us, fl)

if "misc" in which:
    gbuf, scratch, out1 = torch.randn(13_000_000, device=dev), nv.grad_norm_scratch(dev), torch.zeros((), device=dev)
    report("grad_norm [13 M fp32]", timeit(lambda: nv.grad_norm(gbuf, scratch, out1)), 0, 4.0 * gbuf.numel())
    for m in (M, Md):
        dy, xh, rs, gm = rnd(m, d), rnd(m, d), rnd(m, dtype=F32).abs() + 0.5, rnd(d, dtype=F32)
        dx = torch.empty(m, d, dtype=BF16, device=dev)
        a, b, c = (torch.zeros(d, dtype=F32, device=dev) for _ in range(3))
        us = timeit(lambda: nv.ln_bwd(dy, xh, rs, gm, dx, a, b, c))
        report("ln_bwd [%d,%d]" % (m, d), us, 0, 2.0 * 3 * m * d)

if "chain" in which:
    # row chains (csrc/st_rowchain.hip): the encoder layer's chain at 24,060 rows and the decoder's at 1,206, forward and
    # backward, next to the separate kernels they replace (tools/dev/chain_bench.py has the element-wise comparison)
    from st_amd import chains
    d_, dff = 256, 1024
    wo, wqkv, w1, w2 = rnd(d_, d_) * 0.1, rnd(3 * d_, d_) * 0.1, rnd(dff, d_) * 0.1, rnd(d_, dff) * 0.1
    vec = lambda n: torch.randn(n, device=dev) * 0.1
    bo, bqkv, b1, b2, g0, be0, g1, be1 = vec(d_), vec(3 * d_), vec(dff), vec(d_), vec(d_) + 1, vec(d_), vec(d_) + 1, vec(d_)
    cs = chains.ChainSet(dev)
    cf = cs.add(chains.blocks_of(wo) + chains.ffn_blocks(w1, w2) + chains.blocks_of(wqkv))
    cb = cs.add(chains.t_blocks(chains.blocks_of(wqkv)) + chains.ffn_blocks_bwd(w1, w2) + chains.t_blocks(chains.blocks_of(wo)))
    cs.finalize().rebuild()
    chf, chb = cs.chain(cf), cs.chain(cb)
    E = lambda *s, dtype=BF16: torch.empty(*s, dtype=dtype, device=dev)
    for rows in (M, Md) + tuple(int(r) for r in os.environ.get("ST_CHAIN_ROWS", "").split(",") if r):
        ctx, x = rnd(rows, d_), rnd(rows, d_)
        c_, xc, rc, h_, y_, xy, ry, p_ = E(rows, d_), E(rows, d_), E(rows, dtype=F32), E(rows, dff), E(rows, d_), E(rows, d_), E(rows, dtype=F32), E(rows, 3 * d_)
        us = timeit(lambda: nv.row_chain(ctx, chf, pre=(x, bo, g0, be0, c_, xc, rc), ffn=(dff, b1, b2, g1, be1, h_, y_, xy, ry, None, None),
                                         post=(3, bqkv, p_)))
        fl = 2.0 * rows * 12 * 256 * 256
        report("row_chain fwd  wo+LN, FFN, qkv [%d]" % rows, us, fl, 2.0 * rows * (2 * d_ + 4 * d_ + dff + 3 * d_))
        dqkv, dss, Hm, O_, Or = rnd(rows, 3 * d_), rnd(rows, d_), torch.relu(rnd(rows, dff)), rnd(rows, d_), rnd(rows, d_) * 0.004
        ra, rb = torch.rand(rows, device=dev) + 0.5, torch.rand(rows, device=dev) + 0.5
        dsa, dH, dsb, dctx, delta = E(rows, d_), E(rows, dff), E(rows, d_), E(rows, d_), E(4 * rows, dtype=F32)
        acc = [torch.zeros(d_, device=dev) for _ in range(6)]
        if os.environ.get("ST_NO_DBIAS"):      # (development: what the chains' bias-gradient atomics cost)
            acc[2] = acc[5] = None
        if os.environ.get("ST_NO_COLSUMS"):
            acc = [None] * 6
        bits = nv.relu_bits_from(Hm)
        us = timeit(lambda: nv.row_chain_bwd(chb, rows, head=(3, dqkv, dss, xc, ra, g0, None, dsa, acc[0], acc[1], acc[2]),
                                             ffn=(dff, bits, 1.0, dH, xy, rb, g1, dsb, acc[3], acc[4], acc[5]), tail=(O_, Or, dctx, delta)))
        report("row_chain bwd  qkv^T+LNbwd, FFN^T+LNbwd, wo^T+delta [%d]" % rows, us, fl,
               2.0 * rows * (3 * d_ + 2 * d_ + dff + d_ + 2 * d_ + d_ + dff + d_ + d_))

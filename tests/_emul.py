"""TEST-ONLY torch emulation of the C-ABI kernels (st_amd.native).

The container that builds this repo has no GPU, so the *composition* of the
kernels (st_amd.functional: which buffer feeds which GEMM, which gradient lands
in which arena slot) is checked on CPU by swapping every native entry point for
a plain-torch emulation with the same signature and the same bf16 storage
rounding.  Nothing in the product imports this file; on a GPU box the real
kernels run and are compared with the oracle directly (-m gpu tests).
"""
import contextlib
import math

import torch

from st_amd import arena as st_arena
from st_amd import native as nv

BF16, F32 = torch.bfloat16, torch.float32

# ---- dropout: the same counter-based masks as csrc/st_common.cuh (st_hash32, Drop, drop_counter_*) ----------
_M32 = 0xFFFFFFFF


def _hash32(x):
    """lowbias32 on an int64 tensor holding uint32 values (int64 products wrap; the low 32 bits are exact)."""
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7feb352d) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846ca68b) & _M32
    x = x ^ (x >> 16)
    return x


class Drop:
    """CPU stand-in for st_amd.native.Drop (same fields, seed may live on the CPU)."""

    def __init__(self, seed, salt, p):
        self.seed, self.salt = seed, int(salt) & _M32
        self.thresh = min(255, int(round(256.0 * p)))
        self.scale = 256.0 / (256 - self.thresh)


def _on(d):
    return d is not None and d.thresh > 0


def _key(d):
    seed = int(d.seed.reshape(-1)[0].item()) & _M32
    return int(_hash32(torch.tensor((seed + d.salt * 0x9e3779b9) & _M32, dtype=torch.int64)))


def keep_rc(d, rows, cols, ncols):
    """bool [len(rows), len(cols)]: element (row, col) of a row matrix with ``ncols`` columns survives."""
    rows, cols = rows.to(torch.int64).view(-1, 1), cols.to(torch.int64).view(1, -1)
    cnt = (rows * (ncols >> 2) + (cols >> 2)) & _M32
    bits = _hash32(cnt ^ _key(d))
    return ((bits >> (8 * (cols & 3))) & 0xFF) >= d.thresh


def keep_qk(d, bh, nq, nk):
    """bool [nq, nk]: attention probability (q, k) of head slot bh survives."""
    q, k = torch.arange(nq, dtype=torch.int64).view(-1, 1), torch.arange(nk, dtype=torch.int64).view(1, -1)
    cnt = ((((q >> 1) << 15) | (k >> 1)) + bh * 0x85ebca6b) & _M32
    bits = _hash32(cnt ^ _key(d))
    return ((bits >> (8 * (2 * (q & 1) + (k & 1)))) & 0xFF) >= d.thresh



def gemm(X, Y, out, bias=None, aux=None, epi=nv.EPI_BF16, x_cmajor=False, y_cmajor=False, splits=1, m=None, n=None,
         kc=None, drop=None, delta=None, head_dim=0, stack=None, aux2=None):
    if stack is not None:       # Y / bias: the first of `blocks` equally spaced blocks (st_gemm_stacked)
        blocks, y_stride, b_stride = stack
        Y = torch.as_strided(Y, (blocks, Y.shape[0], Y.shape[1]), (y_stride, Y.stride(0), 1)).reshape(-1, Y.shape[1])
        if bias is not None:
            bias = torch.as_strided(bias, (blocks, bias.shape[0]), (b_stride, 1)).reshape(-1)
    Xl = (X.t() if x_cmajor else X).float()      # logical [M, Kc]
    Yl = (Y.t() if y_cmajor else Y).float()      # logical [N, Kc]
    M = m if m is not None else Xl.shape[0]
    N = n if n is not None else Yl.shape[0]
    Kc = kc if kc is not None else Xl.shape[1]
    Xl = torch.nn.functional.pad(Xl, (0, max(0, Kc - Xl.shape[1]), 0, max(0, M - Xl.shape[0])))[:M, :Kc]
    Yl = torch.nn.functional.pad(Yl, (0, max(0, Kc - Yl.shape[1]), 0, max(0, N - Yl.shape[0])))[:N, :Kc]
    acc = Xl @ Yl.t()
    if epi == nv.EPI_F32_ATOMIC_T:
        if bias is not None:
            bias[:N] += Yl.sum(1)      # fused bias gradient
    elif bias is not None:
        acc = acc + bias[:N]
    if epi == nv.EPI_BF16_RELU:
        acc = torch.relu(acc)
        if _on(drop):
            acc = acc * keep_rc(drop, torch.arange(M), torch.arange(N), N) * drop.scale
    elif epi == nv.EPI_BF16_MASK:
        acc = acc * (aux[:M, :N].float() > 0) * (drop.scale if _on(drop) else 1.0)
    elif epi == nv.EPI_BF16_ADD:
        acc = acc + aux[:M, :N].float()
    if epi == nv.EPI_F32_ATOMIC:
        out[:M, :N] += acc
    elif epi == nv.EPI_F32_ATOMIC_T:
        out[:N, :M] += acc.t()
    else:
        out[:M, :N] = acc.to(out.dtype)
        if epi == nv.EPI_BF16_DELTA:
            o_hi = aux[:M, :N].float() + (aux2[:M, :N].float() if aux2 is not None else 0.0)
            prod = out[:M, :N].float() * o_hi
            delta.view(N // head_dim, -1)[:, :M] = prod.view(M, N // head_dim, head_dim).sum(-1).t()
    return out


def gemm_splitk(X, Y, out, splits, y_cmajor=False):
    return gemm(X, Y, out, y_cmajor=y_cmajor)


def gemm_kscale(X, W, out, bias, col_lo, col_hi, scale):
    acc = X.float() @ W.float().t() + (bias.float() if bias is not None else 0.0)
    acc[:, col_lo:col_hi] *= float(scale)
    out[:X.shape[0], :W.shape[0]] = acc.to(BF16)
    return out


def gemm_ws(X, W, out, bias=None, relu=False, drop=None, stack=None):
    return gemm(X, W, out, bias=bias, epi=nv.EPI_BF16_RELU if relu else nv.EPI_BF16, drop=drop, stack=stack)


def adam_clip(p, g, m, v, lr, step, gnorm, max_norm, beta1, beta2, eps, grad_scale=1.0):
    coef = (1.0 if gnorm is None else min(float(max_norm) / (float(gnorm) + 1e-6), 1.0)) * float(grad_scale)
    g.mul_(coef)
    m.lerp_(g, 1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    t = float(step)
    bc1, bc2 = 1.0 - beta1 ** t, 1.0 - beta2 ** t
    p.addcdiv_(m, v.sqrt() / math.sqrt(bc2) + eps, value=-float(lr) / bc1)


def wgrad_group(problems, wide=False):
    for x, dy, gw, gb, sp, rows in problems:
        gemm(x, dy, gw, bias=gb, epi=nv.EPI_F32_ATOMIC_T, x_cmajor=True, y_cmajor=True, splits=sp, n=rows)


def gemm_ln(X, W, bias, res, gamma, beta, out, xhat, rstd, eps=1e-6, relu=False, pe=None, pos=None, pre=None,
            drop=None, drop_where=0):
    v = X.float() @ W.float().t() + bias
    M, N = v.shape
    if relu:
        v = torch.relu(v)
    if _on(drop) and drop_where == 1:
        v = v * keep_rc(drop, torch.arange(M), torch.arange(N), N) * drop.scale
    if res is not None:
        v = v + res.float()
    mu = v.mean(-1, keepdim=True)
    var = ((v - mu) ** 2).mean(-1, keepdim=True)
    rs = torch.rsqrt(var + eps)
    h = (v - mu) * rs
    y = h * gamma + beta
    if pe is not None:
        y = y + pe[pos.long()]
    if _on(drop) and drop_where == 2:
        y = y * keep_rc(drop, torch.arange(M), torch.arange(N), N) * drop.scale
    out.copy_(y.to(BF16))
    if xhat is not None:
        xhat.copy_(h.to(BF16))
    if rstd is not None:
        rstd.copy_(rs.squeeze(-1))
    if pre is not None:
        pre.copy_(v.to(BF16))
    return out


def wfrag_depth():
    return 16


def wfrag_build(table):
    return None       # the emulated chain reads the weight blocks themselves (chains.Chain.blocks)


def chain_mask_words(M, d_ff, d_model=256):
    """The real kernels' buffer size (include/st_hip.h: st_row_chain_mask_words); the emulation packs one bit per hidden
    value row-major into its first M * d_ff / 64 words."""
    if d_model == 512:
        return ((M + 63) // 64) * (d_ff // 256) * 8 * 64
    mt = 1 if (M + 31) // 32 <= 256 else 2 if (M + 63) // 64 <= 256 else 3
    return ((M + 32 * mt - 1) // (32 * mt)) * (d_ff // 256) * 8 * 64


def _pack_bits(mask, out):
    import numpy as np
    w = np.packbits(mask.cpu().numpy().astype(np.uint8).reshape(-1), bitorder="little")
    w = np.concatenate([w, np.zeros((-len(w)) % 8, dtype=np.uint8)]).view(np.int64)
    out[:len(w)] = torch.from_numpy(w.copy())


def _unpack_bits(words, M, d_ff):
    import numpy as np
    b = np.unpackbits(words.cpu().numpy().view(np.uint8), bitorder="little")[:M * d_ff]
    return torch.from_numpy(b.astype(np.bool_)).view(M, d_ff)


def relu_bits_from(H, d_model=256):
    out = torch.zeros(chain_mask_words(H.shape[0], H.shape[1], d_model), dtype=torch.int64)
    _pack_bits(H.float() > 0, out)
    return out


def row_chain(A, chain, pre=None, ffn=None, post=None, eps=1e-6, post_kscale=0.0):
    """csrc/st_rowchain.hip as a composition of the emulated kernels it replaces (same roundings: every intermediate the
    separate kernels round to bf16 is rounded here too)."""
    blocks = list(chain.blocks)
    if A.shape[1] == 512:      # csrc/st_rowchain_pipe512.cuh: the blocks of one GEMM are pieces of ONE weight tensor (chains.encoder512_blocks)
        R, bo, g0, be0, out0, xhat0, rstd0 = pre
        relu_bits = ffn[11] if len(ffn) == 12 else None
        d_ff, b1, b2, g1, be1, H, out1, xhat1, rstd1, drop1, drop2 = ffn[:11]
        wo, w1, w2 = blocks[0][0], blocks[4][0], blocks[6][0]
        assert len(blocks) == 4 + 4 * (d_ff // 256) + (12 if post else 0)
        M = A.shape[0]
        out0 = torch.empty(M, 512, dtype=BF16, device=A.device) if out0 is None else out0
        gemm_ln(A, wo, bo, R[:M], g0, be0, out0, xhat0, rstd0, eps=eps)
        H = torch.empty(M, d_ff, dtype=BF16, device=A.device) if H is None else H
        gemm(out0, w1, H, bias=b1, epi=nv.EPI_BF16_RELU, drop=drop1)
        if relu_bits is not None:
            _pack_bits(H[:M].float() > 0, relu_bits)
        gemm_ln(H, w2, b2, out0, g1, be1, out1, xhat1, rstd1, eps=eps, drop=drop2, drop_where=2 if _on(drop2) else 0)
        if post:
            nb, bp, P = post
            wp = blocks[4 + 4 * (d_ff // 256)][0]
            if post_kscale not in (0.0, 1.0):
                acc = out1[:M].float() @ wp.float().t() + bp.float()
                acc[:, 512:1024] *= float(post_kscale)
                P[:M] = acc.to(BF16)
            else:
                gemm(out1, wp, P, bias=bp)
        return

    def take(n):
        out = [w[n0:n0 + 256, k0:k0 + 256] for w, n0, k0 in blocks[:n]]
        del blocks[:n]
        return out

    cur = A
    if pre:
        R, bo, g0, be0, out0, xhat0, rstd0 = pre
        (wo,) = take(1)
        if out0 is None:
            out0 = torch.empty(A.shape[0], 256, dtype=BF16, device=A.device)
        gemm_ln(A, wo, bo, R[:A.shape[0]], g0, be0, out0, xhat0, rstd0, eps=eps)
        cur = out0
    if ffn:
        relu_bits = ffn[11] if len(ffn) == 12 else None
        d_ff, b1, b2, g1, be1, H, out1, xhat1, rstd1, drop1, drop2 = ffn[:11]
        ws = take(2 * (d_ff // 256))
        w1 = torch.cat(ws[0::2], 0)
        w2 = torch.cat(ws[1::2], 1)
        if H is None:          # inference: the hidden activation is not handed out
            H = torch.empty(A.shape[0], d_ff, dtype=BF16, device=A.device)
        gemm(cur, w1, H, bias=b1, epi=nv.EPI_BF16_RELU, drop=drop1)
        if relu_bits is not None:
            _pack_bits(H[:A.shape[0]].float() > 0, relu_bits)
        gemm_ln(H, w2, b2, cur, g1, be1, out1, xhat1, rstd1, eps=eps, drop=drop2, drop_where=2 if _on(drop2) else 0)
        cur = out1
    if post:
        nb, bp, P = post
        if nb == 3 and post_kscale not in (0.0, 1.0):       # the key block leaves scaled, in fp32, before its one rounding
            wp = torch.cat(take(nb), 0)
            acc = cur[:A.shape[0]].float() @ wp.float().t() + bp.float()
            acc[:, 256:512] *= float(post_kscale)
            P[:A.shape[0]] = acc.to(BF16)
        else:
            gemm(cur, torch.cat(take(nb), 0), P, bias=bp)
    assert not blocks


def row_chain_bwd(chain, M, head=None, ds_in=None, ffn=None, tail=None):
    """csrc/st_rowchain.hip's backward chain as a composition of the emulated kernels it replaces."""
    blocks = list(chain.blocks)
    if ffn is not None and ffn[4] is not None and ffn[4].shape[1] == 512:      # csrc/st_rowchain_pipe512_bwd.cuh (chains.encoder512_blocks_bwd)
        nb, dP, G, xa, ra, ga, drop, dsa, dga, dba, dbia = head
        d_ff, relu_bits, msc, dH, xb, rb, gb, dsb, dgb, dbb, dbib = ffn
        O, Ores, dctx, delta = tail
        assert all(b[3] for b in blocks) and len(blocks) == 2 * nb + 4 * (d_ff // 256) + 4
        w2, w1, wo = blocks[2 * nb][0], blocks[2 * nb + 2][0], blocks[-1][0]
        if nb:
            gemm_lnbwd(dP[:M], blocks[0][0], None if G is None else G[:M], xa, ra[:M], ga, dsa[:M], dga, dba, dbia, drop=drop)
        else:
            ln_bwd(G[:M], xa[:M], ra[:M], ga, dsa[:M], dga, dba, dbia, drop=drop)
        ds = dsa[:M]
        acc = ds.float() @ w2.float()
        dH[:M] = ((acc * msc).to(BF16).float() * _unpack_bits(relu_bits, M, d_ff)).to(BF16)
        gemm_lnbwd(dH[:M], w1, ds, xb, rb[:M], gb, dsb[:M], dgb, dbb, dbib)
        dl = torch.zeros(8 * M, dtype=torch.float32)
        out = torch.zeros(M, 512, dtype=BF16)
        gemm(dsb[:M], wo, out, epi=nv.EPI_BF16_DELTA, aux=O[:M], aux2=None if Ores is None else Ores[:M], y_cmajor=True, delta=dl, head_dim=64)
        dctx[:M] = out
        delta.view(8, -1)[:, :M] = dl.view(8, M)
        return

    def take(n):
        out = [w[n0:n0 + 256, k0:k0 + 256] for w, n0, k0, tr in blocks[:n]]
        assert all(b[3] for b in blocks[:n]), "backward chains read transposed blocks"
        del blocks[:n]
        return out

    ds = ds_in[:M] if ds_in is not None else None
    if head:
        nb, dP, G, xa, ra, ga, drop, dsa, dga, dba, dbia = head
        if nb:
            wp = torch.cat(take(nb), 0)                          # [256 nb (contraction), 256]
            gemm_lnbwd(dP[:M], wp, None if G is None else G[:M], xa, ra[:M], ga, dsa[:M], dga, dba, dbia, drop=drop)
        else:
            ln_bwd(G[:M], xa[:M], ra[:M], ga, dsa[:M], dga, dba, dbia, drop=drop)
        ds = dsa[:M]
    if ffn:
        d_ff, relu_bits, msc, dH, xb, rb, gb, dsb, dgb, dbb, dbib = ffn
        ws = take(2 * (d_ff // 256))
        w2 = torch.cat(ws[0::2], 1)                          # [256 (contraction), d_ff]
        w1 = torch.cat(ws[1::2], 0)                          # [d_ff (contraction), 256]
        acc = ds.float() @ w2.float()
        dH[:M] = ((acc * msc).to(BF16).float() * _unpack_bits(relu_bits, M, d_ff)).to(BF16)
        gemm_lnbwd(dH[:M], w1, ds, xb, rb[:M], gb, dsb[:M], dgb, dbb, dbib)
        ds = dsb[:M]
    if tail:
        O, Ores, dctx, delta = tail
        (wo,) = take(1)
        dl = torch.zeros(4 * M, dtype=torch.float32)
        out = torch.zeros(M, 256, dtype=BF16)
        gemm(ds, wo, out, epi=nv.EPI_BF16_DELTA, aux=O[:M], aux2=None if Ores is None else Ores[:M], y_cmajor=True, delta=dl,
             head_dim=64)
        dctx[:M] = out
        delta.view(4, -1)[:, :M] = dl.view(4, M)
    assert not blocks


def gemm_lnbwd(dY, W, aux, xhat, rstd, gamma, dx, dgamma=None, dbeta=None, dbias=None, drop=None):
    dy = torch.zeros(dY.shape[0], W.shape[1], dtype=BF16)
    gemm(dY, W, dy, aux=aux, epi=nv.EPI_BF16_ADD if aux is not None else nv.EPI_BF16, y_cmajor=True)
    return ln_bwd(dy, xhat[:dY.shape[0]], rstd, gamma, dx, dgamma, dbeta, dbias, drop=drop)


def ln_bwd(dy, xhat, rstd, gamma, dx, dgamma=None, dbeta=None, dbias=None, mask=None, drop=None, mask_scale=1.0):
    d, xh = dy.float(), xhat.float()
    if _on(drop):
        d = d * keep_rc(drop, torch.arange(d.shape[0]), torch.arange(d.shape[1]), d.shape[1]) * drop.scale
    g = d * gamma
    v = rstd.unsqueeze(-1) * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))
    if mask is not None:
        v = v * (mask.float() > 0) * mask_scale
    dx.copy_(v.to(BF16))
    if dgamma is not None:
        dgamma += (d * xh).sum(0)
    if dbeta is not None:
        dbeta += d.sum(0)
    if dbias is not None:
        dbias += v.sum(0)
    return dx


def _attn_core(Q, K, V, q_off, q_len, k_off, k_len, H, causal, scale, b, h, k_prescaled=False):
    dk = Q.shape[1] // H
    qo, ql, ko, kl = int(q_off[b]), int(q_len[b]), int(k_off[b]), int(k_len[b])
    cs = slice(h * dk, (h + 1) * dk)
    q = Q[qo:qo + ql, cs].float()
    k = K[ko:ko + kl, cs].float()
    if k_prescaled:       # K holds scale * log2(e) * k: the emulation works on k (the kernels never undo the scale - they skip theirs)
        k = k / (scale * nv.K_LOG2_SCALE)
    v = V[ko:ko + kl, cs].float()
    s = q @ k.t() * scale
    if causal:
        s = s.masked_fill(torch.ones(ql, kl, dtype=torch.bool).triu(1), float("-inf"))
    return q, k, v, s, slice(qo, qo + ql), slice(ko, ko + kl), cs


def attn_fwd(Q, K, V, O, lse, q_off, q_len, k_off, k_len, n_head, max_q, causal, scale, work=None, drop=None,
             max_k=0, ores=None, k_prescaled=False):
    rows = Q.shape[0]
    for b in range(q_off.numel()):
        for h in range(n_head):
            q, k, v, s, qs, ks, cs = _attn_core(Q, K, V, q_off, q_len, k_off, k_len, n_head, causal, scale, b, h, k_prescaled)
            p = torch.softmax(s, -1)
            if _on(drop):   # the kernel drops un-normalised weights and folds 1/(1-p) into the final 1/l
                p = p * keep_qk(drop, b * n_head + h, *s.shape)
                o32 = p.to(BF16).float() @ v * drop.scale
                O[qs, cs] = o32.to(BF16)
                if ores is not None:
                    ores[qs, cs] = (o32 - O[qs, cs].float()).to(BF16)
                lse.view(n_head, rows)[h, qs] = torch.logsumexp(s, -1) / math.log(2.0)
                continue
            if ores is not None and max_q <= 64:     # the kernel's psplit mode: P as hi + lo bf16 terms
                hi = p.to(BF16).float()
                o32 = (hi + (p - hi).to(BF16).float()) @ v
            else:
                o32 = p.to(BF16).float() @ v
            O[qs, cs] = o32.to(BF16)
            if ores is not None:
                ores[qs, cs] = (o32 - O[qs, cs].float()).to(BF16)
            lse.view(n_head, rows)[h, qs] = torch.logsumexp(s, -1) / math.log(2.0)
    return O


def attn_f1_fwd(A, chain, pre, post, K, V, O, lse, q_off, q_len, k_off, k_len, n_head, max_q, scale, work=None, drop=None,
                max_k=0, ores=None, eps=1e-6):
    """st_attn_f1_fwd = the two launches it stands for."""
    row_chain(A, chain, pre=pre, post=post, eps=eps)
    return attn_fwd(post[2], K, V, O, lse, q_off, q_len, k_off, k_len, n_head, max_q, False, scale, work=work, drop=drop,
                    max_k=max_k, ores=ores)


def attn_sf1_fwd(qkv, Os, lses, pre, post, chain, K, V, O, lse, q_off, q_len, k_off, k_len, n_head, max_q, scale, work_self=None,
                 work=None, drop_self=None, drop=None, max_k=0, ores_self=None, ores=None, eps=1e-6):
    """st_attn_sf1_fwd = the three launches it stands for."""
    d = qkv.shape[1] // 3
    attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], Os, lses, q_off, q_len, q_off, q_len, n_head, max_q, True, scale,
             work=work_self, drop=drop_self, max_k=max_q, ores=ores_self)
    return attn_f1_fwd(Os, chain, pre, post, K, V, O, lse, q_off, q_len, k_off, k_len, n_head, max_q, scale, work=work, drop=drop,
                       max_k=max_k, ores=ores, eps=eps)


def attn_bwd(Q, K, V, O, dO, lse, delta, dQ, dK, dV, q_off, q_len, k_off, k_len, n_head, max_q, max_k, causal, scale,
             parts=3, work_q=None, work_k=None, drop=None, k_prescaled=False):
    rows = Q.shape[0]
    for b in range(q_off.numel()):
        for h in range(n_head):
            q, k, v, s, qs, ks, cs = _attn_core(Q, K, V, q_off, q_len, k_off, k_len, n_head, causal, scale, b, h, k_prescaled)
            p = torch.exp2(s / math.log(2.0) - lse.view(n_head, rows)[h, qs].unsqueeze(-1))
            do = dO[qs, cs].float()
            if O is not None:
                dl = (do * O[qs, cs].float()).sum(-1, keepdim=True)
                delta.view(n_head, rows)[h, qs] = dl.squeeze(-1)
            else:
                dl = delta.view(n_head, rows)[h, qs].unsqueeze(-1)
            dp = do @ v.t()
            pv = p
            if _on(drop):
                keep = keep_qk(drop, b * n_head + h, *s.shape) * drop.scale
                dp, pv = dp * keep, p * keep
            ds = (p * (dp - dl)).to(BF16).float()
            dQ[qs, cs] = (ds @ k * scale).to(BF16)
            dK[ks, cs] = (ds.t() @ q * scale).to(BF16)
            dV[ks, cs] = (pv.to(BF16).float().t() @ do).to(BF16)


def _dense_core(Q, K, V, mask, b, h, n_head, Lq, Lk, scale, drop):
    dk = Q.shape[1] // n_head
    cs = slice(h * dk, (h + 1) * dk)
    qs, ks = slice(b * Lq, (b + 1) * Lq), slice(b * Lk, (b + 1) * Lk)
    q, k, v = Q[qs, cs].float(), K[ks, cs].float(), V[ks, cs].float()
    s = q @ k.t() * scale
    if mask is not None:
        s = s.masked_fill(mask[b].bool(), float("-inf"))
    dead = torch.isinf(s).all(-1, keepdim=True)                 # every key masked: zero context, zero gradients
    p = torch.where(dead, torch.zeros_like(s), torch.softmax(torch.where(dead, torch.zeros_like(s), s), -1))
    keep = keep_qk(drop, b * n_head + h, Lq, Lk).float() * drop.scale if _on(drop) else torch.ones_like(p)
    return q, k, v, s, p, keep, qs, ks, cs, dead


def attn_dense_fwd(Q, K, V, mask, O, lse, B, n_head, Lq, Lk, scale, drop=None, want_probs=False):
    P = torch.zeros(B, n_head, Lq, Lk) if want_probs else None
    for b in range(B):
        for h in range(n_head):
            q, k, v, s, p, keep, qs, ks, cs, dead = _dense_core(Q, K, V, mask, b, h, n_head, Lq, Lk, scale, drop)
            O[qs, cs] = ((p * keep) @ v).to(BF16)
            l = torch.logsumexp(torch.where(dead, torch.zeros_like(s), s), -1)
            lse.view(n_head, B * Lq)[h, qs] = torch.where(dead.squeeze(-1), torch.full_like(l, float("inf")), l)
            if P is not None:
                P[b, h] = p
    return P


def attn_dense_bwd(Q, K, V, mask, dO, lse, delta, dQ, dK, dV, B, n_head, Lq, Lk, scale, drop=None):
    for b in range(B):
        for h in range(n_head):
            q, k, v, s, p, keep, qs, ks, cs, dead = _dense_core(Q, K, V, mask, b, h, n_head, Lq, Lk, scale, drop)
            do = dO[qs, cs].float()
            dp = (do @ v.t()) * keep
            dl = (p * dp).sum(-1, keepdim=True)
            delta.view(n_head, B * Lq)[h, qs] = dl.squeeze(-1)
            ds = p * (dp - dl)
            dQ[qs, cs] = (ds @ k * scale).to(BF16)
            dK[ks, cs] = (ds.t() @ q * scale).to(BF16)
            dV[ks, cs] = ((p * keep).t() @ do).to(BF16)


def ctc_gather(logits, rowmap, T, cols, lse, lp, V=None):
    V = logits.shape[1] if V is None else V
    l = torch.logsumexp(logits[:, :V].float(), dim=1)
    lse.copy_(l)
    ok = rowmap >= 0
    b = torch.div(rowmap[ok], T, rounding_mode="floor")
    rows = torch.nonzero(ok).flatten()
    lp.view(-1, lp.shape[2])[rowmap[ok]] = logits[rows.view(-1, 1), cols[b].long()] - l[rows].view(-1, 1)


def ctc_dlogits(logits, lse, rowmap, T, roww, scat, gsmall, grad_out, dlogits, V=None):
    V = logits.shape[1] if V is None else V
    ok = rowmap >= 0
    b = torch.div(rowmap.clamp_min(0), T, rounding_mode="floor")
    w = torch.where(ok, roww[b], torch.zeros_like(lse)) * grad_out.reshape(())
    dense = torch.exp(logits[:, :V].float() - lse.view(-1, 1)) * w.view(-1, 1)
    dlogits.zero_()
    dlogits[:, :V] = dense.to(BF16)
    g = gsmall.view(-1, gsmall.shape[2])
    for r in torch.nonzero(ok).flatten().tolist():
        sc = scat[int(b[r])]
        keep = sc >= 0
        dlogits[r, sc[keep].long()] = (g[int(rowmap[r])][keep] * grad_out.reshape(())).to(BF16)


def attn_probs(Q, K, q_off, q_len, k_off, k_len, n_head, Lq, Lk, causal, scale, k_prescaled=False):
    B = q_off.numel()
    d_k = Q.shape[1] // n_head
    P = torch.zeros(B, n_head, int(Lq), int(Lk))
    for b in range(B):
        lq, lk, qo, ko = int(q_len[b]), int(k_len[b]), int(q_off[b]), int(k_off[b])
        for h in range(n_head):
            q = Q[qo:qo + lq, h * d_k:(h + 1) * d_k].float()
            k = K[ko:ko + lk, h * d_k:(h + 1) * d_k].float()
            s = q @ k.t() * (math.log(2.0) if k_prescaled else scale)
            if causal:
                s = s.masked_fill(torch.ones(lq, lk, dtype=torch.bool).triu(1), float("-inf"))
            P[b, h, :lq, :lk] = torch.softmax(s, -1)
    return P


def feat_stack(x, in_len, stats, left, right, interval, out_off, out_len, max_out_len, out):
    B, T, F = x.shape
    for b in range(B):
        n = int(in_len[b])
        v = x[b, :n].double()
        if stats is not None:
            st = stats[b].double()
            mean = st[0, :-1] / st[0, -1]
            v = (v - mean) / torch.sqrt(st[1, :-1] / st[0, -1] - mean * mean)
        v = v.float()
        stk = torch.zeros(n, F * (1 + left + right))
        stk[:, left * F:(left + 1) * F] = v
        for i in range(left):
            stk[i + 1:n, (left - i - 1) * F:(left - i) * F] = v[0:n - i - 1]
        for i in range(right):
            stk[0:n - i - 1, (right + i + 1) * F:(right + i + 2) * F] = v[i + 1:n]
        sub = stk[::interval]
        o = int(out_off[b])
        out[o:o + sub.shape[0], :sub.shape[1]] = sub.to(BF16)
    return out


def row_index(off, length, max_len, row_pos, row_seq=None):
    for b in range(off.numel()):
        n, o = int(length[b]), int(off[b])
        row_pos[o:o + n] = torch.arange(n, dtype=row_pos.dtype)
        if row_seq is not None:
            row_seq[o:o + n] = b
    return row_pos


def pack_rows(x, off, length, out):
    for b in range(off.numel()):
        n, o = int(length[b]), int(off[b])
        out[o:o + n] = x[b, :n].to(BF16)
    return out


def unpack_rows(x, off, length, out):
    out.zero_()
    for b in range(off.numel()):
        n, o = int(length[b]), int(off[b])
        out[b, :n] = x[o:o + n].float()
    return out


def pack_grad(g, off, length, out):
    for b in range(off.numel()):
        n, o = int(length[b]), int(off[b])
        out[o:o + n] = g[b, :n].to(BF16)
    return out


def embed_pe_fwd(tok, emb, pe, off, length, out):
    for b in range(off.numel()):
        n, o = int(length[b]), int(off[b])
        out[o:o + n] = (emb[tok[b, :n]] + pe[:n]).to(BF16)
    return out


def embed_bwd(tok, dy, off, length, pad_idx, demb):
    for b in range(off.numel()):
        n, o = int(length[b]), int(off[b])
        ids = tok[b, :n]
        keep = ids != pad_idx
        demb.index_add_(0, ids[keep], dy[o:o + n][keep].float())
    return demb


def embed_step(tokens, emb, pe, step, out):
    out.copy_((emb.index_select(0, tokens) + pe.index_select(0, step)).to(BF16))
    return out


def decode_self_attn(qkv, cache, step, ctx, n_head, scale, anc=None):
    n, S, w = cache.shape
    d = w // 2
    t = int(step)
    cache[:, t] = qkv[:, d:]
    hist = cache[:, :t + 1]
    if anc is not None:                   # lineage table: position p < t of hypothesis i lives in cache row anc[i][p]
        rows = anc[:, :t + 1].long().clone()
        rows[:, t] = torch.arange(n)
        hist = cache[rows, torch.arange(t + 1).unsqueeze(0).expand(n, t + 1)]
    k = hist[:, :, :d].float().view(n, t + 1, n_head, d // n_head)
    v = hist[:, :, d:].float().view(n, t + 1, n_head, d // n_head)
    q = qkv[:, :d].float().view(n, 1, n_head, d // n_head)
    sc = (q * k).sum(-1) * scale                         # [n, t + 1, H]
    p = torch.softmax(sc, dim=1).unsqueeze(-1)
    ctx.copy_((p * v).sum(1).reshape(n, d).to(BF16))


def beam_work_words(B, beam):
    return B * beam * beam + 1 + B


def beam_advance(logits, V, beam, step, eos, scores, tokens, done, lengths, hist_scores, back, toks, order, work=None, anc=None,
                 advance_step=False, embed=None):
    """Beam.advance for all utterances with torch ops (the formulation transformer/Decode.py used before st_beam_advance)."""
    B = scores.shape[0]
    word_lk = torch.log_softmax(logits[:, :V].float(), dim=-1)
    table = (word_lk.view(B, beam, V) + scores.unsqueeze(2)).view(B, beam * V)
    best_scores, best_flat = table.topk(beam, 1, True, True)
    origin = best_flat // V
    token = best_flat - origin * V
    live = ~done
    lv = live.unsqueeze(1)
    hist_scores.index_copy_(0, step, scores.unsqueeze(0))
    scores.copy_(torch.where(lv, best_scores, scores))
    slot_ids = torch.arange(beam).unsqueeze(0).expand(B, beam)
    origin = torch.where(lv, origin, slot_ids)
    back.index_copy_(0, step, origin.unsqueeze(0))
    toks.index_copy_(0, step, token.unsqueeze(0))
    tokens.copy_(torch.where(lv, token, tokens.view(B, beam)).view(-1))
    lengths.add_(live.to(lengths.dtype))
    done.logical_or_(live & (token[:, 0] == eos))
    order.copy_((origin + (torch.arange(B) * beam).unsqueeze(1)).view(-1))
    if anc is not None:
        t = int(step)
        anc[:, :t] = anc[order][:, :t].clone()
        anc[:, t] = order.to(anc.dtype)
    if embed is not None:
        emb, pe, x_next = embed
        if int(step) + 1 < pe.shape[0]:
            x_next.copy_((emb.index_select(0, tokens) + pe.index_select(0, step + 1)).to(BF16))
    if advance_step:
        step.add_(1)


def ce_fwd(logits, target, ignore_index, lse, sums, V=None, index=None):
    V = logits.shape[1] if V is None else V
    target = target if index is None else target.reshape(-1)[index]
    l = torch.logsumexp(logits[:, :V].float(), -1)
    lse.copy_(l)
    valid = target != ignore_index
    picked = logits[:, :V].float().gather(1, target.clamp(0, V - 1).view(-1, 1)).squeeze(1)
    sums[0] = ((l - picked) * valid).sum()
    sums[1] = valid.sum().float()
    sums[2] = sums[0] / sums[1]


def ce_bwd(logits, target, ignore_index, lse, sums, grad_out, dlogits, V=None, index=None):
    V = logits.shape[1] if V is None else V
    target = target if index is None else target.reshape(-1)[index]
    valid = (target != ignore_index).float().view(-1, 1)
    g = torch.exp(logits[:, :V].float() - lse.view(-1, 1))
    g.scatter_add_(1, target.clamp(0, V - 1).view(-1, 1), -torch.ones(len(target), 1))
    dlogits.zero_()
    dlogits[:, :V] = (g * valid * (grad_out.float() / sums[1])).to(BF16)


def zero_tails(table, n_max):
    raise RuntimeError("zero_tails: the emulation never registers tail buffers (functional.TailBuffers serves captures only)")


def grad_norm_scratch(device):
    return torch.zeros(1025, dtype=torch.float32, device=device)


def grad_norm(g, scratch, out, step=None, grad_scale=1.0):
    out.copy_((torch.linalg.vector_norm(g.double()) * float(grad_scale)).float())
    if step is not None:
        step.add_(1)
    return out


def cache_reorder(cache, order, step, beam):
    t = int(step.reshape(-1)[0]) + 1
    cache[:, :, :t] = cache[:, order][:, :, :t].clone()
    return cache


def cast_bf16(src, dst):
    dst.copy_(src.to(BF16))
    return dst


_NAMES = ["gemm", "gemm_kscale", "gemm_splitk", "gemm_ws", "adam_clip", "wgrad_group", "feat_stack", "gemm_lnbwd", "gemm_ln", "ln_bwd", "attn_fwd", "attn_f1_fwd", "attn_sf1_fwd", "attn_bwd", "attn_probs", "attn_dense_fwd", "attn_dense_bwd", "ctc_gather", "ctc_dlogits", "row_index", "pack_rows", "unpack_rows",
          "pack_grad", "embed_pe_fwd", "embed_bwd", "cast_bf16", "cache_reorder", "wfrag_depth", "wfrag_build", "row_chain", "row_chain_bwd", "chain_mask_words", "relu_bits_from", "beam_advance", "beam_work_words", "ce_fwd", "ce_bwd", "grad_norm", "grad_norm_scratch", "zero_tails", "decode_self_attn", "embed_step"]


@contextlib.contextmanager
def emulated_kernels():
    """Swap st_amd.native's entry points for the emulations above (CPU tensors allowed)."""
    saved = {n: getattr(nv, n) for n in _NAMES}
    saved_req = st_arena.ParamArena._require_gpu
    saved_drop = nv.Drop
    import transformer.Optim as st_optim
    saved_cpu_arena = st_optim.ScheduledOptim._allow_cpu_arena
    try:
        nv.Drop = Drop
        st_optim.ScheduledOptim._allow_cpu_arena = True     # flat-arena Adam (st_adam_clip emulation) on CPU tensors
        for n in _NAMES:
            setattr(nv, n, torch.no_grad()(globals()[n]))    # raw kernels know nothing of autograd (the grouped
                                                              # weight gradients are flushed outside backward())
        st_arena.ParamArena._require_gpu = staticmethod(lambda dev: None)
        yield
    finally:
        for n, f in saved.items():
            setattr(nv, n, f)
        nv.Drop = saved_drop
        st_optim.ScheduledOptim._allow_cpu_arena = saved_cpu_arena
        st_arena.ParamArena._require_gpu = staticmethod(saved_req)

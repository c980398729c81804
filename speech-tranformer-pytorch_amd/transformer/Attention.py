"""Multi-head attention sub-layer backed by the fused gfx950 kernels.

Drop-in for reference transformer/Attention.py:40-96 - same constructor,
``forward(q, k, v, mask=None) -> (out, attns)`` and ``state_dict`` keys.
"""
import math

import torch
import torch.nn as nn

from st_amd import functional as F_
from st_amd import rng
from st_amd.arena import arena_of, bundle
from transformer.Utils import LengthMask, lengths_from_mask


def _expand_mask(mask, B, Lq, Lk):
    """A dense mask as the reference's ``masked_fill_`` takes it: anything that broadcasts to [B, Lq, Lk] ([B, 1, Lk] key-padding
    masks, [Lq, Lk] causal masks ...)."""
    if mask.dim() == 2:
        mask = mask.unsqueeze(0)
    if mask.dim() != 3:
        raise ValueError("MultiHeadAttention: mask must broadcast to [batch, len_q, len_k], got shape %s" % (tuple(mask.shape),))
    return mask.expand(B, Lq, Lk)


class MultiHeadAttention(nn.Module):
    """q/k/v projections, masked scaled-dot-product attention per head, output
    projection, residual add and LayerNorm(eps=1e-6).

    Differences from the reference, all documented in DESIGN.md:
      * the residual is ``q`` (the reference writes ``+ v``, identical for
        self-attention and ill-formed otherwise - repair R2);
      * ``attns`` is ``None`` unless ``self.return_attn`` is set (the fused
        kernel never materialises the [B, h, Lq, Lk] tensor; every caller in the
        reference discards it, train.py:39);
      * ``mask`` may be a :class:`LengthMask`; a dense mask is analysed back
        into lengths (device sync), and a dense mask that is neither a key-padding nor a
        key-padding | causal mask - or ``k is not v`` - takes the general slow path
        (st_amd.functional.DenseMhaFn: no reference call site does);
      * training-mode dropout draws counter-based masks inside the kernels (st_amd/rng.py): same distribution as
        nn.Dropout with p quantised to 1/256, a different random stream.
    """

    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.1):
        super(MultiHeadAttention, self).__init__()
        assert d_model % n_head == 0
        assert d_k == d_model // n_head and d_v == d_model // n_head
        self.n_head, self.d_model, self.d_k, self.d_v = n_head, d_model, d_k, d_v
        self.scaled = math.sqrt(d_k)
        self.linear_q = nn.Linear(d_model, n_head * d_k)
        self.linear_k = nn.Linear(d_model, n_head * d_k)
        self.linear_v = nn.Linear(d_model, n_head * d_v)
        self.softmax = nn.Softmax(dim=-1)
        self.dropout = nn.Dropout(dropout)
        self.output_linear = nn.Linear(d_model, d_model)
        self.layernorm = nn.LayerNorm(d_model, eps=1e-6)
        self.return_attn = False

    # ---- arena plumbing ----------------------------------------------------------------------
    def _st_param_order(self):
        return [self.linear_q.weight, self.linear_k.weight, self.linear_v.weight,
                self.linear_q.bias, self.linear_k.bias, self.linear_v.bias,
                self.output_linear.weight, self.output_linear.bias,
                self.layernorm.weight, self.layernorm.bias]

    def _st_bind(self, a):
        d = self.d_model
        if d not in (128, 256, 512) or self.d_k not in (32, 64, 128):
            raise NotImplementedError(
                "HIP path: d_model must be 128, 256 or 512 and d_k = d_model / n_head 32, 64 or 128 (got d_model %d, d_k %d)"
                % (d, self.d_k))
        q, k, o, ln = self.linear_q, self.linear_k, self.output_linear, self.layernorm
        params = self._st_param_order()
        lo, hi = a.span(params)
        return bundle(
            d_model=d, n_head=self.n_head, params=params, lo=lo, hi=hi,
            w_qkv=a.bf16(q.weight, 3 * d), b_qkv=a.master(q.bias, 3 * d),
            w_q=a.bf16(q.weight), b_q=a.master(q.bias),
            w_kv=a.bf16(k.weight, 2 * d), b_kv=a.master(k.bias, 2 * d),
            w_o=a.bf16(o.weight), b_o=a.master(o.bias), gamma=a.master(ln.weight), beta=a.master(ln.bias),
            g_w_qkv=a.grad_view(q.weight, 3 * d), g_b_qkv=a.grad_view(q.bias, 3 * d),
            g_w_q=a.grad_view(q.weight), g_b_q=a.grad_view(q.bias),
            g_w_kv=a.grad_view(k.weight, 2 * d), g_b_kv=a.grad_view(k.bias, 2 * d),
            g_w_o=a.grad_view(o.weight), g_b_o=a.grad_view(o.bias),
            g_gamma=a.grad_view(ln.weight), g_beta=a.grad_view(ln.bias))

    def _drop(self, device):
        """Attention-probability dropout (Attention.py:89) of this call, or None in eval mode / p = 0."""
        return rng.site(device, self.dropout.p) if self.training else None

    # ---- fast path: bf16 row matrices ----------------------------------------------------------
    def forward_rows(self, x_q, x_kv, q_rows, k_rows, causal, kv_acc=None, up=None, down=None, pre=None):
        """x_q [Mq, d] (and x_kv [Mk, d] for cross-attention, else None) bf16 row matrices.
        up / down: st_amd.functional.LnLink shared with the sublayer before / after this one (layer stacks only).
        pre (st_amd.chains.SubPre): the forward values were already computed by fused launches - only record the
        autograd node."""
        arena = arena_of(self)
        with arena.scope():
            drop = pre.drop if pre is not None else self._drop(x_q.device)
            return F_.MhaFn.apply(x_q, x_kv, self.linear_q.weight, self, q_rows, k_rows, bool(causal), False,
                                  drop, kv_acc, up, down, pre)

    # ---- reference API -------------------------------------------------------------------------
    def _forward_dense(self, q, k, v, mask):
        """The general form (any dense mask - anything that broadcasts to [B, Lq, Lk], as the reference's masked_fill_ takes it -, k and
        v different tensors): st_amd.functional.DenseMhaFn.  Two documented deviations from Attention.py:64-96 on this path
        (INTEGRATION.md section 1): the residual added in front of the LayerNorm is ``q`` (R2: the reference adds ``v``, a shape
        error whenever Lq != Lk and the same tensor for self-attention), and the returned ``attns`` are the probabilities BEFORE
        dropout (A1: the reference returns them after ``self.dropout`` in training mode)."""
        B, Lq, d = q.shape
        Lk = k.shape[1]
        if v.shape[1] != Lk or k.shape[0] != B or v.shape[0] != B:
            raise ValueError("MultiHeadAttention: k and v must have the same batch size and length")
        if isinstance(mask, LengthMask):       # (built from the longest utterance: pad to this call's [Lq, Lk], padded keys masked)
            m = mask.dense(q.device)
            mask = torch.ones(B, Lq, Lk, dtype=torch.bool, device=q.device)
            mask[:, :m.shape[1], :m.shape[2]] = m[:, :Lq, :Lk]
            if m.shape[1] < Lq:
                mask[:, m.shape[1]:] = mask[:, m.shape[1] - 1:m.shape[1]]
        m8 = None if mask is None else _expand_mask(mask.to(device=q.device), B, Lq, Lk).ne(0).to(torch.uint8).contiguous()
        arena = arena_of(self)
        with arena.scope():
            args = [t.reshape(-1, d).to(torch.bfloat16) for t in (q, k, v)]
            res = F_.DenseMhaFn.apply(*args, self.linear_q.weight, self, m8, B, Lq, Lk, self._drop(q.device), bool(self.return_attn))
        out, attns = res if self.return_attn else (res, None)
        return out.to(q.dtype).view(B, Lq, d), attns

    def forward(self, q, k, v, mask=None):
        if k is not v:
            return self._forward_dense(q, k, v, mask)
        B, Lq, d = q.shape
        Lk = k.shape[1]
        dev = q.device
        if mask is None:
            k_len, causal = None, False
        elif isinstance(mask, LengthMask):
            k_len, causal = mask.k_len, mask.causal
        else:
            mask = _expand_mask(mask, B, Lq, Lk)
            fam = lengths_from_mask(mask)
            if fam is None:            # neither of the two mask families of Utils.py: the general slow path
                return self._forward_dense(q, k, v, mask)
            k_len, causal = fam
        q_rows = F_.Rows.padded(B, Lq, dev)
        k_rows = F_.Rows.padded(B, Lk, dev, k_len)
        xq = q.reshape(B * Lq, d).to(torch.bfloat16)
        xkv = None if k is q else k.reshape(B * Lk, d).to(torch.bfloat16)
        if self.return_attn:       # Attention.py:96: the probabilities, materialised by st_attn_probs (dropout not applied)
            with F_.AttnTap() as tap:
                out = self.forward_rows(xq, xkv, q_rows, k_rows, causal)
            attns = tap.of(self, Lq, Lk)[0]
        else:
            out, attns = self.forward_rows(xq, xkv, q_rows, k_rows, causal), None
        return out.to(q.dtype).view(B, Lq, d), attns

"""Weight-fragment streams for the row-chain kernel (csrc/st_rowchain.hip, ``native.row_chain``).

A chain is a list of 256 x 256 weight blocks in the order the kernel multiplies them.  ``ChainSet`` collects the chains
of a model, owns ONE bf16 buffer with all their per-wave fragment streams and rebuilds it with one launch
(``native.wfrag_build``) whenever the bf16 weights change - it registers itself with the parameter arena, whose
``refresh()`` (fp32 master -> bf16 shadow, once per forward) then also refreshes the streams.

``DecoderChains`` plans the decoder (Layers.py:37-44, one layer = self-attention -> encoder-decoder attention ->
feed-forward): per layer
  * F1: output_linear + residual + layernorm of the self-attention, then the q projection of the encoder-decoder
    attention (Attention.py:92-94, 74);
  * F2: output_linear + residual + layernorm of the encoder-decoder attention, the whole feed-forward sublayer
    (SubLayers.py:24-28) and the NEXT layer's q|k|v projection (Attention.py:74-76),
so a decoder layer is four launches (two attentions, two chains) instead of eight.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import torch

from . import native as nv
from . import rng

BLK = 256
BF16, F32 = torch.bfloat16, torch.float32


class Chain:
    """One chain's fragment streams (a slice of its ChainSet's buffer) and the weight blocks they were built from."""
    __slots__ = ("stream", "n_blocks", "blocks", "next_blocks")

    def __init__(self, stream, n_blocks, blocks, next_blocks=0):
        # next_blocks: size of the chain stored right behind this one IF it is also the one that runs next (the kernel
        # then warms the L2 with it as well), else 0
        self.stream, self.n_blocks, self.blocks, self.next_blocks = stream, n_blocks, blocks, next_blocks


def blocks_of(w: torch.Tensor, order: str = "rows") -> List[Tuple[torch.Tensor, int, int]]:
    """The 256 x 256 blocks of an nn.Linear weight [N, K] as (weight, n0, k0): ``rows`` = one block per 256 output rows
    (K must be 256); used by the callers below to spell chains."""
    n, k = w.shape
    if n % BLK or k % BLK:
        raise ValueError("chains: weight shape %s is not a multiple of 256" % (tuple(w.shape),))
    if order == "rows":
        if k != BLK:
            raise ValueError("chains: row blocks need K = 256")
        return [(w, r, 0) for r in range(0, n, BLK)]
    raise ValueError(order)


def ffn_blocks(w1: torch.Tensor, w2: torch.Tensor) -> List[Tuple[torch.Tensor, int, int]]:
    """W1 [d_ff, 256], W2 [256, d_ff] in the kernel's chunk order: (W1 rows c*256.., W2 columns c*256..) for c = 0.."""
    d_ff = w1.shape[0]
    if w1.shape[1] != BLK or tuple(w2.shape) != (BLK, d_ff) or d_ff % BLK:
        raise ValueError("chains: feed-forward weights must be [d_ff, 256] and [256, d_ff], d_ff a multiple of 256")
    out = []
    for c in range(0, d_ff, BLK):
        out += [(w1, c, 0), (w2, 0, c)]
    return out


class ChainSet:
    def __init__(self, device):
        self.device = torch.device(device)
        self.depth = nv.wfrag_depth()
        self._rows: List[List[int]] = []
        self._chains: List[Tuple[int, int, list]] = []      # (element offset, n_blocks, blocks)
        self._elems = 0
        self.table = None
        self.buf = None

    def add(self, blocks: Sequence[Tuple[torch.Tensor, int, int]]) -> int:
        """-> chain id.  blocks: (row-major bf16 weight, first row, first column) per 256 x 256 block, in order."""
        if self.table is not None:
            raise RuntimeError("ChainSet: already finalised")
        n = len(blocks)
        wave_frags = n * 16 + self.depth
        base = self._elems
        for i, (w, n0, k0) in enumerate(blocks):
            if w.dtype != torch.bfloat16 or w.dim() != 2 or w.stride(1) != 1:
                raise ValueError("ChainSet: weights must be row-major bf16 matrices")
            if n0 + BLK > w.shape[0] or k0 + BLK > w.shape[1] or (w.stride(0) % 8):
                raise ValueError("ChainSet: block (%d, %d) outside weight %s" % (n0, k0, tuple(w.shape)))
            self._rows.append([w.data_ptr() + 2 * (n0 * w.stride(0) + k0), w.stride(0), i * 16, base | (wave_frags << 40)])
        self._chains.append((base, n, list(blocks)))     # (keeps the weight tensors alive: the table holds raw addresses)
        self._elems += 8 * wave_frags * 512
        return len(self._chains) - 1

    def finalize(self) -> "ChainSet":
        self.table = torch.tensor(self._rows, dtype=torch.int64).to(self.device)
        self.buf = torch.zeros(self._elems, dtype=torch.bfloat16, device=self.device)
        return self

    def rebuild(self) -> None:
        nv.wfrag_build(self.table, self.buf)

    def chain(self, cid: int, runs_before_next: bool = False) -> Chain:
        """runs_before_next: chain cid + 1 (stored right behind) is the next chain to run after this one."""
        base, n, blocks = self._chains[cid]
        nxt = self._chains[cid + 1][1] if runs_before_next and cid + 1 < len(self._chains) else 0
        return Chain(self.buf[base:base + 8 * (n * 16 + self.depth) * 512], n, blocks, nxt)


class SubPre:
    """What a sublayer's autograd Function (functional.MhaFn / FfnFn) takes INSTEAD of launching its forward kernels:
    the tensors those kernels would have produced, and the dropout sites that were used."""
    __slots__ = ("qkv", "kvbuf", "ctx", "ores", "lse", "h", "out", "xhat", "rstd", "drop", "drop1", "drop2")

    def __init__(self):
        for k in self.__slots__:
            setattr(self, k, None)


class EncoderChains:
    """The chains of an encoder's layer stack (Layers.py:16-22, one layer = self-attention -> feed-forward): per layer ONE
    chain - output_linear + residual + layernorm of the self-attention, the feed-forward sublayer, the NEXT layer's q|k|v
    projection - so an encoder layer is two launches (attention, chain) instead of five.  At encoder-sized row counts the
    kernel gives a workgroup 96 rows (24,060 rows = 251 workgroups = one round of the CUs)."""

    def __init__(self, layers, arena):
        self.arena = arena
        self.set = ChainSet(arena.device)
        ids = []
        n = len(layers)
        for l, layer in enumerate(layers):
            sa, ff = layer.slf_attn._st, layer.pos_ffn._st
            nxt = blocks_of(layers[l + 1].slf_attn._st.w_qkv) if l + 1 < n else []
            ids.append(self.set.add(blocks_of(sa.w_o) + ffn_blocks(ff.w1, ff.w2) + nxt))
        self.set.finalize()
        self.e = [self.set.chain(c, True) for c in ids]

    @staticmethod
    def plan(layers, arena):
        if not layers:
            return None
        for layer in layers:
            sa, ff = layer.slf_attn, layer.pos_ffn
            if any(getattr(m, "_st_arena", None) is not arena for m in (sa, ff)):
                return None
            if sa._st.d_model != BLK or ff._st.d_ff % BLK:
                return None
        return EncoderChains(layers, arena)

    def refresh(self) -> None:
        self.set.rebuild()

    def forward(self, layers, x, rows, need_bwd: bool):
        """The encoder's layer stack on frame rows x [M, 256] (front-end output): per layer self-attention, then the chain.
        -> (output rows, [(self-attention, feed-forward) SubPre per layer]); nothing here is recorded by autograd."""
        from .functional import attn_work, linear_fwd
        M, d = x.shape
        dev = x.device

        def E(*shape, dt=BF16):
            return torch.empty(*shape, dtype=dt, device=dev)

        s0 = layers[0].slf_attn._st
        H = s0.n_head
        scale = 1.0 / math.sqrt(d // H)
        work = attn_work(rows, rows, False)[0]
        qkv = E(M, 3 * d)
        linear_fwd(x, s0.w_qkv, qkv, s0.b_qkv)
        pres = []
        n = len(layers)
        for l, layer in enumerate(layers):
            sa, ff = layer.slf_attn, layer.pos_ffn
            a, f = SubPre(), SubPre()
            a.qkv, a.drop = qkv, sa._drop(dev)
            a.ctx, a.lse, a.ores = E(M, d), E(H * M, dt=F32), (E(M, d) if need_bwd else None)
            nv.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], a.ctx, a.lse, rows.off, rows.len, rows.off, rows.len,
                        H, rows.max_len, False, scale, work=work, drop=a.drop, max_k=rows.max_len, ores=a.ores)
            a.out, f.out, f.h = E(M, d), E(M, d), E(M, ff._st.d_ff)
            if need_bwd:
                a.xhat, a.rstd, f.xhat, f.rstd = E(M, d), E(M, dt=F32), E(M, d), E(M, dt=F32)
            f.drop1, f.drop2 = ff._drops(dev)
            nxt = layers[l + 1].slf_attn._st if l + 1 < n else None
            qkv = E(M, 3 * d) if nxt is not None else None
            nv.row_chain(a.ctx, self.e[l], pre=(x, sa._st.b_o, sa._st.gamma, sa._st.beta, a.out, a.xhat, a.rstd),
                         ffn=(ff._st.d_ff, ff._st.b1, ff._st.b2, ff._st.gamma, ff._st.beta, f.h, f.out, f.xhat, f.rstd, f.drop1,
                              f.drop2),
                         post=(3, nxt.b_qkv, qkv) if nxt is not None else None)
            x = f.out
            pres.append((a, f))
        return x, pres


class DecoderChains:
    """The chains of a decoder's layer stack (see the module docstring); ``None`` from ``plan`` when a layer does not fit
    the kernel (d_model != 256, d_ff not a multiple of 256, parameters outside one arena)."""

    def __init__(self, layers, arena):
        self.arena = arena
        self.set = ChainSet(arena.device)
        f1, f2 = [], []
        n = len(layers)
        for l, layer in enumerate(layers):
            sa, ca, ff = layer.slf_attn._st, layer.enc_attn._st, layer.pos_ffn._st
            f1.append(self.set.add(blocks_of(sa.w_o) + blocks_of(ca.w_q)))
            nxt = blocks_of(layers[l + 1].slf_attn._st.w_qkv) if l + 1 < n else []
            f2.append(self.set.add(blocks_of(ca.w_o) + ffn_blocks(ff.w1, ff.w2) + nxt))
        self.set.finalize()
        self.f1 = [self.set.chain(c, True) for c in f1]      # stored in running order: F1(0), F2(0), F1(1), ...
        self.f2 = [self.set.chain(c, True) for c in f2]

    @staticmethod
    def plan(layers, arena):
        if not layers:
            return None
        for layer in layers:
            sa, ca, ff = layer.slf_attn, layer.enc_attn, layer.pos_ffn
            if any(getattr(m, "_st_arena", None) is not arena for m in (sa, ca, ff)):
                return None
            if sa._st.d_model != BLK or ff._st.d_ff % BLK:
                return None
        return DecoderChains(layers, arena)

    def refresh(self) -> None:
        self.set.rebuild()

    def forward(self, layers, x, kv, t_rows, in_rows, need_bwd: bool):
        """The decoder's layer stack on target rows x [M, 256] (embedding + positional encoding), kv = CrossKv's buffer
        (every layer's K | V projection of the encoder output): per layer  causal self-attention, chain F1,
        encoder-decoder attention, chain F2.  -> (output rows, [(self-attention, encoder-decoder attention, feed-forward)
        SubPre per layer]) - nothing here is recorded by autograd; the caller replays the Functions over the SubPre's."""
        from .functional import attn_work
        M, d = x.shape
        dev = x.device

        def E(*shape, dt=BF16):
            return torch.empty(*shape, dtype=dt, device=dev)

        s0 = layers[0].slf_attn._st
        H = s0.n_head
        scale = 1.0 / math.sqrt(d // H)
        work_self, work_cross = attn_work(t_rows, t_rows, True)[0], attn_work(t_rows, in_rows, False)[0]
        qkv = E(M, 3 * d)
        nv.gemm(x, s0.w_qkv, qkv, bias=s0.b_qkv)
        pres = []
        n = len(layers)
        for l, layer in enumerate(layers):
            sa, ca, ff = layer.slf_attn, layer.enc_attn, layer.pos_ffn
            a, b, f = SubPre(), SubPre(), SubPre()
            # ---- causal self-attention (Attention.py:82-90)
            a.qkv, a.drop = qkv, sa._drop(dev)
            a.ctx, a.lse, a.ores = E(M, d), E(H * M, dt=F32), (E(M, d) if need_bwd else None)
            nv.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], a.ctx, a.lse, t_rows.off, t_rows.len, t_rows.off, t_rows.len,
                        H, t_rows.max_len, True, scale, work=work_self, drop=a.drop, max_k=t_rows.max_len, ores=a.ores)
            # ---- F1: its output_linear + residual + LayerNorm, the next attention's q
            a.out = E(M, d)
            if need_bwd:
                a.xhat, a.rstd = E(M, d), E(M, dt=F32)
            b.qkv = E(M, d)
            nv.row_chain(a.ctx, self.f1[l], pre=(x, sa._st.b_o, sa._st.gamma, sa._st.beta, a.out, a.xhat, a.rstd),
                         post=(1, ca._st.b_q, b.qkv))
            # ---- encoder-decoder attention over this layer's column block of kv
            b.kvbuf, b.drop = kv[:, l * 2 * d:(l + 1) * 2 * d], ca._drop(dev)
            b.ctx, b.lse, b.ores = E(M, d), E(H * M, dt=F32), (E(M, d) if need_bwd else None)
            nv.attn_fwd(b.qkv, b.kvbuf[:, :d], b.kvbuf[:, d:], b.ctx, b.lse, t_rows.off, t_rows.len, in_rows.off, in_rows.len,
                        H, t_rows.max_len, False, scale, work=work_cross, drop=b.drop, max_k=in_rows.max_len, ores=b.ores)
            # ---- F2: its output_linear + residual + LayerNorm, the feed-forward sublayer, the next layer's q|k|v
            b.out, f.out, f.h = E(M, d), E(M, d), E(M, ff._st.d_ff)
            if need_bwd:
                b.xhat, b.rstd, f.xhat, f.rstd = E(M, d), E(M, dt=F32), E(M, d), E(M, dt=F32)
            f.drop1, f.drop2 = ff._drops(dev)
            nxt = layers[l + 1].slf_attn._st if l + 1 < n else None
            qkv = E(M, 3 * d) if nxt is not None else None
            nv.row_chain(b.ctx, self.f2[l], pre=(a.out, ca._st.b_o, ca._st.gamma, ca._st.beta, b.out, b.xhat, b.rstd),
                         ffn=(ff._st.d_ff, ff._st.b1, ff._st.b2, ff._st.gamma, ff._st.beta, f.h, f.out, f.xhat, f.rstd, f.drop1,
                              f.drop2),
                         post=(3, nxt.b_qkv, qkv) if nxt is not None else None)
            x = f.out
            pres.append((a, b, f))
        return x, pres

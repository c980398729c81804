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

import os
import torch

from . import native as nv
from . import rng

BLK = 256
BF16, F32 = torch.bfloat16, torch.float32


class Chain:
    """One chain's fragment streams (a slice of its ChainSet's buffer) and the weight blocks they were built from."""
    __slots__ = ("stream", "n_blocks", "blocks", "next_blocks", "split_work")

    def __init__(self, stream, n_blocks, blocks, next_blocks=0):
        # next_blocks: size of the chain stored right behind this one IF it is also the one that runs next (the kernel
        # then warms the L2 with it as well), else 0
        self.stream, self.n_blocks, self.blocks, self.next_blocks = stream, n_blocks, blocks, next_blocks
        # split_work: scratch (zero-initialised int32 tensor, native.split_work_words() elements) that lets st_row_chain cut
        # the feed-forward's hidden dimension over several workgroups per row block at decoder-sized row counts; chains
        # that run one after the other on one stream may share it.  None: one workgroup per row block.
        self.split_work = None


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


def t_blocks(blocks):
    """The same blocks read transposed (the data gradient's operand)."""
    return [(w, n0, k0, True) for w, n0, k0 in blocks]


def ffn_blocks_bwd(w1: torch.Tensor, w2: torch.Tensor):
    """The backward chain's order through the feed-forward weights, all transposed: (W2 columns c*256.., W1 rows c*256..)."""
    out = []
    for c in range(0, w1.shape[0], BLK):
        out += [(w2, 0, c, True), (w1, c, 0, True)]
    return out


def ffn_blocks(w1: torch.Tensor, w2: torch.Tensor) -> List[Tuple[torch.Tensor, int, int]]:
    """W1 [d_ff, 256], W2 [256, d_ff] in the kernel's chunk order: (W1 rows c*256.., W2 columns c*256..) for c = 0.."""
    d_ff = w1.shape[0]
    if w1.shape[1] != BLK or tuple(w2.shape) != (BLK, d_ff) or d_ff % BLK:
        raise ValueError("chains: feed-forward weights must be [d_ff, 256] and [256, d_ff], d_ff a multiple of 256")
    out = []
    for c in range(0, d_ff, BLK):
        out += [(w1, c, 0), (w2, 0, c)]
    return out


def encoder512_blocks(wo: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, wp=None):
    """The block order of st_row_chain512 (csrc/st_rowchain_pipe512.cuh) for one encoder layer at d_model 512: Wo [512, 512] as
    (h, j) -> 2 h + j; per hidden chunk c of 256: W1 rows c (j = 0, 1), W2 columns c (h = 0, 1); then the next layer's q | k | v
    projection Wp [1536, 512] as (u, j) (h / u: 256-row block of the weight = output columns, j: 256-column block = input half)."""
    d_ff = w1.shape[0]
    if tuple(wo.shape) != (512, 512) or w1.shape[1] != 512 or tuple(w2.shape) != (512, d_ff) or d_ff % BLK:
        raise ValueError("chains: d_model-512 weights must be [512, 512], [d_ff, 512] and [512, d_ff], d_ff a multiple of 256")
    out = [(wo, h * BLK, j * BLK) for h in range(2) for j in range(2)]
    for c in range(0, d_ff, BLK):
        out += [(w1, c, 0), (w1, c, BLK), (w2, 0, c), (w2, BLK, c)]
    if wp is not None:
        if tuple(wp.shape) != (1536, 512):
            raise ValueError("chains: the d_model-512 projection must be [1536, 512]")
        out += [(wp, u * BLK, j * BLK) for u in range(6) for j in range(2)]
    return out


def encoder512_blocks_bwd(wo: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, wp=None):
    """The block order of st_row_chain512_bwd, all read transposed: Wp [1536, 512] as (h, u) -> 6 h + u (contraction over the
    weight's rows 256 u .., output = its columns 256 h ..); per hidden chunk c: W2^T (j = 0, 1: rows 256 j .. of W2, columns c),
    W1^T (h = 0, 1: rows c of W1, columns 256 h ..); Wo^T as (h, j) -> 2 h + j (rows 256 j .., columns 256 h ..)."""
    d_ff = w1.shape[0]
    out = []
    if wp is not None:
        out += [(wp, u * BLK, h * BLK, True) for h in range(2) for u in range(6)]
    for c in range(0, d_ff, BLK):
        out += [(w2, 0, c, True), (w2, BLK, c, True), (w1, c, 0, True), (w1, c, BLK, True)]
    out += [(wo, j * BLK, h * BLK, True) for h in range(2) for j in range(2)]
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
        for i, blk in enumerate(blocks):
            w, n0, k0 = blk[:3]
            tr = len(blk) > 3 and blk[3]
            if w.dtype != torch.bfloat16 or w.dim() != 2 or w.stride(1) != 1:
                raise ValueError("ChainSet: weights must be row-major bf16 matrices")
            if n0 + BLK > w.shape[0] or k0 + BLK > w.shape[1] or (w.stride(0) % 8):
                raise ValueError("ChainSet: block (%d, %d) outside weight %s" % (n0, k0, tuple(w.shape)))
            self._rows.append([w.data_ptr() + 2 * (n0 * w.stride(0) + k0), w.stride(0) | (int(tr) << 32), (i * 16) | (wave_frags << 32),
                               2 * base])      # [3]: byte offset of the chain for now; finalize() adds the buffer's address
        self._chains.append((base, n, list(blocks)))     # (keeps the weight tensors alive: the table holds raw addresses)
        self._elems += 8 * wave_frags * 512
        return len(self._chains) - 1

    def finalize(self) -> "ChainSet":
        self.buf = torch.zeros(self._elems, dtype=torch.bfloat16, device=self.device)
        rows = torch.tensor(self._rows, dtype=torch.int64)
        rows[:, 3] += self.buf.data_ptr()
        self.table = rows.to(self.device)
        return self

    def rebuild(self) -> None:
        nv.wfrag_build(self.table)


    def chain(self, cid: int, runs_before_next: bool = False) -> Chain:
        """runs_before_next: chain cid + 1 (stored right behind) is the next chain to run after this one."""
        base, n, blocks = self._chains[cid]
        nxt = self._chains[cid + 1][1] if runs_before_next and cid + 1 < len(self._chains) else 0
        return Chain(self.buf[base:base + 8 * (n * 16 + self.depth) * 512], n, blocks, nxt)


class ChainHub:
    """All ChainSets of one parameter arena, rebuilt with ONE launch after every ``arena.refresh()`` (the tables hold
    absolute addresses, so they concatenate)."""

    def __init__(self):
        self.sets: List[ChainSet] = []
        self.table = None

    @staticmethod
    def of(arena) -> "ChainHub":
        for d in arena._derived:
            if isinstance(d, ChainHub):
                return d
        hub = ChainHub()
        arena._derived.append(hub)
        return hub

    def add(self, *sets: ChainSet) -> None:
        self.sets += list(sets)
        self.table = torch.cat([s.table for s in self.sets], 0).contiguous()
        nv.wfrag_build(torch.cat([s.table for s in sets], 0).contiguous())      # the new ones, from the current shadow

    def refresh(self) -> None:
        if self.table is not None:
            nv.wfrag_build(self.table)


class SubPre:
    """What a sublayer's autograd Function (functional.MhaFn / FfnFn) takes INSTEAD of launching its forward kernels:
    the tensors those kernels would have produced, and the dropout sites that were used."""
    __slots__ = ("qkv", "kvbuf", "ctx", "ores", "lse", "h", "bits", "out", "xhat", "rstd", "drop", "drop1", "drop2", "bwd", "key",
                 "kpre")      # kpre: the key columns of qkv hold scale * log2(e) * k (native.attn_fwd's k_prescaled)

    def __init__(self):
        for k in self.__slots__:
            setattr(self, k, None)


class ChainBackward:
    """One backward pass over a layer stack whose forward ran as row chains: the sublayers' autograd Functions
    (functional.MhaFn / FfnFn) ask it for the pieces of their backward that lie between two attention-backward kernels,
    and it runs those as ONE st_row_chain_bwd launch per gap:

      * ``input_grad(key, dproj, ds)`` - called where an attention's backward would compute its input gradient
        (dproj W + ds, pushed through the LayerNorm backward of the sublayer in front): launches the backward chain of
        everything in front of it down to the previous attention (LayerNorm backward, feed-forward backward, the
        previous attention's d(context) + delta) and returns the normalised gradient that flows on;
      * ``stored(key)`` - what such a launch already produced for the sublayer ``key`` (its normalised output gradient
        ds, the hidden gradient, d(context), delta ...), so that its Function only adds its weight gradients;
      * ``ffn_tail(l, dout)`` - the same for the LAST feed-forward of the stack, whose output gradient arrives from outside
        (its LayerNorm backward is the head of that chain).

    Weight gradients stay with the Functions (they are deferred into the grouped launch)."""

    def __init__(self, owner, layers, pres):
        self.owner, self.layers, self.pres = owner, layers, pres
        self.done = {}

    def stored(self, key):
        return self.done.pop(key, None)

    def _empty(self, like, cols=None, dt=BF16, rows=None):
        return torch.empty(like.shape[0] if rows is None else rows, *(() if cols is None else (cols,)), dtype=dt, device=like.device)

    def _ffn_tail_chain(self, chain, l, ds_f, head, attn_key, attn_pre, attn_mod):
        """[HEAD +] feed-forward backward of layer l + d(context) / delta of the attention in front of it."""
        f = self.pres[l][-1]
        ff = self.layers[l].pos_ffn
        fs, as_ = ff._st, attn_mod._st
        arena = ff._st_arena
        arena.attach_grads(fs.params, fs.lo, fs.hi)
        arena.attach_grads(as_.params, as_.lo, as_.hi)
        M = f.out.shape[0]
        dm = f.out.shape[1]
        dH, ds_b, dctx = self._empty(f.out, fs.d_ff), self._empty(f.out, dm), self._empty(f.out, dm)
        delta = torch.empty(as_.n_head * M, dtype=F32, device=f.out.device)
        scale = f.drop1.scale if f.drop1 is not None and f.drop1.thresh else 1.0
        nv.row_chain_bwd(chain, M, head=head, ds_in=None if head else ds_f,
                         ffn=(fs.d_ff, f.bits, scale, dH, attn_pre.xhat, attn_pre.rstd, as_.gamma, ds_b, as_.g_gamma, as_.g_beta,
                              as_.g_b_o),
                         tail=(attn_pre.ctx, attn_pre.ores, dctx, delta))
        self.done[("ffn", l)] = dict(ds=ds_f, dh=dH, dx=ds_b)
        self.done[(attn_key, l)] = dict(ds=ds_b, dctx=dctx, delta=delta)

    def _ffn_head(self, l, dqkv, ds_s):
        """The LayerNorm backward of feed-forward l's output: of dqkv W_qkv + ds_s (the next layer's self-attention), or
        - dqkv None - of the raw gradient ds_s that reaches the last layer from outside the stack."""
        f = self.pres[l][-1]
        fs = self.layers[l].pos_ffn._st
        ds_f = self._empty(f.out, f.out.shape[1])
        return ds_f, (3 * f.out.shape[1] // BLK if dqkv is not None else 0, dqkv, ds_s, f.xhat, f.rstd, fs.gamma, f.drop2, ds_f, fs.g_gamma,
                      fs.g_beta, fs.g_b2)


class EncoderBackward(ChainBackward):
    def input_grad(self, key, dproj, ds):
        kind, l = key
        if kind == "self" and self.owner.layer_hook is not None:
            # the backward of every layer above l has returned: its weight gradients are all registered (deferred) and its
            # LayerNorm / bias gradients are enqueued - a data-parallel step flushes and starts exchanging them here
            self.owner.layer_hook(l + 1)
        if kind != "self" or l == 0:
            return None
        ds_f, head = self._ffn_head(l - 1, dproj, ds)
        self._ffn_tail_chain(self.owner.bwd[l - 1], l - 1, ds_f, head, "self", self.pres[l - 1][0], self.layers[l - 1].slf_attn)
        return ds_f

    def ffn_tail(self, l, dout):
        """dout: the raw gradient of the stack's output (its LayerNorm backward is the chain's head)."""
        ds_f, head = self._ffn_head(l, None, dout)
        self._ffn_tail_chain(self.owner.bwd[l], l, ds_f, head, "self", self.pres[l][0], self.layers[l].slf_attn)
        return self.stored(("ffn", l))


class DecoderBackward(ChainBackward):
    def input_grad(self, key, dproj, ds):
        kind, l = key
        if kind == "self":
            if l == 0:
                return None
            ds_f, head = self._ffn_head(l - 1, dproj, ds)
            self._ffn_tail_chain(self.owner.bwd2[l - 1], l - 1, ds_f, head, "cross", self.pres[l - 1][1],
                                 self.layers[l - 1].enc_attn)
            return ds_f
        # after the encoder-decoder attention's backward kernel: LayerNorm backward of the self-attention sublayer and its
        # d(context) / delta
        a = self.pres[l][0]
        sa = self.layers[l].slf_attn._st
        arena = self.layers[l].slf_attn._st_arena
        arena.attach_grads(sa.params, sa.lo, sa.hi)
        M = a.out.shape[0]
        ds_s, dctx = self._empty(a.out, BLK), self._empty(a.out, BLK)
        delta = torch.empty(sa.n_head * M, dtype=F32, device=a.out.device)
        nv.row_chain_bwd(self.owner.bwd1[l], M,
                         head=(1, dproj, ds, a.xhat, a.rstd, sa.gamma, None, ds_s, sa.g_gamma, sa.g_beta, sa.g_b_o),
                         tail=(a.ctx, a.ores, dctx, delta))
        self.done[("self", l)] = dict(ds=ds_s, dctx=dctx, delta=delta)
        return ds_s

    def ffn_tail(self, l, dout):
        ds_f, head = self._ffn_head(l, None, dout)
        self._ffn_tail_chain(self.owner.bwd2[l], l, ds_f, head, "cross", self.pres[l][1], self.layers[l].enc_attn)
        return self.stored(("ffn", l))


class EncoderChains:
    """The chains of an encoder's layer stack (Layers.py:16-22, one layer = self-attention -> feed-forward): per layer ONE
    chain - output_linear + residual + layernorm of the self-attention, the feed-forward sublayer, the NEXT layer's q|k|v
    projection - so an encoder layer is two launches (attention, chain) instead of five.  At encoder-sized row counts the
    kernel gives a workgroup 96 rows (24,060 rows = 251 workgroups = one round of the CUs)."""

    PRESCALE_KEYS = True      # (tests switch it off to compare the chain path bit for bit with the per-GEMM path)

    def __init__(self, layers, arena):
        self.arena = arena
        self.set = ChainSet(arena.device)
        ids = []
        n = len(layers)
        self.d = layers[0].slf_attn._st.d_model
        if self.d == 512:
            # BASELINE config 3's width (round 6): forward chains only (st_row_chain512: output_linear + LayerNorm, feed-forward,
            # the next layer's q | k | v projection per launch, 64-row workgroups); layer 0's projection and the whole backward
            # stay on the per-GEMM kernels, which read the tensors the chain saved
            for l, layer in enumerate(layers):
                sa, ff = layer.slf_attn._st, layer.pos_ffn._st
                ids.append(self.set.add(encoder512_blocks(sa.w_o, ff.w1, ff.w2, layers[l + 1].slf_attn._st.w_qkv if l + 1 < n else None)))
            self.set.finalize()
            self.q0 = None
            self.e = [self.set.chain(c, True) for c in ids]
            # backward chains in running order (last layer first): [next layer's q|k|v projection] + feed-forward + output_linear
            self.bset = ChainSet(arena.device)
            bids = {}
            for l in range(n - 1, -1, -1):
                sa, ff = layers[l].slf_attn._st, layers[l].pos_ffn._st
                bids[l] = self.bset.add(encoder512_blocks_bwd(sa.w_o, ff.w1, ff.w2, layers[l + 1].slf_attn._st.w_qkv if l + 1 < n else None))
            self.bset.finalize()
            self.bwd = [self.bset.chain(bids[l], True) for l in range(n)]
            # The backward chain (st_row_chain512_bwd) is built and tested but OFF by default: at config 3 it measured 253 us per
            # launch against 216 for the per-GEMM kernels it replaces (gemm_lnbwd + dgrad GEMMs: 11.57 vs 11.12 ms per step, same
            # box, round 6) - 64-row workgroups run two rounds on 256 CUs and the LayerNorm backward's column sums cost what the
            # separate kernels' do.  ST_CHAIN512=1 turns it on, =0 turns the forward chains off as well.
            self.use_bwd = sa.n_head * 64 == self.d and os.environ.get("ST_CHAIN512", "f") == "1"
            self.layer_hook = None
            ChainHub.of(arena).add(self.set, self.bset)
            return
        # layer 0's q | k | v projection: a chain of its own (three blocks, no PRE / FFN) - every encoder layer's keys leave
        # their projection PRE-SCALED by scale * log2(e) in the chain's fp32 epilogue (st_row_chain's post_kscale), so that no
        # attention kernel multiplies per score and forward and backward exponentiate identical scores (st_attn_common.cuh)
        q0 = self.set.add(blocks_of(layers[0].slf_attn._st.w_qkv))
        for l, layer in enumerate(layers):
            sa, ff = layer.slf_attn._st, layer.pos_ffn._st
            nxt = blocks_of(layers[l + 1].slf_attn._st.w_qkv) if l + 1 < n else []
            ids.append(self.set.add(blocks_of(sa.w_o) + ffn_blocks(ff.w1, ff.w2) + nxt))
        self.set.finalize()
        self.q0 = self.set.chain(q0, False)
        self.e = [self.set.chain(c, True) for c in ids]
        # backward chains, stored in running order (last layer first): [next layer's q|k|v projection] + feed-forward +
        # output_linear, all read transposed
        self.bset = ChainSet(arena.device)
        ids = {}
        for l in range(n - 1, -1, -1):
            sa, ff = layers[l].slf_attn._st, layers[l].pos_ffn._st
            head = t_blocks(blocks_of(layers[l + 1].slf_attn._st.w_qkv)) if l + 1 < n else []
            ids[l] = self.bset.add(head + ffn_blocks_bwd(ff.w1, ff.w2) + t_blocks(blocks_of(sa.w_o)))
        self.bset.finalize()
        self.bwd = [self.bset.chain(ids[l], True) for l in range(n)]
        # (no split_work for the encoder's chains: cutting the hidden dimension over workgroups changes the fp32 summation order
        # per row, and WHICH split a launch takes depends on its row count - an utterance's activations would then depend on what
        # else is in the batch / on the data-parallel shard size.  Measured: a 4-utterance shard 1.667 -> 1.650 ms with a two-way
        # split, tests/test_modules_gpu.py::test_full_size_batch_split_invariance 5e-3 -> 1.6e-2.  Not worth it.  LABNOTES round 5.)
        if os.environ.get("ST_CHAIN_SPLIT", "1") == "e":      # (development switch: the measurement above)
            work = torch.zeros(nv.split_work_words(), dtype=torch.int32, device=arena.device)
            for ch in self.e + self.bwd:
                ch.split_work = work
        self.use_bwd = sa.n_head * 64 == BLK        # the delta epilogue's heads are 64 columns
        self.layer_hook = None      # optional callable(first_finished_layer): EncoderBackward.input_grad / trainer.TrainStep
        ChainHub.of(arena).add(self.set, self.bset)

    @staticmethod
    def plan(layers, arena):
        if not layers:
            return None
        for layer in layers:
            sa, ff = layer.slf_attn, layer.pos_ffn
            if any(getattr(m, "_st_arena", None) is not arena for m in (sa, ff)):
                return None
            if sa._st.d_model not in (BLK, 512) or ff._st.d_ff % BLK:
                return None
            if sa._st.d_model == 512 and os.environ.get("ST_CHAIN512", "f") == "0":      # (development switch: the per-GEMM path)
                return None
        return EncoderChains(layers, arena)

    def forward(self, layers, x, rows, need_bwd: bool):
        """The encoder's layer stack on frame rows x [M, 256] (front-end output): per layer self-attention, then the chain.
        -> (output rows, [(self-attention, feed-forward) SubPre per layer]); nothing here is recorded by autograd."""
        from .functional import attn_work, rows_buffer
        M, d = x.shape
        dev = x.device

        def E(*shape, dt=BF16):
            return torch.empty(*shape, dtype=dt, device=dev)

        def EZ(m, n, layout):      # attention outputs: rows past a length are never written - zeros on padded layouts
            return rows_buffer(m, n, layout, dev)

        s0 = layers[0].slf_attn._st
        H = s0.n_head
        scale = 1.0 / math.sqrt(d // H)
        kpre = bool(self.PRESCALE_KEYS)
        kscale = scale * nv.K_LOG2_SCALE if kpre else 0.0
        work = attn_work(rows, rows, False, d // H, H)[0]
        qkv = E(M, 3 * d)
        if self.q0 is not None:
            nv.row_chain(x, self.q0, post=(3, s0.b_qkv, qkv), post_kscale=kscale)
        elif kpre:
            nv.gemm_kscale(x, s0.w_qkv, qkv, s0.b_qkv, d, 2 * d, kscale)
        else:
            nv.gemm(x, s0.w_qkv, qkv, bias=s0.b_qkv)
        pres = []
        n = len(layers)
        for l, layer in enumerate(layers):
            sa, ff = layer.slf_attn, layer.pos_ffn
            a, f = SubPre(), SubPre()
            a.qkv, a.drop, a.kpre = qkv, sa._drop(dev), kpre
            a.ctx, a.lse, a.ores = EZ(M, d, rows), E(H * M, dt=F32), (EZ(M, d, rows) if need_bwd else None)
            nv.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], a.ctx, a.lse, rows.off, rows.len, rows.off, rows.len,
                        H, rows.max_len, False, scale, work=work, drop=a.drop, max_k=rows.max_len, ores=a.ores, k_prescaled=kpre)
            a.out, f.out, f.h = E(M, d), E(M, d), E(M, ff._st.d_ff)
            if need_bwd:
                a.xhat, a.rstd, f.xhat, f.rstd = E(M, d), E(M, dt=F32), E(M, d), E(M, dt=F32)
            f.drop1, f.drop2 = ff._drops(dev)
            if need_bwd and self.use_bwd:       # the backward chain masks with these bits instead of reading f.h
                f.bits = torch.empty(nv.chain_mask_words(M, ff._st.d_ff, d), dtype=torch.int64, device=dev)
            nxt = layers[l + 1].slf_attn._st if l + 1 < n else None
            qkv = E(M, 3 * d) if nxt is not None else None
            nv.row_chain(a.ctx, self.e[l], pre=(x, sa._st.b_o, sa._st.gamma, sa._st.beta, a.out, a.xhat, a.rstd),
                         ffn=(ff._st.d_ff, ff._st.b1, ff._st.b2, ff._st.gamma, ff._st.beta, f.h, f.out, f.xhat, f.rstd, f.drop1,
                              f.drop2, f.bits),
                         post=(3 * d // BLK, nxt.b_qkv, qkv) if nxt is not None else None, post_kscale=kscale)
            x = f.out
            pres.append((a, f))
        if need_bwd and self.use_bwd:
            cb = EncoderBackward(self, list(layers), pres)
            for l, (a, f) in enumerate(pres):
                a.bwd = f.bwd = cb
                a.key, f.key = ("self", l), ("ffn", l)
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
        if self.f2 and os.environ.get("ST_CHAIN_SPLIT", "1") != "0":      # (development switch)
            work = torch.zeros(nv.split_work_words(), dtype=torch.int32, device=self.f2[0].stream.device)
            for ch in self.f2:
                ch.split_work = work
        # backward chains in running order (last layer first): B2(l) = [q|k|v projection of layer l + 1] + feed-forward +
        # the encoder-decoder attention's output_linear; B1(l) = its q projection + the self-attention's output_linear
        self.bset = ChainSet(arena.device)
        b1, b2 = {}, {}
        for l in range(n - 1, -1, -1):
            sa, ca, ff = layers[l].slf_attn._st, layers[l].enc_attn._st, layers[l].pos_ffn._st
            head = t_blocks(blocks_of(layers[l + 1].slf_attn._st.w_qkv)) if l + 1 < n else []
            b2[l] = self.bset.add(head + ffn_blocks_bwd(ff.w1, ff.w2) + t_blocks(blocks_of(ca.w_o)))
            b1[l] = self.bset.add(t_blocks(blocks_of(ca.w_q)) + t_blocks(blocks_of(sa.w_o)))
        self.bset.finalize()
        self.bwd2 = [self.bset.chain(b2[l], True) for l in range(n)]
        if self.f2 and self.f2[0].split_work is not None and os.environ.get("ST_CHAIN_SPLIT", "1") != "f":      # ("f": forward only)
            for ch in self.bwd2:
                ch.split_work = self.f2[0].split_work       # (forward and backward launches are ordered on one stream)
        self.bwd1 = [self.bset.chain(b1[l], True) for l in range(n)]
        self.use_bwd = sa.n_head * 64 == BLK        # the delta epilogue's heads are 64 columns
        self.layer_hook = None      # optional callable(first_finished_layer): EncoderBackward.input_grad / trainer.TrainStep
        ChainHub.of(arena).add(self.set, self.bset)

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

    def forward(self, layers, x, kv, t_rows, in_rows, need_bwd: bool):
        """The decoder's layer stack on target rows x [M, 256] (embedding + positional encoding), kv = CrossKv's buffer
        (every layer's K | V projection of the encoder output): per layer  causal self-attention, chain F1,
        encoder-decoder attention, chain F2.  -> (output rows, [(self-attention, encoder-decoder attention, feed-forward)
        SubPre per layer]) - nothing here is recorded by autograd; the caller replays the Functions over the SubPre's."""
        from .functional import attn_work, rows_buffer
        M, d = x.shape
        dev = x.device

        def E(*shape, dt=BF16):
            return torch.empty(*shape, dtype=dt, device=dev)

        def EZ(m, n, layout):      # attention outputs: rows past a length are never written - zeros on padded layouts
            return rows_buffer(m, n, layout, dev)

        s0 = layers[0].slf_attn._st
        H = s0.n_head
        scale = 1.0 / math.sqrt(d // H)
        work_self, work_cross = attn_work(t_rows, t_rows, True, d // H, H)[0], attn_work(t_rows, in_rows, False, d // H, H)[0]
        qkv = E(M, 3 * d)
        nv.gemm(x, s0.w_qkv, qkv, bias=s0.b_qkv)
        pres = []
        n = len(layers)
        for l, layer in enumerate(layers):
            sa, ca, ff = layer.slf_attn, layer.enc_attn, layer.pos_ffn
            a, b, f = SubPre(), SubPre(), SubPre()
            # ---- causal self-attention (Attention.py:82-90)
            a.qkv, a.drop = qkv, sa._drop(dev)
            a.ctx, a.lse, a.ores = EZ(M, d, t_rows), E(H * M, dt=F32), (EZ(M, d, t_rows) if need_bwd else None)
            # (launched below, together with the chain stage behind it and the encoder-decoder attention: nv.attn_sf1_fwd)
            # ---- F1: its output_linear + residual + LayerNorm, the next attention's q
            # (nv.attn_f1_fwd writes utterance rows only - the chain launch it replaces wrote every row of the matrix: on
            # padded / bucket layouts the rows outside the utterances must read as zeros, they are operands of the row-wise
            # chains behind and of the weight-gradient contractions)
            a.out = EZ(M, d, t_rows)
            if need_bwd:
                a.xhat = EZ(M, d, t_rows)
                a.rstd = E(M, dt=F32) if t_rows.dense else torch.zeros(M, dtype=F32, device=dev)
            b.qkv = EZ(M, d, t_rows)
            # ---- the self-attention, that stage and the encoder-decoder attention over this layer's column block of kv: ONE launch
            #      where the few-queries kernel serves the shape (nv.attn_sf1_fwd; the three launches otherwise)
            b.kvbuf, b.drop = kv[:, l * 2 * d:(l + 1) * 2 * d], ca._drop(dev)
            b.ctx, b.lse, b.ores = EZ(M, d, t_rows), E(H * M, dt=F32), (EZ(M, d, t_rows) if need_bwd else None)
            nv.attn_sf1_fwd(qkv, a.ctx, a.lse, (x, sa._st.b_o, sa._st.gamma, sa._st.beta, a.out, a.xhat, a.rstd),
                            (1, ca._st.b_q, b.qkv), self.f1[l], b.kvbuf[:, :d], b.kvbuf[:, d:], b.ctx, b.lse, t_rows.off, t_rows.len,
                            in_rows.off, in_rows.len, H, t_rows.max_len, scale, work_self=work_self, work=work_cross,
                            drop_self=a.drop, drop=b.drop, max_k=in_rows.max_len, ores_self=a.ores, ores=b.ores)
            # ---- F2: its output_linear + residual + LayerNorm, the feed-forward sublayer, the next layer's q|k|v
            b.out, f.out, f.h = E(M, d), E(M, d), E(M, ff._st.d_ff)
            if need_bwd:
                b.xhat, b.rstd, f.xhat, f.rstd = E(M, d), E(M, dt=F32), E(M, d), E(M, dt=F32)
            f.drop1, f.drop2 = ff._drops(dev)
            if need_bwd and self.use_bwd:
                f.bits = torch.empty(nv.chain_mask_words(M, ff._st.d_ff), dtype=torch.int64, device=dev)
            nxt = layers[l + 1].slf_attn._st if l + 1 < n else None
            qkv = E(M, 3 * d) if nxt is not None else None
            nv.row_chain(b.ctx, self.f2[l], pre=(a.out, ca._st.b_o, ca._st.gamma, ca._st.beta, b.out, b.xhat, b.rstd),
                         ffn=(ff._st.d_ff, ff._st.b1, ff._st.b2, ff._st.gamma, ff._st.beta, f.h, f.out, f.xhat, f.rstd, f.drop1,
                              f.drop2, f.bits),
                         post=(3, nxt.b_qkv, qkv) if nxt is not None else None)
            x = f.out
            pres.append((a, b, f))
        if need_bwd and self.use_bwd:
            cb = DecoderBackward(self, list(layers), pres)
            for l, (a, b, f) in enumerate(pres):
                a.bwd = b.bwd = f.bwd = cb
                a.key, b.key, f.key = ("self", l), ("cross", l), ("ffn", l)
        return x, pres

"""torch.autograd.Functions composed from the C-ABI HIP kernels.

One Function per reference sub-layer (forward + hand-written backward):

  MhaFn       MultiHeadAttention.forward      transformer/Attention.py:64-96
  FfnFn       PositionwiseFeedForward.forward transformer/SubLayers.py:24-28
  FrontendFn  Encoder.input_proj + PE add     transformer/Models.py:28-33,42-44
  EmbedFn     tgt_word_emb + PE add           transformer/Models.py:84,87 (repair R3)
  VocabFn     tgt_word_proj                   transformer/Models.py:145,151
  Pack/Unpack padded [B,T,*] <-> ragged row matrix (train.py:31-35 trimming)

Activations are bf16 row matrices [rows, d]; parameter gradients are written
straight into the arena's fp32 gradient buffer (atomic accumulate), which is
what ``param.grad`` views - the Functions therefore return ``None`` for their
parameter anchor (the Megatron "main_grad" idiom).
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch

from . import native as nv

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32
LN_EPS = 1e-6  # Attention.py:62, SubLayers.py:18, Models.py:32


_ROWS_CACHE = {}


class Rows:
    """How utterances map onto the rows of an activation matrix.

    Building one costs two small host->device copies; pageable H2D copies block the host
    until the stream drains, so layouts are cached by their length vector (a training
    loader revisits the same bucket shapes, bench.py the same batch) and the model builds
    every layout it needs BEFORE launching the first kernel of a step."""

    __slots__ = ("B", "off", "len", "max_len", "total", "_pos", "lens_host", "dense", "_scatter", "_work", "_seq", "valid_rows")

    def __init__(self, off, length, max_len, total, lens_host=None, dense=True):
        self.dense = dense  # every row of the matrix belongs to some utterance (no padding rows)
        self.B = off.numel()
        self.off, self.len = off, length
        self.max_len, self.total = int(max_len), int(total)
        self._pos = None
        self._scatter = None
        self._work = {}
        self._seq = None      # (packed bucket layouts) row -> utterance, -1 on the unassigned tail rows
        self.valid_rows = None   # (packed bucket layouts) device int32: rows owned by the current batch's utterances
        self.lens_host = lens_host

    @staticmethod
    def packed(lengths: torch.Tensor, device) -> "Rows":
        """Ragged layout: utterance b owns rows cumsum(len)[b-1] ... (no padding rows)."""
        host = lengths.detach().to("cpu", torch.int64)  # one D2H copy if the lengths live on the GPU
        key = ("packed", str(device), host.numpy().tobytes())
        hit = _ROWS_CACHE.get(key)
        if hit is not None:
            return hit
        off = torch.zeros_like(host)
        off[1:] = torch.cumsum(host, 0)[:-1]
        r = Rows(off.to(device=device, dtype=I32), host.to(device=device, dtype=I32), int(host.max()),
                 int(host.sum()), host)
        if len(_ROWS_CACHE) > 256:
            _ROWS_CACHE.clear()
        _ROWS_CACHE[key] = r
        return r

    @staticmethod
    def padded(B: int, T: int, device, lengths: Optional[torch.Tensor] = None) -> "Rows":
        """Padded layout: utterance b owns rows b*T ... b*T+len[b]-1 (len = T when not given)."""
        key = ("padded", str(device), B, T, None if lengths is None else lengths.detach().cpu().numpy().tobytes())
        hit = _ROWS_CACHE.get(key)
        if hit is not None:
            return hit
        r = Rows._padded(B, T, device, lengths)
        if len(_ROWS_CACHE) > 256:
            _ROWS_CACHE.clear()
        _ROWS_CACHE[key] = r
        return r

    @staticmethod
    def _padded(B, T, device, lengths):
        off = torch.arange(B, dtype=I32, device=device) * T
        if lengths is None:
            ln = torch.full((B,), T, dtype=I32, device=device)
            host = None
        else:
            host = lengths.detach().to("cpu", torch.int64)
            ln = host.to(device=device, dtype=I32)
        return Rows(off, ln, T, B * T, host, dense=lengths is None)

    @staticmethod
    def bucket(B: int, T: int, device, rows: Optional[int] = None) -> "Rows":
        """Layout whose LENGTHS live only on the device (``set_lengths`` refreshes them in place): the layout of a
        captured step that serves every batch of a (B, T) bucket - nothing the kernels are launched with depends on the
        lengths (grids and work lists cover every tile an utterance of T rows could have; the kernels skip what lies past a
        length).  rows = None: padded, utterance b owns rows b*T ... (B x T rows).  rows = R: PACKED into R rows - the
        offsets live on the device too, utterance b starts where b - 1 ends, and the rows past the batch's total belong to
        nobody (they carry zero inputs and zero gradients, like the padding rows of the padded form): a step then costs
        what R rows cost, not B x T."""
        if rows is None:
            off = torch.arange(B, dtype=I32, device=device) * T
            r = Rows(off, torch.full((B,), T, dtype=I32, device=device), T, B * T, None, dense=False)
        else:
            if rows < T or rows > B * T:
                raise ValueError("Rows.bucket: a packed capacity of %d rows does not suit %d utterances of up to %d" % (rows, B, T))
            r = Rows(torch.zeros(B, dtype=I32, device=device), torch.zeros(B, dtype=I32, device=device), T, int(rows), None,
                     dense=False)
            r._seq = torch.full((int(rows),), -1, dtype=I32, device=device)
            r.valid_rows = torch.zeros(1, dtype=I32, device=device)
        r.pos        # the position table exists before the capture; set_lengths rewrites it in place
        return r

    _PIN = {}      # device -> ring of (pinned int32 staging buffer, event of the copy that last read it)
    _PIN_SLOTS = 32    # a packed-bucket step stages 9 vectors (two layouts x off / valid_rows / len + 3 work lists): the ring
                       # holds three steps' worth, so a slot's previous copy belongs to a step that finished long ago

    @staticmethod
    def _h2d(dst: torch.Tensor, src: torch.Tensor) -> None:
        """dst (device, int32) <- src (host, any integer dtype), WITHOUT blocking the host: a copy from pageable memory waits
        for the stream to drain (the host then cannot stage batch k + 1 while the GPU runs batch k), so the values go
        through a small ring of pinned buffers; a slot is reused only after the copy that read it has run."""
        if not dst.is_cuda:
            dst.copy_(src.to(dtype=dst.dtype))
            return
        key = str(dst.device)
        ring = Rows._PIN.setdefault(key, {"slots": [], "next": 0})
        n = src.numel()
        if len(ring["slots"]) < Rows._PIN_SLOTS:
            ring["slots"].append([torch.empty(max(n, 1024), dtype=I32).pin_memory(), None])
        i = ring["next"] % len(ring["slots"])
        ring["next"] += 1
        slot = ring["slots"][i]
        if slot[0].numel() < n:
            slot[0], slot[1] = torch.empty(n, dtype=I32).pin_memory(), None
        if slot[1] is not None:
            slot[1].synchronize()
        slot[0][:n].copy_(src.reshape(-1))
        dst.copy_(slot[0][:n].view(dst.shape), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dst.device))
        slot[1] = ev

    def set_lengths(self, lengths: torch.Tensor) -> None:
        """(bucket layout) new utterance lengths, 1 <= len <= T: device vectors and position table rewritten in place."""
        if self.lens_host is not None or self.dense:
            raise RuntimeError("Rows.set_lengths: only for Rows.bucket layouts")
        if int(lengths.min()) < 1 or int(lengths.max()) > self.max_len or lengths.numel() != self.B:
            raise ValueError("Rows.set_lengths: lengths must lie in [1, %d] for %d utterances" % (self.max_len, self.B))
        if self._seq is not None:      # packed: the offsets follow the lengths
            host = lengths.detach().to("cpu", torch.int64)
            if int(host.sum()) > self.total:
                raise ValueError("Rows.set_lengths: %d rows do not fit the packed capacity of %d" % (int(host.sum()), self.total))
            off = torch.zeros_like(host)
            off[1:] = torch.cumsum(host, 0)[:-1]
            Rows._h2d(self.off, off)
            Rows._h2d(self.valid_rows, host.sum().reshape(1))
            self._seq.fill_(-1)
        if lengths.is_cuda:
            self.len.copy_(lengths.to(dtype=I32), non_blocking=True)
        else:
            Rows._h2d(self.len, lengths)
        self._pos.zero_()
        nv.row_index(self.off, self.len, self.max_len, self._pos, self._seq)
        if self._seq is not None and self._scatter is not None:
            self._fill_scatter()

    def _fill_scatter(self) -> None:
        """(packed bucket) row r -> position seq[r] * L + pos[r] of the padded [B, L] layout; B * L (one element past it: the
        caller appends an ignored target there) on the unassigned tail rows."""
        L, buf = self._scatter[0][1], self._scatter[1]
        seq = self._seq.long()
        torch.where(seq >= 0, seq * L + self._pos.long(), torch.full_like(seq, self.B * L), out=buf)

    @property
    def is_bucket(self) -> bool:
        return self.lens_host is None and not self.dense

    def scatter_index(self, L: int) -> torch.Tensor:
        """Row b*L + t of a padded [B, L, *] tensor for every packed row (int64, on device)."""
        if self.is_bucket and self._seq is not None:
            if self._scatter is None or self._scatter[0] != ("scatter", L):
                self._scatter = (("scatter", L), torch.empty(self.total, dtype=torch.int64, device=self.off.device))
                self._fill_scatter()
            return self._scatter[1]
        if self.is_bucket:
            if L != self.max_len:
                raise ValueError("Rows.scatter_index: a bucket layout of %d rows per utterance cannot index a [B, %d] tensor"
                                 % (self.max_len, L))
            if self._scatter is None:
                self._scatter = (("scatter", L), torch.arange(self.total, device=self.off.device))
            return self._scatter[1]
        key = ("scatter", L)
        cache = getattr(self, "_scatter", None)
        if cache is None or cache[0] != key:
            lens = self.lens_host
            seq = torch.repeat_interleave(torch.arange(self.B), lens)
            idx = seq * L + torch.cat([torch.arange(int(n)) for n in lens])
            cache = (key, idx.to(self.off.device))
            self._scatter = cache
        return cache[1]

    @property
    def pos(self) -> torch.Tensor:
        if self._pos is None:
            p = torch.zeros(self.total, dtype=I32, device=self.off.device)
            nv.row_index(self.off, self.len, self.max_len, p)
            self._pos = p
        return self._pos


def attn_work(q_rows: Rows, k_rows: Rows, causal: bool, d_k: int = 64, n_head: int = 0):
    """Work lists of the attention kernels for one (query layout, key layout, causal, head width) combination:
    int32 device vectors of (b << 16) | tile over query tiles (forward; backward dQ) and key tiles (backward dK/dV),
    sorted by decreasing number of streamed 64-row tiles - the dispatcher hands workgroups out in this order, i.e.
    longest-first list scheduling of the ragged batch.  The rows per tile are the kernels' own (``native.attn_tile_rows``:
    the kernel is chosen by problem shape).  -> (forward, backward dQ, backward dK/dV).  Cached on the query layout;
    the model calls this while it builds the layouts, before the first kernel of a step (no H2D copy mid-step).

    XCD affinity (n_head in {1, 2, 4, 8}): workgroup ``bid`` runs on XCD ``bid % 8`` and the kernels enumerate
    ``bid = position * n_head + head``, so with the utterances dealt into 8 / n_head cost-balanced groups and group g
    holding the list positions = g (mod 8 / n_head), every (utterance, head) is served by ONE XCD: its K / V (Q / dO)
    rows are fetched into one L2 instead of up to eight.  Shorter groups are padded with no-op entries (tile 0xffff)."""
    key = (id(k_rows), bool(causal), int(d_k), int(n_head))
    hit = q_rows._work.get(key)
    if hit is not None and hit[0] is k_rows:
        return hit[1]
    lq = q_rows.lens_host.tolist() if q_rows.lens_host is not None else [q_rows.max_len] * q_rows.B
    lk = k_rows.lens_host.tolist() if k_rows.lens_host is not None else [k_rows.max_len] * k_rows.B
    dev = q_rows.off.device
    out = tuple(torch.tensor(flat, dtype=I32).to(dev) for flat in _work_lists(q_rows, k_rows, causal, d_k, n_head, lq, lk))
    q_rows._work[key] = (k_rows, out)
    return out


def _work_lists(q_rows, k_rows, causal, d_k, n_head, lq, lk, xcd_groups=True):
    """The three lists of attn_work as Python lists, for the utterance lengths lq / lk."""
    rows = [nv.attn_tile_rows(w, d_k, q_rows.max_len, k_rows.max_len, causal) for w in range(3)]
    ng = 8 // n_head if xcd_groups and n_head in (1, 2, 4, 8) and not os.environ.get("ST_NO_XCD_AFFINITY") else 1
    # deal the utterances into ng groups of equal total cost (longest first onto the lightest group)
    group, load = [0] * q_rows.B, [0.0] * ng
    if ng > 1:
        for b in sorted(range(q_rows.B), key=lambda b: -lq[b] * lk[b]):
            g = min(range(ng), key=lambda g: load[g])
            group[b] = g
            load[g] += lq[b] * lk[b]
    lists = tuple([[] for _ in range(ng)] for _ in range(3))
    for b in range(q_rows.B):
        for w in (0, 1):          # query tiles: cost = 64-key tiles streamed
            R = rows[w]
            for t in range((lq[b] + R - 1) // R):
                seen = min(lk[b], (t + 1) * R) if causal else lk[b]
                lists[w][group[b]].append(((seen + 63) // 64, (b << 16) | t))
        R = rows[2]               # key tiles: cost = 64-query tiles streamed
        for t in range((lk[b] + R - 1) // R):
            q_begin = (t * R // 64) * 64 if causal else 0
            lists[2][group[b]].append(((lq[b] - q_begin + 63) // 64, (b << 16) | t))
    out = []
    for per_group in lists:
        for g in per_group:
            g.sort(key=lambda c: -c[0])
        n = max(len(g) for g in per_group)
        flat = []
        for i in range(n):        # position i * ng + g <- group g's i-th heaviest item (a no-op where the group ran out)
            for g in per_group:
                flat.append(g[i][1] if i < len(g) else 0xffff)
        out.append(flat)
    return out


def refresh_attn_work(q_rows: Rows, k_rows: Rows, causal: bool, d_k: int, n_head: int, lq, lk) -> None:
    """(bucket layouts) the cached work lists of this combination re-ordered for the current batch's lengths lq / lk (host
    lists) - longest first again, the tiles the batch does not have as no-op entries at the end - and copied over the
    device vectors in place (their size is fixed: every tile an utterance of max_len rows could have), without blocking the
    host.  A captured step then dispatches this batch's workgroups in this batch's best order."""
    hit = q_rows._work.get((id(k_rows), bool(causal), int(d_k), int(n_head)))
    if hit is None or hit[0] is not k_rows:
        raise RuntimeError("refresh_attn_work: no cached lists for this combination (call attn_work first)")
    lists = _work_lists(q_rows, k_rows, causal, d_k, n_head, list(lq), list(lk))
    if any(len(flat) > dst.numel() for dst, flat in zip(hit[1], lists)):
        # the XCD groups of this batch are less even than the padding of the fixed-size vectors allows: one plain
        # longest-first list (never longer than the vectors: they hold every tile of B utterances of max_len rows)
        lists = _work_lists(q_rows, k_rows, causal, d_k, n_head, list(lq), list(lk), xcd_groups=False)
    for dst, flat in zip(hit[1], lists):
        Rows._h2d(dst, torch.tensor(flat + [0xffff] * (dst.numel() - len(flat)), dtype=I32))


def _splits(M: int, N: int, K: int) -> int:
    """Split count of the token (contraction) axis of a weight-gradient GEMM: a power of two so that the
    kernel's XCD-local tile walk keeps whole splits on one XCD; one 8-wave workgroup per CU (its two wave
    groups split the range once more internally - fewer splits = fewer fp32 atomics, the bound here)."""
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    want = max(1, min((M + 255) // 256, (256 + tiles - 1) // tiles))
    s = 1
    while s * 2 <= want:
        s *= 2
    return s


class _Deferred:
    """Weight-gradient launches produce nothing the rest of the backward pass reads (only parameter
    gradients).  Inside ``deferred_wgrads()`` they are therefore collected and issued as grouped launches
    (``st_wgrad_group``): the decoder's when its backward is done (``flush_deferred_wgrads`` from
    EmbedFn.backward), the encoder's on exit.  One by one the decoder-sized ones are pure launch latency
    (~20 workgroups each, 36 per step) and the encoder-sized ones one round of ~250 workgroups whose prologue and
    atomic epilogue nothing overlaps; grouped, ~2300 workgroups stream through the CUs back to back.
    Measured on config 2: 1.15 ms in 32 launches -> ~0.55 ms in 2.  Moving the launches to
    a second stream inside the captured graph was tried and is slower: a cross-queue dependency costs ~50 us in a
    HIP graph.  The operand tensors stay referenced by the list until the launch is enqueued."""
    active = False
    pending = []
    WIDE_ROWS = int(os.environ.get("ST_WIDE_ROWS", "2048"))   # token counts from here up take st_wgrad_wide (8192 until round 5; a 4-utterance shard's
                                                              # 3,120 rows: step 1.675 -> 1.639 ms with the wide kernel; the variable: development)
    ROWS_PER_SPLIT = 3072     # ~48 k-steps per workgroup; measured on config 2 (24060 rows): 8 splits beat 4, 12 and 16

    @staticmethod
    def splits(rows: int) -> int:
        """Power of two (the kernel's XCD-local tile walk keeps whole splits on one XCD); 1 for decoder-sized problems."""
        want, s = max(1, round(rows / _Deferred.ROWS_PER_SPLIT)), 1
        while s * 2 <= want:
            s *= 2
        return s


WIDE_CHUNK = 48    # problems per st_wgrad_wide launch (csrc/st_wgrad.hip WIDE_MAX: the descriptors travel as kernel arguments)


_CUS = {}


def _device_cus(device) -> int:
    """Compute units of the device the problems live on (256 on an MI355X; cached)."""
    idx = torch.device(device).index if torch.device(device).type == "cuda" else None
    if idx is None:
        return 256
    if idx not in _CUS:
        _CUS[idx] = int(torch.cuda.get_device_properties(idx).multi_processor_count)
    return _CUS[idx]


def _wide_plan(problems, cus: int = 0):
    """Launches of the 256 x 256-tile kernel (st_wgrad_wide): the problems in chunks of WIDE_CHUNK, each chunk with the token-split
    count that finishes first.  Items of one launch are equally long (tokens / splits) and one workgroup owns a CU, so a launch takes
    ceil(tiles x splits / CUs) rounds of tokens / splits each.  Cost model (round 6, tools/dev/wgrad_small_m.py: 12- and 85-tile
    launches at 3,120 .. 24,060 tokens, 1 .. 21 splits, two boxes): 17.3 ns per token of an item's k-loop, 12 us per launch, and 0.25 us per item
    for its 256 KB of fp32 atomics (the term that grows with the splits: until round 6 the model charged a flat 50 us per round and
    took as many splits as fit one round - a DP step's per-layer launches, 12 tiles, ran 21 splits where 12 are 23 % faster, 12 where
    4 are 28 % faster at a 4-utterance shard).  Config 2 (85 tiles): 3 splits = 255 items, one round; its 4-utterance shard: 2.
    Config 3 (12 + 6 layers of width 512: 384 tiles in the first launch, 52 in the second): 2 and 4 splits.  At least 256 tokens
    per item.  Returns a list of argument lists for nv.wgrad_group(wide=True)."""
    launches = []
    if cus <= 0:
        cus = _device_cus(problems[0][0].device) if problems else 256
    for c0 in range(0, len(problems), WIDE_CHUNK):
        chunk = problems[c0:c0 + WIDE_CHUNK]
        tiles = sum(-(-p[0].shape[1] // 256) * -(-p[5] // 256) for p in chunk)
        rows = min(p[0].shape[0] for p in chunk)
        loop_us = 17.3e-3 * rows                # one item over all tokens (54 us at 3,120 tokens, 416 at 24,060)
        best, best_cost = 1, None
        for sp in range(1, max(1, min(rows // 256, 4 * cus // max(tiles, 1) + 1)) + 1):
            rounds = -(-tiles * sp // cus)
            cost = rounds * loop_us / sp + 12.0 + 0.25 * tiles * sp
            if best_cost is None or cost < best_cost - 1e-9:
                best, best_cost = sp, cost
        launches.append([p[:4] + (best, p[5]) for p in chunk])
    return launches


def flush_deferred_wgrads(final: bool = True):
    """Encoder-sized problems go to the wide-tile kernel, decoder-sized ones to the 128 x 128 grouped launch.
    final=False (the end of the decoder's backward): the encoder-sized problems collected so far - the decoder's
    encoder-decoder key/value projections - stay pending and join the encoder's launch, which then has one workgroup
    per CU at 3 token splits (config 2: 73 + 12 tiles); on their own they are 12 tiles at 21 splits.
    The column sums the encoder's backward chains left in workspaces (nv.row_chain_bwd) are folded into their gradients here too."""
    nv.flush_colsum_folds()
    if _Deferred.pending:
        wide = [p for p in _Deferred.pending if p[0].shape[0] >= _Deferred.WIDE_ROWS]
        rest = [p for p in _Deferred.pending if p[0].shape[0] < _Deferred.WIDE_ROWS]
        if rest:
            nv.wgrad_group(rest)
        if wide and final:
            for launch in _wide_plan(wide):
                nv.wgrad_group(launch, wide=True)
            wide = []
        _Deferred.pending[:] = wide


class deferred_wgrads:
    """Context manager around loss.backward(): batch the weight-gradient GEMMs into grouped launches
    (flushed at the end of the decoder's backward and on exit)."""

    def __init__(self, enable: bool = True):
        self.enable = enable

    def __enter__(self):
        _Deferred.active = self.enable
        nv.fold_deferred = self.enable
        return self

    def __exit__(self, *exc):
        flush_deferred_wgrads()
        _Deferred.active = False
        nv.fold_deferred = False
        return False


def wgrad(dY, X, gW, rows=None, gB=None):
    """gW[n][k] += sum_m dY[m][n] X[m][k]   (both operands contraction-major).  The kernel's
    lane axis is k (the contiguous axis of gW) so the split-K atomics are coalesced.
    ``gB``: the same launch also accumulates the bias gradient gB[n] += sum_m dY[m][n]."""
    N = dY.shape[1] if rows is None else rows
    splits = _splits(dY.shape[0], N, X.shape[1])
    if _Deferred.active:
        _Deferred.pending.append((X, dY, gW, gB, _Deferred.splits(dY.shape[0]), N))   # the argument tuple of nv.wgrad_group
        return
    nv.gemm(X, dY, gW, bias=gB, epi=nv.EPI_F32_ATOMIC_T, x_cmajor=True, y_cmajor=True, splits=splits, n=N)


def linear_fwd(x, w, out, bias, relu=False, drop=None, stack=None):
    """out = act(x w^T + bias): the weight-stationary streaming kernel for encoder-sized inputs of width 256
    (st_gemm_ws), the tiled kernel otherwise."""
    n = w.shape[0] * (stack[0] if stack is not None else 1)
    if nv.ws_ok(x.shape[0], n, x.shape[1]) and (stack is None or (w.shape[0] % 256 == 0 and not (w.shape[0] // 256) & (w.shape[0] // 256 - 1))):
        return nv.gemm_ws(x, w, out, bias=bias, relu=relu, drop=drop, stack=stack)
    return nv.gemm(x, w, out, bias=bias, epi=nv.EPI_BF16_RELU if relu else nv.EPI_BF16, drop=drop, stack=stack)


def dgrad(dY, W, out, epi=nv.EPI_BF16, aux=None, kc=None, drop=None):
    """out[m][k] = sum_n dY[m][n] W[n][k]  (+ aux | masked by aux > 0, survivors scaled by drop.scale)."""
    return nv.gemm(dY, W, out, epi=epi, aux=aux, y_cmajor=True, kc=kc, drop=drop)


def dgrad_long_k(dY, W, out):
    """dgrad for FEW output tiles and a LONG contraction (the vocabulary projection's input gradient: 1,206 target rows x 256
    columns over 4,344 vocabulary entries = 20 output tiles that each walk the whole contraction: 45 us on 20 compute
    units): the contraction cut 6 ways over workgroups, merged by the last one per tile (st_gemm_splitk: 40 -> 23 us; 2 / 4 / 8 /
    12 splits: 29 / 24 / 26 / 36 us - the exchange costs what the shorter loop saves).  Elsewhere: dgrad."""
    M, K = dY.shape
    tiles = ((M + 127) // 128) * ((out.shape[1] + 127) // 128)
    if tiles > 32 or K < 2048:
        return dgrad(dY, W, out)
    return nv.gemm_splitk(dY, W, out, max(2, min(6, 128 // tiles)), y_cmajor=True)


def _empty(rows, cols, like, dtype=BF16):
    return torch.empty(rows, cols, dtype=dtype, device=like.device)


class TailBuffers:
    """The row matrices of a step whose rows outside every utterance must read as zeros (attention outputs and gradients:
    the kernels write utterance rows only).  On a dense layout nothing needs zeroing; on a padded one the whole buffer is
    zero-filled.  On a PACKED bucket layout only the tail behind the batch's last utterance is unassigned, and a captured
    step can do better than one memset node per buffer: while the step runs eagerly once, ``record`` notes every request;
    before the capture ``materialize`` allocates them all, persistently and exclusively (a buffer from the graph's pool
    could alias memory another tensor used earlier in the step, and its tail would then not be zero when it is read);
    during the capture ``request`` hands them out in the same order, and the captured step begins with ONE
    ``st_zero_tails`` launch over the whole list."""
    active = None
    N_MAX = 256

    def __init__(self):
        self.mode, self.specs, self.bufs, self.i, self.table, self.missed = "record", [], [], 0, None, 0

    def request(self, m, n, layout, device):
        if self.mode == "record":
            self.specs.append((int(m), int(n), layout))
            return torch.zeros(m, n, dtype=BF16, device=device)
        if self.i < len(self.bufs) and self.specs[self.i][:2] == (int(m), int(n)) and self.specs[self.i][2] is layout:
            self.i += 1
            return self.bufs[self.i - 1]
        self.missed += 1          # (a request the recorded step did not make: served the plain way)
        return torch.zeros(m, n, dtype=BF16, device=device)

    def materialize(self, device):
        self.specs = self.specs[:TailBuffers.N_MAX]
        self.bufs = [torch.zeros(m, n, dtype=BF16, device=device) for m, n, _ in self.specs]
        rows = []
        for buf, (m, n, layout) in zip(self.bufs, self.specs):
            rows += [buf.data_ptr(), n * buf.element_size(), m, layout.valid_rows.data_ptr()]
        rows += [0, 0, 0, 0] * (TailBuffers.N_MAX - len(self.bufs))
        self.table = torch.tensor(rows, dtype=torch.int64).to(device)
        self.mode, self.i = "serve", 0
        return self


def rows_buffer(m, n, layout, device):
    """bf16 [m, n] for a kernel that writes utterance rows only: uninitialised on a dense layout, else zero outside the
    utterances (see TailBuffers)."""
    if layout.dense:
        return torch.empty(m, n, dtype=BF16, device=device)
    tb = TailBuffers.active
    if tb is not None and layout._seq is not None and n * 2 % 16 == 0:
        return tb.request(m, n, layout, device)
    return torch.zeros(m, n, dtype=BF16, device=device)


# ------------------------------------------------------------------------------------------------
class CrossGradAcc:
    """Shared by the N decoder-encoder attentions of one forward pass (they all read the same encoder output):
    their backward passes run in reverse layer order and add their encoder gradients into ``buf`` inside the
    dgrad GEMM's epilogue; the N-th one returns the sum.  Saves N - 1 full-size adds and their launches."""
    __slots__ = ("n", "seen", "buf")

    def __init__(self, n: int):
        self.n, self.seen, self.buf = n, 0, None


class CrossKv:
    """The key/value projections of the encoder output for ALL decoder layers (Attention.py:75-76 of every
    decoder-encoder attention) as one GEMM, and their input gradient as one GEMM: the layers' [2d, d] weights are
    equally spaced in the parameter arena, so ``st_gemm_stacked`` reads them as one [n * 2d, d] operand.  Takes 2n
    full-size launches out of the decoder's (latency-bound) chain.  Layer l reads columns [l * 2d, (l + 1) * 2d) of
    ``kv`` and writes its dK | dV into the same columns of ``dkv``; the last layer to run backward (layer 0) hands
    ``dkv`` to CrossKvFn.backward as THE gradient of ``kv`` (the other layers return None for it)."""
    __slots__ = ("mods", "n", "w_stride", "b_stride", "seen", "dkv")

    def __init__(self, mods, w_stride, b_stride):
        self.mods, self.n, self.w_stride, self.b_stride, self.seen, self.dkv = mods, len(mods), w_stride, b_stride, 0, None

    @staticmethod
    def plan(mods):
        """-> CrossKv for these decoder-encoder attention modules, or None when one GEMM cannot serve them
        (different shapes, unequal spacing, live gradient-ready hooks that expect per-layer weight gradients)."""
        if len(mods) < 2:
            return None
        a = mods[0]._st_arena
        if a is None or a._grad_ready_cb is not None:
            return None
        cached = getattr(mods[0], "_st_crosskv", None)           # the layout check is done once per arena
        if cached is None or cached[0] is not a:
            cached = (a, CrossKv._strides(mods, a))
            mods[0]._st_crosskv = cached
        return None if cached[1] is None else CrossKv(list(mods), *cached[1])

    @staticmethod
    def _strides(mods, a):
        if any(m._st_arena is not a for m in mods):
            return None
        d = mods[0]._st.d_model
        if any(m._st.d_model != d or m._st.n_head != mods[0]._st.n_head for m in mods) or (2 * d) & (2 * d - 1) or d < 64:
            return None
        w_off = [a.offset[id(m.linear_k.weight)] for m in mods]
        b_off = [a.offset[id(m.linear_k.bias)] for m in mods]
        ws, bs = w_off[1] - w_off[0], b_off[1] - b_off[0]
        if ws <= 0 or bs <= 0 or any(w_off[i + 1] - w_off[i] != ws or b_off[i + 1] - b_off[i] != bs for i in range(len(mods) - 1)):
            return None
        return ws, bs


class CrossKvSlot:
    """What layer ``idx`` is handed instead of a CrossGradAcc: its column block of the shared CrossKv buffers."""
    __slots__ = ("state", "idx")

    def __init__(self, state, idx):
        self.state, self.idx = state, idx


class CrossKvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, anchor, state: CrossKv):
        s0 = state.mods[0]._st
        kv = _empty(enc.shape[0], state.n * 2 * s0.d_model, enc)
        linear_fwd(enc, s0.w_kv, kv, s0.b_kv, stack=(state.n, state.w_stride, state.b_stride))
        ctx.save_for_backward(enc)
        ctx.state = state
        return kv

    @staticmethod
    def backward(ctx, dkv):
        enc, = ctx.saved_tensors
        state = ctx.state
        s0 = state.mods[0]._st
        w = 2 * s0.d_model
        dkv = dkv.contiguous()
        for l, m in enumerate(state.mods):
            s = m._st
            m._st_arena.attach_grads(s.params, s.lo, s.hi)
            wgrad(dkv[:, l * w:(l + 1) * w], enc, s.g_w_kv, gB=s.g_b_kv)
        d_enc = _empty(enc.shape[0], s0.d_model, enc)
        nv.gemm(dkv, s0.w_kv, d_enc, y_cmajor=True, stack=(state.n, state.w_stride, 0))
        return d_enc, None, None


class LnLink:
    """Joins two consecutive sublayers S1 -> S2 of one forward pass so that S2's backward can run S1's LayerNorm
    backward inside its own last dgrad GEMM (``nv.gemm_lnbwd``): S1 (the LayerNorm's owner) records what that needs
    in its forward; S2's backward consumes it, leaves the result in ``ds`` and returns it as its input gradient;
    S1's backward recognises that tensor and skips its own ``ln_bwd``.  Valid only while S1's output has S2 as its
    single consumer (the layer stacks guarantee it); anything else arriving at S1 raises."""
    __slots__ = ("ok", "st", "arena", "xhat", "rstd", "g_bias", "drop", "ds")

    def __init__(self):
        self.ok, self.ds = False, None

    def offer(self, mod, xhat, rstd, g_bias, drop=None):
        """S1.forward: what S2 needs to run this LayerNorm's backward (drop: dropout applied to the LN output)."""
        self.ok, self.st, self.arena, self.xhat, self.rstd, self.g_bias = True, mod._st, mod._st_arena, xhat, rstd, g_bias
        self.drop = drop

    def fused_dgrad(self, dY, W, aux):
        """S2.backward: d(S2 input) = dY W + aux, pushed through S1's LayerNorm backward in the same launch."""
        st = self.st
        self.arena.attach_grads(st.params, st.lo, st.hi)      # before the kernel accumulates into S1's slots
        ds = _empty(dY.shape[0], W.shape[1], dY)
        nv.gemm_lnbwd(dY, W, aux, self.xhat, self.rstd, st.gamma, ds, st.g_gamma, st.g_beta, self.g_bias, drop=self.drop)
        self.ds = ds
        return ds

    def claim(self, dout):
        """S1.backward: the already-normalised gradient if S2 produced it, else None (run ln_bwd as usual)."""
        ds, self.ds = self.ds, None
        if ds is None:
            return None
        if dout.data_ptr() != ds.data_ptr() or dout.shape != ds.shape:
            raise RuntimeError("LnLink: the gradient reaching this sublayer is not the one its consumer fused the "
                               "LayerNorm backward into (its output has more than one consumer?)")
        return ds


def _input_grad(link, dY, W, aux):
    """The last GEMM of a sublayer's backward: its input gradient, fused with the upstream LayerNorm when linked."""
    if link is not None and link.ok:
        return link.fused_dgrad(dY, W, aux)
    dx = _empty(dY.shape[0], W.shape[1], dY)
    dgrad(dY, W, dx, epi=nv.EPI_BF16_ADD, aux=aux)
    return dx


class CtcPlan:
    """What the CTC head needs to know about a batch, computed once per batch signature (tiny eager torch ops; a captured
    step reads these tensors by address): ctc_loss is fed a SMALL alphabet per utterance - class 0 = blank, class 1 + j = the
    label first seen at target position j (equal labels share a class, so the repeated-label rule of CTC is untouched) -
    and only those columns of the [rows, V] logits are ever turned into log-probabilities."""

    def __init__(self, targets, target_lengths, input_lengths, in_rows: Rows, blank: int, v_pad: int):
        dev = targets.device
        B, L = targets.shape
        self.B, self.L, self.C, self.T = B, L, L + 1, in_rows.max_len
        self.blank = int(blank)
        self._tl = target_lengths.to(device=dev, dtype=torch.int64)
        self._il = input_lengths.to(device=dev, dtype=torch.int64)
        # the label-derived tensors (filled by refresh_labels; a captured step reads them by address)
        self.classes = torch.zeros(B, L, dtype=torch.int64, device=dev)               # the targets ctc_loss is given
        self.cols = torch.zeros(B, L + 1, dtype=I32, device=dev)                      # class -> vocabulary column
        self.scat = torch.zeros(B, L + 1, dtype=I32, device=dev)                      # gradients return through first occurrences only
        self.finite = torch.zeros(B, dtype=torch.bool, device=dev)
        self.tl = self._tl.clamp_min(1)
        self.in_len_host = [int(v) for v in input_lengths.tolist()]
        self.tgt_len_host = [int(v) for v in target_lengths.tolist()]
        self.refresh_labels(targets)
        self.rowmap = in_rows.scatter_index(self.T)
        self.lp = torch.zeros(B, self.T, self.C, dtype=F32, device=dev)              # frames past a length stay 0 (never read)
        self.g_lp = torch.zeros(B, self.T, self.C, dtype=F32, device=dev)            # ctc_loss's gradient, staged for the backward graph
        self.roww = torch.zeros(B, dtype=F32, device=dev)                            # weight of utterance b's softmax term
        self.one = torch.ones(1, dtype=F32, device=dev)
        self.v_pad = v_pad


    def refresh_labels(self, targets) -> None:
        """(Re)derive everything that depends on the label VALUES, written in place: a loader that refills the same static
        target buffer with new labels of the same lengths keeps its batch signature (addresses, shapes, lengths), so
        JointTrainStep calls this on every step - a handful of tiny device ops - instead of trusting a plan built from the
        first batch's contents."""
        dev, B, L = targets.device, self.B, self.L
        tl = self._tl
        valid = torch.arange(L, device=dev).view(1, -1) < tl.view(-1, 1)
        # ext = [blank, label 0, label 1, ...]: the class of an entry is the index of the FIRST entry with the same vocabulary
        # id (a label equal to the blank id - this repository's synthetic ground truth ends every utterance with id 0 - is the
        # blank's class 0, exactly as ctc_loss treats it on the dense alphabet)
        ext = torch.cat([torch.full((B, 1), self.blank, dtype=torch.int64, device=dev), targets.to(torch.int64)], 1)     # [B, L + 1]
        first = (ext.unsqueeze(2) == ext.unsqueeze(1)).to(torch.int8).argmax(dim=2)                                      # [B, L + 1]
        self.classes.copy_(first[:, 1:])
        self.cols.copy_(ext)
        idx = torch.arange(L + 1, device=dev).view(1, -1)
        keep = (first == idx) & torch.cat([torch.ones(B, 1, dtype=torch.bool, device=dev), valid], 1)
        self.scat.copy_(torch.where(keep, ext, torch.full_like(ext, -1)))
        # an utterance whose frames cannot spell its labels has an infinite loss: zero_infinity drops it from the objective
        rep = ((targets[:, 1:] == targets[:, :-1]) & valid[:, 1:]).sum(1) if L > 1 else torch.zeros(B, dtype=torch.int64, device=dev)
        self.finite.copy_(self._il >= tl + rep)


class CtcProjFn(torch.autograd.Function):
    """lp = log_softmax(enc W^T + b)[the columns ctc_loss reads]  (transformer/Loss.py:CTCAttentionLoss, BASELINE config 4): the
    encoder-side vocabulary projection as an st_gemm over the ragged encoder rows, st_ctc_gather instead of the [T, B, V]
    log-softmax; backward: st_ctc_dlogits (softmax term + ctc_loss's small gradient -> bf16 logits gradient), then the
    projection's two backward GEMMs.  The head's parameters live outside the model's arena: their bf16 / padded copies and
    fp32 gradient buffers belong to the head (`head._st_*`), W.grad / b.grad are views of those buffers."""

    @staticmethod
    def forward(ctx, enc, W, b, head, plan: CtcPlan):
        V, d = W.shape
        wb, bias = head._st_wb, head._st_bias
        wb[:V].copy_(W)
        bias[:V].copy_(b)
        logits = torch.empty(enc.shape[0], plan.v_pad, dtype=F32, device=enc.device)
        nv.gemm(enc, wb, logits, epi=nv.EPI_F32, bias=bias)          # padding columns at -1e30: probability 0
        lse = torch.empty(enc.shape[0], dtype=F32, device=enc.device)
        nv.ctc_gather(logits, plan.rowmap, plan.T, plan.cols, lse, plan.lp)
        ctx.save_for_backward(enc, logits, lse)
        ctx.head, ctx.plan, ctx.V = head, plan, V
        return plan.lp.view_as(plan.lp)

    @staticmethod
    def backward(ctx, g_lp):
        enc, logits, lse = ctx.saved_tensors
        head, plan, V = ctx.head, ctx.plan, ctx.V
        dl = torch.empty(logits.shape, dtype=BF16, device=logits.device)
        nv.ctc_dlogits(logits, lse, plan.rowmap, plan.T, plan.roww, plan.scat, g_lp.contiguous(), plan.one, dl, V=V)
        wgrad(dl, enc, head._st_gw, gB=head._st_gb)
        dx = _empty(enc.shape[0], enc.shape[1], enc)
        dgrad(dl, head._st_wb, dx)
        return dx, None, None, None, None


class AttnTap:
    """``with AttnTap() as tap:`` - every attention sublayer whose forward runs inside materialises its probabilities
    (native.attn_probs: f32 [B, h, max q_len, max k_len], dropout not applied) and appends (module, map) to ``tap.maps``
    in call order.  The return_attns path of transformer.Models (reference Models.py:53-54,107-109); the fused kernels
    themselves never write these tensors."""
    active = None

    def __enter__(self):
        self.maps, self._prev = [], AttnTap.active
        AttnTap.active = self
        return self

    def __exit__(self, *exc):
        AttnTap.active = self._prev
        return False

    @staticmethod
    def record(mod, Q, K, q_rows, k_rows, causal, kpre=False):
        """append `mod`'s probability map for projected queries Q / keys K (row matrices, head h at columns h * d_k; kpre: the
        keys are pre-scaled - native.attn_fwd's k_prescaled)"""
        s = mod._st
        scale = 1.0 / math.sqrt(s.d_model // s.n_head)
        AttnTap.active.maps.append((mod, nv.attn_probs(Q, K, q_rows.off, q_rows.len, k_rows.off, k_rows.len, s.n_head, q_rows.max_len,
                                                       k_rows.max_len, causal, scale, k_prescaled=kpre)))

    @staticmethod
    def record_chain(layers, pres, q_rows, kv_rows):
        """The maps of a layer stack whose forward ran as row chains WITHOUT the autograd replay (no gradient wanted: the
        sublayer Functions, which feed the tap otherwise, are never called).  pres: chains.SubPre tuples per layer -
        (self-attention, feed-forward) for an encoder, (self-attention, encoder-decoder attention, feed-forward) for a decoder."""
        if AttnTap.active is None or pres is None:
            return
        for layer, pre in zip(layers, pres):
            a = pre[0]
            d = layer.slf_attn._st.d_model
            AttnTap.record(layer.slf_attn, a.qkv[:, :d], a.qkv[:, d:2 * d], q_rows, q_rows, len(pre) == 3, bool(a.kpre))
            if len(pre) == 3:
                b = pre[1]
                AttnTap.record(layer.enc_attn, b.qkv, b.kvbuf[:, :d], q_rows, kv_rows, False)

    def of(self, module, Lq, Lk):
        """the maps of `module`'s calls, zero-padded to [B, h, Lq, Lk] (the padded batch layout of the reference)"""
        out = []
        for m, P in self.maps:
            if m is module:
                out.append(torch.nn.functional.pad(P, (0, Lk - P.shape[3], 0, Lq - P.shape[2])) if (P.shape[2] != Lq or P.shape[3] != Lk) else P)
        return out


class MhaFn(torch.autograd.Function):
    """out = LN(attn(x_q W_q, x_kv W_k, x_kv W_v) W_o + b_o + x_q)   (Attention.py:64-96, R2)."""

    @staticmethod
    def forward(ctx, x_q, x_kv, anchor, mod, q_rows: Rows, k_rows: Rows, causal: bool, want_attn: bool, drop=None,
                kv_acc=None, up=None, down=None, pre=None):
        """up / down: LnLink to the sublayer that produced x_q / that consumes the output (see LnLink).
        pre (chains.SubPre): everything below was already computed by the fused decoder launches - record only."""
        s = mod._st
        d, H = s.d_model, s.n_head
        Mq = x_q.shape[0]
        scale = 1.0 / math.sqrt(d // H)
        if pre is not None:
            qkv, kvbuf, attn_ctx, ores, lse, out, xhat, rstd = (pre.qkv, pre.kvbuf, pre.ctx, pre.ores, pre.lse, pre.out,
                                                                pre.xhat, pre.rstd)
            kpre = bool(pre.kpre)
        else:
            # non-causal self-attention (an encoder layer on the per-GEMM path: config 3's width, the stand-alone module): the
            # projection hands the attention kernels PRE-SCALED keys, as the encoder's row chains do (chains.EncoderChains)
            # (pre-scaled keys on the per-GEMM path go through st_gemm's column-range epilogue; where the weight-stationary
            # st_gemm_ws serves the projection - width 256, encoder-sized row counts, 1.2-1.4x the tiled kernel - it is kept and
            # the keys stay plain: both forms are exact to their one rounding, the attention kernels take either.  ADVICE r5)
            kpre = (x_kv is None and not causal and MhaFn.PRESCALE_KEYS
                    and not nv.ws_ok(x_q.shape[0], 3 * s.d_model, x_q.shape[1]))
            qkv, kvbuf, attn_ctx, ores, lse, out, xhat, rstd = MhaFn.compute(
                x_q, x_kv, s, q_rows, k_rows, causal, drop, kv_acc, any(ctx.needs_input_grad), scale, kpre)
        if AttnTap.active is not None:
            Qm, Km = (qkv[:, :d], qkv[:, d:2 * d]) if x_kv is None else (qkv, kvbuf[:, :d])
            AttnTap.record(mod, Qm, Km, q_rows, k_rows, causal, kpre)
        ctx.save_for_backward(x_q, x_kv, qkv, kvbuf, attn_ctx, lse, xhat, rstd, ores)
        ctx.mod, ctx.q_rows, ctx.k_rows, ctx.causal, ctx.scale = mod, q_rows, k_rows, causal, scale
        ctx.drop = drop          # attention-probability dropout (Attention.py:89): the backward regenerates the mask
        ctx.kv_acc = kv_acc      # decoder-encoder attention: the layers share one encoder-gradient buffer (or CrossKv's)
        ctx.kpre = kpre          # the saved keys are pre-scaled: the backward must be told
        ctx.up, ctx.down = up, down
        ctx.chain = (pre.bwd, pre.key) if pre is not None and pre.bwd is not None else None   # chains.ChainBackward
        if down is not None:
            down.offer(mod, xhat, rstd, s.g_b_o)
        if pre is not None:
            # autograd puts its grad_fn on the tensor object returned here.  Were that pre.out itself, the reference
            # chain out -> grad_fn -> ctx.chain -> ChainBackward.pres -> SubPre.out would close THROUGH C++, where
            # Python's gc cannot see it: every grad-enabled forward would leak the stack's activations.  Return an
            # alias (same storage: ChainBackward recognises gradients by address) and drop the back-reference.
            pre.bwd = None
            return out.view_as(out)
        return out

    PRESCALE_KEYS = True      # (tests switch it off together with chains.EncoderChains.PRESCALE_KEYS)

    @staticmethod
    def compute(x_q, x_kv, s, q_rows, k_rows, causal, drop, kv_acc, need_bwd, scale, kpre=False):
        """The forward launches of one attention sublayer -> (qkv, kvbuf, attn_ctx, ores, lse, out, xhat, rstd).
        kpre: the key block of the q | k | v projection leaves pre-scaled by scale * log2(e) (native.attn_fwd's k_prescaled)."""
        d, H = s.d_model, s.n_head
        Mq = x_q.shape[0]
        self_attn = x_kv is None
        if self_attn:
            qkv = _empty(Mq, 3 * d, x_q)
            if kpre:
                nv.gemm_kscale(x_q, s.w_qkv, qkv, s.b_qkv, d, 2 * d, scale * nv.K_LOG2_SCALE)
            else:
                linear_fwd(x_q, s.w_qkv, qkv, s.b_qkv)
            Q, K, V = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            kvbuf = None
        else:
            qkv = _empty(Mq, d, x_q)
            nv.gemm(x_q, s.w_q, qkv, bias=s.b_q)
            if isinstance(kv_acc, CrossKvSlot):       # x_kv is CrossKv's buffer: this layer's K | V are already in it
                kvbuf = x_kv[:, kv_acc.idx * 2 * d:(kv_acc.idx + 1) * 2 * d]
            else:
                kvbuf = _empty(x_kv.shape[0], 2 * d, x_q)
                nv.gemm(x_kv, s.w_kv, kvbuf, bias=s.b_kv)
            Q, K, V = qkv, kvbuf[:, :d], kvbuf[:, d:]
        # (rows past a length are never written by the attention kernel: on padded layouts they must still hold finite
        # values - they are operands of the weight-gradient GEMMs, multiplied by exact zeros)
        attn_ctx = rows_buffer(Mq, d, q_rows, x_q.device)
        # what rounding the context to bf16 drops (kept only when a backward follows): delta = rowsum(dO * O) is a
        # difference partner of dP in dS = P (dP - delta); with O to ~16 bits the two stay consistent
        # (every attention takes it: at config 3's depth the late ENCODER layers' keys are nearly identical across
        # positions too, and their q / k gradients come out 5x off without it - tests/test_fullsize_gpu.py)
        ores = rows_buffer(Mq, d, q_rows, x_q.device) if need_bwd else None
        lse = torch.empty(H * Mq, dtype=F32, device=x_q.device)
        nv.attn_fwd(Q, K, V, attn_ctx, lse, q_rows.off, q_rows.len, k_rows.off, k_rows.len, H, q_rows.max_len, causal,
                    scale, work=attn_work(q_rows, k_rows, causal, d // H, H)[0], drop=drop, max_k=k_rows.max_len, ores=ores,
                    k_prescaled=kpre)
        out, xhat = _empty(Mq, d, x_q), _empty(Mq, d, x_q)
        rstd = torch.empty(Mq, dtype=F32, device=x_q.device)
        nv.gemm_ln(attn_ctx, s.w_o, s.b_o, x_q, s.gamma, s.beta, out, xhat, rstd, eps=LN_EPS)
        return qkv, kvbuf, attn_ctx, ores, lse, out, xhat, rstd

    @staticmethod
    def backward(ctx, dout):
        x_q, x_kv, qkv, kvbuf, attn_ctx, lse, xhat, rstd, ores = ctx.saved_tensors
        mod, q_rows, k_rows = ctx.mod, ctx.q_rows, ctx.k_rows
        s, arena = mod._st, mod._st_arena
        d, H = s.d_model, s.n_head
        Mq = x_q.shape[0]
        dout = dout.contiguous()
        arena.attach_grads(s.params, s.lo, s.hi)
        # row-chain stacks: a backward chain launched by the sublayer behind this one already produced this sublayer's
        # LayerNorm backward, d(context) and delta (chains.ChainBackward)
        got = ctx.chain[0].stored(ctx.chain[1]) if ctx.chain is not None else None
        if got is not None:
            ds, dctx, delta = got["ds"], got["dctx"], got["delta"]
            if dout.data_ptr() != ds.data_ptr():
                raise RuntimeError("MhaFn.backward: the gradient reaching this sublayer is not the one its backward chain "
                                   "produced (its output has more than one consumer?)")
        else:
            ds = ctx.down.claim(dout) if ctx.down is not None else None
            if ds is None:
                ds = _empty(Mq, d, x_q)
                nv.ln_bwd(dout, xhat, rstd, s.gamma, ds, s.g_gamma, s.g_beta, s.g_b_o)
        wgrad(ds, attn_ctx, s.g_w_o)
        if got is None:
            dctx = _empty(Mq, d, x_q)
            delta = torch.empty(H * Mq, dtype=F32, device=x_q.device)
            # d(context) and, in the same launch, delta = rowsum(d(context) * context) per head: the two attention
            # backward kernels then depend on nothing but finished buffers and run as ONE launch
            nv.gemm(ds, s.w_o, dctx, epi=nv.EPI_BF16_DELTA, aux=attn_ctx, y_cmajor=True, delta=delta, head_dim=d // H, aux2=ores)
        if x_kv is None:
            Q, K, V = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            # key rows past k_len (padded layout only) get no gradient: they must read as zeros
            dqkv = rows_buffer(Mq, 3 * d, k_rows, x_q.device)
            dQ, dK, dV = dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:]
        else:
            Q, K, V = qkv, kvbuf[:, :d], kvbuf[:, d:]
            # (query rows past a length - padded layouts - get no gradient from the kernel: they must read as zeros)
            dqkv = rows_buffer(Mq, d, q_rows, x_q.device)
            slot = ctx.kv_acc if isinstance(ctx.kv_acc, CrossKvSlot) else None
            if slot is not None:
                st = slot.state
                if st.dkv is None:
                    st.dkv = rows_buffer(x_kv.shape[0], x_kv.shape[1], k_rows, x_q.device)
                dkv = st.dkv[:, slot.idx * 2 * d:(slot.idx + 1) * 2 * d]
            else:
                dkv = rows_buffer(x_kv.shape[0], 2 * d, k_rows, x_q.device)
            dQ, dK, dV = dqkv, dkv[:, :d], dkv[:, d:]
        _, work_q, work_k = attn_work(q_rows, k_rows, ctx.causal, d // H, H)
        nv.attn_bwd(Q, K, V, None, dctx, lse, delta, dQ, dK, dV, q_rows.off, q_rows.len, k_rows.off, k_rows.len, H,
                    q_rows.max_len, k_rows.max_len, ctx.causal, ctx.scale, work_q=work_q, work_k=work_k, drop=ctx.drop,
                    k_prescaled=ctx.kpre)
        dx_kv = None
        # row-chain stacks: everything down to the previous attention's backward kernel is one launch
        dx_q = ctx.chain[0].input_grad(ctx.chain[1], dqkv, ds) if ctx.chain is not None and ctx.needs_input_grad[0] else None
        if x_kv is None:
            wgrad(dqkv, x_q, s.g_w_qkv, gB=s.g_b_qkv)
            if dx_q is None:
                dx_q = _input_grad(ctx.up if ctx.needs_input_grad[0] else None, dqkv, s.w_qkv, ds)
        else:
            wgrad(dqkv, x_q, s.g_w_q, gB=s.g_b_q)
            if dx_q is None:
                dx_q = _input_grad(ctx.up if ctx.needs_input_grad[0] else None, dqkv, s.w_q, ds)
            acc = ctx.kv_acc
            if slot is not None:
                # weight and input gradients of all layers' K/V projections: CrossKvFn.backward, from the whole dkv
                st.seen += 1
                if st.seen == st.n:
                    dx_kv, st.dkv, st.seen = st.dkv, None, 0
            elif acc is None:
                wgrad(dkv, x_kv, s.g_w_kv, gB=s.g_b_kv)
                dx_kv = _empty(x_kv.shape[0], d, x_q)
                dgrad(dkv, s.w_kv, dx_kv)
            else:
                # every decoder layer attends the same encoder output: instead of handing autograd N gradients
                # to add, the GEMM epilogues accumulate into one buffer and the last layer to run (layer 0) returns it
                wgrad(dkv, x_kv, s.g_w_kv, gB=s.g_b_kv)
                if acc.buf is None:
                    acc.buf = _empty(x_kv.shape[0], d, x_q)
                    dgrad(dkv, s.w_kv, acc.buf)
                else:
                    dgrad(dkv, s.w_kv, acc.buf, epi=nv.EPI_BF16_ADD, aux=acc.buf)
                acc.seen += 1
                if acc.seen == acc.n:
                    dx_kv, acc.buf, acc.seen = acc.buf, None, 0
        arena.grads_ready(s.lo, s.hi)
        return dx_q, dx_kv, None, None, None, None, None, None, None, None, None, None, None


class DenseMhaFn(torch.autograd.Function):
    """MultiHeadAttention.forward in its GENERAL form (Attention.py:64-96): an arbitrary dense mask [B, Lq, Lk] and / or keys and
    values projected from different tensors.  No reference call site needs it (they pass k == v and one of the two mask families
    the fused kernels serve through lengths); it closes the module boundary with a slow path: three projection GEMMs,
    st_attn_dense_fwd / _bwd (one workgroup per query / key, no MFMA), then the usual output_linear + residual (q, repair R2) +
    LayerNorm.  Padded layouts only; rows are [B Lq] / [B Lk]."""

    @staticmethod
    def forward(ctx, xq, xk, xv, anchor, mod, mask, B, Lq, Lk, drop=None, want_attn=False):
        s = mod._st
        d, H = s.d_model, s.n_head
        scale = 1.0 / math.sqrt(d // H)
        Q, K, V = _empty(B * Lq, d, xq), _empty(B * Lk, d, xq), _empty(B * Lk, d, xq)
        nv.gemm(xq, s.w_q, Q, bias=s.b_q)
        nv.gemm(xk, s.w_kv[:d], K, bias=s.b_kv[:d])
        nv.gemm(xv, s.w_kv[d:], V, bias=s.b_kv[d:])
        attn_ctx = _empty(B * Lq, d, xq)
        lse = torch.empty(H * B * Lq, dtype=F32, device=xq.device)
        probs = nv.attn_dense_fwd(Q, K, V, mask, attn_ctx, lse, B, H, Lq, Lk, scale, drop=drop, want_probs=want_attn)
        out, xhat = _empty(B * Lq, d, xq), _empty(B * Lq, d, xq)
        rstd = torch.empty(B * Lq, dtype=F32, device=xq.device)
        nv.gemm_ln(attn_ctx, s.w_o, s.b_o, xq, s.gamma, s.beta, out, xhat, rstd, eps=LN_EPS)
        ctx.save_for_backward(xq, xk, xv, Q, K, V, attn_ctx, lse, xhat, rstd, mask)
        ctx.mod, ctx.dims, ctx.scale, ctx.drop = mod, (B, Lq, Lk), scale, drop
        ctx.mark_non_differentiable(*([probs] if probs is not None else []))
        return (out, probs) if want_attn else out

    @staticmethod
    def backward(ctx, dout, *_):
        xq, xk, xv, Q, K, V, attn_ctx, lse, xhat, rstd, mask = ctx.saved_tensors
        mod = ctx.mod
        s, arena = mod._st, mod._st_arena
        d, H = s.d_model, s.n_head
        B, Lq, Lk = ctx.dims
        arena.attach_grads(s.params, s.lo, s.hi)
        ds = _empty(B * Lq, d, xq)
        nv.ln_bwd(dout.contiguous(), xhat, rstd, s.gamma, ds, s.g_gamma, s.g_beta, s.g_b_o)
        wgrad(ds, attn_ctx, s.g_w_o)
        dctx = _empty(B * Lq, d, xq)
        dgrad(ds, s.w_o, dctx)
        dQ, dK, dV = _empty(B * Lq, d, xq), _empty(B * Lk, d, xq), _empty(B * Lk, d, xq)
        delta = torch.empty(H * B * Lq, dtype=F32, device=xq.device)
        nv.attn_dense_bwd(Q, K, V, mask, dctx, lse, delta, dQ, dK, dV, B, H, Lq, Lk, ctx.scale, drop=ctx.drop)
        wgrad(dQ, xq, s.g_w_q, gB=s.g_b_q)
        wgrad(dK, xk, s.g_w_kv[:d], gB=s.g_b_kv[:d])
        wgrad(dV, xv, s.g_w_kv[d:], gB=s.g_b_kv[d:])
        dxq = _empty(B * Lq, d, xq)
        dgrad(dQ, s.w_q, dxq, epi=nv.EPI_BF16_ADD, aux=ds)              # + the residual's gradient
        dxk, dxv = _empty(B * Lk, d, xq), _empty(B * Lk, d, xq)
        dgrad(dK, s.w_kv[:d], dxk)
        dgrad(dV, s.w_kv[d:], dxv)
        arena.grads_ready(s.lo, s.hi)
        return dxq, dxk, dxv, None, None, None, None, None, None, None, None


class FfnFn(torch.autograd.Function):
    """out = LN(x + fc2(relu(fc1(x))))   (SubLayers.py:24-28)."""

    @staticmethod
    def forward(ctx, x, anchor, mod, drop1=None, drop2=None, up=None, down=None, pre=None):
        """drop1: dropout after the ReLU (SubLayers.py:25); drop2: on the LayerNorm output (SubLayers.py:27);
        up / down: LnLink to the sublayer that produced x / that consumes the output;
        pre (chains.SubPre): the forward values were already computed by a fused launch - record only."""
        s = mod._st
        M, d = x.shape
        if pre is not None:
            h, out, xhat, rstd = pre.h, pre.out, pre.xhat, pre.rstd
        else:
            h = _empty(M, s.d_ff, x)
            linear_fwd(x, s.w1, h, s.b1, relu=True, drop=drop1)
            out, xhat = _empty(M, d, x), _empty(M, d, x)
            rstd = torch.empty(M, dtype=F32, device=x.device)
            nv.gemm_ln(h, s.w2, s.b2, x, s.gamma, s.beta, out, xhat, rstd, eps=LN_EPS, drop=drop2,
                       drop_where=2 if drop2 is not None else 0)
        ctx.save_for_backward(x, h, xhat, rstd)
        ctx.mod, ctx.drop1, ctx.drop2 = mod, drop1, drop2
        ctx.up, ctx.down = up, down
        ctx.chain = (pre.bwd, pre.key) if pre is not None and pre.bwd is not None else None   # chains.ChainBackward
        if down is not None:
            down.offer(mod, xhat, rstd, s.g_b2, drop2)
        if pre is not None:      # see MhaFn.forward: never hand autograd the tensor object the SubPre owns
            pre.bwd = None
            return out.view_as(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, h, xhat, rstd = ctx.saved_tensors
        mod = ctx.mod
        s, arena = mod._st, mod._st_arena
        M, d = x.shape
        dout = dout.contiguous()
        arena.attach_grads(s.params, s.lo, s.hi)
        got = ctx.chain[0].stored(ctx.chain[1]) if ctx.chain is not None else None
        if got is not None and dout.data_ptr() != got["ds"].data_ptr():
            raise RuntimeError("FfnFn.backward: the gradient reaching this sublayer is not the one its backward chain "
                               "produced (its output has more than one consumer?)")
        if got is None and ctx.chain is not None and ctx.needs_input_grad[0]:
            # the last feed-forward of a row-chain stack: its LayerNorm backward, hidden gradient and input gradient are one
            # chain launch (with the d(context) / delta of the attention in front)
            got = ctx.chain[0].ffn_tail(ctx.chain[1][1], dout)
        ds = got["ds"] if got is not None else (ctx.down.claim(dout) if ctx.down is not None else None)
        if ds is None:
            ds = _empty(M, d, x)
            nv.ln_bwd(dout, xhat, rstd, s.gamma, ds, s.g_gamma, s.g_beta, s.g_b2, drop=ctx.drop2)
        if got is not None:
            wgrad(ds, h, s.g_w2)
            wgrad(got["dh"], x, s.g_w1, gB=s.g_b1)
            arena.grads_ready(s.lo, s.hi)
            return got["dx"], None, None, None, None, None, None, None
        wgrad(ds, h, s.g_w2)
        dh = _empty(M, s.d_ff, x)
        dgrad(ds, s.w2, dh, epi=nv.EPI_BF16_MASK, aux=h, drop=ctx.drop1)   # h is the dropped activation: 0 where dropped
        wgrad(dh, x, s.g_w1, gB=s.g_b1)
        dx = _input_grad(ctx.up if ctx.needs_input_grad[0] else None, dh, s.w1, ds)
        arena.grads_ready(s.lo, s.hi)
        return dx, None, None, None, None, None, None, None


class FrontendFn(torch.autograd.Function):
    """e = LN(relu(x W_in^T + b_in)) + PE[pos]   (Models.py:28-33 eval-mode, 42-44)."""

    @staticmethod
    def forward(ctx, xp, anchor, mod, rows: Rows, drop=None):
        """drop: the Dropout between the ReLU and the LayerNorm (Models.py:31, p = 0.5 in training mode)."""
        s = mod._st
        M = xp.shape[0]
        d = s.d_model
        out, xhat, pre = _empty(M, d, xp), _empty(M, d, xp), _empty(M, d, xp)
        rstd = torch.empty(M, dtype=F32, device=xp.device)
        nv.gemm_ln(xp, s.w_in, s.b_in, None, s.gamma_in, s.beta_in, out, xhat, rstd, eps=LN_EPS, relu=True, pe=s.pe,
                   pos=rows.pos, pre=pre, drop=drop, drop_where=1 if drop is not None else 0)
        ctx.save_for_backward(xp, xhat, rstd, pre)
        ctx.mod = mod
        ctx.mask_scale = drop.scale if drop is not None else 1.0   # `pre` is 0 where the ReLU or the dropout zeroed
        ctx.need_dx = xp.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        xp, xhat, rstd, pre = ctx.saved_tensors
        mod = ctx.mod
        s, arena = mod._st, mod._st_arena
        M, d = xhat.shape
        dout = dout.contiguous()
        arena.attach_grads(s.front_params, s.front_lo, s.front_hi)
        dz = _empty(M, d, xp)
        nv.ln_bwd(dout, xhat, rstd, s.gamma_in, dz, s.g_gamma_in, s.g_beta_in, s.g_b_in, mask=pre,
                  mask_scale=ctx.mask_scale)
        wgrad(dz, xp, s.g_w_in)
        dxp = None
        if ctx.need_dx:
            dxp = _empty(M, xp.shape[1], xp)
            dgrad(dz, s.w_in, dxp)
        arena.grads_ready(s.front_lo, s.front_hi)
        return dxp, None, None, None, None


class EmbedFn(torch.autograd.Function):
    """y = Emb(tokens) + PE[pos]   (Models.py:84,87 with repair R3; padding_idx row gets no grad)."""

    @staticmethod
    def forward(ctx, anchor, mod, tokens, rows: Rows):
        s = mod._st
        out = rows_buffer(rows.total, s.d_model, rows, tokens.device)
        nv.embed_pe_fwd(tokens, s.emb, s.pe, rows.off, rows.len, out)
        ctx.mod, ctx.rows = mod, rows
        ctx.save_for_backward(tokens)
        return out

    @staticmethod
    def backward(ctx, dout):
        (tokens,) = ctx.saved_tensors
        mod = ctx.mod
        s, arena = mod._st, mod._st_arena
        arena.attach_grads(s.emb_params, s.emb_lo, s.emb_hi)
        nv.embed_bwd(tokens, dout.contiguous(), ctx.rows.off, ctx.rows.len, s.pad_idx, s.g_emb)
        flush_deferred_wgrads(final=False)      # the decoder's backward ends here: its deferred weight gradients go out as one launch
        arena.grads_ready(s.emb_lo, s.emb_hi)
        return None, None, None, None


class VocabFn(torch.autograd.Function):
    """logits = dec W_vocab^T, fp32, columns padded to a multiple of 8 (Models.py:145,151)."""

    @staticmethod
    def forward(ctx, x, anchor, mod, mask_pad=False):
        """mask_pad: the padding columns (zero weight rows -> logit 0) come out as -1e30 instead."""
        s = mod._st
        logits = torch.empty(x.shape[0], s.v_pad, dtype=F32, device=x.device)
        nv.gemm(x, s.w_vocab, logits, epi=nv.EPI_F32, bias=s.pad_bias if mask_pad else None)
        ctx.save_for_backward(x)
        ctx.mod = mod
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        (x,) = ctx.saved_tensors
        mod = ctx.mod
        s, arena = mod._st, mod._st_arena
        arena.attach_grads(s.vocab_params, s.vocab_lo, s.vocab_hi)
        dl = dlogits if dlogits.dtype == BF16 else dlogits.to(BF16)      # (CeFn hands over bf16 directly)
        wgrad(dl, x, s.g_w_vocab)
        dx = _empty(x.shape[0], x.shape[1], x)
        dgrad_long_k(dl, s.w_vocab, dx)
        arena.grads_ready(s.vocab_lo, s.vocab_hi)
        return dx, None, None, None


class VocabCeFn(torch.autograd.Function):
    """loss = CrossEntropy(dec W_vocab^T, target) (Models.py:151 + train.py:40,120) as ONE autograd node: the bf16 logits
    gradient st_ce_bwd writes goes straight into the vocabulary projection's two backward GEMMs (as separate nodes autograd
    would cast it to the logits' fp32 and VocabFn back to bf16: two more passes over [rows, V])."""

    @staticmethod
    def forward(ctx, x, anchor, mod, target, ignore_index, index=None):
        s = mod._st
        R = x.shape[0]
        logits = torch.empty(R, s.v_pad, dtype=F32, device=x.device)
        nv.gemm(x, s.w_vocab, logits, epi=nv.EPI_F32, bias=s.pad_bias)      # padding columns at -1e30: probability 0
        lse = torch.empty(R, dtype=F32, device=x.device)
        sums = torch.empty(3, dtype=F32, device=x.device)
        nv.ce_fwd(logits, target, ignore_index, lse, sums, index=index)
        ctx.save_for_backward(x, logits, target, lse, sums, index)
        ctx.mod, ctx.ignore_index = mod, ignore_index
        return sums[2]

    @staticmethod
    def backward(ctx, go):
        x, logits, target, lse, sums, index = ctx.saved_tensors
        mod = ctx.mod
        s, arena = mod._st, mod._st_arena
        arena.attach_grads(s.vocab_params, s.vocab_lo, s.vocab_hi)
        dl = torch.empty(logits.shape, dtype=BF16, device=logits.device)
        nv.ce_bwd(logits, target, ctx.ignore_index, lse, sums, go.reshape(1).float(), dl, index=index)
        wgrad(dl, x, s.g_w_vocab)
        dx = _empty(x.shape[0], x.shape[1], x)
        dgrad_long_k(dl, s.w_vocab, dx)
        arena.grads_ready(s.vocab_lo, s.vocab_hi)
        return dx, None, None, None, None, None


class CeFn(torch.autograd.Function):
    """nn.CrossEntropyLoss(ignore_index) (train.py:40,120: mean over the non-ignored tokens) over ragged fp32 logits rows as
    two launches: forward = per-row log-sum-exp + the loss sum and token count, backward = (softmax - onehot) * grad / count
    written straight in bf16 - the operand VocabFn.backward feeds to its two GEMMs (PyTorch runs log-softmax, nll-loss,
    their two backwards and a cast: five passes over the [rows, V] logits)."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        R = logits.shape[0]
        lse = torch.empty(R, dtype=F32, device=logits.device)
        sums = torch.empty(3, dtype=F32, device=logits.device)
        nv.ce_fwd(logits, target, ignore_index, lse, sums)
        ctx.save_for_backward(logits, target, lse, sums)
        ctx.ignore_index = ignore_index
        return sums[2]

    @staticmethod
    def backward(ctx, go):
        logits, target, lse, sums = ctx.saved_tensors
        dl = torch.empty(logits.shape, dtype=BF16, device=logits.device)
        nv.ce_bwd(logits, target, ctx.ignore_index, lse, sums, go.reshape(1).float(), dl)
        return dl, None, None


def cross_entropy_rows(logits, target, ignore_index=0):
    """Mean cross-entropy of fp32 logits rows [R, V] (contiguous; padding columns at -1e30 are fine) against int64 targets,
    rows with ``ignore_index`` excluded (nn.CrossEntropyLoss(ignore_index=0), train.py:120)."""
    return CeFn.apply(logits, target.contiguous(), ignore_index)


class PackFn(torch.autograd.Function):
    """Padded fp32 [B, T, D] -> bf16 row matrix [rows.total, D]."""

    @staticmethod
    def forward(ctx, x, rows: Rows):
        # (a packed layout has no row outside an utterance: nothing to zero)
        out = rows_buffer(rows.total, x.shape[2], rows, x.device)
        nv.pack_rows(x.contiguous(), rows.off, rows.len, out)
        ctx.rows, ctx.shape = rows, x.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        g = torch.empty(ctx.shape, dtype=F32, device=dout.device)
        nv.unpack_rows(dout.contiguous(), ctx.rows.off, ctx.rows.len, g)
        return g, None


class UnpackFn(torch.autograd.Function):
    """bf16 row matrix -> padded fp32 [B, T, D] (zeros past each length)."""

    @staticmethod
    def forward(ctx, x, rows: Rows, T: int):
        out = torch.empty(rows.B, T, x.shape[1], dtype=F32, device=x.device)
        nv.unpack_rows(x, rows.off, rows.len, out)
        ctx.rows, ctx.n = rows, x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        out = torch.zeros(ctx.n, dtype=BF16, device=g.device)
        nv.pack_grad(g.contiguous(), ctx.rows.off, ctx.rows.len, out)
        return out, None, None

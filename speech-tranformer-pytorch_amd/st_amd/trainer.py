"""The training step of the reference's ``train()`` loops, as one callable.

Sequence (train.py:25-46 / train_multi.py:47-68): trim the batch to its longest
utterance, zero_grad, forward, ``CrossEntropyLoss(ignore_index=0)`` over
``logits.view(-1, V)`` vs ``ground_truth.view(-1)``, backward, [gradient average
over ranks], global-norm clip, Noam-Adam update.  The per-step ``.item()`` syncs of
the reference (train.py:31-32,42) are not reproduced: lengths are taken on the host
(where the loader produced them) and the loss / grad-norm come back as device tensors.

HIP-graph mode (``use_graph=True``): the step launches ~190 kernels; the decoder half
(M ~ 1.2k rows) is launch-bound from Python (~10 us of host time per launch vs 2-4 us of
GPU time).  The step is therefore captured per batch SIGNATURE (same tensors, same length
vectors) into a HIP graph and replayed; the Noam rate is a device scalar updated before each
replay.  Up to ``max_graphs`` signatures are kept (default 4; least recently used evicted; every
signature owns a private memory pool with a full step of activations - ~5 GB at config 2 -
so the cap is a memory bound), each with its own memory pool and with references to the ragged layouts its kernels read by address,
so a loader that cycles through a set of pre-collated batches (bench.py; an epoch over a
bucketed, cached dataset) replays, and one whose length vectors never repeat runs the eager
path - whose ms/step bench.py reports next to the replay figure.  With data parallelism the
backward is cut at the encoder output: forward + loss + decoder backward | encoder backward |
clip+Adam.  The decoder's and the vocabulary projection's gradients (the tail ~2/3 of the flat
buffer) are final after the first part, so their RCCL all-reduce runs while the encoder's
backward - ~40 % of the step - is executing.  The bucket all-reduces are captured INSIDE the
step graph (``dp_in_graph``; RCCL collectives are stream-capturable), so a DP step is one graph
replay; where that capture is refused the three parts become three graphs with eager
collectives between them.
"""
from __future__ import annotations

import collections
import os
import time
from typing import Optional

import torch
import torch.distributed
import torch.nn as nn

from .arena import arena_of
from .dp import GradReducer
from . import rng
from . import native as nv
from .functional import TailBuffers, deferred_wgrads


def clip_grad_norm_flat(arena, max_norm: float) -> torch.Tensor:
    """``nn.utils.clip_grad_norm_`` (train.py:45) over the flat gradient buffer: one
    reduction + one scale instead of a pass per tensor; padding elements are zero."""
    total = torch.linalg.vector_norm(arena.grad)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    arena.grad.mul_(coef)
    return total


class TrainStep:
    def __init__(self, model: nn.Module, optimizer, vocab_size: int, max_grad_norm: float,
                 reducer: Optional[GradReducer] = None, use_graph: bool = False, graph_warmup: int = 2,
                 max_graphs: int = 4, bucket=None, dp_in_graph: bool = True, bucket_rows=None):
        self.model, self.optimizer = model, optimizer
        self.vocab_size, self.max_grad_norm = vocab_size, max_grad_norm
        self.crit = nn.CrossEntropyLoss(ignore_index=0)          # train.py:120
        self.reducer = reducer
        self.global_step = 0
        self.use_graph, self.graph_warmup, self.max_graphs = use_graph, graph_warmup, max_graphs
        # data parallelism in graph mode: the bucket all-reduces are CAPTURED with the step (RCCL collectives are
        # stream-capturable: ProcessGroupNCCL forks its stream off the capturing one and joins it again in wait()), so a DP
        # step is ONE graph replay with no host in the loop; False (or a capture that fails) = three graphs with eager
        # collectives in between
        self.dp_in_graph = dp_in_graph
        self.dp_mode = None                            # "in-graph" / "split" after the first DP capture (introspection)
        # bucket = (T_cap, L_cap): ONE captured step serves every batch of B utterances with at most T_cap frames and L_cap
        # target tokens - the batch is staged into static padded buffers, the lengths live on the device (Rows.bucket), so
        # a loader whose batches never repeat a length signature still replays a graph.  Costs the padding rows
        # (B * T_cap instead of sum(len) rows through the row-wise kernels): for length-bucketed loaders.
        self.bucket = None if bucket is None else (int(bucket[0]), int(bucket[1]))
        # bucket_rows = (input rows, target rows): the bucket's layouts are PACKED into that many rows (Rows.bucket(rows=...)) -
        # offsets on the device as well - so the captured step runs over about the batch's real row count instead of
        # B * T_cap.  Several capacities may be given, [(rows, rows), ...] in ascending order: a batch takes the first one it
        # fits (one capture each); a batch with more rows than the largest takes the padded bucket.  Pick the capacities
        # from the loader's length statistics AND the kernels' tile rounds: the encoder-sized row chains run one 96-row
        # workgroup per CU, so 256 x 96 = 24,576 rows are one round and 24,577 are two (of 64-row workgroups: up to 32,768).
        if bucket_rows is not None and not isinstance(bucket_rows[0], (tuple, list)):
            bucket_rows = [bucket_rows]
        self.bucket_rows = None if bucket_rows is None else [(int(a), int(b)) for a, b in bucket_rows]
        if self.bucket_rows is not None and self.bucket is None:
            raise ValueError("TrainStep: bucket_rows needs bucket=(T_cap, L_cap)")
        self._buckets = {}
        self._graphs = collections.OrderedDict()      # signature -> captured step (LRU)
        self._seen = collections.OrderedDict()        # signature -> eager sightings before the capture
        self._g_fb = self._g_enc = self._g_opt = None
        self._loss = self._gnorm = None
        self._cut = None
        self._seed = None

    def _backward(self, loss):
        """loss.backward() with a resident gradient seed (autograd's own ones_like(loss) is a fill launch per step)."""
        if self._seed is None or self._seed.device != loss.device or self._seed.dtype != loss.dtype:
            self._seed = torch.ones((), dtype=loss.dtype, device=loss.device)
        torch.autograd.backward(loss, grad_tensors=self._seed)

    # ---- the two halves of a step -------------------------------------------------------------
    @staticmethod
    def _zero_tails():
        """First launch of a captured packed-bucket step: the unassigned tail rows of every registered buffer (one launch)."""
        tb = TailBuffers.active
        if tb is not None and tb.mode == "serve" and tb.bufs:
            nv.zero_tails(tb.table, TailBuffers.N_MAX)

    def _single_graph(self) -> bool:
        """The whole step is captured as ONE graph (no data-parallel split into several captures that share a memory pool)."""
        split = self.reducer is not None and self.reducer.active and hasattr(self.model, "forward_packed") \
            and hasattr(self.model, "encoder")
        return not split or self.dp_in_graph

    def _forward_backward(self, inputs, input_lengths, targets, target_lengths, ground_truth, captured=False, layouts=None):
        self._zero_tails()
        self.optimizer.zero_grad()
        rng.advance()                          # next step's dropout masks (an in-place device add: capturable)
        if layouts is not None:
            loss, _ = self.model.forward_packed(inputs, input_lengths, targets, target_lengths, ce_truth=ground_truth,
                                                ignore_index=self.crit.ignore_index, layouts=layouts)
        elif hasattr(self.model, "forward_packed"):
            # loss over the valid tokens only: the kernels' ragged logits rows against the matching ground-truth
            # entries.  Identical to train.py:40 on the padded [B, L, V] tensor: its padded positions carry
            # ground truth 0 = ignore_index, and the mean is over non-ignored tokens either way.
            # (projection + cross-entropy as one autograd node: functional.VocabCeFn)
            loss, t_rows = self.model.forward_packed(inputs, input_lengths, targets, target_lengths, ce_truth=ground_truth,
                                                     ignore_index=self.crit.ignore_index)
        else:
            logits, _ = self.model(inputs, input_lengths, targets, target_lengths)
            loss = self.crit(logits.contiguous().view(-1, self.vocab_size), ground_truth.contiguous().view(-1))
        # the decoder's weight gradients are deferred into one grouped launch - unless eager gradient-ready hooks
        # are live (they assume a weight gradient is enqueued when its layer's backward returns)
        hooks_live = self.reducer is not None and not captured
        with deferred_wgrads(not hooks_live):
            self._backward(loss)
        return loss.detach()

    # ---- data-parallel graph mode: the backward in two captures -----------------------------------
    def _forward_decoder_backward(self, inputs, input_lengths, targets, target_lengths, ground_truth, layouts=None):
        """zero_grad, forward, loss, and the backward down to the encoder output (weight gradients of the decoder
        flushed).  Leaves (encoder output, its gradient) in ``self._cut`` for :meth:`_encoder_backward`."""
        self._zero_tails()
        self.optimizer.zero_grad()
        rng.advance()
        loss, t_rows, enc, enc_leaf = self.model.forward_packed(inputs, input_lengths, targets, target_lengths,
                                                                cut_encoder=True, ce_truth=ground_truth,
                                                                ignore_index=self.crit.ignore_index, layouts=layouts)
        with deferred_wgrads(True):
            self._backward(loss)
        self._cut = (enc, enc_leaf.grad)
        return loss.detach()

    def _encoder_backward(self, fire_layers: bool = False):
        """fire_layers (the collectives are being captured with the step): every time the backward has left an encoder layer,
        the weight gradients deferred so far are flushed and the buckets that lie wholly in the finished layers start their
        all-reduce - the exchange of the upper encoder layers runs under the backward of the lower ones instead of behind the
        whole backward (the deferred weight gradients otherwise become final in ONE launch at its end)."""
        enc, d_enc = self._cut
        self._cut = None
        chains = self._encoder_chains() if fire_layers else None
        if chains is not None:
            chains.layer_hook = self._encoder_layer_done
        try:
            with deferred_wgrads(True):
                enc.backward(d_enc)
        finally:
            if chains is not None:
                chains.layer_hook = None

    def _encoder_chains(self):
        enc = getattr(self.model, "encoder", None)
        if enc is None or not hasattr(enc, "row_chains") or self._encoder_layer_offsets() is None:
            return None
        return enc.row_chains(arena_of(self.model))

    def _encoder_layer_offsets(self):
        """First element of every encoder layer in the flat gradient buffer - None unless the layers lie in stack order with
        everything that is not the encoder behind them (the layout ParamArena builds from module order)."""
        hit = getattr(self, "_enc_layer_lo", False)
        if hit is not False:
            return hit
        arena = arena_of(self.model)
        los = [min(arena.offset[id(p)] for p in layer.parameters()) for layer in self.model.encoder.layer_stack]
        his = [max(arena.offset[id(p)] + arena.size[id(p)] for p in layer.parameters()) for layer in self.model.encoder.layer_stack]
        cut = self._decoder_grad_start()
        ok = all(his[i] <= los[i + 1] for i in range(len(los) - 1)) and cut < arena.total and his[-1] <= cut
        self._enc_layer_lo = los if ok else None
        return self._enc_layer_lo

    def _encoder_layer_done(self, first_finished: int) -> None:
        los = self._encoder_layer_offsets()
        if los is None or first_finished >= len(los):
            return
        lo = los[first_finished]
        red = self.reducer
        if any(blo >= lo and not red._fired[i] for i, (blo, _) in enumerate(red.buckets)):
            from .functional import flush_deferred_wgrads
            flush_deferred_wgrads()          # everything registered so far: the finished layers' weight (+ bias) gradients
            red.fire_from(lo)

    def _decoder_grad_start(self) -> int:
        """First element of the flat gradient buffer that the encoder's backward no longer touches."""
        arena = arena_of(self.model)
        enc = {id(p) for p in self.model.encoder.parameters()}
        cut = max(arena.offset[i] + arena.size[i] for i in enc)
        others = [arena.offset[id(p)] for p in arena.params if id(p) not in enc]
        return cut if not others or min(others) >= cut else arena.total      # interleaved layout: nothing fires early

    def _fold_scale(self) -> bool:
        """The rank average's 1 / world is folded into the norm + clip + Adam kernels (no pass of its own over the flat buffer)."""
        return getattr(self.optimizer, "arena", None) is not None

    def _clip_and_update(self, grad_scale: float = 1.0):
        if getattr(self.optimizer, "arena", None) is not None:
            # global norm (one launch over the flat buffer: st_grad_norm, which also advances the step count), then clip +
            # Adam as one pass (st_adam_clip); grad_scale: the reducer left the rank sum (GradReducer.synchronize(divide=False))
            return self.optimizer.step_captured(grad_norm=True, max_norm=self.max_grad_norm, grad_scale=grad_scale)
        if grad_scale != 1.0:
            arena_of(self.model).grad.mul_(grad_scale)
        grad_norm = clip_grad_norm_flat(arena_of(self.model), self.max_grad_norm)
        self.optimizer.step_captured()
        return grad_norm

    def _eager(self, batch, layouts=None):
        loss = self._forward_backward(*batch, layouts=layouts)
        scale = self.reducer.synchronize(divide=not self._fold_scale()) if self.reducer is not None else 1.0
        self.optimizer.update_learning_rate(self.global_step)
        return loss, self._clip_and_update(scale)

    def __call__(self, inputs, input_lengths, targets, target_lengths, ground_truth):
        """inputs [B, T, F] / targets, ground_truth [B, L] on the GPU; lengths on host or GPU."""
        t_max, l_max = int(input_lengths.max()), int(target_lengths.max())     # host ints when lengths are CPU tensors
        self.global_step += 1
        batch = (inputs[:, :t_max], input_lengths, targets[:, :l_max], target_lengths, ground_truth[:, :l_max])
        if self.bucket is not None:
            return self._bucket_call(*batch)
        if not self.use_graph:
            return self._eager(batch)

        sig = (inputs.data_ptr(), targets.data_ptr(), ground_truth.data_ptr(), tuple(inputs.shape),
               tuple(targets.shape), input_lengths.cpu().numpy().tobytes(), target_lengths.cpu().numpy().tobytes())
        cap = self._graphs.get(sig)
        if cap is None:
            n = self._seen.get(sig, 0) + 1
            self._seen[sig] = n
            while len(self._seen) > 64:
                self._seen.popitem(last=False)
            if n <= self.graph_warmup:                   # lazy init (arena, layouts, allocator) happens eagerly
                return self._eager(batch)
            cap = self._capture(batch)
            self._graphs[sig] = cap
            self._seen.pop(sig, None)
            while len(self._graphs) > self.max_graphs:
                self._graphs.popitem(last=False)         # least recently used: its graphs, pool and pinned layouts go
        self._graphs.move_to_end(sig)
        return self._replay(cap)

    def _bucket_call(self, inputs, input_lengths, targets, target_lengths, ground_truth):
        """Bucket mode: stage the batch into the bucket's static buffers, refresh the device-resident lengths, replay."""
        T_cap, L_cap = self.bucket
        B, T, Fd = inputs.shape
        L = targets.shape[1]
        if T > T_cap or L > L_cap or ground_truth.shape[1] > L_cap:
            raise ValueError("TrainStep(bucket=%r): batch of %d frames / %d tokens does not fit" % (self.bucket, T, L))
        caps = None                   # the first packed capacity the batch fits (None: the padded bucket)
        if self.bucket_rows is not None:
            n_in, n_tgt = int(input_lengths.sum()), int(target_lengths.sum())
            caps = next((c for c in self.bucket_rows if n_in <= c[0] and n_tgt <= c[1]), None)
        packed = caps is not None
        key = (B, Fd, inputs.dtype, str(inputs.device), caps)
        st = self._buckets.get(key)
        if st is None:
            from .functional import Rows
            dev = inputs.device
            st = _Bucket()
            st.x = torch.zeros(B, T_cap, Fd, dtype=inputs.dtype, device=dev)
            st.tok = torch.zeros(B, L_cap, dtype=targets.dtype, device=dev)
            # (one spare element behind the padded ground truth: the target of the rows a packed bucket leaves unassigned;
            # it stays 0 = ignore_index)
            st.gt_flat = torch.zeros(B * L_cap + 1, dtype=ground_truth.dtype, device=dev)
            st.gt = st.gt_flat[:B * L_cap].view(B, L_cap)
            in_cap, tgt_cap = caps if packed else (None, None)
            st.layouts = (Rows.bucket(B, T_cap, dev, rows=in_cap), Rows.bucket(B, L_cap, dev, rows=tgt_cap))
            if hasattr(self.model, "prepare_layouts"):          # chain plans, work lists: before any capture
                from .functional import attn_work
                self.model.prepare_layouts(torch.full((B,), T_cap), torch.full((B,), L_cap), L_cap, dev)
                ir, tr = st.layouts
                dk, nh = getattr(self.model, "_d_k", 64), getattr(self.model, "_n_head", 0)
                attn_work(ir, ir, False, dk, nh), attn_work(tr, tr, True, dk, nh), attn_work(tr, ir, False, dk, nh)
                tr.scatter_index(L_cap)
            self._buckets[key] = st
        st.x[:, :T].copy_(inputs, non_blocking=True)
        if T < T_cap:
            st.x[:, T:].zero_()                                  # no frame of an earlier batch survives behind a length
        st.tok.zero_()
        st.tok[:, :L].copy_(targets, non_blocking=True)
        st.gt.zero_()                                            # padding positions: ground truth 0 = ignore_index
        st.gt[:, :ground_truth.shape[1]].copy_(ground_truth, non_blocking=True)
        st.layouts[0].set_lengths(input_lengths)
        st.layouts[1].set_lengths(target_lengths)
        if hasattr(self.model, "prepare_layouts"):
            # the encoder self-attention's work lists follow the batch (longest utterance first); the decoder's attentions
            # are a handful of short workgroups either way
            from .functional import refresh_attn_work
            ir = st.layouts[0]
            refresh_attn_work(ir, ir, False, getattr(self.model, "_d_k", 64), getattr(self.model, "_n_head", 0),
                              input_lengths.tolist(), input_lengths.tolist())
        batch = (st.x, input_lengths, st.tok, target_lengths, st.gt_flat if packed else st.gt)
        if not self.use_graph:
            return self._eager(batch, layouts=st.layouts)
        if st.cap is None:
            st.seen += 1
            if st.seen <= self.graph_warmup:
                if packed and st.seen == self.graph_warmup:
                    # the last eager step notes the buffers whose unassigned tail rows must read as zeros: the capture
                    # serves them from persistent memory and zeroes all tails with one launch (functional.TailBuffers)
                    st.tails = TailBuffers()
                    TailBuffers.active = st.tails
                    try:
                        return self._eager(batch, layouts=st.layouts)
                    finally:
                        TailBuffers.active = None
                return self._eager(batch, layouts=st.layouts)
            if packed and st.tails is not None and self._single_graph():
                TailBuffers.active = st.tails.materialize(inputs.device)
            try:
                st.cap = self._capture(batch, layouts=st.layouts)
            finally:
                TailBuffers.active = None
        return self._replay(st.cap)

    def _replay(self, cap):
        self._g_fb, self._g_enc, self._g_opt, self._dec_lo = cap.g_fb, cap.g_enc, cap.g_opt, cap.dec_lo     # (introspection / tests)
        self.optimizer.update_learning_rate(self.global_step)
        cap.g_fb.replay()
        if cap.g_enc is not None:
            self.reducer.fire_from(cap.dec_lo)           # decoder-side buckets: exchanged while the encoder's backward runs
            cap.g_enc.replay()
            self.reducer.synchronize(divide=not self._fold_scale())      # (the captured update applies 1 / world: _capture)
        elif self.reducer is not None and cap.g_opt is not None:
            self.reducer.reduce_all(divide=not self._fold_scale())
        # (cap.g_opt None with a reducer: the collectives were captured inside g_fb - nothing happens on the host)
        if cap.g_opt is not None:
            cap.g_opt.replay()
        return cap.loss, cap.gnorm

    def _capture(self, batch, layouts=None):
        if hasattr(self.optimizer, "_flat_state") and getattr(self.optimizer, "arena", None) is not None:
            self.optimizer._flat_state()      # Adam's lazily created state must exist BEFORE the capture (a captured
                                              # zero-fill would reset it on every replay)
        # ... and so must the process-wide scratch the step's kernels share (split-K partials and tickets, the norm's block
        # partials, the backward seed): created inside a capture they would live in that graph's private pool while later
        # captures and eager calls keep using them
        dev = batch[0].device
        nv.splitk_scratch(dev)
        if hasattr(self.optimizer, "norm_scratch"):
            self.optimizer.norm_scratch(dev)
        if self._seed is None or self._seed.device != dev:
            self._seed = torch.ones((), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        if self.reducer is not None and self.reducer.active:
            drain_collective_watchdog()
        cap = _Captured()
        # the captured kernels read the ragged layouts (offsets, lengths, positions, attention work lists, scatter
        # index) by ADDRESS: pin the layout objects of this batch for as long as its graphs live (the layout cache may
        # be flushed by other shapes meanwhile)
        if layouts is not None:
            cap.keep = layouts
        elif hasattr(self.model, "prepare_layouts"):
            cap.keep = self.model.prepare_layouts(batch[1], batch[3], batch[2].shape[1], batch[0].device)
        # ... and the batch tensors themselves: the signature matches on their addresses, so the memory must not be
        # recycled for something else while a capture that reads it is alive
        cap.keep = (cap.keep, batch[0], batch[2], batch[4])
        pool = torch.cuda.graph_pool_handle()           # one pool per signature: replays of different signatures interleave freely
        cap.g_fb, cap.g_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        split = self.reducer is not None and self.reducer.active and hasattr(self.model, "forward_packed") \
            and hasattr(self.model, "encoder")
        if self.reducer is not None:
            self.reducer.detach()                       # bucket all-reduces are issued explicitly between the graphs
        # ProcessGroupNCCL's watchdog thread polls the events of outstanding collectives (hipEventQuery); under the
        # default "global" capture mode that call from ANOTHER thread aborts the process ("operation not permitted
        # when stream is capturing").  With a process group alive, only this thread's unsafe calls are policed.
        mode = dict(capture_error_mode="thread_local") if torch.distributed.is_initialized() else {}
        if split and self.dp_in_graph:
            # ONE graph: forward + loss + decoder backward | decoder-side buckets start on RCCL's stream (a forked branch
            # of the graph) | encoder backward runs beside them | remaining buckets | join | clip + Adam
            failure = None
            try:
                g_all = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_all, pool=pool, **mode):
                    cap.loss = self._forward_decoder_backward(*batch, layouts=layouts)
                    self.reducer.fire_from(self._decoder_grad_start())
                    self._encoder_backward(fire_layers=True)
                    cap.gnorm = self._clip_and_update(self.reducer.synchronize(divide=not self._fold_scale()))
            except Exception as e:  # noqa: BLE001 - a process group / RCCL build that cannot be captured: eager collectives
                failure = e
            # every rank must take the same mode (a rank replaying captured collectives beside one that issues them from the
            # host would deadlock): the ranks agree before anything else is exchanged - capture only records, so this is
            # the next collective on every rank either way
            if self.reducer.world > 1:
                ok = torch.tensor([0.0 if failure is not None else 1.0], device=batch[0].device)
                torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=self.reducer.group)
                if float(ok) == 0.0 and failure is None:
                    failure = RuntimeError("another rank could not capture its collectives")
            if failure is None:
                cap.g_fb, cap.g_enc, cap.g_opt = g_all, None, None
                self.dp_mode = "in-graph"
                return cap
            import warnings
            warnings.warn("TrainStep: capturing the gradient all-reduces failed (%s: %s); using eager collectives "
                          "between three graphs" % (type(failure).__name__, failure))
            # the failed capture may have stopped anywhere: collectives noted as fired, tail buffers handed out, the
            # encoder cut stored, weight gradients still deferred - the recapture starts from a clean slate
            from .functional import _Deferred
            self.reducer._work, self.reducer._fired = [], [False] * len(self.reducer.buckets)
            self._cut = None
            _Deferred.pending.clear()
            _Deferred.active = False
            nv._fold_pending.clear()
            nv.fold_deferred = False
            if TailBuffers.active is not None:
                TailBuffers.active.i = 0
            del g_all
            torch.cuda.synchronize()
            drain_collective_watchdog()      # (the agreement all-reduce above: see the wait before the capture)
            pool = torch.cuda.graph_pool_handle()
            cap.g_fb, cap.g_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        if split:
            self.dp_mode = "split"
            cap.g_enc, cap.dec_lo = torch.cuda.CUDAGraph(), self._decoder_grad_start()
            with torch.cuda.graph(cap.g_fb, pool=pool, **mode):
                cap.loss = self._forward_decoder_backward(*batch, layouts=layouts)
            with torch.cuda.graph(cap.g_enc, pool=pool, **mode):
                self._encoder_backward()
        elif self.reducer is not None and self.reducer.active:
            with torch.cuda.graph(cap.g_fb, pool=pool, **mode):
                cap.loss = self._forward_backward(*batch, captured=True, layouts=layouts)
        else:                                           # nothing happens between backward and the update: ONE graph
            cap.g_opt = None
            with torch.cuda.graph(cap.g_fb, pool=pool, **mode):
                cap.loss = self._forward_backward(*batch, captured=True, layouts=layouts)
                cap.gnorm = self._clip_and_update()
            return cap
        # (the collectives run eagerly between the graphs and leave the rank SUM: the captured update carries the 1 / world)
        left = 1.0 / self.reducer.world if (self.reducer is not None and self.reducer.active and self.reducer.world > 1
                                            and self._fold_scale()) else 1.0
        with torch.cuda.graph(cap.g_opt, pool=pool, **mode):
            cap.gnorm = self._clip_and_update(left)
        # capture only records; the step that triggered it is executed by the replay that follows
        return cap


class _Bucket:
    """Bucket mode: static staging buffers, device-length layouts and the captured step of one (B, feature) shape."""
    __slots__ = ("x", "tok", "gt", "gt_flat", "layouts", "cap", "seen", "tails")

    def __init__(self):
        self.x = self.tok = self.gt = self.gt_flat = self.layouts = self.cap = self.tails = None
        self.seen = 0


def drain_collective_watchdog(limit_s: float = 5.0) -> str:
    """Before a capture that contains collectives: wait until ProcessGroupNCCL's watchdog thread has RETIRED every eager collective.

    The eager steps' all-reduces are finished on the GPU after torch.cuda.synchronize(), but the watchdog retires them on its own
    clock - and its hipEventQuery of an eager collective's end event fails with "operation not permitted on an event last recorded
    in a capturing stream" once RCCL's stream has joined a capture, which terminates the process from the watchdog thread (round
    5: tools/dev/dp_capture_stress.py - 6 of 16 processes of 12 captures each died).  Round 5 waited a fixed 300 ms; that is not a
    synchronisation.  Now: the process group's flight recorder says for every collective whether the watchdog thread
    has RETIRED it (the ``retired`` flag of ``_dump_nccl_trace`` entries is set by that thread, when it drops the work from its
    list - unlike ``onlyActive`` and the ``state`` strings, which the dump call fills in itself by querying the events) - poll until
    every recorded collective is retired.  The recorder must be on (``TORCH_NCCL_TRACE_BUFFER_SIZE`` > 0 before the process group is
    created: ``enable_collective_recorder()``, called by bench.py and dp.GradReducer's callers); where it is off or absent the fixed
    wait remains as the fallback (``ST_DP_DRAIN_MS``, default 300).  -> which of the two happened ("recorder" / "sleep")."""
    torch.cuda.synchronize()
    fallback_s = float(os.environ.get("ST_DP_DRAIN_MS", "300")) / 1e3
    dump = getattr(torch._C._distributed_c10d, "_dump_nccl_trace", None) if torch.distributed.is_available() else None
    if dump is not None and os.environ.get("ST_DP_DRAIN", "recorder") == "recorder":
        import pickle
        try:
            t0 = time.time()
            while time.time() - t0 < limit_s:
                entries = pickle.loads(dump(includeCollectives=True, includeStackTraces=False, onlyActive=False)).get("entries")
                if not entries or "retired" not in entries[0]:
                    break                      # the recorder is off (or speaks another format): the fixed wait below
                if all(e["retired"] for e in entries):
                    return "recorder"
                time.sleep(0.005)
        except Exception:      # (an unexpected dump format: fall through to the fixed wait)
            pass
    time.sleep(fallback_s)
    return "sleep"


def enable_collective_recorder(entries: int = 512) -> None:
    """Call BEFORE torch.distributed.init_process_group: turns ProcessGroupNCCL's flight recorder on (a ring of the last `entries`
    collectives; off by default in this build), which is what lets drain_collective_watchdog() wait for the watchdog instead of
    sleeping."""
    os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", str(entries))


class _Captured:
    """One batch signature's captured step: its graph(s), static result tensors and pinned ragged layouts."""
    __slots__ = ("g_fb", "g_enc", "g_opt", "loss", "gnorm", "dec_lo", "keep")

    def __init__(self):
        self.g_fb = self.g_enc = self.g_opt = self.loss = self.gnorm = self.keep = None
        self.dec_lo = 0


class JointTrainStep:
    """One optimisation step of the joint CTC + attention objective (BASELINE config 4; the reference's
    train_attn_and_ctc.py is an empty file - the step is train.py:25-46 with transformer/Loss.py:CTCAttentionLoss as the
    criterion):  loss = w * CTC(encoder output) + (1 - w) * CE(decoder logits).

    Everything except ``ctc_loss`` itself runs as HIP kernels and, in graph mode, inside three captured graphs that share a
    memory pool:
      graph A1  zero_grad, encoder forward, the CTC head's projection over the ragged encoder rows and st_ctc_gather (the
                <= L + 1 log-probabilities per frame ctc_loss reads - the [T, B, V] log-softmax tensor of the module-level
                path is never built);
      eager     ``torch.nn.functional.ctc_loss`` on that small tensor + its gradient, ON A SIDE STREAM (PyTorch-ROCm, as the
                task prescribes for the loss heads; torch's CTC kernels build their length tables from pageable host memory,
                which a stream capture refuses).  Its three kernels walk the 1,000 frames sequentially on 32 workgroups -
                3.1 ms at config 2 with the chip idle beside them - so meanwhile the main stream replays
      graph A2  decoder forward, vocabulary projection + cross-entropy (one node), and the backward of that branch down to the
                encoder output (the decoder never sees the CTC branch);
      graph B   (after the side stream's event) backward from BOTH roots into the encoder - the encoder output with the
                decoder's gradient, the log-probabilities with w * ctc_loss's gradient (st_ctc_dlogits -> the head's backward
                GEMMs) -, the deferred weight gradients, clip + Adam over the model's arena, the head's own Adam.
    One batch signature at a time (a new signature re-captures)."""

    def __init__(self, model: nn.Module, optimizer, head, max_grad_norm: float, head_optimizer=None, use_graph: bool = True,
                 graph_warmup: int = 2):
        self.model, self.optimizer, self.head, self.head_optimizer = model, optimizer, head, head_optimizer
        self.max_grad_norm, self.use_graph, self.graph_warmup = max_grad_norm, use_graph, graph_warmup
        self.global_step = 0
        self._sig, self._seen, self._cap, self._plan = None, 0, None, None
        self._seed = None
        self._side = None

    # ---- the parts ------------------------------------------------------------------------------------------------------
    def _part_a1(self, batch, plan, layouts):
        inputs, in_len, targets, tgt_len, gt = batch
        in_rows, t_rows = layouts
        # the plan's label-derived tensors are a derived copy of the label buffer: re-made in place at the head of every step
        # (a loader may refill the buffer with new labels of the same lengths) - inside graph A1 when the step is captured, so
        # the dozen tiny launches cost no host time (eagerly they were 0.36 ms of launch gaps per step)
        plan.refresh_labels(gt)
        self.optimizer.zero_grad()
        self.head.zero_grad_buffers()
        rng.advance()
        arena = arena_of(self.model)
        with arena.scope():
            enc, _ = self.model.encoder.forward_rows(inputs, in_len, in_rows)
        lp = self.head.project_rows(enc, plan)
        return enc, lp

    def _part_a2(self, batch, enc, layouts):
        from . import functional as F_
        inputs, in_len, targets, tgt_len, gt = batch
        in_rows, t_rows = layouts
        arena = arena_of(self.model)
        enc_in = enc.detach().requires_grad_(True)
        arena._depth += 1                     # the bf16 shadow of this step's weights exists (graph A1 refreshed it)
        try:
            dec, _ = self.model.decoder.forward_rows(targets, tgt_len, enc_in, in_rows, t_rows)
            att = F_.VocabCeFn.apply(dec, self.model.tgt_word_proj.weight, self.model, gt.contiguous().view(-1), 0,
                                     t_rows.scatter_index(gt.shape[1]))
        finally:
            arena._depth -= 1
        if self._seed is None or self._seed.device != att.device:
            self._seed = torch.empty((), dtype=att.dtype, device=att.device)
        self._seed.fill_(1.0 - float(self.head.ctc_weight))
        with deferred_wgrads(True):
            torch.autograd.backward(att, self._seed)
        return att.detach(), enc_in

    def _ctc(self, lp, plan):
        w = float(self.head.ctc_weight)
        ctc, g = self.head.ctc_rows(lp, plan)
        plan.g_lp.copy_(g)
        plan.g_lp.mul_(w)
        torch.mul(plan.finite.to(plan.roww.dtype), w / plan.B, out=plan.roww)
        plan.roww.div_(plan.tl.to(plan.roww.dtype))
        return ctc

    def _part_b(self, enc, enc_in, lp, plan):
        with deferred_wgrads(True):
            torch.autograd.backward([enc, lp], [enc_in.grad, plan.g_lp])
        if getattr(self.optimizer, "arena", None) is None:
            if self.head_optimizer is not None:
                self.head_optimizer.step()
            return None
        if self.head_optimizer is None:
            return self.optimizer.step_captured(grad_norm=True, max_norm=self.max_grad_norm)
        # train.py:45 clips everything that is being optimised: the global norm runs over the model's flat gradient AND the
        # CTC head's (its buffers live outside the arena), and both are scaled by the same coefficient
        arena = arena_of(self.model)
        self.optimizer.norm_scratch(arena.grad.device)
        g_model = nv.grad_norm(arena.grad, self.optimizer._norm_scratch, torch.empty((), dtype=torch.float32, device=arena.grad.device))
        gw, gb = self.head._st_gw, self.head._st_gb
        gnorm = torch.sqrt(g_model * g_model + (gw * gw).sum() + (gb * gb).sum())
        coef = torch.clamp(self.max_grad_norm / (gnorm + 1e-6), max=1.0)
        gw.mul_(coef)
        gb.mul_(coef)
        self.optimizer.step_captured(grad_norm=gnorm, max_norm=self.max_grad_norm)       # (advances the step count itself)
        self.head_optimizer.step()
        return gnorm

    def _joint(self, att, ctc):
        w = float(self.head.ctc_weight)
        return w * ctc + (1.0 - w) * att

    def _side_ctc(self, lp, plan):
        """ctc_loss + its gradient on the side stream, behind everything the main stream has queued so far; -> (loss, event)"""
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            ctc = self._ctc(lp, plan)
            done = torch.cuda.Event()
            done.record(self._side)
        return ctc, done

    def __call__(self, inputs, input_lengths, targets, target_lengths, ground_truth):
        """-> (joint loss, attention CE, CTC loss, clip norm): device tensors."""
        t_max, l_max = int(input_lengths.max()), int(target_lengths.max())
        self.global_step += 1
        batch = (inputs[:, :t_max], input_lengths, targets[:, :l_max], target_lengths, ground_truth[:, :l_max])
        sig = (inputs.data_ptr(), targets.data_ptr(), ground_truth.data_ptr(), tuple(inputs.shape), tuple(targets.shape),
               input_lengths.cpu().numpy().tobytes(), target_lengths.cpu().numpy().tobytes())
        if sig != self._sig:
            self._sig, self._seen, self._cap = sig, 0, None
            self._layouts = self.model.prepare_layouts(batch[1], batch[3], l_max, inputs.device)
            # the CTC labels are the ground truth of train.py:40 (label ids, PAD = blank = 0 past each length)
            self._plan = self.head.plan(batch[4], batch[3], batch[1], self._layouts[0])
            self._keep = batch
        plan, layouts = self._plan, self._layouts
        self.optimizer.update_learning_rate(self.global_step)
        main = torch.cuda.current_stream()
        if not self.use_graph or self._seen < self.graph_warmup:
            self._seen += 1
            enc, lp = self._part_a1(batch, plan, layouts)
            ctc, done = self._side_ctc(lp, plan)
            att, enc_in = self._part_a2(batch, enc, layouts)
            main.wait_event(done)
            gnorm = self._part_b(enc, enc_in, lp, plan)
            return self._joint(att, ctc), att, ctc, gnorm
        if self._cap is None:
            # as TrainStep._capture: everything lazily created must exist BEFORE the capture - Adam's flat state (a captured
            # zero-fill would reset it on every replay), the process-wide scratch the step's kernels share (split-K partials and
            # tickets, the norm's block partials: inside a capture they would land in this graph's private pool), the seed
            if hasattr(self.optimizer, "_flat_state") and getattr(self.optimizer, "arena", None) is not None:
                self.optimizer._flat_state()
            nv.splitk_scratch(inputs.device)
            if hasattr(self.optimizer, "norm_scratch"):
                self.optimizer.norm_scratch(inputs.device)
            if self._seed is None or self._seed.device != inputs.device:
                self._seed = torch.empty((), dtype=torch.float32, device=inputs.device)
            if self.head_optimizer is not None:
                groups = self.head_optimizer.param_groups
                if not all(g.get("capturable", False) for g in groups):
                    raise ValueError("JointTrainStep(use_graph=True): head_optimizer must be built with capturable=True "
                                     "(its step is captured in graph B)")
                if any(len(self.head_optimizer.state.get(q, {})) == 0 for g in groups for q in g["params"]):
                    raise ValueError("JointTrainStep(use_graph=True): head_optimizer has no state yet - run at least one eager "
                                     "step first (graph_warmup >= 1): state created inside the capture would be reset by "
                                     "every replay")
            torch.cuda.synchronize()
            pool = torch.cuda.graph_pool_handle()
            ga1, ga2, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga1, pool=pool):
                enc, lp = self._part_a1(batch, plan, layouts)
            with torch.cuda.graph(ga2, pool=pool):
                att, enc_in = self._part_a2(batch, enc, layouts)
            with torch.cuda.graph(gb, pool=pool):
                gnorm = self._part_b(enc, enc_in, lp, plan)
            self._cap = (ga1, ga2, gb, att, lp, gnorm)
        ga1, ga2, gb, att, lp, gnorm = self._cap
        ga1.replay()
        ctc, done = self._side_ctc(lp, plan)
        ga2.replay()
        main.wait_event(done)
        gb.replay()
        return self._joint(att, ctc), att, ctc, gnorm

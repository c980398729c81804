"""Beam-search decode on the HIP path: drop-in for the reference's transformer/Decode.py (BASELINE config 5).

Semantics follow Decode.py:48-179 - encode once, at most 100 steps, every unfinished utterance's beam is advanced
with the log-softmax of the last position, an utterance leaves the batch when the top of its beam is EOS, the
``n_best`` hypotheses come from the final score order - with two structural changes that do not alter results:

* **KV cache** instead of re-running the decoder over the whole prefix every step (Decode.py:96-98): each
  layer's self-attention keys / values of the tokens decoded so far are kept per hypothesis (re-gathered by the
  beam's back-pointers each step); the decoder is causal, so position t only ever needed the new token.
* the encoder output is **not repeated per beam** (Decode.py:57-66): the encoder-decoder keys / values are
  projected once per utterance and every hypothesis's attention points its (offset, length) at its utterance's rows.

* the **beam state lives on the device** (scores, back-pointers, tokens, finished flags as [.., B, beam] tensors; the
  arithmetic of transformer/Beam.py for all utterances in one top-k) and so do the step counter and the cache length,
  so ONE decoder step - ~60 launches - is captured into a HIP graph after the first step and replayed; the host reads
  the finished flags every 8 steps.  A finished utterance keeps its rows and is frozen (the reference drops it from
  the batch, Decode.py:112-165: identical results); hypotheses inherit their parent's K|V history through
  ``st_cache_reorder`` (the back-pointers of Beam.py:65 applied to the cache).

One decode step reuses the training kernels through the C-ABI (``st_gemm``, ``st_gemm_ln``, ``st_attn_fwd``: the
hypotheses of one utterance are one ``beam``-query problem of the key-split attention kernel).  The reference's constructor
cannot run (obsolete ``Transformer(...)`` signature, undefined ``prob_projection``): pass the model."""
import math

import torch

import transformer.Constants as Constants
from st_amd import functional as F_
from st_amd import native as nv
from st_amd.arena import arena_of
from transformer.Beam import Beam  # noqa: F401  (re-exported: the reference's Decode module exposes it)

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32
LN_EPS = 1e-6


class Decode(object):
    ''' Beam search over a trained Transformer. '''

    def __init__(self, opt, device, model=None):
        """opt: attribute-style (beam_size, n_best, [max_steps=100], [use_graph]); model: a transformer.Models.Transformer
        on ``device`` (the reference loads a checkpoint here with an API that no longer exists)."""
        if model is None:
            raise NotImplementedError("Decode(HIP): pass the Transformer instance (the reference's checkpoint loader "
                                      "calls an obsolete constructor and cannot run)")
        self.opt, self.device = opt, device
        self.model = model.to(device).eval()
        self.max_steps = int(getattr(opt, "max_steps", None) or 100)
        ug = getattr(opt, "use_graph", None)
        self.use_graph = torch.device(device).type == "cuda" if ug is None else bool(ug)
        self._side = None
        self._warm = False        # a step has run eagerly on this object: later calls capture before their step 0

    # ---- one decoder step for all n = B * beam hypotheses; everything that changes from step to step is DEVICE state ---
    @torch.no_grad()
    def _step(self, st):
        """-> vocabulary logits [n, v_pad] fp32 of the next token; writes the step's self-attention K|V into the caches.
        ``st`` (a _DecodeState) holds: tokens [n] int64, step [1] int64, c_len [n] int32 (= step + 1), caches
        [L, n, S, 2d], the per-utterance encoder keys / values ``cross[l]`` and the attention layouts."""
        dec = self.model.decoder
        n, d, S = st.n, dec.d_model, self.max_steps
        dst = dec._st
        if st.x_in is not None:    # (search: st_beam_advance left the decoder input of the tokens it chose; step 0: _init_state)
            x = st.x_in
        else:
            x = nv.embed_step(st.tokens, dst.emb, dst.pe, st.step, torch.empty(n, d, dtype=BF16, device=st.tokens.device))  # Models.py:84,87
        dc = st.chains
        layers = list(dec.layer_stack)

        def E(cols):
            return torch.empty(n, cols, dtype=BF16, device=x.device)

        qkv = None
        for l, layer in enumerate(layers):
            # -- masked self-attention over the cache (the causal mask is implicit: only the past is cached)
            s = layer.slf_attn._st
            c = layer.enc_attn._st
            f = layer.pos_ffn._st
            H = s.n_head
            scale = 1.0 / math.sqrt(d // H)
            if qkv is None:
                qkv = E(3 * d)
                nv.gemm(x, s.w_qkv, qkv, bias=s.b_qkv)
            ctx = E(d)
            if d // H == 64 and S <= 128:
                # decode-shaped kernel: one wave per (hypothesis, head); it also appends this step's K | V to the cache
                nv.decode_self_attn(qkv, st.caches[l], st.step, ctx, H, scale, anc=st.anc)
            else:
                st.caches[l].index_copy_(1, st.step, qkv[:, d:].unsqueeze(1))
                kv = st.caches[l].view(n * S, 2 * d)
                nv.attn_fwd(qkv[:, :d], kv[:, :d], kv[:, d:], ctx, st.lse, st.q_off, st.q_one, st.c_off, st.c_len, H, 1, False,
                            scale, max_k=S)
            # -- output_linear + LayerNorm, then the encoder-decoder attention's q projection: separate launches, or ONE
            #    row chain (csrc/st_rowchain.hip) when the layers fit it
            # -- encoder-decoder attention: keys / values projected once per utterance; the beam's hypotheses of one
            #    utterance are consecutive rows = ONE attention problem of `beam` queries (one pass over its keys).  With the
            #    row chains the chain stage and the attention are one launch (nv.attn_f1_fwd) where the shape allows
            y, q, ctx2 = E(d), E(d), E(d)
            if dc is not None:
                nv.attn_f1_fwd(ctx, dc.f1[l], (x, s.b_o, s.gamma, s.beta, y, None, None), (1, c.b_q, q), st.cross[l][:, :d],
                               st.cross[l][:, d:], ctx2, st.lse, st.u_off, st.u_len, st.k_off, st.k_len, H, st.beam, scale,
                               max_k=st.max_k)
            else:
                nv.gemm_ln(ctx, s.w_o, s.b_o, x, s.gamma, s.beta, y, None, None, eps=LN_EPS)
                nv.gemm(y, c.w_q, q, bias=c.b_q)
                nv.attn_fwd(q, st.cross[l][:, :d], st.cross[l][:, d:], ctx2, st.lse, st.u_off, st.u_len, st.k_off, st.k_len, H,
                            st.beam, False, scale, max_k=st.max_k)
            ctx = ctx2
            # -- its output_linear + LayerNorm, the position-wise feed-forward, the next layer's q|k|v projection
            x_next = E(d)
            z, h = (None, None) if dc is not None else (E(d), E(f.d_ff))      # (inside the chain neither leaves the chip)
            nxt = layers[l + 1].slf_attn._st if l + 1 < len(layers) else None
            qkv = E(3 * d) if nxt is not None else None
            if dc is not None:
                nv.row_chain(ctx, dc.f2[l], pre=(y, c.b_o, c.gamma, c.beta, z, None, None),
                             ffn=(f.d_ff, f.b1, f.b2, f.gamma, f.beta, h, x_next, None, None, None, None),
                             post=(3, nxt.b_qkv, qkv) if nxt is not None else None)
            else:
                nv.gemm_ln(ctx, c.w_o, c.b_o, y, c.gamma, c.beta, z, None, None, eps=LN_EPS)
                nv.gemm(z, f.w1, h, bias=f.b1, epi=nv.EPI_BF16_RELU)
                nv.gemm_ln(h, f.w2, f.b2, z, f.gamma, f.beta, x_next, None, None, eps=LN_EPS)
                if nxt is not None:
                    nv.gemm(x_next, nxt.w_qkv, qkv, bias=nxt.b_qkv)
            x = x_next
        ms = self.model._st
        logits = torch.empty(n, ms.v_pad, dtype=F32, device=x.device)
        nv.gemm(x, ms.w_vocab, logits, epi=nv.EPI_F32)
        return logits           # the log-softmax (the undefined `prob_projection`) is taken by st_beam_advance

    @torch.no_grad()
    def _advance(self, st, logits):
        """Beam.advance (Beam.py:43-74) for every utterance at once, on the device (``st_beam_advance``: the best of every
        hypothesis row, then one wave per utterance merges them):
        log-softmax of the logits, top-k over beam x vocab of score + log-probability, back-pointer = index // vocab,
        token = index % vocab; an utterance is finished once the top of its beam emits EOS - after which its state is
        frozen (the reference removes it from the batch, Decode.py:112-165; here it keeps its rows and is ignored).
        Step 0 expands slot 0 only (Beam.py:48-51): the other slots start at -inf.  Then the self-attention histories
        follow the back-pointers - through the lineage table the same launch maintains (``st.anc``: the cache rows stay
        where they were written), or, on the fall-back attention path, by permuting the caches - and the step counter
        advances."""
        nv.beam_advance(logits, self.model.vocab_size, st.beam, st.step, Constants.EOS, st.scores, st.tokens, st.done,
                        st.lengths, st.hist_scores, st.back, st.toks, st.order, work=st.beam_work, anc=st.anc,
                        advance_step=st.anc is not None,
                        embed=(self.model.decoder._st.emb, self.model.decoder._st.pe, st.x_in) if st.x_in is not None else None)
        if st.anc is None:         # (with the lineage table the cache rows stay where they were written, and the merge
            nv.cache_reorder(st.caches, st.order, st.step, st.beam)             # launch has advanced the step counter)
            st.step.add_(1)
        if st.need_c_len:          # (only the fall-back self-attention path reads the cache length vector)
            st.c_len.add_(1)

    @torch.no_grad()
    def decode_batch(self, src_batch):
        """src_batch = (inputs [B, T, F] fp32, input_lengths [B]) -> (all_hyp, all_scores) as Decode.py:168-177:
        all_hyp[b] = the n_best token lists of utterance b, all_scores[b] = their scores (tensor[n_best])."""
        if not self.use_graph:
            return self._decode_batch(src_batch)
        # the whole call runs on this object's side stream: a step can then be captured where it is (graph capture needs
        # a non-default stream) without hopping streams in the middle of the batch
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        cur = torch.cuda.current_stream()
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            out = self._decode_batch(src_batch)
        cur.wait_stream(self._side)
        return out

    def _init_state(self, src_batch, beam, arena, search=True):
        """Encoder pass + the device state of a search over ``beam`` hypotheses per utterance (call inside ``arena.scope()``)."""
        inputs, in_len = src_batch
        model, dev = self.model, self.device
        inputs = inputs.to(dev)
        B = inputs.shape[0]
        t_max = int(in_len.max())
        enc, in_rows = model.encoder.forward_rows(inputs[:, :t_max], in_len)        # packed [sum T, d]
        dec = model.decoder
        d, S, n = dec.d_model, self.max_steps, B * beam
        if S > dec.position_enc.pe.shape[1]:
            raise ValueError("Decode: max_steps %d exceeds the decoder's positional-encoding table" % S)
        st = _DecodeState()
        st.B, st.beam, st.n = B, beam, n
        st.chains = dec.row_chains(arena)          # None: the layers do not fit the row-chain kernel
        st.need_c_len = not (dec.d_model // dec.layer_stack[0].slf_attn.n_head == 64 and self.max_steps <= 128)
        st.cross = []
        for layer in dec.layer_stack:                                               # once per utterance
            s = layer.enc_attn._st
            kv = torch.empty(enc.shape[0], 2 * d, dtype=BF16, device=dev)
            nv.gemm(enc, s.w_kv, kv, bias=s.b_kv)
            st.cross.append(kv)
        H = dec.layer_stack[0].slf_attn.n_head
        ar = torch.arange(n, dtype=I32, device=dev)
        st.q_off, st.q_one, st.c_off = ar, torch.ones(n, dtype=I32, device=dev), ar * S
        st.c_len = torch.ones(n, dtype=I32, device=dev)
        st.u_off = torch.arange(B, dtype=I32, device=dev) * beam
        st.u_len = torch.full((B,), beam, dtype=I32, device=dev)
        st.k_off, st.k_len, st.max_k = in_rows.off, in_rows.len, int(in_rows.max_len)
        st.lse = torch.empty(H * n, dtype=F32, device=dev)
        st.caches = torch.zeros(len(dec.layer_stack), n, S, 2 * d, dtype=BF16, device=dev)
        st.tokens = torch.full((n,), Constants.BOS, dtype=torch.long, device=dev)
        st.step = torch.zeros(1, dtype=torch.long, device=dev)
        st.scores = torch.full((B, beam), float("-inf"), dtype=F32, device=dev)
        st.scores[:, 0] = 0.0
        st.done = torch.zeros(B, dtype=torch.bool, device=dev)
        st.lengths = torch.zeros(B, dtype=torch.long, device=dev)
        st.hist_scores = torch.zeros(S, B, beam, dtype=F32, device=dev)
        st.back = torch.zeros(S, B, beam, dtype=torch.long, device=dev)
        st.toks = torch.zeros(S, B, beam, dtype=torch.long, device=dev)
        st.order = torch.zeros(n, dtype=torch.long, device=dev)
        st.beam_work = torch.zeros(nv.beam_work_words(B, beam), dtype=torch.long, device=dev)  # st_beam_advance's keys + tickets
        # the lineage table of the decode-shaped self-attention (cache row of every earlier position of every hypothesis):
        # the caches are then never permuted.  The fall-back self-attention reads its own rows: st_cache_reorder stays.
        st.anc = None if st.need_c_len or model.vocab_size > 5120 else \
            torch.arange(n, dtype=I32, device=dev).unsqueeze(1).repeat(1, S).contiguous()
        # the decoder input of the current step: written by st_beam_advance for the tokens it chose (no st_embed_step launch per
        # step); step 0's is made here.  Only a search advances that way (search=False: the caller feeds the tokens).
        st.x_in = None
        if search and st.anc is not None:
            st.x_in = nv.embed_step(st.tokens, dec._st.emb, dec._st.pe, st.step, torch.empty(n, d, dtype=BF16, device=dev))
        return st

    @torch.no_grad()
    def score_hypotheses(self, src_batch, hyps):
        """Teacher-forced log-probability of ONE given token list per utterance, computed by the DECODE path (the step
        kernels of ``decode_batch``: KV cache, shared encoder keys, row chains) with the search switched off - the tokens
        are fed, not chosen.  -> tensor [B] fp32.  Not in the reference (its Decode has no scoring entry point): this is what
        lets the decode kernels be held to the fp64 oracle WITHOUT the selection bias of an arg-max over noisy scores
        (tests/test_decode_cpu.py::run_decode), and it rescores n-best lists."""
        B = src_batch[0].shape[0]
        assert len(hyps) == B and all(len(h) <= self.max_steps for h in hyps)
        V, dev = self.model.vocab_size, self.device
        arena = arena_of(self.model)
        total = torch.zeros(B, dtype=torch.float64, device=dev)
        with arena.scope():
            st = self._init_state(src_batch, 1, arena, search=False)
            steps = max(len(h) for h in hyps)
            fed = torch.full((steps, B), Constants.PAD, dtype=torch.long)
            live = torch.zeros(steps, B, dtype=torch.float64)
            for b, h in enumerate(hyps):
                fed[:len(h), b] = torch.tensor(h, dtype=torch.long)
                live[:len(h), b] = 1.0
            fed, live = fed.to(dev), live.to(dev)
            for t in range(steps):
                lp = torch.log_softmax(self._step(st)[:, :V].double(), -1)
                total += lp.gather(1, fed[t].unsqueeze(1)).squeeze(1) * live[t]
                st.tokens.copy_(fed[t])                 # beam 1: no back-pointers, the cache row stays where it is
                st.step.add_(1)
                if st.need_c_len:
                    st.c_len.add_(1)
        return total.float()

    def _decode_batch(self, src_batch):
        inputs, in_len = src_batch
        model, dev = self.model, self.device
        B, beam, n_best = inputs.shape[0], int(self.opt.beam_size), int(self.opt.n_best)
        arena = arena_of(model)
        with arena.scope():
            st = self._init_state(src_batch, beam, arena)
            S = self.max_steps

            def one_step():
                self._advance(st, self._step(st))

            graph, steps_done = None, 0
            # the first call on this object runs step 0 eagerly (kernel modules loaded, allocator warm) and captures at step
            # 1; later calls capture before step 0 - the host records the step while the GPU is still busy with the encoder
            capture_at = 0 if self._warm else 1
            while steps_done < S:
                if self.use_graph and steps_done == capture_at and graph is None:
                    # every later step is a replay of ONE captured step: the position, the cache length, the tokens and the
                    # beams are device state
                    # (captured by hand, on decode_batch's side stream: `with torch.cuda.graph(...)` also runs gc.collect() and
                    # torch.cuda.empty_cache() - 4 ms of every decode_batch call, a tenth of a 32-utterance batch)
                    if not self._warm:
                        torch.cuda.current_stream().synchronize()      # (as torch.cuda.graph does before a capture)
                    graph = torch.cuda.CUDAGraph()
                    graph.capture_begin(**(dict(capture_error_mode="thread_local") if torch.distributed.is_initialized() else {}))
                    try:
                        one_step()
                    except BaseException:
                        # a wrapper raised mid-capture (non-zero status, out of memory in the private pool): leave
                        # capture mode before the error travels on, or every later call on this stream fails with
                        # errors that hide this one
                        try:
                            graph.capture_end()
                        except Exception:  # noqa: BLE001 - the original error is the one to report
                            pass
                        graph = None
                        raise
                    graph.capture_end()
                    self._warm = True
                if graph is not None:
                    graph.replay()
                else:
                    one_step()
                steps_done += 1
                # the EOS test of Beam.py:70 lives on the device; the host looks at it every few steps only
                if (steps_done % 8 == 0 or not self.use_graph) and bool(st.done.all()):
                    break

        # ---- read-out (Beam.sort_scores / Beam.get_hypothesis, Beam.py:76-116) from the device trellis ------------------
        # The trellis crosses to the host ONCE and is walked as plain lists: per utterance the scores are sorted as
        # Beam.sort_scores does and each of the n_best slots is followed back through the back-pointers.  (Building a Beam
        # object per utterance - three lists of per-step tensor views, a stack + tolist per hypothesis - was 3.0 of the
        # 28.6 ms of a 32-utterance call; before that the read-out ran on the device: a sort, two stacks and three
        # synchronising .tolist() per utterance.)
        lengths = st.lengths.tolist()
        back_h, toks_h, scores_h = st.back.cpu().tolist(), st.toks.cpu().tolist(), st.scores.cpu()
        all_hyp, best = [], []
        for b in range(B):
            scores, slots = torch.sort(scores_h[b], 0, True)
            best.append(scores[:n_best])
            hyps = []
            for slot in slots[:n_best].tolist():
                out = []
                for step in range(lengths[b] - 1, -1, -1):
                    out.append(toks_h[step][b][slot])
                    slot = back_h[step][b][slot]
                out.reverse()
                hyps.append(out)
            all_hyp.append(hyps)
        all_scores = list(torch.stack(best).to(dev).unbind(0)) if best else []
        self.beams = None
        return all_hyp, all_scores


class _DecodeState(object):
    """Device-resident state of one decode_batch call (attribute bag)."""
    pass

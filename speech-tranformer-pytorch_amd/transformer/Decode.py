"""Beam-search decode on the HIP path: drop-in for the reference's transformer/Decode.py (BASELINE config 5).

Semantics follow Decode.py:48-179 - encode once, at most 100 steps, every unfinished utterance's beam is advanced
with the log-softmax of the last position, an utterance leaves the batch when the top of its beam is EOS, the
``n_best`` hypotheses come from the final score order - with two structural changes that do not alter results:

* **KV cache** instead of re-running the decoder over the whole prefix every step (Decode.py:96-98): each
  layer's self-attention keys / values of the tokens decoded so far are kept per hypothesis (re-gathered by the
  beam's back-pointers each step); the decoder is causal, so position t only ever needed the new token.
* the encoder output is **not repeated per beam** (Decode.py:57-66): the encoder-decoder keys / values are
  projected once per utterance and every hypothesis's attention points its (offset, length) at its utterance's rows.

One decode step is a chain of small launches (attention with one query row per hypothesis); it reuses the
training kernels through the C-ABI (``st_gemm``, ``st_gemm_ln``, ``st_attn_fwd``).  The reference's constructor
cannot run (obsolete ``Transformer(...)`` signature, undefined ``prob_projection``): pass the model."""
import math

import torch

import transformer.Constants as Constants
from st_amd import functional as F_
from st_amd import native as nv
from st_amd.arena import arena_of
from transformer.Beam import Beam

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32
LN_EPS = 1e-6


class Decode(object):
    ''' Beam search over a trained Transformer. '''

    def __init__(self, opt, device, model=None):
        """opt: attribute-style (beam_size, n_best, [max_steps=100]); model: a transformer.Models.Transformer on
        ``device`` (the reference loads a checkpoint here with an API that no longer exists)."""
        if model is None:
            raise NotImplementedError("Decode(HIP): pass the Transformer instance (the reference's checkpoint loader "
                                      "calls an obsolete constructor and cannot run)")
        self.opt, self.device = opt, device
        self.model = model.to(device).eval()
        self.max_steps = int(getattr(opt, "max_steps", 100))

    # ---- one decoder step for `n` hypotheses -----------------------------------------------------
    @torch.no_grad()
    def _step(self, tokens, step, caches, cross, hyp_koff, hyp_klen, max_k):
        """tokens [n] int64 (last token of every hypothesis), step = its position; caches[l] bf16
        [n, max_steps, 2d] (self-attention K|V of positions < step, already in hypothesis order);
        cross[l] = bf16 [enc_rows, 2d]; hyp_koff / hyp_klen int32 [n]: the utterance rows each hypothesis attends.
        -> log-probabilities [n, V] fp32; the step's K|V are written into the caches."""
        dec = self.model.decoder
        n, d = tokens.numel(), dec.d_model
        dev = tokens.device
        st = dec._st
        x = (st.emb[tokens] + st.pe[step]).to(BF16)                                 # Models.py:84,87 (repair R3)
        q_off = torch.arange(n, dtype=I32, device=dev)
        q_len = torch.ones(n, dtype=I32, device=dev)
        c_off = q_off * self.max_steps                                               # cache rows of hypothesis j
        c_len = torch.full((n,), step + 1, dtype=I32, device=dev)
        lse = torch.empty(dec.layer_stack[0].slf_attn.n_head * n, dtype=F32, device=dev)
        for l, layer in enumerate(dec.layer_stack):
            # -- masked self-attention over the cache (the causal mask is implicit: only the past is cached)
            s = layer.slf_attn._st
            H = s.n_head
            scale = 1.0 / math.sqrt(d // H)
            qkv = torch.empty(n, 3 * d, dtype=BF16, device=dev)
            nv.gemm(x, s.w_qkv, qkv, bias=s.b_qkv)
            caches[l][:, step] = qkv[:, d:]
            kv = caches[l].view(n * self.max_steps, 2 * d)
            ctx = torch.empty(n, d, dtype=BF16, device=dev)
            nv.attn_fwd(qkv[:, :d], kv[:, :d], kv[:, d:], ctx, lse, q_off, q_len, c_off, c_len, H, 1, False, scale,
                        max_k=step + 1)
            y = torch.empty(n, d, dtype=BF16, device=dev)
            nv.gemm_ln(ctx, s.w_o, s.b_o, x, s.gamma, s.beta, y, None, None, eps=LN_EPS)
            # -- encoder-decoder attention: keys / values projected once per utterance
            s = layer.enc_attn._st
            q = torch.empty(n, d, dtype=BF16, device=dev)
            nv.gemm(y, s.w_q, q, bias=s.b_q)
            nv.attn_fwd(q, cross[l][:, :d], cross[l][:, d:], ctx, lse, q_off, q_len, hyp_koff, hyp_klen, H, 1, False,
                        scale, max_k=max_k)
            z = torch.empty(n, d, dtype=BF16, device=dev)
            nv.gemm_ln(ctx, s.w_o, s.b_o, y, s.gamma, s.beta, z, None, None, eps=LN_EPS)
            # -- position-wise feed-forward
            s = layer.pos_ffn._st
            h = torch.empty(n, s.d_ff, dtype=BF16, device=dev)
            nv.gemm(z, s.w1, h, bias=s.b1, epi=nv.EPI_BF16_RELU)
            x = torch.empty(n, d, dtype=BF16, device=dev)
            nv.gemm_ln(h, s.w2, s.b2, z, s.gamma, s.beta, x, None, None, eps=LN_EPS)
        ms = self.model._st
        logits = torch.empty(n, ms.v_pad, dtype=F32, device=dev)
        nv.gemm(x, ms.w_vocab, logits, epi=nv.EPI_F32)
        return torch.log_softmax(logits[:, :self.model.vocab_size], dim=-1)         # the undefined `prob_projection`

    @torch.no_grad()
    def decode_batch(self, src_batch):
        """src_batch = (inputs [B, T, F] fp32, input_lengths [B]) -> (all_hyp, all_scores) as Decode.py:168-177:
        all_hyp[b] = the n_best token lists of utterance b, all_scores[b] = their scores (tensor[n_best])."""
        inputs, in_len = src_batch
        model, dev = self.model, self.device
        inputs = inputs.to(dev)
        B, beam, n_best = inputs.shape[0], int(self.opt.beam_size), int(self.opt.n_best)
        arena = arena_of(model)
        with arena.scope():
            t_max = int(in_len.max())
            enc, in_rows = model.encoder.forward_rows(inputs[:, :t_max], in_len)        # packed [sum T, d]
            dec = model.decoder
            d = dec.d_model
            cross = []
            for layer in dec.layer_stack:                                               # once per utterance
                s = layer.enc_attn._st
                kv = torch.empty(enc.shape[0], 2 * d, dtype=BF16, device=dev)
                nv.gemm(enc, s.w_kv, kv, bias=s.b_kv)
                cross.append(kv)
            in_off_h, in_len_h = in_rows.off.cpu(), in_rows.len.cpu()

            beams = [Beam(beam, dev) for _ in range(B)]
            active = list(range(B))
            caches = [torch.zeros(B * beam, self.max_steps, 2 * d, dtype=BF16, device=dev) for _ in dec.layer_stack]
            koff = klen = None
            max_k = 0
            for step in range(self.max_steps):
                n = len(active) * beam
                tokens = torch.cat([beams[b].next_ys[-1] for b in active])              # slot order = score order
                if koff is None:                                                        # (re)built only when the batch shrinks
                    idx = torch.tensor(active).repeat_interleave(beam)
                    koff, klen = in_off_h[idx].to(dev, I32), in_len_h[idx].to(dev, I32)
                    max_k = int(in_len_h[idx].max())
                word_lk = self._step(tokens, step, [c[:n] for c in caches], cross, koff, klen,
                                     max_k).view(len(active), beam, -1)
                done = Beam.advance_batch([beams[b] for b in active], word_lk)          # one top-k, one host read
                still, origins = [], []
                for i, b in enumerate(active):
                    if not done[i]:
                        still.append(b)
                        origins.append(beams[b].get_current_origin() + i * beam)        # rows of the step's layout
                if not still:
                    break
                if len(still) != len(active):
                    koff = None
                # finished utterances leave the batch (Decode.py:112-165); surviving hypotheses inherit the
                # cache rows of the hypothesis they extend
                order = torch.cat(origins)
                for l in range(len(caches)):
                    caches[l][:order.numel(), :step + 1] = caches[l][:n].index_select(0, order)[:, :step + 1]
                active = still

        all_hyp, all_scores = [], []
        for b in range(B):
            scores, tail_idxs = beams[b].sort_scores()
            all_scores += [scores[:n_best]]
            all_hyp += [[beams[b].get_hypothesis(i) for i in tail_idxs[:n_best].tolist()]]
        return all_hyp, all_scores

"""Encoder / Decoder / Transformer assembly (drop-in for reference transformer/Models.py).

Same class names, constructor arguments, forward signatures, return arity and
``state_dict`` keys as the reference; inside, utterances are packed into ragged
bf16 row matrices once at the front-end and stay packed through every layer (no
padded frame is ever multiplied), lengths replace the dense masks, and the
positional-encoding add is fused into the front-end / embedding kernels.

Decoder semantics follow the documented minimal repairs R3/R4 of the reference's
unfinished ``Decoder.forward`` (embeddings PLUS positional
encoding; masks from the length vectors; 2-tuple unpack).
"""
import torch
import torch.nn as nn

import transformer.Constants as Constants
from st_amd import functional as F_
from st_amd import rng
from st_amd.arena import arena_of, bundle
from st_amd.chains import DecoderChains, EncoderChains
from transformer.Embedding import PositionalEncoding
from transformer.Layers import EncoderLayer, DecoderLayer


def _check_lengths(lengths, limit, what):
    if int(lengths.min()) < 1:
        raise ValueError("%s: every utterance needs length >= 1 (a fully masked softmax row is NaN in the reference)" % what)
    if int(lengths.max()) > limit:
        raise ValueError("%s: length %d exceeds the positional-encoding table (%d)" % (what, int(lengths.max()), limit))


class Encoder(nn.Module):
    """Linear+ReLU+Dropout+LayerNorm front-end, += PE, N x EncoderLayer (Models.py:14-56)."""

    def __init__(self, input_size, n_max_seq, n_layers=6, n_head=8, d_k=64, d_v=64,
                 d_model=512, d_inner_hid=1024, dropout=0.1, emb_scale=1):
        super(Encoder, self).__init__()
        self.n_max_seq, self.d_model, self.emb_scale = n_max_seq, d_model, emb_scale
        self.position_enc = PositionalEncoding(dropout, d_model, self.n_max_seq)
        # nn.Dropout() here is p=0.5 whatever the config says (reference Models.py:31)
        self.input_proj = nn.Sequential(nn.Linear(input_size, d_model, bias=True), nn.ReLU(), nn.Dropout(),
                                        nn.LayerNorm(d_model, eps=1e-6))
        self.layer_stack = nn.ModuleList([
            EncoderLayer(d_model, d_inner_hid, n_head, d_k, d_v, dropout=dropout) for _ in range(n_layers)])
        self.use_row_chains = True      # False: every GEMM of the layer stack is its own launch (A/B runs, tests)

    def _st_bind(self, a):
        lin, ln = self.input_proj[0], self.input_proj[3]
        params = [lin.weight, lin.bias, ln.weight, ln.bias]
        lo, hi = a.span(params)
        return bundle(d_model=self.d_model, front_params=params, front_lo=lo, front_hi=hi,
                      pe=self.position_enc.pe[0],
                      w_in=a.bf16(lin.weight), b_in=a.master(lin.bias), gamma_in=a.master(ln.weight),
                      beta_in=a.master(ln.bias), g_w_in=a.grad_view(lin.weight), g_b_in=a.grad_view(lin.bias),
                      g_gamma_in=a.grad_view(ln.weight), g_beta_in=a.grad_view(ln.bias))

    def forward_rows(self, inputs, inputs_length, rows=None, packed=None):
        """inputs [B, T, F] fp32 (zero past each length) -> (packed bf16 [sum(len), d], Rows).
        ``packed = (row_matrix bf16 [sum(len), F], Rows)``: features already in the ragged layout
        (st_amd.features.stack_frames) - ``inputs`` / ``inputs_length`` are then ignored."""
        arena = arena_of(self)
        with arena.scope():
            if packed is not None:
                xp, rows = packed
                if rows.max_len > self.n_max_seq:
                    raise ValueError("Encoder: utterance of %d frames exceeds max_inputs_length %d" % (rows.max_len, self.n_max_seq))
            else:
                _check_lengths(inputs_length, min(self.n_max_seq, inputs.shape[1]), "Encoder")
                if rows is None:
                    rows = F_.Rows.packed(inputs_length, inputs.device)
            rows.pos                                      # position table built before the first launch
            if packed is None:
                xp = F_.PackFn.apply(inputs.float(), rows)
            drop = rng.site(xp.device, self.input_proj[2].p) if self.training else None   # Models.py:31: p = 0.5
            e = F_.FrontendFn.apply(xp, self.input_proj[0].weight, self, rows, drop)
            link = None                         # FrontendFn's LayerNorm output is masked/offset: not linked
            # everything between two attention kernels as ONE row-chain launch (st_amd.chains); the Functions below then
            # only record the autograd nodes over those values
            ec = self.row_chains(arena)
            pres = None
            if ec is not None:
                need_bwd = torch.is_grad_enabled() and e.requires_grad
                with torch.no_grad():
                    e_out, pres = ec.forward(self.layer_stack, e, rows, need_bwd)
                if not need_bwd:
                    # (no autograd replay below, so the sublayer Functions - whose forward feeds an active AttnTap - never
                    # run: return_attns under torch.no_grad() takes its maps from the chains' own q | k | v buffers)
                    F_.AttnTap.record_chain(self.layer_stack, pres, rows, rows)
                    return e_out, rows
            for l, layer in enumerate(self.layer_stack):
                e, link = layer.forward_rows(e, rows, link, pre=pres[l] if pres is not None else None)
        return e, rows

    def row_chains(self, arena):
        """This encoder's row-chain plan (st_amd.chains.EncoderChains) for ``arena`` or None; see Decoder.row_chains."""
        hit = getattr(self, "_st_chains", None)
        if hit is None or hit[0] is not arena:
            ec = EncoderChains.plan(list(self.layer_stack), arena) if self.use_row_chains else None
            hit = (arena, ec)
            self._st_chains = hit
        return hit[1] if self.use_row_chains else None

    def forward(self, inputs, inputs_length, return_attns=False):
        """-> (enc_output [B, T, d], enc_slf_attns): with return_attns (Models.py:53-54) one f32 [B, h, T, T] map per layer
        (st_attn_probs recomputes it from the layer's projected queries and keys; zeros at masked keys and in the rows of
        padding frames; training-mode dropout is not applied to the returned maps), else []."""
        if not return_attns:
            e, rows = self.forward_rows(inputs, inputs_length)
            return F_.UnpackFn.apply(e, rows, inputs.shape[1]), []
        T = inputs.shape[1]
        with F_.AttnTap() as tap:
            e, rows = self.forward_rows(inputs, inputs_length)
        return F_.UnpackFn.apply(e, rows, T), [tap.of(layer.slf_attn, T, T)[0] for layer in self.layer_stack]


class Decoder(nn.Module):
    """Embedding + PE, N x DecoderLayer (Models.py:59-111 with repairs R3/R4)."""

    def __init__(self, vocab_size, n_max_seq, n_layers=6, n_head=8, d_k=64, d_v=64,
                 d_model=512, d_inner_hid=1024, dropout=0.1, emb_scale=1):
        super(Decoder, self).__init__()
        self.n_max_seq, self.output_dim, self.d_model, self.emb_scale = n_max_seq, vocab_size, d_model, emb_scale
        self.position_enc = PositionalEncoding(dropout, d_model, self.n_max_seq)
        self.tgt_word_emb = nn.Embedding(vocab_size, d_model, Constants.PAD)
        self.layer_stack = nn.ModuleList([
            DecoderLayer(d_model, d_inner_hid, n_head, d_k, d_v, dropout=dropout) for _ in range(n_layers)])
        self.use_row_chains = True      # False: every GEMM of the layer stack is its own launch (A/B runs, tests)

    def _st_bind(self, a):
        emb = self.tgt_word_emb.weight
        lo, hi = a.span([emb])
        return bundle(d_model=self.d_model, pad_idx=Constants.PAD, emb_params=[emb], emb_lo=lo, emb_hi=hi,
                      emb=a.master(emb), g_emb=a.grad_view(emb), pe=self.position_enc.pe[0])

    def row_chains(self, arena):
        """This decoder's row-chain plan (st_amd.chains.DecoderChains) for ``arena``, or None when the layers do not fit the
        chain kernel.  Built once per arena (its fragment buffers join the arena's ChainHub: filled now from the current
        bf16 shadow, then by every ``arena.refresh()``); must first be called outside a HIP-graph capture
        (Transformer.prepare_layouts does)."""
        hit = getattr(self, "_st_chains", None)
        if hit is None or hit[0] is not arena:
            dc = DecoderChains.plan(list(self.layer_stack), arena) if self.use_row_chains else None
            hit = (arena, dc)
            self._st_chains = hit
        return hit[1] if self.use_row_chains else None

    def forward_rows(self, tokens, tgt_len, enc_rows_mat, in_rows, t_rows=None):
        _check_lengths(tgt_len, min(self.n_max_seq, tokens.shape[1]), "Decoder")
        arena = arena_of(self)
        with arena.scope():
            if t_rows is None:
                t_rows = F_.Rows.packed(tgt_len, tokens.device)
            y = F_.EmbedFn.apply(self.tgt_word_emb.weight, self, tokens.contiguous(), t_rows)
            link = None
            ckv = F_.CrossKv.plan([layer.enc_attn for layer in self.layer_stack])
            if ckv is not None:
                # the K/V projections of the encoder output for all layers: one GEMM now (and one in the backward)
                kv = F_.CrossKvFn.apply(enc_rows_mat, self.layer_stack[0].enc_attn.linear_k.weight, ckv)
                # decoder-sized row counts: everything between two attention kernels is ONE row-chain launch
                # (st_amd.chains); the Functions below then only record the autograd nodes over those values
                dc = self.row_chains(arena)
                pres = None
                if dc is not None:
                    need_bwd = torch.is_grad_enabled() and (y.requires_grad or kv.requires_grad)
                    with torch.no_grad():
                        y_out, pres = dc.forward(self.layer_stack, y, kv, t_rows, in_rows, need_bwd)
                    if not need_bwd:
                        F_.AttnTap.record_chain(self.layer_stack, pres, t_rows, in_rows)      # (see Encoder.forward_rows)
                        return y_out, t_rows
                for l, layer in enumerate(self.layer_stack):
                    y, link = layer.forward_rows(y, kv, t_rows, in_rows, F_.CrossKvSlot(ckv, l), link,
                                                 pre=pres[l] if pres is not None else None)
            else:
                # one accumulator for the encoder gradient of all layers (only when the encoder output needs one)
                acc = F_.CrossGradAcc(len(self.layer_stack)) if enc_rows_mat.requires_grad else None
                for layer in self.layer_stack:
                    y, link = layer.forward_rows(y, enc_rows_mat, t_rows, in_rows, acc, link)
        return y, t_rows

    def forward(self, outputs_data, outputs_pos, input_pos, enc_output, return_attns=False):
        """outputs_data [B, L] tokens, outputs_pos [B] target lengths, input_pos [B]
        input lengths, enc_output [B, T, d] -> (dec_output [B, L, d], [], [])."""
        in_rows = F_.Rows.packed(input_pos, enc_output.device)
        enc = F_.PackFn.apply(enc_output.float(), in_rows)
        if not return_attns:
            y, t_rows = self.forward_rows(outputs_data, outputs_pos, enc, in_rows)
            return F_.UnpackFn.apply(y, t_rows, outputs_data.shape[1]), [], []
        # Models.py:107-109: per layer the causal self-attention map [B, h, L, L] and the encoder-decoder map [B, h, L, T]
        L, T = outputs_data.shape[1], enc_output.shape[1]
        with F_.AttnTap() as tap:
            y, t_rows = self.forward_rows(outputs_data, outputs_pos, enc, in_rows)
        return (F_.UnpackFn.apply(y, t_rows, L), [tap.of(layer.slf_attn, L, L)[0] for layer in self.layer_stack],
                [tap.of(layer.enc_attn, L, T)[0] for layer in self.layer_stack])


class Transformer(nn.Module):
    """encoder -> decoder -> bias-free vocabulary projection (Models.py:114-153).

    ``config`` is attribute-style with the reference's keys (Models.py:120-143):
    feature_dim, max_inputs_length (``max_input_length`` - the spelling the shipped
    YAML uses - is accepted too), max_target_length, num_enc_layer, num_dec_layer,
    n_heads, d_k, d_v, d_model, d_inner_hid, dropout, vocab_size, [emb_scale],
    [return_attns].  Missing required keys raise instead of silently reading None.
    """

    def __init__(self, config):
        super(Transformer, self).__init__()

        def need(*names):
            for n in names:
                v = getattr(config, n, None) if not isinstance(config, dict) else config.get(n)
                if v is not None:
                    return v
            raise KeyError("Transformer(config): missing required key %s" % "/".join(names))

        def opt(name, default):
            v = getattr(config, name, None) if not isinstance(config, dict) else config.get(name)
            return default if v is None else v

        self.return_attns = opt('return_attns', None)
        common = dict(n_head=need('n_heads'), d_k=need('d_k'), d_v=need('d_v'), d_model=need('d_model'),
                      d_inner_hid=need('d_inner_hid'), dropout=need('dropout'), emb_scale=opt('emb_scale', 1))
        self.vocab_size, self.d_model = need('vocab_size'), common['d_model']
        self._d_k, self._n_head = common['d_k'], common['n_head']
        self.encoder = Encoder(input_size=need('feature_dim'), n_max_seq=need('max_inputs_length', 'max_input_length'),
                               n_layers=need('num_enc_layer'), **common)
        self.decoder = Decoder(vocab_size=self.vocab_size, n_max_seq=need('max_target_length'),
                               n_layers=need('num_dec_layer'), **common)
        self.tgt_word_proj = nn.Linear(self.d_model, self.vocab_size, bias=False)

    def _st_row_padding(self):
        return [(self.tgt_word_proj.weight, (self.vocab_size + 7) // 8 * 8)]

    def _st_bind(self, a):
        d_model, d_k = self.d_model, self._d_k
        if d_model not in (128, 256, 512) or d_k not in (32, 64, 128):
            # raised when the HIP path is bound (first forward on a GPU), not at construction: building the module tree,
            # loading / saving its state_dict and handing its weights to the oracle work for any size the reference
            # accepts.  The attention kernels keep a whole head row per lane (d_k 32, 64 or 128 - the last is the
            # reference's shipped config/character.yaml), the LayerNorm-fused GEMMs own full rows (d_model 128 / 256 / 512);
            # the fused row chains serve d_model 256, other widths run one launch per GEMM (README.md "Supported shapes").
            raise NotImplementedError(
                "Transformer(HIP path): d_model must be 128, 256 or 512 and d_k = d_v = d_model / n_heads 32, 64 or 128; got "
                "d_model %d (d_k %d)." % (d_model, d_k))
        w = self.tgt_word_proj.weight
        v_pad = (self.vocab_size + 7) // 8 * 8
        lo, hi = a.span([w])
        # additive mask of the padded vocabulary columns: -1e30 there makes log-softmax over the whole padded row equal
        # to log-softmax over the vocabulary (their probability is exactly 0), so a loss can read the buffer in place
        pad_bias = torch.zeros(v_pad, dtype=torch.float32, device=w.device)
        pad_bias[self.vocab_size:] = -1e30
        return bundle(v_pad=v_pad, vocab_params=[w], vocab_lo=lo, vocab_hi=hi, pad_bias=pad_bias,
                      w_vocab=a.bf16(w, v_pad), g_w_vocab=a.grad_view(w, v_pad))

    def forward_joint(self, inputs, inputs_pos, targets, targets_pos):
        """For the joint CTC + attention objective (BASELINE config 4; transformer/Loss.py:CTCAttentionLoss):
        -> (seq_logit [B, L, V], enc_output [B, T, d] fp32, zero past each length).  Both are differentiable;
        the encoder receives the sum of the decoder's and the CTC branch's gradients through autograd."""
        B, L = targets.shape
        logits, t_rows, enc, in_rows = self.forward_packed(inputs, inputs_pos, targets, targets_pos, want_enc=True)
        padded = logits.new_zeros(B * L, logits.shape[1]).index_copy(0, t_rows.scatter_index(L), logits)
        return padded.view(B, L, -1), F_.UnpackFn.apply(enc, in_rows, int(in_rows.max_len))

    def prepare_layouts(self, inputs_pos, targets_pos, l_max, device):
        """Every ragged layout of a batch (row offsets / lengths, position tables, the scatter index of the padded target
        layout, the attention work lists) and its host->device copies, set up BEFORE the first kernel launch of a step -
        and before a HIP-graph capture, whose kernels then hold these tensors by address (trainer.TrainStep pins the
        returned objects).  -> (input Rows, target Rows)"""
        in_rows = F_.Rows.packed(inputs_pos, device)
        t_rows = F_.Rows.packed(targets_pos, device)
        t_rows.scatter_index(l_max)
        in_rows.pos, t_rows.pos
        dk = self._d_k
        F_.attn_work(in_rows, in_rows, False, dk, self._n_head)     # encoder self-attention
        F_.attn_work(t_rows, t_rows, True, dk, self._n_head)        # decoder self-attention (causal)
        F_.attn_work(t_rows, in_rows, False, dk, self._n_head)      # decoder-encoder attention
        self.encoder.row_chains(arena_of(self))   # the chain plans' block tables (a host->device copy the first time)
        self.decoder.row_chains(arena_of(self))
        return in_rows, t_rows

    def forward_packed(self, inputs, inputs_pos, targets, targets_pos, want_enc=False, cut_encoder=False,
                       padded_logits=False, ce_truth=None, ignore_index=0, layouts=None):
        """The same computation with the logits left in the ragged layout the kernels produce:
        -> (logits [sum(targets_pos), V] fp32 (a column slice of a [*, v_pad] buffer), Rows of the target side).
        ``Rows.scatter_index(L)`` maps row r to its position b*L + t in the padded layout; trainer.TrainStep uses
        this form so that the loss runs over valid tokens only and nothing is scattered back to [B, L, V]."""
        arena = arena_of(self)
        # layouts = (input Rows, target Rows): given by the caller (trainer.TrainStep's bucket mode: padded layouts whose
        # lengths live on the device), else the packed layouts of this batch
        in_rows, t_rows = layouts if layouts is not None else \
            self.prepare_layouts(inputs_pos, targets_pos, targets.shape[1], inputs.device)
        with arena.scope():
            enc, _ = self.encoder.forward_rows(inputs, inputs_pos, in_rows)
            # cut_encoder: the decoder runs on a detached leaf, so that the backward can be taken in two calls -
            # loss.backward() (decoder; leaves d(enc) in enc_leaf.grad), then enc.backward(enc_leaf.grad)
            enc_in = enc.detach().requires_grad_(True) if cut_encoder else enc
            dec, _ = self.decoder.forward_rows(targets, targets_pos, enc_in, in_rows, t_rows)
            # padded_logits: return the whole [*, v_pad] buffer with the padding columns at -1e30 (a cross-entropy over it
            # equals the one over the vocabulary, and neither the column slice nor its gradient is ever copied)
            if ce_truth is not None:
                # ce_truth [B, L] int64 (the padded ground truth of train.py:40): the first return value is the token-mean
                # cross-entropy (train.py:40,120) instead of the logits - projection + loss as one autograd node
                # (the ragged rows read their ground-truth entries through their padded positions: no gather launch)
                # (a 1-D ce_truth is the padded [B, L] ground truth flattened, L = the target layout's rows per utterance,
                # with spare elements behind it for rows that belong to no utterance: trainer.TrainStep's packed buckets)
                L_gt = ce_truth.shape[1] if ce_truth.dim() == 2 else t_rows.max_len
                logits = F_.VocabCeFn.apply(dec, self.tgt_word_proj.weight, self, ce_truth.contiguous().view(-1), ignore_index,
                                            t_rows.scatter_index(L_gt))
            else:
                logits = F_.VocabFn.apply(dec, self.tgt_word_proj.weight, self, padded_logits)   # [sum(tgt_len), v_pad]
        if not padded_logits and ce_truth is None:
            logits = logits[:, :self.vocab_size]
        if cut_encoder:
            return logits, t_rows, enc, enc_in
        if want_enc:
            return logits, t_rows, enc, in_rows
        return logits, t_rows

    def forward(self, inputs, inputs_pos, targets=None, targets_pos=None):
        """inputs [B, T, F]; inputs_pos [B] input lengths; targets [B, L] tokens;
        targets_pos [B] target lengths (the current train.py:39 calling convention)
        -> (seq_logit [B, L, V] fp32, (enc_slf_attn, dec_slf_attn, dec_enc_attn)): three empty lists unless
        config.return_attns (Models.py:147-153), then one f32 map per layer each ([B, h, T, T], [B, h, L, L], [B, h, L, T])."""
        B, L = targets.shape
        T = inputs.shape[1]
        if self.return_attns:
            with F_.AttnTap() as tap:
                logits, t_rows = self.forward_packed(inputs, inputs_pos, targets, targets_pos)
            attns = ([tap.of(l.slf_attn, T, T)[0] for l in self.encoder.layer_stack],
                     [tap.of(l.slf_attn, L, L)[0] for l in self.decoder.layer_stack],
                     [tap.of(l.enc_attn, L, T)[0] for l in self.decoder.layer_stack])
        else:
            logits, t_rows = self.forward_packed(inputs, inputs_pos, targets, targets_pos)
            attns = ([], [], [])
        # scatter the ragged rows back to the padded [B, L, V] layout train.py:40 expects
        padded = logits.new_zeros(B * L, logits.shape[1]).index_copy(0, t_rows.scatter_index(L), logits)
        return padded.view(B, L, -1), attns

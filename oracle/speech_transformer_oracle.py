"""CPU restatement of the reference's encoder/decoder forward-backward.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Every function restates, in plain functional PyTorch (device-agnostic: the tests run it in float64 on the
CPU for the small configurations and, unchanged, in float64 ON THE GPU for the benchmark-sized ones), what one
reference symbol computes; the docstring cites the reference file:line it
follows (paths relative to the reference checkout).  Parameters are looked up
in a flat ``dict`` keyed by the *reference's own* ``state_dict`` names, so a
reference ``state_dict`` (the golden fixtures under ``tests/golden``) drives
the oracle directly.  All arithmetic is dtype-generic: run it in float64 for a
"truth" value, in float32 for the timed CPU baseline.

Pinning: the reference's own tests hold no golden vectors, so
the oracle is pinned against fixtures produced by importing the reference
modules in the build container (``tools/make_goldens.py``); encoder-side units
are pinned to the reference *as written*, decoder-side units to the reference
with the documented minimal repairs R1-R4 (SURVEY.md section 8c / DESIGN.md).

Gradients come from autograd over these functions - the reference does the
same (``train.py:44``).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

PAD, UNK, BOS, EOS = 0, 1, 2, 3  # transformer/Constants.py:1-4

__all__ = [
    "PAD", "UNK", "BOS", "EOS",
    "pe_table", "positional_encoding", "padding_info_mask", "feature_info_mask",
    "layer_norm", "multi_head_attention", "positionwise_ffn", "encoder_layer",
    "decoder_layer", "encoder", "decoder", "transformer", "cross_entropy",
    "label_smoothing_loss", "noam_lr", "xavier_init_", "make_params",
    "param_names", "train_step", "dp_average_grads", "synthetic_batch",
    "count_step_flops", "dropout_masks",
]


# --------------------------------------------------------------------------
# Positional encoding / masks
# --------------------------------------------------------------------------
def pe_table(max_len: int, dim: int, dtype=torch.float32) -> torch.Tensor:
    """Sinusoid table ``pe[1, max_len, dim]`` (Embedding.py:10-17).

    ``pe[0,p,2i] = sin(p * exp(-2i ln(1e4)/dim))``, ``pe[0,p,2i+1] = cos(same)``.
    The reference builds the table in float32 (``dtype=torch.float`` on the
    frequency vector, Embedding.py:12); we build it the same way and cast, so a
    float64 oracle sees bit-identical table values to the reference buffer.
    """
    pos = torch.arange(0, max_len).unsqueeze(1).float()
    freq = torch.exp(torch.arange(0, dim, 2, dtype=torch.float) * -(math.log(10000.0) / dim))
    tab = torch.zeros(max_len, dim)
    tab[:, 0::2] = torch.sin(pos * freq)
    tab[:, 1::2] = torch.cos(pos * freq)
    return tab.unsqueeze(0).to(dtype)


def positional_encoding(pe: torch.Tensor, lengths: torch.Tensor, step: Optional[int] = None) -> torch.Tensor:
    """``PositionalEncoding.forward`` (Embedding.py:21-29): returns the PE rows
    ``[B, max(lengths), dim]`` (or the single row ``step``); the caller adds."""
    bsz = lengths.size(0)
    if step is None:
        return pe[:, : int(lengths.max())].repeat(bsz, 1, 1)
    return pe[:, step].repeat(bsz, 1, 1)


def padding_info_mask(q_len: torch.Tensor, k_len: torch.Tensor) -> torch.Tensor:
    """Key-padding mask ``[B, max(q_len), max(k_len)]``, True = masked
    (Utils.py:41-57): ``mask[b,i,j] = j >= k_len[b]``; queries are never masked."""
    assert q_len.dim() == 1 and k_len.dim() == 1
    lq, lk = int(q_len.max()), int(k_len.max())
    col = torch.arange(lk).view(1, 1, lk)
    return (col >= k_len.view(-1, 1, 1)).expand(k_len.size(0), lq, lk)


def feature_info_mask(lengths: torch.Tensor) -> torch.Tensor:
    """Causal mask ``[B, L, L]``, True above the diagonal (Utils.py:60-70)."""
    assert lengths.dim() == 1
    n = int(lengths.max())
    tri = torch.triu(torch.ones(n, n, dtype=torch.bool), diagonal=1)
    return tri.unsqueeze(0).expand(lengths.size(0), n, n)


# --------------------------------------------------------------------------
# Sub-layers
# --------------------------------------------------------------------------
# Training-mode ``nn.Dropout`` (Attention.py:89, SubLayers.py:25,27, Models.py:31) restated with the Bernoulli
# draw SUPPLIED by the caller: ``out = x * keep / (1 - p)``.  Parity tests hand the oracle exactly the masks the
# HIP kernels draw (their counter-based generator cannot match torch's stream); without a provider dropout is
# the identity, i.e. the reference in eval() mode - the default parity mode.
_DROPOUT_PROVIDER = None


class dropout_masks:
    """``with dropout_masks(fn):`` - ``fn(site, shape)`` returns ``keep / (1 - p)`` (a tensor broadcastable to
    ``shape``) or None; sites are visited in forward order: "front", then per layer "attn" (per attention),
    "ffn1", "ffn2"."""

    def __init__(self, provider):
        self.provider = provider

    def __enter__(self):
        global _DROPOUT_PROVIDER
        self.saved, _DROPOUT_PROVIDER = _DROPOUT_PROVIDER, self.provider
        return self

    def __exit__(self, *exc):
        global _DROPOUT_PROVIDER
        _DROPOUT_PROVIDER = self.saved
        return False


def _dropout(x: torch.Tensor, site: str) -> torch.Tensor:
    if _DROPOUT_PROVIDER is None:
        return x
    m = _DROPOUT_PROVIDER(site, x.shape)
    return x if m is None else x * m.to(x.dtype)


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """``nn.LayerNorm(d, eps=1e-6)`` (Attention.py:62, SubLayers.py:18,
    Models.py:32): biased variance over the last dim."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)   # same definition, one fused pass on the CPU


def multi_head_attention(p: Params, pre: str, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                         mask: Optional[torch.Tensor], n_head: int,
                         return_attn: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """``MultiHeadAttention.forward`` (Attention.py:64-96), dropout = identity unless
    a ``dropout_masks`` provider is active.

    Q/K/V projections with bias (:74-76), split heads (:68-69,78-80),
    ``scores = QK^T / sqrt(d_k)`` (:82), ``masked_fill(-inf)`` (:84-87), softmax
    over keys (:89), ``context = attn V`` merged heads (:90), output projection
    (:92), then post-LN of ``output + residual`` (:94).  The reference writes the
    residual as ``+ v``; that equals the layer input for self-attention and is a
    shape error for cross-attention, so (repair R2) the residual is ``q`` - the
    same tensor whenever the reference line is well-formed.
    """
    bsz, lq, d = q.shape
    dk = d // n_head

    def split(x):
        return x.view(bsz, -1, n_head, dk).transpose(1, 2)

    qh = split(F.linear(q, p[pre + "linear_q.weight"], p[pre + "linear_q.bias"]))
    kh = split(F.linear(k, p[pre + "linear_k.weight"], p[pre + "linear_k.bias"]))
    vh = split(F.linear(v, p[pre + "linear_v.weight"], p[pre + "linear_v.bias"]))
    scores = torch.matmul(qh, kh.transpose(2, 3)).div_(math.sqrt(dk))      # in place, as Attention.py:82-87
    if mask is not None:
        scores.masked_fill_(mask.unsqueeze(1).to(scores.device), float("-inf"))   # masks are built on the host (Utils.py:41-70)
    attn = _dropout(torch.softmax(scores, dim=-1), "attn")                  # Attention.py:89
    ctx = torch.matmul(attn, vh).transpose(1, 2).contiguous().view(bsz, lq, d)
    out = F.linear(ctx, p[pre + "output_linear.weight"], p[pre + "output_linear.bias"])
    out = layer_norm(out + q, p[pre + "layernorm.weight"], p[pre + "layernorm.bias"])
    return out, (attn if return_attn else None)


def positionwise_ffn(p: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """``PositionwiseFeedForward.forward`` (SubLayers.py:24-28), dropout =
    identity: ``LN(x + fc2(relu(fc1(x))))``."""
    h = _dropout(torch.relu(F.linear(x, p[pre + "fc1.weight"], p[pre + "fc1.bias"])), "ffn1")   # SubLayers.py:25
    y = F.linear(h, p[pre + "fc2.weight"], p[pre + "fc2.bias"])
    return _dropout(layer_norm(x + y, p[pre + "layernorm.weight"], p[pre + "layernorm.bias"]), "ffn2")   # :27


def encoder_layer(p: Params, pre: str, x: torch.Tensor, mask, n_head: int, return_attn: bool = False):
    """``EncoderLayer.forward`` (Layers.py:18-22)."""
    a, w = multi_head_attention(p, pre + "slf_attn.", x, x, x, mask, n_head, return_attn)
    return positionwise_ffn(p, pre + "pos_ffn.", a), w


def decoder_layer(p: Params, pre: str, y: torch.Tensor, enc: torch.Tensor, slf_mask, enc_mask,
                  n_head: int, return_attn: bool = False):
    """``DecoderLayer.forward`` (Layers.py:37-44): masked self-attention,
    cross-attention (q = decoder stream, k = v = encoder output), FFN."""
    s, w1 = multi_head_attention(p, pre + "slf_attn.", y, y, y, slf_mask, n_head, return_attn)
    c, w2 = multi_head_attention(p, pre + "enc_attn.", s, enc, enc, enc_mask, n_head, return_attn)
    return positionwise_ffn(p, pre + "pos_ffn.", c), (w1, w2)


# --------------------------------------------------------------------------
# Model assembly
# --------------------------------------------------------------------------
def _count_layers(p: Params, pre: str) -> int:
    n = 0
    while (pre + "layer_stack.%d.slf_attn.linear_q.weight" % n) in p:
        n += 1
    return n


def encoder(p: Params, x: torch.Tensor, in_len: torch.Tensor, n_head: int,
            pre: str = "encoder.", return_attns: bool = False):
    """``Encoder.forward`` (Models.py:40-56), dropout = identity.

    ``e = LN(relu(x W_in^T + b_in))`` (:28-33; the Dropout() at :31 is identity
    in eval), ``e += PE(lengths)`` (:43-44), key-padding mask (:46), N layers."""
    e = _dropout(torch.relu(F.linear(x, p[pre + "input_proj.0.weight"], p[pre + "input_proj.0.bias"])), "front")
    e = layer_norm(e, p[pre + "input_proj.3.weight"], p[pre + "input_proj.3.bias"])
    e = e + positional_encoding(p[pre + "position_enc.pe"].to(e.dtype), in_len)
    mask = padding_info_mask(in_len, in_len)
    attns = []
    for i in range(_count_layers(p, pre)):
        e, w = encoder_layer(p, pre + "layer_stack.%d." % i, e, mask, n_head, return_attns)
        if return_attns:
            attns.append(w)
    return e, attns


def decoder(p: Params, tokens: torch.Tensor, tgt_len: torch.Tensor, in_len: torch.Tensor,
            enc_out: torch.Tensor, n_head: int, pre: str = "decoder.", return_attns: bool = False):
    """``Decoder.forward`` (Models.py:81-111) with repairs R3/R4 (the function
    cannot execute as written): ``y = Emb(tokens) + PE(tgt_len)`` (:84,87 -
    intended add, cf. the encoder at :42-44); self mask = key-padding(tgt_len)
    OR causal (:89-94, masks built from the *length vectors*); cross mask =
    key-padding(tgt_len, in_len) (:96-97); N decoder layers (:101-109)."""
    y = F.embedding(tokens, p[pre + "tgt_word_emb.weight"])
    y = y + positional_encoding(p[pre + "position_enc.pe"].to(y.dtype), tgt_len)
    slf = padding_info_mask(tgt_len, tgt_len) | feature_info_mask(tgt_len)
    crs = padding_info_mask(tgt_len, in_len)
    a1, a2 = [], []
    for i in range(_count_layers(p, pre)):
        y, (w1, w2) = decoder_layer(p, pre + "layer_stack.%d." % i, y, enc_out, slf, crs, n_head, return_attns)
        if return_attns:
            a1.append(w1)
            a2.append(w2)
    return y, a1, a2


def transformer(p: Params, x: torch.Tensor, in_len: torch.Tensor, tokens: torch.Tensor,
                tgt_len: torch.Tensor, n_head: int, return_attns: bool = False):
    """``Transformer.forward`` (Models.py:147-153): encoder, decoder, bias-free
    vocabulary projection (:145,151).  ``in_len`` / ``tgt_len`` are the length
    vectors the current training loop passes (train.py:39)."""
    enc, ea = encoder(p, x, in_len, n_head, return_attns=return_attns)
    dec, da, dc = decoder(p, tokens, tgt_len, in_len, enc, n_head, return_attns=return_attns)
    return F.linear(dec, p["tgt_word_proj.weight"]), (ea, da, dc)


# --------------------------------------------------------------------------
# Losses / optimiser / step
# --------------------------------------------------------------------------
def cross_entropy(logits: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """``nn.CrossEntropyLoss(ignore_index=0)`` on ``logits.view(-1,V)`` vs
    ``gt.view(-1)`` (train.py:40,120): mean over non-PAD targets."""
    v = logits.size(-1)
    lp = torch.log_softmax(logits.reshape(-1, v), dim=-1)
    t = gt.reshape(-1)
    keep = t != PAD
    nll = -lp[torch.arange(t.numel(), device=t.device), t]
    return (nll * keep).sum() / keep.sum()


def label_smoothing_loss(logits2d: torch.Tensor, target: torch.Tensor, smoothing: float,
                         ignore_index: int = -1) -> torch.Tensor:
    """``LabelSmoothingLoss`` + its dense ``CrossEntropyLoss`` (Loss.py:6-73),
    including the reference's quirks: the PAD column of the smoothing row is
    zeroed only when ``ignore_index == 0`` (``if not ignore_index``, :20-21);
    rows whose target equals ``ignore_index`` are zeroed when ``ignore_index >=
    0`` (:35-37); the sum is divided by *all* rows, PAD rows included (:69)."""
    n, v = logits2d.shape
    row = torch.full((v,), smoothing / (v - 1), dtype=logits2d.dtype)
    if not ignore_index:
        row[ignore_index] = 0
    prob = row.unsqueeze(0).repeat(n, 1)
    prob.scatter_(1, target.unsqueeze(1), 1.0 - smoothing)
    if ignore_index >= 0:
        prob.masked_fill_((target == ignore_index).unsqueeze(1), 0)
    return -(torch.log_softmax(logits2d, -1) * prob).sum() / n


def noam_lr(d_model: int, warmup: int, step: int) -> float:
    """``ScheduledOptim.update_learning_rate`` (Optim.py:36-45)."""
    return d_model ** -0.5 * min(step ** -0.5, warmup ** -1.5 * step)


def param_names(n_enc: int, n_dec: int) -> List[str]:
    """``named_parameters()`` order of the reference ``Transformer``."""
    mha = ["linear_q", "linear_k", "linear_v", "output_linear", "layernorm"]
    ffn = ["fc1", "fc2", "layernorm"]
    names = ["encoder.input_proj.0", "encoder.input_proj.3"]
    for i in range(n_enc):
        names += ["encoder.layer_stack.%d.slf_attn.%s" % (i, m) for m in mha]
        names += ["encoder.layer_stack.%d.pos_ffn.%s" % (i, m) for m in ffn]
    out = []
    for n in names:
        out += [n + ".weight", n + ".bias"]
    out.append("decoder.tgt_word_emb.weight")
    names = []
    for i in range(n_dec):
        names += ["decoder.layer_stack.%d.slf_attn.%s" % (i, m) for m in mha]
        names += ["decoder.layer_stack.%d.enc_attn.%s" % (i, m) for m in mha]
        names += ["decoder.layer_stack.%d.pos_ffn.%s" % (i, m) for m in ffn]
    for n in names:
        out += [n + ".weight", n + ".bias"]
    out.append("tgt_word_proj.weight")
    return out


def make_params(feature_dim: int, vocab: int, d_model: int, d_ff: int, n_enc: int, n_dec: int,
                max_in: int, max_tgt: int, dtype=torch.float32) -> Params:
    """Parameter/buffer dict shaped like ``Transformer(config).state_dict()``
    (Models.py:117-145): LayerNorm weight 1 / bias 0, everything else zero until
    :func:`xavier_init_` fills it."""
    p: Params = {}
    for n in param_names(n_enc, n_dec):
        leaf = n.rsplit(".", 2)[-2]
        kind = n.rsplit(".", 1)[-1]
        if leaf in ("layernorm", "3"):
            shape = (d_model,)
        elif leaf == "0":
            shape = (d_model, feature_dim) if kind == "weight" else (d_model,)
        elif leaf == "fc1":
            shape = (d_ff, d_model) if kind == "weight" else (d_ff,)
        elif leaf == "fc2":
            shape = (d_model, d_ff) if kind == "weight" else (d_model,)
        elif leaf in ("tgt_word_emb", "tgt_word_proj"):
            shape = (vocab, d_model)
        else:
            shape = (d_model, d_model) if kind == "weight" else (d_model,)
        is_ln_w = leaf in ("layernorm", "3") and kind == "weight"
        p[n] = torch.ones(shape, dtype=dtype) if is_ln_w else torch.zeros(shape, dtype=dtype)
    p["encoder.position_enc.pe"] = pe_table(max_in, d_model, dtype)
    p["decoder.position_enc.pe"] = pe_table(max_tgt, d_model, dtype)
    return p


def xavier_init_(p: Params, seed: int = 0) -> Params:
    """``init_parameters`` (Utils.py:101-104): xavier_normal on every
    parameter with >= 2 dims, in ``named_parameters`` order; 1-D parameters keep
    their module defaults.  NOTE: the module defaults of nn.Linear *biases* are
    random in the reference; the oracle's stand-alone initialiser draws them
    uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) like nn.Linear does but does not
    replay the reference's RNG stream - goldens carry their own weights."""
    g = torch.Generator().manual_seed(seed)
    for n, t in p.items():
        if n.endswith(".pe"):
            continue
        if t.dim() >= 2:
            fan_out, fan_in = t.shape[0], t.shape[1]
            std = math.sqrt(2.0 / (fan_in + fan_out))
            t.copy_(torch.randn(t.shape, generator=g, dtype=torch.float32).to(t.dtype) * std)
    for n, t in p.items():
        if n.endswith(".bias") and ("layernorm" not in n) and (".3." not in n):
            w = p[n[:-4] + "weight"]
            bound = 1.0 / math.sqrt(w.shape[1])
            t.copy_(((torch.rand(t.shape, generator=g, dtype=torch.float32) * 2 - 1) * bound).to(t.dtype))
    return p


def _trainable(p: Params) -> List[str]:
    return [n for n in p if not n.endswith(".pe")]


def train_step(p: Params, batch: dict, n_head: int, d_model: int, warmup: int, step: int,
               max_grad_norm: float, adam_state: Optional[dict] = None) -> dict:
    """One optimisation step as ``train()`` runs it (train.py:25-46):
    trim to the batch maxima (:31-35), forward (:39), CE with ignore_index=0
    (:40), backward (:44), global-norm clip (:45), Noam-Adam update (:46 ->
    Optim.py:11-16,36-45: betas (0.9, 0.98), eps 1e-9).

    Returns loss, logits, grads (pre-clip), grad_norm, lr and the updated
    parameters (``p`` itself is left untouched)."""
    names = _trainable(p)
    leaves = {n: (p[n].detach().clone().requires_grad_(True) if n in names else p[n]) for n in p}
    ti, tl = int(batch["in_len"].max()), int(batch["tgt_len"].max())
    x = batch["x"][:, :ti]
    tokens = batch["tokens"][:, :tl]
    gt = batch["gt"][:, :tl]
    logits, _ = transformer(leaves, x, batch["in_len"], tokens, batch["tgt_len"], n_head)
    loss = cross_entropy(logits, gt)
    grads = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(p[n])) for n, g in zip(names, grads)}
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).to(loss.dtype)
    coef = torch.clamp(max_grad_norm / (total + 1e-6), max=1.0)
    lr = noam_lr(d_model, warmup, step)
    st = adam_state if adam_state is not None else {"t": 0, "m": {}, "v": {}}
    st["t"] += 1
    b1, b2, eps = 0.9, 0.98, 1e-9
    new_p = dict(p)
    for n in names:
        g = grads[n] * coef
        m = st["m"].get(n, torch.zeros_like(g)) * b1 + (1 - b1) * g
        v = st["v"].get(n, torch.zeros_like(g)) * b2 + (1 - b2) * g * g
        st["m"][n], st["v"][n] = m, v
        mhat = m / (1 - b1 ** st["t"])
        denom = v.sqrt() / math.sqrt(1 - b2 ** st["t"]) + eps
        new_p[n] = p[n] - lr * mhat / denom
    return {"loss": loss.detach(), "logits": logits.detach(), "grads": grads, "grad_norm": total.detach(),
            "lr": lr, "params": new_p, "adam": st}


def dp_average_grads(p: Params, batch: dict, n_head: int, world: int) -> Tuple[torch.Tensor, Params]:
    """Data-parallel semantics of ``train_multi.py`` (:60-68,136-139,161-163):
    the global minibatch is split contiguously into ``world`` shards, every
    shard computes its *own* token-mean CE and gradients, and Horovod averages
    the gradients over ranks (not a global token mean).  Returns (mean of the
    per-rank losses, averaged gradients)."""
    bsz = batch["x"].size(0)
    assert bsz % world == 0
    per = bsz // world
    acc: Params = {}
    losses = []
    for r in range(world):
        sl = slice(r * per, (r + 1) * per)
        shard = {k: v[sl] for k, v in batch.items()}
        res = train_step(p, shard, n_head, d_model=1, warmup=1, step=1, max_grad_norm=float("inf"))
        losses.append(res["loss"])
        for n, g in res["grads"].items():
            acc[n] = acc.get(n, 0) + g / world
    return torch.stack(losses).mean(), acc


# --------------------------------------------------------------------------
# Synthetic workload (BASELINE.md section 3 / SURVEY.md section 8d)
# --------------------------------------------------------------------------
def synthetic_batch(bsz: int, t_max: int, l_max: int, feat: int, vocab: int, seed: int = 0,
                    t_min: Optional[int] = None, l_min: Optional[int] = None,
                    dtype=torch.float32) -> dict:
    """The seeded synthetic 80-d fbank batch of BASELINE.md section 3: lengths
    uniform in [t_min, t_max] / [l_min, l_max] with utterance 0 at the maximum,
    features zero past ``in_len``, tokens in [4, V) zero (PAD) past ``tgt_len``,
    ``gt = roll(tokens, -1)`` with a PAD tail (the ``targets[1:]`` convention of
    tests/random_character_loader.py:94-96)."""
    t_min = t_max // 2 if t_min is None else t_min
    l_min = l_max // 2 if l_min is None else l_min
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(bsz, t_max, feat, generator=g)
    in_len = torch.randint(t_min, t_max + 1, (bsz,), generator=g)
    in_len[0] = t_max
    tgt_len = torch.randint(l_min, l_max + 1, (bsz,), generator=g)
    tgt_len[0] = l_max
    tokens = torch.randint(4, vocab, (bsz, l_max), generator=g)
    ar_t = torch.arange(t_max).unsqueeze(0)
    ar_l = torch.arange(l_max).unsqueeze(0)
    x = x * (ar_t < in_len.unsqueeze(1)).unsqueeze(-1)
    tokens = tokens * (ar_l < tgt_len.unsqueeze(1))
    gt = torch.roll(tokens, -1, dims=1)
    gt[:, -1] = PAD
    gt = gt * (ar_l < (tgt_len - 1).unsqueeze(1))
    return {"x": x.to(dtype), "in_len": in_len, "tokens": tokens, "tgt_len": tgt_len, "gt": gt}


def count_step_flops(in_len, tgt_len, feat: int, d: int, d_ff: int, vocab: int, n_enc: int, n_dec: int) -> float:
    """Algorithmic FLOPs of one training step at the *valid* lengths
    (SURVEY.md section 8d): multiply-add = 2, backward = 2 x forward, no recompute."""
    fwd = 0.0
    for t, l in zip([int(v) for v in in_len], [int(v) for v in tgt_len]):
        fwd += 2 * t * feat * d
        fwd += n_enc * (8 * t * d * d + 4 * t * t * d + 4 * t * d * d_ff)
        fwd += n_dec * (8 * l * d * d + 4 * l * l * d + 4 * l * d * d + 4 * t * d * d + 4 * l * t * d
                        + 4 * l * d * d_ff)
        fwd += 2 * l * d * vocab
    return 3.0 * fwd

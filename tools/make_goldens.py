#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference); nothing here is
shipped or imported at run time.  The reference's files are never copied: the
modules are imported from where they lie, with the documented minimal-repair
shim:

  D1  stub the unused ``editdistance`` import (Utils.py:2)
  R1  uint8 masks -> bool (torch >= 2 rejects uint8 in masked_fill_)
  R2  MultiHeadAttention residual ``output + q`` instead of ``+ v``
      (identical for self-attention; Attention.py:94)
  R3  Decoder.forward: ``emb + PE(target_lengths)``, masks built from the
      length vectors (Models.py:87,89-97)
  R4  unpack DecoderLayer's ``(out, (w1, w2))`` 2-tuple (Models.py:102)

Fixtures whose ``repair`` field is "none" come from the reference exactly as
written (plus D1/R1, which do not touch arithmetic); "R2".."R4" mark the
repaired ones.  All fixtures are eval() mode (every Dropout = identity).

Usage:  PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/make_goldens.py
"""
import os
import sys
import types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = os.environ.get("ST_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.modules.setdefault("editdistance", types.ModuleType("editdistance"))  # D1

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import transformer.Attention as A  # noqa: E402
import transformer.Embedding as E  # noqa: E402
import transformer.Layers as L  # noqa: E402
import transformer.Loss as LS  # noqa: E402
import transformer.Models as M  # noqa: E402
import transformer.Optim as O  # noqa: E402
import transformer.SubLayers as S  # noqa: E402
import transformer.Utils as U  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)

# ---- R1: bool masks (Models.py:11 binds the builders by name) ---------------
_pad, _feat = U.padding_info_mask, U.feature_info_mask


def pad_mask(a, b):
    return _pad(a, b).bool()


def feat_mask(a):
    return _feat(a).bool()


M.padding_info_mask = pad_mask
M.feature_info_mask = feat_mask

_orig_mha_forward = A.MultiHeadAttention.forward


def install_repairs():
    """R2-R4.  The bodies call the reference's own sub-modules; only the
    operand of the residual add and the decoder's glue lines differ."""

    def mha_forward(self, q, k, v, mask=None):  # R2
        bsz = q.size(0)

        def shape(x):
            return x.view(bsz, -1, self.n_head, self.d_k).transpose(1, 2)

        query, key, value = shape(self.linear_q(q)), shape(self.linear_k(k)), shape(self.linear_v(v))
        scores = torch.matmul(query, key.transpose(2, 3)).div(self.scaled)
        if mask is not None:
            scores = scores.masked_fill(mask.unsqueeze(1), -float("inf"))
        attns = self.dropout(self.softmax(scores))
        ctx = torch.matmul(attns, value).transpose(1, 2).contiguous().view(bsz, -1, self.n_head * self.d_k)
        return self.layernorm(self.output_linear(ctx) + q), attns

    def dec_forward(self, tokens, tgt_len, in_len, enc_out, return_attns=False):  # R3 + R4
        x = self.tgt_word_emb(tokens) + self.position_enc(tgt_len)
        slf = M.padding_info_mask(tgt_len, tgt_len) | M.feature_info_mask(tgt_len)
        enc = M.padding_info_mask(tgt_len, in_len)
        for layer in self.layer_stack:
            x, _ = layer(x, enc_out, slf_attn_mask=slf, dec_enc_attn_mask=enc)
        return x, [], []

    A.MultiHeadAttention.forward = mha_forward
    M.Decoder.forward = dec_forward


def remove_repairs():
    A.MultiHeadAttention.forward = _orig_mha_forward


def np32(t):
    return t.detach().to(torch.float32).clone().numpy()


def np64(t):
    return t.detach().to(torch.float64).clone().numpy()


def state_np(mod, conv):
    return {"w/" + k: conv(v) for k, v in mod.state_dict().items()}


def grads_np(mod, conv, tag="g/", compact=False):
    out = {}
    for n, p in mod.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        if compact and g.numel() > 4096:
            for kk, vv in summary(g).items():
                out[tag + n + "/" + kk] = vv
        else:
            out[tag + n] = conv(g)
    return out


def summary(t, n=64, seed=0):
    """(shape, sum, sumsq, sampled flat indices, sampled values) for big tensors."""
    flat = t.detach().double().reshape(-1)
    g = torch.Generator().manual_seed(seed)
    idx = torch.randint(0, flat.numel(), (min(n, flat.numel()),), generator=g)
    return {"shape": np.array(t.shape), "sum": flat.sum().numpy(), "sumsq": (flat * flat).sum().numpy(),
            "idx": idx.numpy(), "val": flat[idx].numpy()}


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def rand_weights(mod, seed):
    """Seeded non-trivial weights for unit fixtures (biases / LN params too, so
    every term of the arithmetic is exercised)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * (1.5 / (p.shape[-1] ** 0.5)))
            elif "layernorm" in n or n.endswith("3.weight") or n.endswith("3.bias"):
                base = 1.0 if n.endswith("weight") else 0.0
                p.copy_(base + 0.2 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))


# ---------------------------------------------------------------------------
def fx_pe_masks():
    pe16 = E.PositionalEncoding(0.1, 16, 64)
    pe128 = E.PositionalEncoding(0.1, 128, 1000)
    lens = torch.tensor([5, 3])
    save("pe_masks",
         pe16=np32(pe16.pe), pe128_rows=np32(pe128.pe[0, ::37]), pe128_row_idx=np.arange(0, 1000, 37),
         pe16_fwd=np32(pe16(lens)), pe16_step=np32(pe16(lens, step=4)),
         lens=lens.numpy(), pad=pad_mask(lens, lens).numpy(), pad_q3_k5=pad_mask(torch.tensor([3, 2]), lens).numpy(),
         causal=feat_mask(lens).numpy(), repair="none")


def mha_case(name, b, lq, lk, d, h, q_len, k_len, causal, cross, seed, full=True, repair="none", slim=False):
    """slim: production-sized heads (d_k = 64) - the fp64 results are stored once, rounded to fp32 (``r32/*``: 6e-8
    relative, irrelevant next to the bf16 tolerances of the GPU parity tests) together with fp64 (sum, sumsq,
    samples) summaries that pin the oracle to 1e-12; no fp32 run, no attention map."""
    torch.manual_seed(seed)
    mha = A.MultiHeadAttention(h, d, d // h, d // h).eval()
    rand_weights(mha, seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    q0 = torch.randn(b, lq, d, generator=g)
    kv0 = torch.randn(b, lk, d, generator=g) if cross else None
    dy0 = torch.randn(b, lq, d, generator=g)
    mask = pad_mask(torch.tensor(q_len), torch.tensor(k_len))
    if causal:
        mask = mask | feat_mask(torch.tensor(q_len))
    out = {"q_len": np.array(q_len), "k_len": np.array(k_len), "causal": np.array(int(causal)),
           "n_head": np.array(h), "repair": repair, "mask": mask.numpy(), "q": np32(q0), "dy": np32(dy0)}
    if cross:
        out["kv"] = np32(kv0)
    for tag, dt in ((("f64", torch.float64),) if slim else (("f32", torch.float32), ("f64", torch.float64))):
        m = A.MultiHeadAttention(h, d, d // h, d // h).eval()
        m.load_state_dict(mha.state_dict())
        m = m.to(dt)
        q = q0.detach().clone().to(dt).requires_grad_(True)
        if cross:
            kv = kv0.detach().clone().to(dt).requires_grad_(True)
            y, attn = m(q, kv, kv, mask)
        else:
            y, attn = m(q, q, q, mask)
        (y * dy0.to(dt)).sum().backward()
        conv = np32 if tag == "f32" else np64
        if slim:
            out["r32/out"], out["r32/dq"] = np32(y), np32(q.grad)
            big = {"out": y, "dq": q.grad}
            if cross:
                out["r32/dkv"] = np32(kv.grad)
                big["dkv"] = kv.grad
            for nm, t in big.items():
                out.update({"f64/%s/%s" % (nm, k): v for k, v in summary(t).items()})
            out.update(grads_np(m, conv, "f64/g/", compact=True))
            continue
        out[tag + "/out"] = conv(y)
        out[tag + "/dq"] = conv(q.grad)
        if cross:
            out[tag + "/dkv"] = conv(kv.grad)
        if full:
            out[tag + "/attn"] = conv(attn)
        else:
            out.update({tag + "/attn_" + k: v for k, v in summary(attn).items()})
        out.update(grads_np(m, conv, tag + "/g/", compact=not full))
    out.update(state_np(mha, np32))
    save(name, **out)


def mha_dense_case(name, b, lq, lk, d, h, seed, split_kv):
    """The GENERAL form of MultiHeadAttention.forward (Attention.py:64-96): an arbitrary dense mask (random here, every query
    keeps at least one key - the reference gives NaN otherwise; the mask of the module's own ``__main__`` demo,
    Attention.py:100-102, is of this kind) and, with split_kv, keys and values projected from DIFFERENT tensors.  No call site
    of the reference uses either; the product serves them through its slow dense path (st_attn_dense_fwd / _bwd).
    Needs repair R2 when lq != lk or k is not v (residual q)."""
    torch.manual_seed(seed)
    mha = A.MultiHeadAttention(h, d, d // h, d // h).eval()
    rand_weights(mha, seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    q0, k0, dy0 = torch.randn(b, lq, d, generator=g), torch.randn(b, lk, d, generator=g), torch.randn(b, lq, d, generator=g)
    v0 = torch.randn(b, lk, d, generator=g) if split_kv else None
    mask = torch.rand(b, lq, lk, generator=g) < 0.4
    mask[:, :, 0] &= ~mask.all(-1)                      # a fully masked row keeps key 0
    out = {"n_head": np.array(h), "repair": "R2", "mask": mask.numpy(), "q": np32(q0), "k": np32(k0), "dy": np32(dy0)}
    if split_kv:
        out["v"] = np32(v0)
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        m = A.MultiHeadAttention(h, d, d // h, d // h).eval()
        m.load_state_dict(mha.state_dict())
        m = m.to(dt)
        q, k = (t.detach().clone().to(dt).requires_grad_(True) for t in (q0, k0))
        v = v0.detach().clone().to(dt).requires_grad_(True) if split_kv else k
        y, attn = m(q, k, v, mask)
        (y * dy0.to(dt)).sum().backward()
        conv = np32 if tag == "f32" else np64
        out[tag + "/out"], out[tag + "/dq"], out[tag + "/dk"], out[tag + "/attn"] = conv(y), conv(q.grad), conv(k.grad), conv(attn)
        if split_kv:
            out[tag + "/dv"] = conv(v.grad)
        out.update(grads_np(m, conv, tag + "/g/"))
    out.update(state_np(mha, np32))
    save(name, **out)


def fx_pffn(name="pffn", d=16, dff=32, shape=(2, 7), seed=7):
    torch.manual_seed(seed)
    ff = S.PositionwiseFeedForward(d, dff).eval()
    rand_weights(ff, seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    x0, dy0 = torch.randn(*shape, d, generator=g), torch.randn(*shape, d, generator=g)
    out = {"x": np32(x0), "dy": np32(dy0), "repair": "none"}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        m = S.PositionwiseFeedForward(d, dff).eval()
        m.load_state_dict(ff.state_dict())
        m = m.to(dt)
        x = x0.detach().clone().to(dt).requires_grad_(True)
        y = m(x)
        (y * dy0.to(dt)).sum().backward()
        conv = np32 if tag == "f32" else np64
        out[tag + "/out"], out[tag + "/dx"] = conv(y), conv(x.grad)
        out.update(grads_np(m, conv, tag + "/g/", compact=name != "pffn"))
    out.update(state_np(ff, np32))
    save(name, **out)


def fx_encoder():
    """Encoder exactly as written (direct pin): 2 layers, d128, h4, d_ff 256, F 80."""
    torch.manual_seed(11)
    enc = M.Encoder(80, 64, n_layers=2, n_head=4, d_k=32, d_v=32, d_model=128, d_inner_hid=256).eval()
    U.init_parameters(enc)
    g = torch.Generator().manual_seed(12)
    lens = torch.tensor([40, 23, 31])
    x0 = torch.randn(3, 40, 80, generator=g) * (torch.arange(40).view(1, -1, 1) < lens.view(-1, 1, 1))
    dy0 = torch.randn(3, 40, 128, generator=g) * (torch.arange(40).view(1, -1, 1) < lens.view(-1, 1, 1))
    out = {"x": np32(x0), "dy": np32(dy0), "in_len": lens.numpy(), "n_head": np.array(4), "repair": "none"}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        m = M.Encoder(80, 64, n_layers=2, n_head=4, d_k=32, d_v=32, d_model=128, d_inner_hid=256).eval()
        m.load_state_dict(enc.state_dict())
        m = m.to(dt)
        x = x0.detach().clone().to(dt).requires_grad_(True)
        y, attns = m(x, lens, return_attns=True)
        (y * dy0.to(dt)).sum().backward()
        conv = np32 if tag == "f32" else np64
        out[tag + "/out"], out[tag + "/dx"] = conv(y), conv(x.grad)
        out.update({tag + "/attn0_" + k: v for k, v in summary(attns[0]).items()})
        out.update(grads_np(m, conv, tag + "/g/", compact=True))
    out.update(state_np(enc, np32))
    save("encoder_2l", **out)


def c1_config(vocab=30):
    return U.AttrDict(dict(feature_dim=80, max_inputs_length=100, max_target_length=20, num_enc_layer=2,
                           num_dec_layer=2, n_heads=4, d_k=32, d_v=32, d_model=128, d_inner_hid=256,
                           dropout=0.1, vocab_size=vocab))


def c1_batch(bsz, seed):
    """BASELINE.md section 3 recipe at the C1 shape: T in [30,60], L in [5,10]."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(bsz, 60, 80, generator=g)
    in_len = torch.randint(30, 61, (bsz,), generator=g)
    in_len[0] = 60
    tgt_len = torch.randint(5, 11, (bsz,), generator=g)
    tgt_len[0] = 10
    tokens = torch.randint(4, 30, (bsz, 10), generator=g)
    x = x * (torch.arange(60).view(1, -1, 1) < in_len.view(-1, 1, 1))
    tokens = tokens * (torch.arange(10).view(1, -1) < tgt_len.view(-1, 1))
    gt = torch.roll(tokens, -1, dims=1)
    gt[:, -1] = 0
    return x, in_len, tokens, tgt_len, gt


def fx_c1_step():
    """Full repaired Transformer, BASELINE config 1: loss + all 90 grads +
    grad-norm + one Noam-Adam step, the train.py:25-46 sequence."""
    cfg = c1_config()
    torch.manual_seed(0)
    ref = M.Transformer(cfg)
    U.init_parameters(ref)
    ref.eval()
    x0, in_len, tokens, tgt_len, gt = c1_batch(4, 0)
    out = {"x": np32(x0), "in_len": in_len.numpy(), "tokens": tokens.numpy(), "tgt_len": tgt_len.numpy(),
           "gt": gt.numpy(), "n_head": np.array(4), "warmup": np.array(100), "step": np.array(1),
           "max_grad_norm": np.array(5.0), "repair": "R1-R4"}
    out["param_order"] = np.array([n for n, _ in ref.named_parameters()])
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        m = M.Transformer(cfg)
        m.load_state_dict(ref.state_dict())
        m = m.to(dt).eval()
        opt = O.ScheduledOptim(m, cfg.d_model, U.AttrDict(dict(n_warmup_steps=100)))
        opt.zero_grad()
        logits, _ = m(x0.to(dt), in_len, tokens, tgt_len)
        loss = nn.CrossEntropyLoss(ignore_index=0)(logits.contiguous().view(-1, cfg.vocab_size), gt.contiguous().view(-1))
        loss.backward()
        conv = np32 if tag == "f32" else np64
        out[tag + "/logits"], out[tag + "/loss"] = conv(logits), conv(loss)
        out.update(grads_np(m, conv, tag + "/g/", compact=True))
        gn = nn.utils.clip_grad_norm_(m.parameters(), 5.0)
        out[tag + "/grad_norm"] = conv(gn)
        opt.step(1)
        out[tag + "/lr"] = np.array(opt.lr)
        for n, p in m.named_parameters():
            for kk, vv in summary(p).items():
                out["%s/after/%s/%s" % (tag, n, kk)] = vv
    out.update(state_np(ref, np32))
    save("transformer_c1_step", **out)
    return ref, cfg


def fx_dp8(ref, cfg):
    """train_multi.py semantics: 8 shards of one utterance each, per-shard
    token-mean CE, gradients averaged over shards (Horovod average)."""
    x0, in_len, tokens, tgt_len, gt = c1_batch(8, 5)
    dt = torch.float64
    m = M.Transformer(cfg)
    m.load_state_dict(ref.state_dict())
    m = m.to(dt).eval()
    acc = {n: torch.zeros_like(p) for n, p in m.named_parameters()}
    losses = []
    for r in range(8):
        m.zero_grad()
        sl = slice(r, r + 1)
        ti, tl = int(in_len[sl].max()), int(tgt_len[sl].max())
        logits, _ = m(x0[sl, :ti].to(dt), in_len[sl], tokens[sl, :tl], tgt_len[sl])
        loss = nn.CrossEntropyLoss(ignore_index=0)(logits.contiguous().view(-1, cfg.vocab_size), gt[sl, :tl].contiguous().view(-1))
        loss.backward()
        losses.append(loss.detach())
        for n, p in m.named_parameters():
            acc[n] += p.grad / 8
    out = {"x": np32(x0), "in_len": in_len.numpy(), "tokens": tokens.numpy(), "tgt_len": tgt_len.numpy(),
           "gt": gt.numpy(), "n_head": np.array(4), "world": np.array(8), "repair": "R1-R4",
           "weights_from": "transformer_c1_step.npz", "f64/mean_loss": np64(torch.stack(losses).mean()),
           "f64/losses": np64(torch.stack(losses))}
    for n, gavg in acc.items():
        for kk, vv in summary(gavg).items():
            out["f64/gavg/%s/%s" % (n, kk)] = vv
    save("dp8_c1", **out)


def fx_ls_loss():
    g = torch.Generator().manual_seed(21)
    logits = torch.randn(12, 30, generator=g)
    target = torch.randint(0, 30, (12,), generator=g)
    target[3] = 0
    target[7] = 0
    out = {"logits": np32(logits), "target": target.numpy(), "repair": "none"}
    for ign in (0, -1, 5):
        crit = LS.LabelSmoothingLoss(0.1, 30, weight=torch.ones(1, 30), ignore_index=ign)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out["loss_ign%d" % ign] = np32(crit(logits, target))
    out["ce_ign0"] = np32(nn.CrossEntropyLoss(ignore_index=0)(logits, target))
    out["noam"] = np.array([[s, w, float(np.power(256, -0.5) * np.min([np.power(s, -0.5), np.power(w, -1.5) * s]))]
                            for s in (1, 10, 4000, 12000, 50000) for w in (100, 12000)])
    save("loss_optim", **out)


if __name__ == "__main__":
    torch.set_num_threads(4)
    fx_pe_masks()
    # direct pins (reference exactly as written; q is k is v)
    mha_case("mha_self_small", 2, 7, 7, 16, 2, [7, 4], [7, 4], False, False, 100)
    mha_case("mha_self_small_causal", 2, 7, 7, 16, 2, [7, 4], [7, 4], True, False, 110)
    mha_case("mha_self_medium", 3, 96, 96, 128, 4, [96, 50, 77], [96, 50, 77], False, False, 120, full=False)
    fx_pffn()
    fx_pffn("pffn_medium", 128, 512, (3, 40), 27)          # a width the HIP path supports (d_model % 64 == 0)
    # production head shape (d_k = 64), +- causal, ragged key lengths
    mha_case("mha_self_causal_c2", 2, 200, 200, 256, 4, [200, 137], [200, 137], True, False, 150, slim=True)
    mha_case("mha_self_causal_dec", 4, 50, 50, 128, 2, [50, 31, 44, 27], [50, 31, 44, 27], True, False, 160, slim=True)
    fx_encoder()
    fx_ls_loss()
    # repaired pins
    install_repairs()
    mha_case("mha_cross_small", 2, 5, 9, 16, 2, [5, 3], [9, 6], False, True, 130, repair="R2")
    mha_case("mha_cross_medium", 2, 20, 150, 128, 4, [20, 11], [150, 97], False, True, 140, full=False, repair="R2")
    mha_case("mha_cross_c2", 2, 50, 400, 128, 2, [50, 33], [400, 273], False, True, 170, repair="R2", slim=True)
    ref, cfg = fx_c1_step()
    fx_dp8(ref, cfg)
    # the general form of MultiHeadAttention.forward (round 5): arbitrary dense masks, keys and values from different tensors
    mha_dense_case("mha_dense_mask", 2, 9, 9, 128, 4, 180, split_kv=False)
    mha_dense_case("mha_dense_mask_kv", 3, 11, 23, 128, 4, 190, split_kv=True)
    remove_repairs()

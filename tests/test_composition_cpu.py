"""Host-logic test (no GPU): the autograd composition in st_amd.functional and the
drop-in module tree, with every C-ABI kernel swapped for its test-only torch
emulation (tests/_emul.py), must reproduce the oracle's C1 training step.

This does NOT claim kernel parity (the -m gpu tests do that on hardware); it pins
the wiring: operand order of every GEMM, residual / LN gradient routing, arena
slots, ragged packing, logits scatter.  Tolerances are the bf16-activation ones."""
import os

import numpy as np
import pytest
import torch

import oracle as orc
from tests._emul import emulated_kernels


# bf16 activations / fp32 accumulate against fp64 truth: SURVEY.md section 8c's tolerances.  (Round 1 needed 1.5e-1
# per tensor for the decoder's q / k projections, whose gradient is a difference of nearly equal terms dP - delta; since
# the attention forward hands the backward O to ~16 bits (Ores) and multiplies P in as hi + lo for the small-Lq
# attentions, delta is consistent with the recomputed P and those tensors sit at 5e-2 - the level the reference itself
# shows under bf16 autocast, profiles/r02_reference_bf16_floor.txt.)
GRAD_TOL_TENSOR, GRAD_TOL_MEDIAN, GRAD_TOL_GLOBAL = 8e-2, 4e-2, 3e-2


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _load_c1(golden_dir):
    fx = dict(np.load(os.path.join(golden_dir, "transformer_c1_step.npz")))
    w = {k[2:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("w/")}
    batch = {k: torch.from_numpy(fx[k]) for k in ("x", "in_len", "tokens", "tgt_len", "gt")}
    return fx, w, batch


def _build(w, n_enc=2, n_dec=2, device="cpu"):
    import transformer.Models as M
    import transformer.Utils as U
    cfg = U.AttrDict(dict(feature_dim=80, max_inputs_length=100, max_target_length=20, num_enc_layer=n_enc,
                          num_dec_layer=n_dec, n_heads=4, d_k=32, d_v=32, d_model=128, d_inner_hid=256,
                          dropout=0.0, vocab_size=30))
    m = M.Transformer(cfg)
    m.load_state_dict(w)
    return m.eval().to(device)


def run_c1_step(golden_dir, device):
    fx, w, batch = _load_c1(golden_dir)
    truth = orc.train_step({k: v.double() for k, v in w.items()}, {k: (v.double() if v.is_floating_point() else v)
                                                                  for k, v in batch.items()}, 4, 128, 100, 1, 5.0)
    if True:
        m = _build(w, device=device)
        logits, _ = m(batch["x"].to(device), batch["in_len"], batch["tokens"].to(device), batch["tgt_len"])
        assert logits.shape == (4, 10, 30)
        loss = torch.nn.CrossEntropyLoss(ignore_index=0)(logits.contiguous().view(-1, 30), batch["gt"].view(-1).to(device))
        loss.backward()
        logits = logits.detach().cpu()
        valid = (torch.arange(10).view(1, -1) < batch["tgt_len"].view(-1, 1))
        assert rel(logits[valid], truth["logits"][valid]) < 2e-2
        assert abs(loss.item() - truth["loss"].item()) < 2e-2 * truth["loss"].item()
        bad, rels, flat_g, flat_t = [], [], [], []
        for n, p in m.named_parameters():
            assert p.grad is not None, n
            g, t = p.grad.detach().cpu(), truth["grads"][n]
            assert torch.isfinite(g).all(), n
            if "linear_k.bias" in n:
                # analytically zero; bf16 rounding of dK leaves noise well below the q-bias gradient scale
                assert g.abs().max().item() < 2.5e-1 * truth["grads"][n.replace("linear_k", "linear_q")].abs().max().item() + 1e-6
                continue
            rels.append(rel(g, t))
            flat_g.append(g.double().reshape(-1))
            flat_t.append(t.double().reshape(-1))
            if rels[-1] > GRAD_TOL_TENSOR:
                bad.append((n, rels[-1]))
        assert not bad, bad
        assert sorted(rels)[len(rels) // 2] < GRAD_TOL_MEDIAN, sorted(rels)[len(rels) // 2]
        assert rel(torch.cat(flat_g), torch.cat(flat_t)) < GRAD_TOL_GLOBAL
        # the vocabulary slot is padded to a multiple of 8 rows in the arena; padding stays zero
        a = m._st_arena
        assert a.grad_view(m.tgt_word_proj.weight, 32)[30:].abs().max().item() == 0


def test_c1_step_composition(golden_dir):
    with emulated_kernels():
        run_c1_step(golden_dir, "cpu")


def run_c1_step_dropout(golden_dir, device):
    """The C1 step in TRAINING mode (dropout 0.2 in the layers, the hard-coded 0.5 of the front-end): the
    product draws its counter-based masks, the fp64 oracle is handed exactly those masks (re-derived from the
    recorded call sites with the bit-identical host implementation in tests/_emul.py) - loss, logits and all
    90 gradients must then agree to the same tolerances as the dropout-free step."""
    from st_amd import rng
    from st_amd.functional import Rows
    from tests import _emul as em
    fx, w, batch = _load_c1(golden_dir)
    import transformer.Models as M
    import transformer.Utils as U
    cfg = U.AttrDict(dict(feature_dim=80, max_inputs_length=100, max_target_length=20, num_enc_layer=2,
                          num_dec_layer=2, n_heads=4, d_k=32, d_v=32, d_model=128, d_inner_hid=256,
                          dropout=0.2, vocab_size=30))
    m = M.Transformer(cfg)
    m.load_state_dict(w)
    m = m.to(device).train()
    rng.seed_tensor(device)
    rng.manual_seed(20260928)        # fixed masks: the tolerances below are statements about ONE draw

    sites, orig_site = [], rng.site

    def recording_site(dev, p):
        d = orig_site(dev, p)
        sites.append(d)
        return d

    rng.site = recording_site
    try:
        logits, _ = m(batch["x"].to(device), batch["in_len"], batch["tokens"].to(device), batch["tgt_len"])
    finally:
        rng.site = orig_site
    loss = torch.nn.CrossEntropyLoss(ignore_index=0)(logits.contiguous().view(-1, 30), batch["gt"].view(-1).to(device))
    loss.backward()
    assert len(sites) == 1 + 2 * 3 + 2 * 4 and all(d is not None for d in sites)
    assert sites[0].thresh == 128 and all(d.thresh == 51 for d in sites[1:])      # p = 0.5 and round(256 * 0.2)

    # ---- the same masks for the oracle ----------------------------------------------------------
    in_len, tgt_len, H = batch["in_len"], batch["tgt_len"], 4
    T, L = int(in_len.max()), int(tgt_len.max())
    rows_of = {T: Rows.packed(in_len, "cpu"), L: Rows.packed(tgt_len, "cpu")}
    queue = [em.Drop(d.seed.detach().cpu(), d.salt, d.thresh / 256.0) for d in sites]

    def provider(site, shape):
        d = queue.pop(0)
        if site == "attn":
            B, h, lq, lk = shape
            keep = torch.stack([torch.stack([em.keep_qk(d, b * H + hh, lq, lk) for hh in range(h)]) for b in range(B)])
            return keep.double() * d.scale
        B, t, n = shape
        r = rows_of[t]
        keep = torch.ones(B, t, n, dtype=torch.float64)
        for b in range(B):
            nb, ob = int(r.lens_host[b]), int(r.off[b])
            keep[b, :nb] = em.keep_rc(d, torch.arange(ob, ob + nb), torch.arange(n), n).double() * d.scale
        return keep

    with orc.dropout_masks(provider):
        truth = orc.train_step({k: v.double() for k, v in w.items()},
                               {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()},
                               4, 128, 100, 1, 5.0)
    assert not queue
    # without the masks the oracle lands somewhere else entirely: the comparison below is not vacuous
    plain = orc.train_step({k: v.double() for k, v in w.items()},
                           {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}, 4, 128, 100, 1, 5.0)
    valid = (torch.arange(L).view(1, -1) < tgt_len.view(-1, 1))
    assert rel(plain["logits"][valid], truth["logits"][valid]) > 1e-1

    logits = logits.detach().cpu()
    assert rel(logits[valid], truth["logits"][valid]) < 2e-2
    assert abs(loss.item() - truth["loss"].item()) < 2e-2 * truth["loss"].item()
    rels, flat_g, flat_t, bad = [], [], [], []
    for n, p in m.named_parameters():
        g, t = p.grad.detach().cpu(), truth["grads"][n]
        assert torch.isfinite(g).all(), n
        if "linear_k.bias" in n:
            continue
        rels.append(rel(g, t))
        flat_g.append(g.double().reshape(-1))
        flat_t.append(t.double().reshape(-1))
        if rels[-1] > 2 * GRAD_TOL_TENSOR:
            bad.append((n, rels[-1]))
    # Training-mode tolerances are twice the eval-mode ones.  A routing mistake (wrong mask, missing 1/(1-p),
    # mask not regenerated identically in the backward) shows up as O(1) - cf. the 0.76 of the mask-free oracle
    # above - whereas what remains here is bf16 noise, uniformly spread over the tensors and larger than in eval
    # mode because dropout thins every reduction (p = 0.5 in front of the first LayerNorm halves its terms and
    # doubles them): measured on this draw 5.1e-2 global / 8e-2 worst tensor with the emulated kernels, 3.4e-2
    # global on MI355X.
    assert not bad, bad
    assert sorted(rels)[len(rels) // 2] < 2 * GRAD_TOL_MEDIAN, sorted(rels)[len(rels) // 2]
    assert rel(torch.cat(flat_g), torch.cat(flat_t)) < 2 * GRAD_TOL_GLOBAL


def test_c1_step_dropout_composition(golden_dir):
    with emulated_kernels():
        run_c1_step_dropout(golden_dir, "cpu")


def run_standalone_modules(golden_dir, device):
    """MultiHeadAttention / PositionwiseFeedForward called through the REFERENCE API (padded tensors + dense mask,
    Attention.py:64 / SubLayers.py:24) against the import-generated goldens at widths the HIP path supports
    (d_model % 64 == 0): self-attention +- causal with ragged keys at the production head shape (d_k = 64, d_model 256
    and 128), decoder-encoder attention with Lq << Lk, and the feed-forward sublayer.  Outputs, input gradients and
    EVERY parameter gradient are compared with the fp64 reference."""
    import transformer.Attention as A
    import transformer.SubLayers as S

    def w_of(fx):
        return {k[2:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("w/")}

    def truth_grad(fx, n):
        key = "f64/g/" + n
        return torch.from_numpy(fx[key]) if key in fx else None

    def check_param_grads(mod, fx, name):
        for n, p in mod.named_parameters():
            if "linear_k.bias" in n:
                continue
            g, t = p.grad.detach().cpu().double(), truth_grad(fx, n)
            if t is not None:
                assert rel(g, t) < GRAD_TOL_TENSOR, (name, n, rel(g, t))
            else:          # big tensors are stored as (sumsq, sampled entries)
                key = "f64/g/" + n
                idx, val = torch.from_numpy(fx[key + "/idx"]), torch.from_numpy(fx[key + "/val"])
                assert rel(g.reshape(-1)[idx], val) < GRAD_TOL_TENSOR, (name, n)
                assert abs(g.norm().item() - float(np.sqrt(fx[key + "/sumsq"]))) < GRAD_TOL_TENSOR * float(np.sqrt(fx[key + "/sumsq"])), (name, n)

    for name in ("mha_self_medium", "mha_self_causal_c2", "mha_self_causal_dec", "mha_cross_medium", "mha_cross_c2"):
        fx = dict(np.load(os.path.join(golden_dir, name + ".npz")))
        d, h, cross = fx["q"].shape[-1], int(fx["n_head"]), "kv" in fx
        assert d % 64 == 0
        tag = "r32" if "r32/out" in fx else "f64"
        mha = A.MultiHeadAttention(h, d, d // h, d // h).eval()
        mha.load_state_dict(w_of(fx))
        mha = mha.to(device)
        q = torch.from_numpy(fx["q"]).to(device).requires_grad_(True)
        kv = torch.from_numpy(fx["kv"]).to(device).requires_grad_(True) if cross else q
        out, attn = mha(q, kv, kv, torch.from_numpy(fx["mask"]).to(device))
        assert attn is None
        (out * torch.from_numpy(fx["dy"]).to(device)).sum().backward()
        # padded query rows carry reference values nobody reads (and the HIP path computes them as well here:
        # the padded layout keeps every row) - compare everything
        assert rel(out.detach().cpu(), torch.from_numpy(fx[tag + "/out"])) < 2e-2, name
        assert rel(q.grad.cpu(), torch.from_numpy(fx[tag + "/dq"])) < 6e-2, (name, rel(q.grad.cpu(), torch.from_numpy(fx[tag + "/dq"])))
        if cross:
            assert rel(kv.grad.cpu(), torch.from_numpy(fx[tag + "/dkv"])) < 6e-2, name
        check_param_grads(mha, fx, name)

    fx = dict(np.load(os.path.join(golden_dir, "pffn_medium.npz")))
    ff = S.PositionwiseFeedForward(128, 512).eval()
    ff.load_state_dict(w_of(fx))
    ff = ff.to(device)
    x = torch.from_numpy(fx["x"]).to(device).requires_grad_(True)
    y = ff(x)
    (y * torch.from_numpy(fx["dy"]).to(device)).sum().backward()
    assert rel(y.detach().cpu(), torch.from_numpy(fx["f64/out"])) < 2e-2
    assert rel(x.grad.cpu(), torch.from_numpy(fx["f64/dx"])) < 6e-2
    check_param_grads(ff, fx, "pffn_medium")


def run_general_attention(golden_dir, device):
    """MultiHeadAttention.forward in its general form (Attention.py:64-96): an arbitrary dense mask, k is not v - the slow dense
    path (st_attn_dense_fwd / _bwd behind functional.DenseMhaFn) against fixtures generated by importing the reference: output,
    the returned probabilities, the input gradients of q / k / v and every parameter gradient."""
    import transformer.Attention as A

    def w_of(fx):
        return {k[2:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("w/")}

    def check_param_grads(mod, fx, name):
        for n, p in mod.named_parameters():
            g, t = p.grad.detach().cpu().double(), torch.from_numpy(fx["f64/g/" + n])
            if "linear_k.bias" in n:          # analytically zero (softmax is shift-invariant): absolute bound
                assert g.abs().max().item() < 0.25 * mod.linear_q.bias.grad.abs().max().item() + 1e-6, (name, n)
                continue
            assert rel(g, t) < GRAD_TOL_TENSOR, (name, n, rel(g, t))

    for name in ("mha_dense_mask", "mha_dense_mask_kv"):
        fx = dict(np.load(os.path.join(golden_dir, name + ".npz")))
        d, h = fx["q"].shape[-1], int(fx["n_head"])
        mha = A.MultiHeadAttention(h, d, d // h, d // h).eval()
        mha.load_state_dict(w_of(fx))
        mha = mha.to(device)
        mha.return_attn = True
        q, k = (torch.from_numpy(fx[n]).to(device).requires_grad_(True) for n in ("q", "k"))
        v = torch.from_numpy(fx["v"]).to(device).requires_grad_(True) if "v" in fx else k
        out, attn = mha(q, k, v, torch.from_numpy(fx["mask"]).to(device))
        (out * torch.from_numpy(fx["dy"]).to(device)).sum().backward()
        assert rel(out.detach().cpu(), torch.from_numpy(fx["f64/out"])) < 2e-2, name
        assert rel(attn.detach().cpu(), torch.from_numpy(fx["f64/attn"])) < 2e-2, name
        assert rel(q.grad.cpu(), torch.from_numpy(fx["f64/dq"])) < 6e-2, (name, rel(q.grad.cpu(), torch.from_numpy(fx["f64/dq"])))
        assert rel(k.grad.cpu(), torch.from_numpy(fx["f64/dk"])) < 6e-2, (name, rel(k.grad.cpu(), torch.from_numpy(fx["f64/dk"])))
        if "v" in fx:
            assert rel(v.grad.cpu(), torch.from_numpy(fx["f64/dv"])) < 6e-2, name
        check_param_grads(mha, fx, name)
    # a row whose keys are ALL masked: zero context (the reference: NaN), finite everywhere, no gradient through that row's scores
    mha = A.MultiHeadAttention(4, 128, 32, 32).eval().to(device)
    x = torch.randn(2, 5, 128, device=device, requires_grad=True)
    mask = torch.zeros(2, 5, 5, dtype=torch.bool, device=device)
    mask[1, 2] = True
    mask[0, 0, 3] = True                      # (and a mask outside the two families, so that the dense path is taken)
    y, _ = mha(x, x, x, mask)
    y.sum().backward()
    assert torch.isfinite(y).all() and torch.isfinite(x.grad).all()
    # masks that BROADCAST to [B, Lq, Lk], as the reference's masked_fill_ takes them: a [B, 1, Lk] key-padding mask
    # (one of the two families -> the fast path) and a [Lq, Lk] foreign mask (-> the dense path) equal their expanded forms
    with torch.no_grad():
        kp = torch.zeros(2, 1, 5, dtype=torch.bool, device=device)
        kp[1, 0, 3:] = True
        ya, _ = mha(x, x, x, kp)
        yb, _ = mha(x, x, x, kp.expand(2, 5, 5).contiguous())
        assert torch.equal(ya, yb)
        fm = torch.zeros(5, 5, dtype=torch.bool, device=device)
        fm[0, 2] = fm[3, 1] = True
        ya, _ = mha(x, x, x, fm)
        yb, _ = mha(x, x, x, fm.unsqueeze(0).expand(2, 5, 5).contiguous())
        assert torch.equal(ya, yb)


def test_general_attention_composition(golden_dir):
    with emulated_kernels():
        run_general_attention(golden_dir, "cpu")


def test_standalone_modules_composition(golden_dir):
    with emulated_kernels():
        run_standalone_modules(golden_dir, "cpu")


def run_encoder_padded_api(golden_dir, device):
    import transformer.Models as M
    fx = dict(np.load(os.path.join(golden_dir, "encoder_2l.npz")))
    if True:
        enc = M.Encoder(80, 64, n_layers=2, n_head=4, d_k=32, d_v=32, d_model=128, d_inner_hid=256, dropout=0.0).eval()
        enc.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("w/")})
        enc = enc.to(device)
        x = torch.from_numpy(fx["x"]).to(device).requires_grad_(True)
        lens = torch.from_numpy(fx["in_len"])
        y, attns = enc(x, lens)
        assert attns == [] and y.shape == (3, 40, 128)
        (y * torch.from_numpy(fx["dy"]).to(device)).sum().backward()
        y, dx = y.detach().cpu(), x.grad.cpu()
        valid = torch.arange(40).view(1, -1) < lens.view(-1, 1)
        assert rel(y[valid], torch.from_numpy(fx["f64/out"])[valid]) < 2e-2
        assert y[~valid].abs().max().item() == 0          # padded frames come back as zeros
        assert rel(dx[valid], torch.from_numpy(fx["f64/dx"])[valid]) < 8e-2


def test_encoder_padded_api_composition(golden_dir):
    with emulated_kernels():
        run_encoder_padded_api(golden_dir, "cpu")


def run_return_attns(golden_dir, device):
    """return_attns (reference Models.py:53-54,107-109,147-153; Attention.py:96): the maps st_attn_probs materialises.
    (1) Encoder.forward(return_attns=True) against the reference's own first-layer map of fixture encoder_2l (sampled
    entries + sum, fp64); (2) MultiHeadAttention.forward with return_attn against the medium fixtures' maps; (3) the
    Transformer with config.return_attns: one map per layer and attention, rows sum to one over the visible keys, zeros at
    masked keys and in padding rows, causal structure in the decoder's self-attention."""
    import transformer.Attention as A
    import transformer.Models as M
    import transformer.Utils as U
    fx = dict(np.load(os.path.join(golden_dir, "encoder_2l.npz")))
    enc = M.Encoder(80, 64, n_layers=2, n_head=4, d_k=32, d_v=32, d_model=128, d_inner_hid=256, dropout=0.0).eval()
    enc.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("w/")})
    enc = enc.to(device)
    lens = torch.from_numpy(fx["in_len"])
    with torch.no_grad():
        y, attns = enc(torch.from_numpy(fx["x"]).to(device), lens, return_attns=True)
        y0, none = enc(torch.from_numpy(fx["x"]).to(device), lens)
    assert none == [] and torch.equal(y, y0)                      # asking for the maps changes nothing else
    assert len(attns) == 2 and tuple(attns[0].shape) == tuple(int(v) for v in fx["f64/attn0_shape"])
    a0 = attns[0].double().cpu()
    # the reference computes a (meaningless) distribution for padding FRAMES too; the product leaves those rows zero
    B, H, T, _ = a0.shape
    rows_valid = (torch.arange(T).view(1, 1, T, 1) < lens.view(B, 1, 1, 1)).expand(B, H, T, T)
    idx = torch.from_numpy(fx["f64/attn0_idx"])
    keep = rows_valid.reshape(-1)[idx]
    assert keep.sum() > 50
    assert rel(a0.reshape(-1)[idx][keep], torch.from_numpy(fx["f64/attn0_val"])[keep]) < 2e-2
    assert abs(a0.sum().item() - float(lens.sum()) * H) < 1e-3 * float(lens.sum()) * H        # every valid row sums to one
    assert a0[~rows_valid].abs().max().item() == 0
    for name in ("mha_self_medium", "mha_cross_medium", "mha_self_causal_c2"):
        fm = dict(np.load(os.path.join(golden_dir, name + ".npz")))
        d, h, cross = fm["q"].shape[-1], int(fm["n_head"]), "kv" in fm
        mha = A.MultiHeadAttention(h, d, d // h, d // h).eval()
        mha.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in fm.items() if k.startswith("w/")})
        mha = mha.to(device)
        mha.return_attn = True
        q = torch.from_numpy(fm["q"]).to(device)
        kv = torch.from_numpy(fm["kv"]).to(device) if cross else q
        with torch.no_grad():
            _, attn = mha(q, kv, kv, torch.from_numpy(fm["mask"]).to(device))
        if "f64/attn_idx" in fm:
            assert tuple(attn.shape) == tuple(int(v) for v in fm["f64/attn_shape"]), name
            flat = attn.double().cpu().reshape(-1)
            assert rel(flat[torch.from_numpy(fm["f64/attn_idx"])], torch.from_numpy(fm["f64/attn_val"])) < 2e-2, name
            assert abs(flat.sum().item() - float(fm["f64/attn_sum"])) < 2e-3 * abs(float(fm["f64/attn_sum"])), name
        else:             # (the fixture carries no map: structure only - rows are distributions over the visible keys)
            a = attn.double().cpu()
            assert (a.sum(-1) - 1).abs().max().item() < 1e-4 and torch.triu(a, 1).abs().max().item() == 0, name
    # the whole model
    fx1, w, batch = _load_c1(golden_dir)
    cfg = U.AttrDict(dict(feature_dim=80, max_inputs_length=100, max_target_length=20, num_enc_layer=2, num_dec_layer=2, n_heads=4,
                          d_k=32, d_v=32, d_model=128, d_inner_hid=256, dropout=0.0, vocab_size=30, return_attns=True))
    m = M.Transformer(cfg)
    m.load_state_dict(w)
    m = m.eval().to(device)
    with torch.no_grad():
        logits, (ea, sa, ca) = m(batch["x"].to(device), batch["in_len"], batch["tokens"].to(device), batch["tgt_len"])
        m.return_attns = None
        logits0, empty = m(batch["x"].to(device), batch["in_len"], batch["tokens"].to(device), batch["tgt_len"])
    assert empty == ([], [], []) and torch.equal(logits, logits0)
    Bm, T, L = batch["x"].shape[0], batch["x"].shape[1], batch["tokens"].shape[1]
    assert [tuple(a.shape) for a in ea] == [(Bm, 4, T, T)] * 2 and [tuple(a.shape) for a in sa] == [(Bm, 4, L, L)] * 2
    assert [tuple(a.shape) for a in ca] == [(Bm, 4, L, T)] * 2
    il, tl = batch["in_len"], batch["tgt_len"]
    for maps, ql, kl in ((ea, il, il), (sa, tl, tl), (ca, tl, il)):
        for a in maps:
            a = a.double().cpu()
            for b in range(Bm):
                nq, nk = int(ql[b]), int(kl[b])
                assert (a[b, :, :nq].sum(-1) - 1).abs().max().item() < 1e-4
                assert a[b, :, nq:].abs().max().item() == 0 if nq < a.shape[2] else True
                assert a[b, :, :, nk:].abs().max().item() == 0 if nk < a.shape[3] else True
    for a in sa:          # feature_info_mask: no probability above the diagonal
        assert torch.triu(a.double().cpu(), 1).abs().max().item() == 0
    # the production width (d_model 256: the layer stacks run as row chains, and under no_grad the sublayer Functions
    # that feed the tap are never replayed - ADVICE r4): the maps must come out, and equal the grad-enabled call's
    import oracle as orc
    cfg2 = U.AttrDict(dict(feature_dim=80, max_inputs_length=100, max_target_length=20, num_enc_layer=2, num_dec_layer=2, n_heads=4,
                           d_k=64, d_v=64, d_model=256, d_inner_hid=512, dropout=0.0, vocab_size=30, return_attns=True))
    m2 = M.Transformer(cfg2).eval().to(device)
    b2 = orc.synthetic_batch(3, 70, 9, 80, 30, seed=4, t_min=30, l_min=4)
    args = (b2["x"].to(device), b2["in_len"], b2["tokens"].to(device), b2["tgt_len"])
    with torch.no_grad():
        lg_n, maps_n = m2(*args)
    lg_g, maps_g = m2(*args)
    assert m2.encoder._st_chains[1] is not None and m2.decoder._st_chains[1] is not None, "row chains were not taken"
    # (not bit-equal: when a backward will follow, the decoder's attentions multiply P in as hi + lo bf16 parts - DESIGN section 3)
    assert rel(lg_n, lg_g.detach()) < 1e-2
    for fam_n, fam_g in zip(maps_n, maps_g):
        assert len(fam_n) == len(fam_g) == 2
        for a, g in zip(fam_n, fam_g):
            assert a.shape == g.shape and (a - g).abs().max().item() < 5e-3 and abs(a.sum().item() - g.sum().item()) < 1e-2
    with torch.no_grad():      # ... and through the stand-alone Encoder / Decoder entry points
        y, ea2 = m2.encoder(args[0], args[1], return_attns=True)
        _, sa2, ca2 = m2.decoder(args[2], args[3], args[1], y, return_attns=True)
    assert len(ea2) == len(sa2) == len(ca2) == 2 and all((a - g).abs().max().item() < 5e-3 for a, g in zip(ea2, maps_g[0]))


def test_return_attns_composition(golden_dir):
    with emulated_kernels():
        run_return_attns(golden_dir, "cpu")


def run_wide_step(device, d_model=512, n_head=8, d_ff=1024, n_enc=2, n_dec=1):
    """The BASELINE config 3 layer shape (d_model 512, 8 heads, d_k 64) on a small batch: loss, logits and every
    gradient against the fp64 oracle.  Exercises the N = 512 LayerNorm-GEMM geometry and 8-head attention."""
    import transformer.Models as M
    import transformer.Utils as U
    p = orc.xavier_init_(orc.make_params(80, 30, d_model, d_ff, n_enc, n_dec, 100, 20, dtype=torch.float64), seed=3)
    batch = orc.synthetic_batch(3, 70, 9, 80, 30, seed=4, t_min=30, l_min=4)
    truth = orc.train_step(p, {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}, n_head,
                           d_model, 100, 1, 5.0)
    cfg = U.AttrDict(dict(feature_dim=80, max_inputs_length=100, max_target_length=20, num_enc_layer=n_enc,
                          num_dec_layer=n_dec, n_heads=n_head, d_k=d_model // n_head, d_v=d_model // n_head,
                          d_model=d_model, d_inner_hid=d_ff, dropout=0.0, vocab_size=30))
    m = M.Transformer(cfg)
    m.load_state_dict({k: v.float() for k, v in p.items()})
    m = m.eval().to(device)
    L = int(batch["tgt_len"].max())
    logits, _ = m(batch["x"].to(device), batch["in_len"], batch["tokens"][:, :L].to(device), batch["tgt_len"])
    loss = torch.nn.CrossEntropyLoss(ignore_index=0)(logits.contiguous().view(-1, 30), batch["gt"][:, :L].reshape(-1).to(device))
    loss.backward()
    valid = (torch.arange(L).view(1, -1) < batch["tgt_len"].view(-1, 1))
    assert rel(logits.detach().cpu()[valid], truth["logits"][valid]) < 2e-2
    assert abs(loss.item() - truth["loss"].item()) < 2e-2 * truth["loss"].item()
    rels, flat_g, flat_t = [], [], []
    for n, q in m.named_parameters():
        g, t = q.grad.detach().cpu(), truth["grads"][n]
        assert torch.isfinite(g).all(), n
        if "linear_k.bias" in n:
            continue
        rels.append(rel(g, t))
        flat_g.append(g.double().reshape(-1))
        flat_t.append(t.double().reshape(-1))
    assert max(rels) < GRAD_TOL_TENSOR, max(rels)
    assert sorted(rels)[len(rels) // 2] < GRAD_TOL_MEDIAN
    assert rel(torch.cat(flat_g), torch.cat(flat_t)) < GRAD_TOL_GLOBAL


def test_wide_step_composition():
    with emulated_kernels():
        run_wide_step("cpu")


def test_shipped_config_head_width_composition():
    """d_model 512 with FOUR heads = d_k 128: the reference's shipped config/character.yaml:28-31."""
    with emulated_kernels():
        run_wide_step("cpu", n_head=4)


def run_joint_ctc_step(device):
    """BASELINE config 4: joint lambda * CTC + (1 - lambda) * attention loss (transformer/Loss.py:CTCAttentionLoss
    on Transformer.forward_joint).  The encoder gets gradient from BOTH branches; everything is compared with the
    same objective evaluated on the fp64 oracle's encoder / decoder."""
    import torch.nn.functional as func
    import transformer.Models as M
    import transformer.Utils as U
    from transformer.Loss import CTCAttentionLoss
    d_model, n_head = 128, 4
    p = orc.xavier_init_(orc.make_params(80, 30, d_model, 256, 2, 2, 100, 20, dtype=torch.float64), seed=5)
    batch = orc.synthetic_batch(3, 70, 9, 80, 30, seed=6, t_min=40, l_min=4)
    x, in_len, tokens, tgt_len, gt = (batch[k] for k in ("x", "in_len", "tokens", "tgt_len", "gt"))
    L = int(tgt_len.max())
    tokens, gt = tokens[:, :L], gt[:, :L]
    # The CTC projection's nn.Linear init draws from the GLOBAL generator, whose seed differs from process to process:
    # unseeded, the worst tensor of this test moved between 0.050 and 0.085 with the drawn head (the product itself is
    # run-to-run and process-to-process identical to 6e-8, tools/dev/determinism_check.py).
    torch.manual_seed(0)
    head = CTCAttentionLoss(d_model, 30, ctc_weight=0.3)
    # ---- fp64 truth
    leaves = {k: (v.clone().requires_grad_(True) if not k.endswith(".pe") else v) for k, v in p.items()}
    w64 = head.ctc_proj.weight.detach().double().clone().requires_grad_(True)
    b64 = head.ctc_proj.bias.detach().double().clone().requires_grad_(True)
    enc, _ = orc.encoder(leaves, x.double(), in_len, n_head)
    dec, _, _ = orc.decoder(leaves, tokens, tgt_len, in_len, enc, n_head)
    logits = func.linear(dec, leaves["tgt_word_proj.weight"])
    att = orc.cross_entropy(logits, gt)
    logp = func.log_softmax(func.linear(enc, w64, b64), -1).transpose(0, 1)
    ctc = func.ctc_loss(logp, gt, in_len, tgt_len, blank=0, reduction="mean", zero_infinity=True)
    truth = 0.3 * ctc + 0.7 * att
    names = [k for k in leaves if not k.endswith(".pe")]
    all_g = torch.autograd.grad(truth, [leaves[k] for k in names] + [w64, b64], allow_unused=True)
    tg = dict(zip(names, all_g))
    g_w64, g_b64 = all_g[-2], all_g[-1]
    # ---- product
    cfg = U.AttrDict(dict(feature_dim=80, max_inputs_length=100, max_target_length=20, num_enc_layer=2,
                          num_dec_layer=2, n_heads=n_head, d_k=32, d_v=32, d_model=d_model, d_inner_hid=256,
                          dropout=0.0, vocab_size=30))
    m = M.Transformer(cfg)
    m.load_state_dict({k: v.float() for k, v in p.items()})
    m, head = m.eval().to(device), head.to(device)
    logits_h, enc_h = m.forward_joint(x.to(device), in_len, tokens.to(device), tgt_len)
    assert enc_h.shape == (3, int(in_len.max()), d_model) and logits_h.shape == (3, L, 30)
    loss, att_h, ctc_h = head(enc_h, in_len, logits_h, gt.to(device), tgt_len, gt.to(device))
    loss.backward()
    assert abs(loss.item() - truth.item()) < 2e-2 * abs(truth.item()), (loss.item(), truth.item())
    assert abs(ctc_h.item() - ctc.item()) < 2e-2 * abs(ctc.item())
    rels = []
    for n, q in m.named_parameters():
        if "linear_k.bias" in n or tg[n] is None:
            continue
        assert q.grad is not None and torch.isfinite(q.grad).all(), n
        rels.append((rel(q.grad.detach().cpu(), tg[n]), n))
    assert max(rels)[0] < GRAD_TOL_TENSOR, max(rels)
    assert sorted(rels)[len(rels) // 2][0] < GRAD_TOL_MEDIAN
    assert rel(head.ctc_proj.weight.grad.cpu(), g_w64) < GRAD_TOL_MEDIAN       # the CTC head sees the bf16 encoder output
    assert rel(head.ctc_proj.bias.grad.cpu(), g_b64) < GRAD_TOL_MEDIAN


def test_joint_ctc_step_composition():
    with emulated_kernels():
        run_joint_ctc_step("cpu")


def run_trainstep_vs_oracle(golden_dir, device, use_graph=False):
    """The TIMED object (st_amd.trainer.TrainStep: zero_grad, forward, CE on the padded-logits buffer, backward with
    grouped weight gradients, global norm, st_adam_clip) against fixture F7 = the repaired reference's one train.py
    step (train.py:25-46, Optim.py:11-16,36-45): loss, pre-clip gradient norm, the Noam rate, and the post-step
    weights.  Adam's first update is -lr * g / (|g| + eps): a sign function, so the post-step weights are compared
    (i) tightly against the oracle's Adam applied to the PRODUCT's gradients (pins clip + Adam arithmetic) and
    (ii) against F7 itself through the normalised update u = (p_new - p_old) / lr in [-1, 1], whose mean absolute
    deviation from the reference's is bounded (elements whose gradient is below the bf16 noise flip sign)."""
    import transformer.Utils as U
    from st_amd.arena import arena_of
    from st_amd.trainer import TrainStep
    from transformer.Optim import ScheduledOptim
    fx, w, batch = _load_c1(golden_dir)
    m = _build(w, device=device)
    opt = ScheduledOptim(m, 128, U.AttrDict(n_warmup_steps=int(fx["warmup"])))
    assert opt.arena is not None
    step = TrainStep(m, opt, 30, float(fx["max_grad_norm"]), use_graph=use_graph, graph_warmup=0)
    x, tok, gt = batch["x"].to(device), batch["tokens"].to(device), batch["gt"].to(device)
    if use_graph:     # lazy set-up (arena, ragged layouts, work lists) must not happen inside the capture
        with torch.no_grad():
            m.forward_packed(x, batch["in_len"], tok[:, :10], batch["tgt_len"])
    before = {n: p.detach().clone().cpu().double() for n, p in m.named_parameters()}
    loss, gnorm = step(x, batch["in_len"], tok, batch["tgt_len"], gt)
    assert step.global_step == int(fx["step"])
    lr = float(fx["f64/lr"])
    assert abs(opt.lr - lr) <= 1e-12 and abs(float(opt.lr_tensor) - lr) <= 1e-6 * lr
    assert abs(float(loss) - float(fx["f64/loss"])) <= 2e-2 * float(fx["f64/loss"])
    assert abs(float(gnorm) - float(fx["f64/grad_norm"])) <= 2e-2 * float(fx["f64/grad_norm"]), (float(gnorm), float(fx["f64/grad_norm"]))
    # (i) clip + Adam arithmetic on the product's own (now clipped, in place) gradients
    coef = min(1.0, float(fx["max_grad_norm"]) / (float(gnorm) + 1e-6))
    dev_sum, n_el = 0.0, 0
    arena = arena_of(m)
    for n, p in m.named_parameters():
        g = arena.grad_view(p).detach().cpu().double()        # already multiplied by coef
        mm, vv = 0.1 * g, 0.02 * g * g
        want = before[n] - lr * (mm / 0.1) / ((vv.sqrt() / (0.02 ** 0.5)) + 1e-9)
        got = p.detach().cpu().double()
        assert (got - want).abs().max().item() <= 2e-3 * lr + 1e-7 * want.abs().max().item(), n
        # (ii) against the reference's post-step samples
        if "linear_k.bias" in n:
            continue          # analytically zero gradient: the update is +-lr of rounding noise on both sides
        key = "f64/after/" + n
        idx = torch.from_numpy(fx[key + "/idx"])
        u = (got.reshape(-1)[idx] - before[n].reshape(-1)[idx]) / lr
        ut = (torch.from_numpy(fx[key + "/val"]) - before[n].reshape(-1)[idx]) / lr
        assert u.abs().max().item() <= 1.0 + 1e-3
        dev_sum += (u - ut).abs().sum().item()
        n_el += idx.numel()
    assert coef < 1.0      # the fixture's norm (11.8) exceeds max_grad_norm (5): the clip path is exercised
    # measured 0.016 with the emulated kernels (a sign flip costs 2): < 3 % of the sampled weights move the other way
    assert dev_sum / n_el < 0.06, dev_sum / n_el
    return float(loss), float(gnorm)


def test_trainstep_vs_oracle_composition(golden_dir):
    with emulated_kernels():
        run_trainstep_vs_oracle(golden_dir, "cpu")


def run_dp_shards_vs_golden(golden_dir, device):
    """Fixture F8 (tests/golden/dp8_c1.npz: the repaired reference on 8 one-utterance shards, per-shard token-mean
    loss, gradients averaged over the shards - train_multi.py:136-139,161-163) against the PRODUCT path run shard by
    shard into the flat gradient arena and averaged - the data a GradReducer's all-reduce produces."""
    from st_amd.arena import arena_of
    from tests.test_oracle_golden import load
    fx = load(golden_dir, "dp8_c1")
    _, w, _ = _load_c1(golden_dir)
    m = _build(w, device=device)
    arena = arena_of(m)
    world = int(fx["world"])
    x, in_len, tokens = torch.from_numpy(fx["x"]), torch.from_numpy(fx["in_len"]), torch.from_numpy(fx["tokens"])
    tgt_len, gt = torch.from_numpy(fx["tgt_len"]), torch.from_numpy(fx["gt"])
    crit = torch.nn.CrossEntropyLoss(ignore_index=0)
    avg, losses = torch.zeros_like(arena.grad), []
    for r in range(world):
        sl = slice(r * (x.shape[0] // world), (r + 1) * (x.shape[0] // world))
        T, L = int(in_len[sl].max()), int(tgt_len[sl].max())
        arena.zero_grads()
        logits, _ = m(x[sl, :T].to(device), in_len[sl], tokens[sl, :L].to(device), tgt_len[sl])
        loss = crit(logits.contiguous().view(-1, 30), gt[sl, :L].reshape(-1).to(device))
        loss.backward()
        losses.append(float(loss.detach()))
        avg += arena.grad / world
    assert abs(sum(losses) / world - float(fx["f64/mean_loss"])) <= 2e-2 * float(fx["f64/mean_loss"])
    # per-tensor: norm and the stored samples of the averaged gradient
    num = den = 0.0
    for n, p in m.named_parameters():
        if "linear_k.bias" in n:
            continue
        key = "f64/gavg/" + n
        off = arena.offset[id(p)]
        g = avg[off:off + p.numel()].detach().cpu().double()
        t_norm = float(np.sqrt(fx[key + "/sumsq"]))
        assert abs(g.norm().item() - t_norm) <= 6e-2 * t_norm, n
        idx = torch.from_numpy(fx[key + "/idx"])
        val = torch.from_numpy(fx[key + "/val"])
        num += ((g[idx] - val) ** 2).sum().item()
        den += (val ** 2).sum().item()
    assert (num / den) ** 0.5 < GRAD_TOL_GLOBAL * 1.5, (num / den) ** 0.5


def test_dp_shards_vs_golden_composition(golden_dir):
    with emulated_kernels():
        run_dp_shards_vs_golden(golden_dir, "cpu")


def run_row_chain_step(device):
    """d_model 256 (the width the row-chain kernel serves): the decoder's layer stack runs as attention kernels + two
    row chains per layer (st_amd/chains.py); loss, logits and every gradient against the fp64 oracle."""
    import transformer.Models as M
    made = []
    plan = M.DecoderChains.plan
    M.DecoderChains.plan = staticmethod(lambda layers, arena: made.append(plan(layers, arena)) or made[-1])
    try:
        run_wide_step(device, d_model=256, n_head=4, d_ff=512, n_enc=1, n_dec=2)
    finally:
        M.DecoderChains.plan = plan
    assert made and made[0] is not None and len(made[0].f1) == 2, "the decoder did not take the row-chain path"


def test_row_chain_step_composition():
    with emulated_kernels():
        run_row_chain_step("cpu")


def run_row_chains_on_off(device, exact):
    """The same model and batch with the decoder's row chains on and off (every GEMM its own launch): loss and gradients
    must agree - bit for bit under the emulation (it composes the same emulated kernels, so this checks the host logic:
    dropout sites, saved tensors, autograd replay), to rounding on the GPU - in eval mode and in training mode (same
    seed -> same masks)."""
    import transformer.Models as M
    import transformer.Utils as U
    from st_amd import functional as F_, rng
    from st_amd.arena import arena_of
    torch.manual_seed(3)
    cfg = U.AttrDict(dict(feature_dim=80, max_inputs_length=100, max_target_length=20, num_enc_layer=1, num_dec_layer=3,
                          n_heads=4, d_k=64, d_v=64, d_model=256, d_inner_hid=512, dropout=0.1, vocab_size=30))
    m = M.Transformer(cfg)
    U.init_parameters(m)
    m = m.to(device)
    rng.seed_tensor(device)
    batch = orc.synthetic_batch(4, 70, 9, 80, 30, seed=4, t_min=30, l_min=4)
    x, tok, gt = batch["x"].to(device), batch["tokens"].to(device), batch["gt"].to(device)
    crit = torch.nn.CrossEntropyLoss(ignore_index=0)

    def run(chains_on):
        m.decoder.use_row_chains = m.encoder.use_row_chains = chains_on
        arena = arena_of(m)
        arena.zero_grads()
        rng.manual_seed(5)
        logits, t_rows = m.forward_packed(x, batch["in_len"], tok, batch["tgt_len"])
        truth = gt.contiguous().view(-1).index_select(0, t_rows.scatter_index(gt.shape[1]))
        loss = crit(logits, truth)
        with F_.deferred_wgrads(True):
            loss.backward()
        return loss.item(), logits.detach().clone(), arena.grad.detach().clone()

    from st_amd.chains import EncoderChains
    try:
        for training in (False, True):
            m.train(training)
            (l1, lg1, g1), (l0, lg0, g0) = run(True), run(False)
            assert m.decoder._st_chains[1] is not None and m.encoder._st_chains[1] is not None
            if exact:
                # the encoder chains hand the attention kernels PRE-SCALED keys (one more fp32 multiply in front of the keys'
                # one rounding: another rounding realisation than the per-GEMM path) - switched off, the two paths must agree
                # bit for bit; switched on (the product), to rounding
                assert abs(l1 - l0) <= 2e-3 * abs(l0) and rel(lg1, lg0) < 2e-2 and rel(g1, g0) < GRAD_TOL_TENSOR, (training, l1, l0)
                EncoderChains.PRESCALE_KEYS = F_.MhaFn.PRESCALE_KEYS = False
                try:
                    (l1, lg1, g1), (l0, lg0, g0) = run(True), run(False)
                finally:
                    EncoderChains.PRESCALE_KEYS = F_.MhaFn.PRESCALE_KEYS = True
                assert l1 == l0 and torch.equal(lg1, lg0) and torch.equal(g1, g0), (training, l1, l0)
            else:
                # two bf16 pipelines with different accumulation orders: rounding flips are amplified layer by layer on
                # this random-weight model (measured 6e-3 on the logits, 4e-2 on the gradients); each path is held to the
                # fp64 oracle separately (run_row_chain_step, run_c1_step ...), the kernel to the separate kernels tightly
                # (tests/test_kernels_gpu.py::test_row_chain_matches_the_separate_kernels)
                assert abs(l1 - l0) <= 1e-3 * abs(l0), (training, l1, l0)
                assert rel(lg1, lg0) < 2e-2 and rel(g1, g0) < GRAD_TOL_TENSOR, (training, rel(lg1, lg0), rel(g1, g0))
    finally:
        m.decoder.use_row_chains = m.encoder.use_row_chains = True


def test_row_chains_on_off_composition():
    with emulated_kernels():
        run_row_chains_on_off("cpu", exact=True)


def run_no_activation_leak(device):
    """with row chains on, every grad-enabled forward used to leak the layer stacks' activations
    (SubPre.out carried the grad_fn, closing a reference cycle through C++ that gc cannot traverse).  The number of live
    tensors must be flat over eager steps - with a backward, and with a forward that is never followed by one."""
    import gc
    import transformer.Models as M
    import transformer.Utils as U
    from st_amd import functional as F_
    from st_amd.arena import arena_of
    torch.manual_seed(3)
    cfg = U.AttrDict(dict(feature_dim=80, max_inputs_length=100, max_target_length=20, num_enc_layer=2, num_dec_layer=2,
                          n_heads=4, d_k=64, d_v=64, d_model=256, d_inner_hid=512, dropout=0.0, vocab_size=30))
    m = M.Transformer(cfg)
    U.init_parameters(m)
    m = m.to(device).eval()
    batch = orc.synthetic_batch(3, 40, 7, 80, 30, seed=4, t_min=20, l_min=4)
    x, tok, gt = batch["x"].to(device), batch["tokens"].to(device), batch["gt"].to(device)
    crit = torch.nn.CrossEntropyLoss(ignore_index=0)

    def live():
        gc.collect()
        n = b = 0
        for o in gc.get_objects():
            try:
                if torch.is_tensor(o) and o.device.type == torch.device(device).type:
                    n += 1
                    b += o.numel() * o.element_size()
            except Exception:
                pass
        return n, b

    def step(backward):
        arena_of(m).zero_grads()
        logits, t_rows = m.forward_packed(x, batch["in_len"], tok, batch["tgt_len"])
        loss = crit(logits, gt.contiguous().view(-1).index_select(0, t_rows.scatter_index(gt.shape[1])))
        if backward:
            with F_.deferred_wgrads(True):
                loss.backward()

    for backward in (True, False):
        step(backward)
        step(backward)
        n0, b0 = live()
        for _ in range(4):
            step(backward)
        n1, b1 = live()
        assert m.decoder._st_chains[1] is not None and m.encoder._st_chains[1] is not None
        assert n1 <= n0 and b1 <= b0, ("activations leak across eager steps", backward, n0, n1, b0, b1)


def test_no_activation_leak_composition():
    with emulated_kernels():
        run_no_activation_leak("cpu")


def run_row_chain_step_narrow_heads(device):
    """d_model 256 with 8 heads (d_k 32): the forward runs as row chains, the backward falls back to the per-GEMM path with the
    LayerNorm hand-over links (the backward chains' delta epilogue is written for 64-wide heads) - against the fp64 oracle."""
    run_wide_step(device, d_model=256, n_head=8, d_ff=512, n_enc=1, n_dec=2)


def test_row_chain_step_narrow_heads_composition():
    with emulated_kernels():
        run_row_chain_step_narrow_heads("cpu")


def run_bucket_mode(device, use_graph, bucket_rows=None, T_cap=96, L_cap=12, t_min=40):
    """TrainStep(bucket=(T_cap, L_cap)): ONE captured step (padded layouts, lengths on the device) must serve batches whose
    lengths never repeat - six seeded batches through it against the eager packed step on a twin model: loss, clip norm
    and the weights after every update.  bucket_rows: the same with the bucket's rows PACKED into a fixed capacity (offsets
    on the device too; the rows behind the batch's total belong to nobody)."""
    import copy
    import transformer.Models as M
    import transformer.Utils as U
    from st_amd.trainer import TrainStep
    from transformer.Optim import ScheduledOptim
    torch.manual_seed(5)
    cfg = U.AttrDict(dict(feature_dim=80, max_inputs_length=T_cap, max_target_length=L_cap, num_enc_layer=2, num_dec_layer=2,
                          n_heads=4, d_k=64, d_v=64, d_model=256, d_inner_hid=512, dropout=0.0, vocab_size=30))
    ma = M.Transformer(cfg)
    U.init_parameters(ma)
    mb = copy.deepcopy(ma)
    ma, mb = ma.eval().to(device), mb.eval().to(device)
    oa = ScheduledOptim(ma, 256, U.AttrDict(n_warmup_steps=50))
    ob = ScheduledOptim(mb, 256, U.AttrDict(n_warmup_steps=50))
    sa = TrainStep(ma, oa, 30, 5.0, use_graph=use_graph, graph_warmup=1, bucket=(T_cap, L_cap), bucket_rows=bucket_rows)
    sb = TrainStep(mb, ob, 30, 5.0, use_graph=False)
    for i in range(6):
        # (packed buckets: batch 3 is four full-length utterances - more rows than the capacity: it takes the padded bucket)
        full = bucket_rows is not None and i == 3
        b = orc.synthetic_batch(4, T_cap, L_cap, 80, 30, seed=20 + i, t_min=T_cap if full else t_min, l_min=L_cap if full else 4)
        T, L = int(b["in_len"].max()), int(b["tgt_len"].max())
        x, tok, gt = b["x"][:, :T].to(device), b["tokens"][:, :L].to(device), b["gt"][:, :L].to(device)
        la, ga = sa(x, b["in_len"], tok, b["tgt_len"], gt)
        lb, gb = sb(x, b["in_len"], tok, b["tgt_len"], gt)
        la, ga, lb, gb = float(la), float(ga), float(lb), float(gb)
        # The FIRST batch through a layout is the real check: same weights, same arithmetic, so the two steps agree to the order
        # of the fp32 atomics (measured 2e-7 on the gradient, tools/dev/bucket_one_batch.py).  Later batches compare two
        # TRAJECTORIES: the twins' weights differ by ~1e-9 after a step (atomics order), and a few steps on one bf16 rounding
        # of a LayerNorm gain or a bias flips in one twin only - a 5e-5 jump of the loss that grows from there (2e-3 by the fifth
        # batch with some kernel realisations, tools/dev/bucket_traj_probe.py; a defect in a layout shows up at O(1e-1))
        assert abs(la - lb) <= (1e-5 if i == 0 else 5e-3) * abs(lb), (i, la, lb)
        assert abs(ga - gb) <= (1e-4 if i == 0 else 3e-2) * abs(gb), (i, ga, gb)
        for (n, p), q in zip(ma.named_parameters(), mb.parameters()):
            assert torch.isfinite(p).all(), (i, n)
        da = torch.cat([p.detach().reshape(-1) for p in ma.parameters()]).double()
        db = torch.cat([p.detach().reshape(-1) for p in mb.parameters()]).double()
        assert float((da - db).norm() / db.norm()) < 2e-3, (i, float((da - db).norm() / db.norm()))
    if use_graph:
        n_caps = 0 if bucket_rows is None else len(bucket_rows) if isinstance(bucket_rows[0], (tuple, list)) else 1
        assert 1 <= len(sa._buckets) <= 1 + n_caps and (n_caps == 0 or len(sa._buckets) >= 2)
        assert sum(st.cap is not None for st in sa._buckets.values()) >= 1


def test_bucket_mode_composition():
    with emulated_kernels():
        run_bucket_mode("cpu", use_graph=False)


def test_packed_bucket_mode_composition():
    with emulated_kernels():
        run_bucket_mode("cpu", use_graph=False, bucket_rows=(340, 44))


def test_packed_bucket_two_capacities_composition():
    """Two packed capacities: a batch takes the first it fits, the full-length batch the padded bucket."""
    with emulated_kernels():
        run_bucket_mode("cpu", use_graph=False, bucket_rows=[(250, 44), (340, 44)])


def run_layerwise_bucket_firing(device):
    """Data-parallel graph mode (trainer.TrainStep._capture, collectives in the step graph): while the encoder's backward
    runs, the buckets that lie wholly in FINISHED encoder layers are handed to the all-reduce early
    (``_encoder_backward(fire_layers=True)`` -> chains.EncoderBackward's layer hook -> flush of the deferred weight gradients
    -> ``fire_from``).  The property that must hold: whatever a fired bucket holds at that moment is FINAL - nothing is
    accumulated into it afterwards.  A recording stand-in for the reducer snapshots every bucket when it is fired; the
    snapshots must equal the gradients after the whole backward, and buckets inside the upper encoder layers must in fact
    fire before the backward ends."""
    import transformer.Models as M
    import transformer.Utils as U
    from st_amd.arena import arena_of
    from st_amd.trainer import TrainStep
    from transformer.Optim import ScheduledOptim
    torch.manual_seed(7)
    cfg = U.AttrDict(dict(feature_dim=80, max_inputs_length=100, max_target_length=20, num_enc_layer=4, num_dec_layer=1,
                          n_heads=4, d_k=64, d_v=64, d_model=256, d_inner_hid=512, dropout=0.0, vocab_size=30))
    m = M.Transformer(cfg)
    U.init_parameters(m)
    m = m.eval().to(device)
    arena = arena_of(m)
    batch = orc.synthetic_batch(4, 70, 9, 80, 30, seed=4, t_min=30, l_min=4)
    x, tok, gt = batch["x"].to(device), batch["tokens"].to(device), batch["gt"].to(device)

    class Recorder:          # the part of dp.GradReducer's interface the step uses in explicit mode
        def __init__(self, total, per):
            self.buckets, hi = [], total
            while hi > 0:
                lo = max(0, hi - per)
                self.buckets.append((lo, hi))
                hi = lo
            self._fired = [False] * len(self.buckets)
            self.active, self.world, self.group = True, 2, None
            self.snap, self.order = {}, []

        def fire_from(self, lo):
            for i, (blo, bhi) in enumerate(self.buckets):
                if blo >= lo and not self._fired[i]:
                    self._fired[i] = True
                    self.snap[i] = arena.grad[blo:bhi].detach().clone()
                    self.order.append(i)

        def synchronize(self):
            self.fire_from(0)

        def detach(self):
            pass

    red = Recorder(arena.total, 96 * 1024)           # 384 KB buckets: several per encoder layer
    opt = ScheduledOptim(m, 256, U.AttrDict(n_warmup_steps=100))
    step = TrainStep(m, opt, 30, max_grad_norm=1e9, reducer=red, use_graph=False)
    in_len, tgt_len = batch["in_len"], batch["tgt_len"]
    step._forward_decoder_backward(x[:, :int(in_len.max())], in_len, tok[:, :int(tgt_len.max())], tgt_len, gt[:, :int(tgt_len.max())])
    lo_dec = step._decoder_grad_start()
    red.fire_from(lo_dec)
    n_dec = len(red.order)
    assert step._encoder_layer_offsets() is not None and step._encoder_chains() is not None
    step._encoder_backward(fire_layers=True)
    early = len(red.order) - n_dec                 # fired from inside the encoder's backward
    red.synchronize()
    los = step._encoder_layer_offsets()
    inside_upper = [i for i, (blo, bhi) in enumerate(red.buckets) if blo >= los[1] and bhi <= lo_dec]
    assert inside_upper and early >= len(inside_upper) > 0, (early, inside_upper)
    assert all(red.order.index(i) < len(red.order) - 1 for i in inside_upper)
    for i, (blo, bhi) in enumerate(red.buckets):
        assert torch.equal(red.snap[i], arena.grad[blo:bhi]), "bucket %d [%d, %d) changed after it was fired" % (i, blo, bhi)
    # ... and the gradients are those of the plain single-graph step
    m2 = M.Transformer(cfg)
    m2.load_state_dict({k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    m2 = m2.eval().to(device)
    opt2 = ScheduledOptim(m2, 256, U.AttrDict(n_warmup_steps=100))
    step2 = TrainStep(m2, opt2, 30, max_grad_norm=1e9, use_graph=False)
    step2._forward_backward(x[:, :int(in_len.max())], in_len, tok[:, :int(tgt_len.max())], tgt_len, gt[:, :int(tgt_len.max())])
    a2 = arena_of(m2)
    assert float((arena.grad - a2.grad).norm() / a2.grad.norm()) < 1e-5


def test_layerwise_bucket_firing_composition():
    with emulated_kernels():
        run_layerwise_bucket_firing("cpu")

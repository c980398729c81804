"""-m gpu: parity of the TIMED object at the BENCHMARKED sizes.

BASELINE config 2 exactly as ``bench.py`` builds it (6+6 layers, d_model 256, 4 heads, d_ff 1024, V 4337, B = 32,
seed-0 synthetic batch: 24,060 packed frame rows / 1,206 target rows, 8-way split grouped weight gradients) and
BASELINE config 3's real depth (12+6 layers, d_model 512, 8 heads, the 4-utterance per-GPU shard of DP = 8) run
through ``st_amd.trainer.TrainStep`` and are compared with ``oracle.train_step`` evaluated in float64 ON THE GPU
(the oracle is plain device-agnostic torch; the [32, 4, 1000, 1000] fp64 score tensors are 1 GB each on a 288 GB
part): loss, logits of the valid rows, EVERY gradient tensor, the pre-clip gradient norm and the post-step weights
(train.py:25-46).  Tolerances are SURVEY.md section 8c's (bf16 activations / fp32 accumulate against fp64 truth):
logits rel-L2 <= 2e-2, per-tensor gradient rel-L2 <= 8e-2, the analytically zero ``linear_k.bias`` by absolute
bound.

Those bounds were measured by the survey on a 6+6 / T = 200 model; at T = 1000 and V = 4337 some gradients are
ill-conditioned for ANY bf16 implementation (near-uniform attention makes the decoder's q / k projection gradients a
1e-4 fraction of the gradient norm: differences of nearly equal terms).  The test therefore ALSO runs the oracle under
``torch.autocast("cuda", bfloat16)`` - the reference arithmetic with bf16 matmuls, fp32 softmax / LayerNorm / residual
stream - on the same batch and prints its per-tensor error next to the HIP path's: a tensor passes when it is within
8e-2 OR within 1.5x what the bf16 reference itself shows on that tensor (round 3: was 2x; the worst ratio among the
tensors above 8e-2 is 1.42, profiles/r03_parity_c2_b32.txt); the global and median figures must be within
3e-2 / 4e-2 OR 1.15x the bf16 reference's (round 4; it was 1.5x).  The table is written to ``gpurun_out/parity_<config>.txt`` (and shown
when an assertion fails); a copy per round lives under ``profiles/``.
"""
import os

import pytest
import torch

import oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C2 = dict(feature_dim=80, max_inputs_length=1000, max_target_length=50, num_enc_layer=6, num_dec_layer=6, n_heads=4,
          d_k=64, d_v=64, d_model=256, d_inner_hid=1024, dropout=0.1, vocab_size=4337)
C3 = dict(feature_dim=80, max_inputs_length=1000, max_target_length=50, num_enc_layer=12, num_dec_layer=6, n_heads=8,
          d_k=64, d_v=64, d_model=512, d_inner_hid=1024, dropout=0.1, vocab_size=4337)

LOGIT_TOL, GRAD_TOL_TENSOR, GRAD_TOL_MEDIAN, GRAD_TOL_GLOBAL = 2e-2, 8e-2, 4e-2, 3e-2
TRAIN_FLOOR_X = 1.15     # training mode: global / median gradient error within this factor of the bf16 reference's (same masks;
                         # round 6: was 1.25 - eleven seeds range 0.86 .. 1.12, profiles/r05_train_parity_seeds.txt)


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


def sharpen_attention(model, factor):
    """Scale every attention's query and key projections (weights and biases) by `factor`: the scores grow by factor^2 and the
    softmax turns peaky - the regime a trained model lives in, and the one in which a re-rounded score operand shows (round 5:
    the backward streams were 3-14x less accurate than the general kernels there until the keys were pre-scaled)."""
    with torch.no_grad():
        for n, p in model.named_parameters():
            if ".linear_q." in n or ".linear_k." in n:
                p.mul_(factor)


def train_on_one_batch_fp64(w, cfg, batch, n_steps, warmup=400, max_grad_norm=5.0):
    """The oracle's weights after `n_steps` fp64 Adam steps on ONE batch (warm-up 400, the trajectory test's schedule: the loss
    falls by about a nat in 30 steps and the gradients stay healthy; 120 steps at warm-up 60 collapse this model onto the unigram
    distribution - gradient norm 1.6e-3, nothing left to measure).  -> state_dict-like dict of fp32 tensors."""
    p64 = {k: v.double().cuda() for k, v in w.items()}
    adam = None
    for k in range(1, n_steps + 1):
        out = orc.train_step(p64, batch, cfg["n_heads"], cfg["d_model"], warmup, k, max_grad_norm, adam_state=adam)
        adam, p64 = out["adam"], out["params"]
    return {k: v.float().cpu() for k, v in p64.items()}, out["loss"].item()


def run_step_parity(cfg, n_utts, tag, use_graph=False, warmup=12000, max_grad_norm=5.0, seed=0, sharp=None, pretrain=0):
    import transformer.Models as M
    import transformer.Utils as U
    from st_amd import synthetic
    from st_amd.arena import arena_of
    from st_amd.trainer import TrainStep
    from transformer.Optim import ScheduledOptim

    torch.manual_seed(seed)                        # (the weights' seed: tools/dev/c3_kpre_ab.py varies it)
    model = M.Transformer(U.AttrDict(cfg))
    U.init_parameters(model)                       # train.py:116
    if sharp:
        sharpen_attention(model, sharp)
    x, tokens, in_len, tgt_len, gt = synthetic.make_batch(32, 1000, 50, cfg["feature_dim"], cfg["vocab_size"], seed=0,
                                                          t_min=500, l_min=25)
    x, tokens, in_len, tgt_len, gt = x[:n_utts], tokens[:n_utts], in_len[:n_utts], tgt_len[:n_utts], gt[:n_utts]
    if n_utts == 32:
        assert int(in_len.sum()) == 24060 and int(tgt_len.sum()) == 1206
    xg, tg, gg = x.cuda(), tokens.cuda(), gt.cuda()
    if pretrain:       # the weights after `pretrain` fp64 steps of the oracle on this very batch
        w0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        wt, l_end = train_on_one_batch_fp64(w0, cfg, {"x": xg.double(), "in_len": in_len, "tokens": tg, "tgt_len": tgt_len, "gt": gg}, pretrain)
        model.load_state_dict(wt)
        print("pretrained %d fp64 steps on the batch: loss %.4f" % (pretrain, l_end))
    w = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.eval().cuda()                    # eval(): every Dropout is the identity (the oracle's parity mode)
    L = int(tgt_len.max())
    valid = (torch.arange(L).view(1, -1) < tgt_len.view(-1, 1)).cuda()

    # ---- fp64 truth on the GPU ---------------------------------------------------------------------------------
    p64 = {k: v.double().cuda() for k, v in w.items()}
    b64 = {"x": xg.double(), "in_len": in_len, "tokens": tg, "tgt_len": tgt_len, "gt": gg}
    truth = orc.train_step(p64, b64, cfg["n_heads"], cfg["d_model"], warmup, 1, max_grad_norm)
    torch.cuda.synchronize()

    # ---- the reference arithmetic in bf16 (autocast): the noise floor of this configuration, tensor by tensor -------
    names = [k for k in w if not k.endswith(".pe")]
    leaves = {k: (v.float().cuda().requires_grad_(True) if k in names else v.float().cuda()) for k, v in w.items()}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lg_ac, _ = orc.transformer(leaves, xg, in_len, tg[:, :L], tgt_len, cfg["n_heads"])
        loss_ac = orc.cross_entropy(lg_ac.float(), gg[:, :L])
    g_ac = dict(zip(names, torch.autograd.grad(loss_ac, [leaves[k] for k in names])))
    floor = {n: rel(g_ac[n], truth["grads"][n]) for n in names if "linear_k.bias" not in n}
    fl_sorted = sorted(floor.values())
    floor_med = fl_sorted[len(fl_sorted) // 2]
    floor_glob = rel(torch.cat([g_ac[n].reshape(-1) for n in floor]), torch.cat([truth["grads"][n].reshape(-1) for n in floor]))
    del leaves, lg_ac, loss_ac

    # ---- the product: logits (no-grad forward, same kernels), then ONE TrainStep call ------------------------------
    with torch.no_grad():
        lg, t_rows = model.forward_packed(xg, in_len, tg[:, :L], tgt_len)
    logit_rel = rel(lg.double(), truth["logits"][valid])       # ragged rows are utterance-major == masked-select order
    before = arena_of(model).flat.detach().clone()
    opt = ScheduledOptim(model, cfg["d_model"], U.AttrDict(n_warmup_steps=warmup))
    if use_graph:      # every kernel module loaded before the capture (a throw-away backward; the step zeroes the gradients)
        lg2, _ = model.forward_packed(xg, in_len, tg[:, :L], tgt_len)
        lg2.float().sum().backward()
        del lg2
    step = TrainStep(model, opt, cfg["vocab_size"], max_grad_norm, use_graph=use_graph, graph_warmup=0)
    loss, gnorm = step(xg, in_len, tg, tgt_len, gg)
    torch.cuda.synchronize()
    loss, gnorm = float(loss), float(gnorm)
    arena = arena_of(model)
    coef = min(1.0, max_grad_norm / (gnorm + 1e-6))        # st_adam_clip scaled the gradients in place
    lr = truth["lr"]

    rows, fg, ft, dev_sum, n_el = [], [], [], 0.0, 0
    kbias = []
    for n, p in model.named_parameters():
        g = arena.grad_view(p).detach().double() / coef
        t = truth["grads"][n]
        if "linear_k.bias" in n:                          # analytically zero (softmax shift invariance)
            q = truth["grads"][n.replace("linear_k", "linear_q")].abs().max().item()
            kbias.append((g.abs().max().item(), q, n))
            continue
        rows.append((rel(g, t), floor[n], n, t.norm().item()))
        fg.append(g.reshape(-1))
        ft.append(t.reshape(-1))
        off = arena.offset[id(p)]
        old = before[off:off + p.numel()].double()
        u = (p.detach().double().reshape(-1) - old) / lr
        ut = (truth["params"][n].reshape(-1) - p64[n].reshape(-1)) / lr
        dev_sum += (u - ut).abs().sum().item()
        n_el += p.numel()
    glob = rel(torch.cat(fg), torch.cat(ft))
    rows.sort(reverse=True)
    med = rows[len(rows) // 2][0]
    lines = ["# %s: TrainStep(use_graph=%s) vs oracle.train_step in fp64 on the GPU" % (tag, use_graph),
             "loss %.6f (oracle %.6f, rel %.2e)  grad_norm %.5f (oracle %.5f, rel %.2e)  logits rel-L2 %.3e"
             % (loss, truth["loss"].item(), abs(loss - truth["loss"].item()) / truth["loss"].item(), gnorm,
                truth["grad_norm"].item(), abs(gnorm - truth["grad_norm"].item()) / truth["grad_norm"].item(), logit_rel),
             "gradients: global rel-L2 %.3e, per-tensor median %.3e, max %.3e; mean |u - u_ref| of the Adam update %.4f"
             % (glob, med, rows[0][0], dev_sum / n_el),
             "reference under bf16 autocast (same batch): global %.3e, median %.3e, max %.3e"
             % (floor_glob, floor_med, fl_sorted[-1]),
             "per-tensor rel-L2 (worst first):   HIP path | reference-in-bf16 | tensor"]
    lines += ["  %.3e  %.3e  %-58s |g| = %.3e" % r for r in rows]
    lines += ["  linear_k.bias |g|max %.2e vs linear_q.bias |g|max %.2e  %s" % k for k in kbias]
    report = "\n".join(lines)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "parity_%s.txt" % tag), "w") as f:
        f.write(report + "\n")
    head = "\n".join(lines[:12])
    assert logit_rel < LOGIT_TOL, head
    assert abs(loss - truth["loss"].item()) < 2e-2 * truth["loss"].item(), head
    assert abs(gnorm - truth["grad_norm"].item()) < 2e-2 * truth["grad_norm"].item(), head
    # (round 4: with the long forward's scores scaled in fp32 both figures lie BELOW the bf16 reference's - 0.96x / 0.92x on
    # config 2, 0.84x / 0.85x on config 3's shard; 1.15x leaves room for another box's rounding realisation)
    assert glob < max(GRAD_TOL_GLOBAL, 1.15 * floor_glob) and med < max(GRAD_TOL_MEDIAN, 1.15 * floor_med), head
    bad = [r for r in rows if r[0] > max(GRAD_TOL_TENSOR, 1.5 * r[1])]
    assert not bad, "\n".join([head, "outside max(8e-2, 1.5 x bf16 reference):"] + ["  %.3e  %.3e  %s" % r[:3] for r in bad])
    for g, q, n in kbias:      # analytically zero: 1e-2 of the query bias' gradient (measured: 5e-4 .. 4e-2 of it where that gradient
        assert g < 1e-2 * q + 1e-7, (n, g, q)     # itself is ~1e-6 - the absolute term is the fp32 noise floor of those sums)
    assert dev_sum / n_el < 0.08, head          # a sign flip of an Adam first-step update costs 2
    return report


def test_config2_trainstep_full_size_vs_fp64_oracle():
    """BASELINE config 2 at the benchmarked size (B = 32, 24,060 rows, 6+6 layers), eager TrainStep."""
    run_step_parity(C2, 32, "c2_b32")


def test_config2_trainstep_graph_full_size_vs_fp64_oracle():
    """The same step replayed from the HIP graph bench.py times."""
    run_step_parity(C2, 32, "c2_b32_graph", use_graph=True)


def test_config2_full_size_sharp_attention_vs_fp64_oracle():
    """VERDICT r5 (next 6a): the step-level table in the regime the pre-scaled keys exist for.  Config 2 at the benchmarked size
    with every attention's q / k projections x 2 (scores x 4: peaky softmax rows), replayed from the graph as bench.py times it -
    same tolerance rules as at Xavier initialisation.  Measured (tools/dev/sharp_parity_scan.py): x 1.25 / 1.5 / 1.75 / 2 give
    global 3.5e-2 / 3.1e-2 / 3.9e-2 / 5.8e-2 against 3.6e-2 / 3.4e-2 / 3.8e-2 / 6.5e-2 for the reference arithmetic under bf16
    autocast.  x 3 (the verdict's figure) is past what this 6+6-layer random-weight model can measure: the gradient norm goes from
    0.8 to 70 and the bf16 REFERENCE is 1.4 (rel-L2) away from fp64 - a one-hot softmax over 1,000 random keys flips its argmax
    under any rounding."""
    run_step_parity(C2, 32, "c2_b32_sharp2", use_graph=True, sharp=2.0)


def test_config2_after_fp64_pretraining_vs_fp64_oracle():
    """... and with weights that TRAINING produced: the oracle's own weights after 30 fp64 Adam steps on the batch (8 utterances,
    warm-up 400), then one step of the product against one step of the oracle from those weights."""
    run_step_parity(C2, 8, "c2_b8_pretrained", use_graph=True, pretrain=30)


def test_config3_depth_trainstep_vs_fp64_oracle():
    """BASELINE config 3's depth and width (12+6 layers, d_model 512, 8 heads) on its per-GPU shard (4 utterances)."""
    run_step_parity(C3, 4, "c3_b4")


def test_config2_trainstep_trajectory_vs_fp64_oracle():
    """TWENTY optimisation steps, not one: the captured step (HIP-graph replay, as bench.py times it) against
    ``oracle.train_step`` in float64 carrying its own Adam state, on one 8-utterance batch of config 2 with a short warm-up
    (warmup 400: the learning rate reaches 1.6e-4 at step 20 and the loss falls by about half a nat - a trajectory that goes
    somewhere, inside the smooth regime: this model's loss surface turns sharp near loss 7.4 - the ORACLE's own clip norm jumps
    0.85 -> 0.92 -> 1.42 there (tools/dev/traj_probe.py, which also shows the HIP gradients agreeing with the oracle's at
    IDENTICAL weights all the way through, global rel-L2 1e-2 .. 4e-2) - and two roundings of the recursion reach that region a
    step apart; with warmup 100 / 200 that happens at step 11 / 15).
    Per step: the loss, the clip norm, and the accumulated weight change w_k - w_0 against the oracle's (rel-L2 over the whole
    parameter vector).  Early Adam steps are sign-like (update ~ lr * g / |g|), so elements whose gradient is bf16 noise can
    move the other way: the weight change agrees to a few per cent, not to rounding - what must NOT happen is drift (the error
    growing step over step) or a loss curve that separates."""
    import transformer.Models as M
    import transformer.Utils as U
    from st_amd import synthetic
    from st_amd.arena import arena_of
    from st_amd.trainer import TrainStep
    from transformer.Optim import ScheduledOptim

    cfg, n_utts, n_steps, warmup = C2, 8, 20, 400
    torch.manual_seed(0)
    model = M.Transformer(U.AttrDict(cfg))
    U.init_parameters(model)
    w0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.eval().cuda()
    x, tokens, in_len, tgt_len, gt = synthetic.make_batch(32, 1000, 50, cfg["feature_dim"], cfg["vocab_size"], seed=0, t_min=500, l_min=25)
    x, tokens, in_len, tgt_len, gt = x[:n_utts], tokens[:n_utts], in_len[:n_utts], tgt_len[:n_utts], gt[:n_utts]
    xg, tg, gg = x.cuda(), tokens.cuda(), gt.cuda()
    opt = ScheduledOptim(model, cfg["d_model"], U.AttrDict(n_warmup_steps=warmup))
    step = TrainStep(model, opt, cfg["vocab_size"], 5.0, use_graph=True, graph_warmup=1)
    names = [n for n, _ in model.named_parameters()]
    p64 = {k: v.double().cuda() for k, v in w0.items()}
    b64 = {"x": xg.double(), "in_len": in_len, "tokens": tg, "tgt_len": tgt_len, "gt": gg}
    flat0 = torch.cat([p64[n].reshape(-1) for n in names])
    adam, lines, worst_loss, worst_w = None, ["# 20 steps of TrainStep(use_graph=True) vs oracle.train_step (fp64, own Adam state); config 2, 8 utterances, warmup 400",
                                              "step  loss(HIP)  loss(oracle)  rel       gnorm(HIP)  gnorm(oracle)  |dw - dw_ref| / |dw_ref|   lr"], 0.0, 0.0
    for k in range(1, n_steps + 1):
        loss, gnorm = step(xg, in_len, tg, tgt_len, gg)
        loss, gnorm = float(loss), float(gnorm)
        truth = orc.train_step(p64, b64, cfg["n_heads"], cfg["d_model"], warmup, k, 5.0, adam_state=adam)
        adam, p64 = truth["adam"], truth["params"]
        sd = model.state_dict()
        flat = torch.cat([sd[n].detach().double().reshape(-1) for n in names])
        ref = torch.cat([p64[n].reshape(-1) for n in names])
        wrel = ((flat - ref).norm() / (ref - flat0).norm()).item()
        lrel = abs(loss - truth["loss"].item()) / truth["loss"].item()
        worst_loss, worst_w = max(worst_loss, lrel), max(worst_w, wrel)
        lines.append("%3d   %.5f    %.5f       %.2e  %.5f     %.5f        %.3e                  %.3e"
                     % (k, loss, truth["loss"].item(), lrel, gnorm, truth["grad_norm"].item(), wrel, truth["lr"]))
        assert abs(gnorm - truth["grad_norm"].item()) < 3e-2 * truth["grad_norm"].item(), "\n".join(lines)
    first, last = float(lines[2].split()[2]), truth["loss"].item()
    lines.append("loss of the oracle: %.4f -> %.4f; worst loss rel %.2e, worst weight-change rel-L2 %.3e" % (first, last, worst_loss, worst_w))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "trajectory_c2_b8.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    report = "\n".join(lines)
    assert first - last > 0.4, report                       # the trajectory goes somewhere
    assert worst_loss < 1e-3 and worst_w < 0.3, report
    wrels = [float(l.split()[6]) for l in lines[2:2 + n_steps]]
    assert wrels[-1] < 1.5 * min(wrels), report            # no drift: the disagreement does not grow along the trajectory


# ------------------------------------------------------------------------------------------------------------------
# Training mode at the benchmarked size: the product draws its counter-based dropout masks (attention probabilities, both
# FFN sites per layer, the front-end's 0.5), the fp64 oracle is handed EXACTLY those masks (re-derived on the GPU from the
# recorded call sites with the host implementation of the hash in tests/_emul.py).  VERDICT r2 "missing 6".
# ------------------------------------------------------------------------------------------------------------------
def _gpu_masks(sites, in_len, tgt_len, H):
    """-> provider(site, shape) for oracle.dropout_masks, replayable (``provider.rewind()``)."""
    from st_amd.functional import Rows
    from tests import _emul as em
    dev = "cuda"
    T, L = int(in_len.max()), int(tgt_len.max())
    rows_of = {T: Rows.packed(in_len, "cpu"), L: Rows.packed(tgt_len, "cpu")}
    drops = [em.Drop(d.seed.detach().cpu(), d.salt, d.thresh / 256.0) for d in sites]
    state = {"i": 0}

    def provider(site, shape):
        d = drops[state["i"]]
        state["i"] += 1
        key = em._key(d)
        if site == "attn":
            B, h, lq, lk = shape
            q = torch.arange(lq, dtype=torch.int64, device=dev).view(1, -1, 1)
            k = torch.arange(lk, dtype=torch.int64, device=dev).view(1, 1, -1)
            bh = torch.arange(B * h, dtype=torch.int64, device=dev).view(-1, 1, 1)
            cnt = ((((q >> 1) << 15) | (k >> 1)) + bh * 0x85ebca6b) & em._M32
            bits = em._hash32(cnt ^ key)
            keep = ((bits >> (8 * (2 * (q & 1) + (k & 1)))) & 0xFF) >= d.thresh
            return keep.view(B, h, lq, lk).double() * d.scale
        B, t, n = shape
        r = rows_of[t]
        off = r.off.to(dev).long().view(-1, 1, 1)
        rows = off + torch.arange(t, dtype=torch.int64, device=dev).view(1, -1, 1)          # packed row of (b, i)
        cols = torch.arange(n, dtype=torch.int64, device=dev).view(1, 1, -1)
        cnt = (rows * (n >> 2) + (cols >> 2)) & em._M32
        bits = em._hash32(cnt ^ key)
        keep = (((bits >> (8 * (cols & 3))) & 0xFF) >= d.thresh).double() * d.scale
        valid = (torch.arange(t, device=dev).view(1, -1) < r.len.to(dev).view(-1, 1)).view(B, t, 1)
        return torch.where(valid, keep, torch.ones_like(keep))      # padded frames: nobody reads them

    provider.rewind = lambda: state.update(i=0)
    provider.count = lambda: state["i"]
    return provider


def test_config2_training_mode_full_size_vs_fp64_oracle():
    """BASELINE config 2 at the benchmarked size under model.train() (how train.py:21 runs the reference): loss, logits and
    every gradient against the fp64 oracle with the kernels' own masks.

    Three mask draws, because ONE draw says little: over eleven mask seeds (profiles/r05_train_parity_seeds.txt) the ratio of the
    HIP path's global gradient error to that of the reference arithmetic under bf16 autocast WITH THE SAME MASKS ranges from
    0.86 to 1.12 (mean 0.94; round 4's "regression" 4.02e-2 -> 4.37e-2 was the draw of seed 20260928, the high end, and is
    the same to four digits with the general attention-backward kernels: ST_ATTN_BWD64=e).  Bounds: every draw within
    TRAIN_FLOOR_X of its floor (and the per-tensor bounds of run_train_mode_parity), the MEAN ratio over the draws <= 1.05
    global and <= 1.08 in the per-tensor median."""
    res = [run_train_mode_parity(seed, out_name="parity_c2_b32_train%s.txt" % ("" if i == 0 else "_seed%d" % seed))
           for i, seed in enumerate((20260928, 1, 2))]
    rg = [r["glob"] / r["floor_glob"] for r in res]
    rm = [r["med"] / r["floor_med"] for r in res]
    note = "ratios to the bf16 reference over three mask draws: global %s, median %s" % (
        " ".join("%.3f" % v for v in rg), " ".join("%.3f" % v for v in rm))
    with open(os.path.join(ROOT, "gpurun_out", "parity_c2_b32_train.txt"), "a") as f:
        f.write("# " + note + "\n")
    assert sum(rg) / len(rg) <= 1.05 and sum(rm) / len(rm) <= 1.08, note


def test_config2_training_mode_sharp_attention_vs_fp64_oracle():
    """Training mode (in-stream dropout masks, the streams' dropout variants) with every attention's q / k projections x 1.5
    (x 2 under dropout is past the measurable: the reference arithmetic under bf16 autocast is 0.67 rel-L2 from fp64 there, the
    HIP path 0.51)."""
    run_train_mode_parity(1, out_name="parity_c2_b32_train_sharp.txt", sharp=1.5)


def run_train_mode_parity(mask_seed, out_name="parity_c2_b32_train.txt", check=True, sharp=None):
    """-> dict(glob, med, worst, floor_glob, floor_med, floor_worst, logits); tools/dev/train_parity_sweep.py runs it over
    several mask seeds and attention-kernel variants (the spread of the figures = their realisation noise)."""
    import transformer.Models as M
    import transformer.Utils as U
    from st_amd import functional as F_, rng, synthetic
    from st_amd.arena import arena_of

    cfg = C2
    torch.manual_seed(0)
    model = M.Transformer(U.AttrDict(cfg))
    U.init_parameters(model)
    if sharp:
        sharpen_attention(model, sharp)
    w = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda().train()
    x, tokens, in_len, tgt_len, gt = synthetic.make_batch(32, 1000, 50, cfg["feature_dim"], cfg["vocab_size"], seed=0,
                                                          t_min=500, l_min=25)
    xg, tg, gg = x.cuda(), tokens.cuda(), gt.cuda()
    L = int(tgt_len.max())
    valid = (torch.arange(L).view(1, -1) < tgt_len.view(-1, 1)).cuda()
    rng.seed_tensor("cuda")
    rng.manual_seed(mask_seed)
    sites, orig_site = [], rng.site

    def recording_site(dev, p):
        d = orig_site(dev, p)
        sites.append(d)
        return d

    rng.site = recording_site
    try:
        arena_of(model).zero_grads()
        lg, t_rows = model.forward_packed(xg, in_len, tg[:, :L], tgt_len)
    finally:
        rng.site = orig_site
    truth_gt = gg[:, :L].contiguous().view(-1).index_select(0, t_rows.scatter_index(L))
    loss = torch.nn.CrossEntropyLoss(ignore_index=0)(lg.float(), truth_gt)
    with F_.deferred_wgrads(True):
        loss.backward()
    torch.cuda.synchronize()
    n_e, n_d = cfg["num_enc_layer"], cfg["num_dec_layer"]
    assert len(sites) == 1 + 3 * n_e + 4 * n_d and all(d is not None for d in sites)
    assert sites[0].thresh == 128 and all(d.thresh == 26 for d in sites[1:])      # p = 0.5 and round(256 * 0.1)

    provider = _gpu_masks(sites, in_len, tgt_len, cfg["n_heads"])
    p64 = {k: v.double().cuda() for k, v in w.items()}
    b64 = {"x": xg.double(), "in_len": in_len, "tokens": tg[:, :L], "tgt_len": tgt_len, "gt": gg[:, :L]}
    with orc.dropout_masks(provider):
        truth = orc.train_step(p64, b64, cfg["n_heads"], cfg["d_model"], 12000, 1, 5.0)
    assert provider.count() == len(sites)
    # the same masks under bf16 autocast: the noise floor of this configuration in training mode
    names = [k for k in w if not k.endswith(".pe")]
    leaves = {k: (v.float().cuda().requires_grad_(True) if k in names else v.float().cuda()) for k, v in w.items()}
    provider.rewind()
    with orc.dropout_masks(provider), torch.autocast("cuda", dtype=torch.bfloat16):
        lg_ac, _ = orc.transformer(leaves, xg, in_len, tg[:, :L], tgt_len, cfg["n_heads"])
        loss_ac = orc.cross_entropy(lg_ac.float(), gg[:, :L])
    g_ac = dict(zip(names, torch.autograd.grad(loss_ac, [leaves[k] for k in names])))
    floor = {n: rel(g_ac[n], truth["grads"][n]) for n in names if "linear_k.bias" not in n}
    fl = sorted(floor.values())
    floor_glob = rel(torch.cat([g_ac[n].reshape(-1) for n in floor]), torch.cat([truth["grads"][n].reshape(-1) for n in floor]))

    arena = arena_of(model)
    logit_rel = rel(lg.double(), truth["logits"][valid])
    rows, fg, ft = [], [], []
    for n, p in model.named_parameters():
        if "linear_k.bias" in n:
            continue
        g, t = arena.grad_view(p).detach().double(), truth["grads"][n]
        assert torch.isfinite(g).all(), n
        rows.append((rel(g, t), floor[n], n, t.norm().item()))
        fg.append(g.reshape(-1))
        ft.append(t.reshape(-1))
    glob = rel(torch.cat(fg), torch.cat(ft))
    rows.sort(reverse=True)
    med = rows[len(rows) // 2][0]
    lines = ["# c2_b32_train: model.train() forward + CE + backward vs the fp64 oracle with the kernels' dropout masks",
             "loss %.6f (oracle %.6f)  logits rel-L2 %.3e  (%d dropout sites)" % (loss.item(), truth["loss"].item(), logit_rel, len(sites)),
             "gradients: global rel-L2 %.3e, per-tensor median %.3e, max %.3e" % (glob, med, rows[0][0]),
             "reference under bf16 autocast, same masks: global %.3e, median %.3e, max %.3e" % (floor_glob, fl[len(fl) // 2], fl[-1]),
             "per-tensor rel-L2 (worst first):   HIP path | reference-in-bf16 | tensor"]
    lines += ["  %.3e  %.3e  %-58s |g| = %.3e" % r for r in rows]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", out_name), "w") as f:
        f.write("\n".join(lines) + "\n")
    head = "\n".join(lines[:12])
    out = dict(glob=glob, med=med, worst=rows[0][0], floor_glob=floor_glob, floor_med=fl[len(fl) // 2], floor_worst=fl[-1],
               logits=logit_rel, loss=loss.item(), loss_oracle=truth["loss"].item())
    if not check:
        return out
    # training-mode tolerances: as eval mode, measured against the reference's own arithmetic under bf16 autocast WITH THE SAME
    # MASKS (the noise floor of this configuration): global / median within TRAIN_FLOOR_X x the floor (the absolute bounds only
    # matter where the floor is tiny); per tensor 1e-1 or 1.5x the bf16 reference's
    assert logit_rel < 2 * LOGIT_TOL, head
    assert abs(loss.item() - truth["loss"].item()) < 2e-2 * truth["loss"].item(), head
    assert glob < max(GRAD_TOL_GLOBAL, TRAIN_FLOOR_X * floor_glob) and med < max(GRAD_TOL_MEDIAN, TRAIN_FLOOR_X * fl[len(fl) // 2]), head
    bad = [r for r in rows if r[0] > max(1.25 * GRAD_TOL_TENSOR, 1.5 * r[1])]      # (measured worst: 7.6e-2 vs 9.5e-2 for the bf16 reference)
    assert not bad, "\n".join([head, "outside max(1e-1, 1.5 x bf16 reference):"] + ["  %.3e  %.3e  %s" % r[:3] for r in bad])
    return out


def test_config4_joint_ctc_attention_at_stated_shape_vs_fp64_oracle():
    """BASELINE config 4 at its stated shape (6+6 layers, d_model 256: the row-chain path): joint 0.3 CTC + 0.7 attention
    objective (train_attn_and_ctc.py); the encoder output receives gradient from the CTC head AND from the decoder's
    shared cross-K/V gradient buffer.  8 utterances of the benchmark batch, oracle in fp64 on the GPU."""
    import torch.nn.functional as func
    import transformer.Models as M
    import transformer.Utils as U
    from st_amd import synthetic
    from transformer.Loss import CTCAttentionLoss

    cfg, n = C2, 8
    torch.manual_seed(0)
    model = M.Transformer(U.AttrDict(cfg))
    U.init_parameters(model)
    w = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.eval().cuda()
    x, tokens, in_len, tgt_len, gt = synthetic.make_batch(32, 1000, 50, cfg["feature_dim"], cfg["vocab_size"], seed=0,
                                                          t_min=500, l_min=25)
    x, tokens, in_len, tgt_len, gt = x[:n], tokens[:n], in_len[:n], tgt_len[:n], gt[:n]
    L, T = int(tgt_len.max()), int(in_len.max())
    xg, tg, gg = x[:, :T].cuda(), tokens[:, :L].cuda(), gt[:, :L].cuda()
    torch.manual_seed(0)
    head = CTCAttentionLoss(cfg["d_model"], cfg["vocab_size"], ctc_weight=0.3).cuda()
    H = cfg["n_heads"]

    def objective(leaves, wc, bc, xin, autocast=False):
        enc, _ = orc.encoder(leaves, xin, in_len, H)
        dec, _, _ = orc.decoder(leaves, tg, tgt_len, in_len, enc, H)
        logits = func.linear(dec, leaves["tgt_word_proj.weight"])
        att = orc.cross_entropy(logits.float() if autocast else logits, gg)
        z = func.linear(enc, wc, bc)
        logp = func.log_softmax(z.float() if autocast else z, -1).transpose(0, 1)
        ctc = func.ctc_loss(logp, gg, in_len, tgt_len, blank=0, reduction="mean", zero_infinity=True)
        return 0.3 * ctc + 0.7 * att, ctc

    names = [k for k in w if not k.endswith(".pe")]
    l64 = {k: (v.double().cuda().requires_grad_(True) if k in names else v.double().cuda()) for k, v in w.items()}
    w64 = head.ctc_proj.weight.detach().double().clone().requires_grad_(True)
    b64 = head.ctc_proj.bias.detach().double().clone().requires_grad_(True)
    truth, ctc64 = objective(l64, w64, b64, xg.double())
    g64 = torch.autograd.grad(truth, [l64[k] for k in names] + [w64, b64], allow_unused=True)
    tgd = dict(zip(names, g64))
    l32 = {k: (v.float().cuda().requires_grad_(True) if k in names else v.float().cuda()) for k, v in w.items()}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        obj_ac, _ = objective(l32, head.ctc_proj.weight.detach().float(), head.ctc_proj.bias.detach().float(), xg, autocast=True)
    gac = dict(zip(names, torch.autograd.grad(obj_ac, [l32[k] for k in names], allow_unused=True)))

    logits_h, enc_h = model.forward_joint(xg, in_len, tg, tgt_len)
    assert enc_h.shape == (n, T, cfg["d_model"]) and logits_h.shape == (n, L, cfg["vocab_size"])
    loss, att_h, ctc_h = head(enc_h, in_len, logits_h, gg, tgt_len, gg)
    loss.backward()
    torch.cuda.synchronize()
    assert model.encoder._st_chains[1] is not None and model.decoder._st_chains[1] is not None      # the row-chain path
    assert abs(loss.item() - truth.item()) < 2e-2 * abs(truth.item()), (loss.item(), truth.item())
    assert abs(ctc_h.item() - ctc64.item()) < 2e-2 * abs(ctc64.item())
    rows = []
    for nme, q in model.named_parameters():
        if "linear_k.bias" in nme or tgd[nme] is None:
            continue
        assert q.grad is not None and torch.isfinite(q.grad).all(), nme
        rows.append((rel(q.grad.detach(), tgd[nme]), rel(gac[nme], tgd[nme]), nme))
    rows.sort(reverse=True)
    lines = ["# c4_b8: joint 0.3 CTC + 0.7 attention objective, 6+6 / d256, 8 utterances: loss %.5f (oracle %.5f), ctc %.4f (%.4f)"
             % (loss.item(), truth.item(), ctc_h.item(), ctc64.item()),
             "per-tensor rel-L2 (worst first):   HIP path | reference-in-bf16 | tensor"] + ["  %.3e  %.3e  %s" % r for r in rows]
    with open(os.path.join(ROOT, "gpurun_out", "parity_c4_b8.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    # (per tensor 1e-1 or 1.5x the bf16 reference's: on this 8-utterance batch decoder layer 1's encoder-decoder q / k tensors move
    # between 7.1e-2 (round 4) and 8.5e-2 (round 5) against 5.5e-2 for the reference-in-bf16 with every change of an upstream
    # rounding - at B = 32 the same tensors sit BELOW the reference's error, 4.99e-2 vs 5.62e-2: profiles/r05_parity_c2_b32.txt)
    bad = [r for r in rows if r[0] > max(1.25 * GRAD_TOL_TENSOR, 1.5 * r[1])]
    assert not bad, "\n".join(lines[:2] + ["  %.3e  %.3e  %s" % r for r in bad])
    med = sorted(r[0] for r in rows)[len(rows) // 2]
    assert med < max(GRAD_TOL_MEDIAN, 1.5 * sorted(r[1] for r in rows)[len(rows) // 2]), med
    assert rel(head.ctc_proj.weight.grad, g64[-2]) < GRAD_TOL_MEDIAN and rel(head.ctc_proj.bias.grad, g64[-1]) < GRAD_TOL_MEDIAN


def test_config4_joint_trainstep_b32_graph_vs_fp64_oracle():
    """BASELINE config 4 AS BENCHMARKED: st_amd.trainer.JointTrainStep (two captured graphs around an eager ctc_loss on the
    small alphabet; the CTC head's projection, st_ctc_gather / st_ctc_dlogits and its backward GEMMs as HIP kernels) on the
    whole B = 32 batch of config 2 - joint loss, CTC loss, attention CE and EVERY gradient (the head's weight and bias
    included) after a graph REPLAY against the fp64 oracle, which builds the [T, B, V] log-softmax the product never does.
    (The learning rate is ~0 and the clip norm huge, so three steps leave the weights and the gradient scale untouched.)"""
    import torch.nn.functional as func
    import transformer.Models as M
    import transformer.Utils as U
    from st_amd import synthetic
    from st_amd.arena import arena_of
    from st_amd.trainer import JointTrainStep
    from transformer.Loss import CTCAttentionLoss
    from transformer.Optim import ScheduledOptim

    cfg, n = C2, 32
    torch.manual_seed(0)
    model = M.Transformer(U.AttrDict(cfg))
    U.init_parameters(model)
    w = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.eval().cuda()
    x, tokens, in_len, tgt_len, gt = synthetic.make_batch(32, 1000, 50, cfg["feature_dim"], cfg["vocab_size"], seed=0, t_min=500, l_min=25)
    L, T = int(tgt_len.max()), int(in_len.max())
    xg, tg, gg = x[:, :T].cuda(), tokens[:, :L].cuda(), gt[:, :L].cuda()
    torch.manual_seed(0)
    head = CTCAttentionLoss(cfg["d_model"], cfg["vocab_size"], ctc_weight=0.3).cuda()
    H = cfg["n_heads"]

    names = [k for k in w if not k.endswith(".pe")]
    l64 = {k: (v.double().cuda().requires_grad_(True) if k in names else v.double().cuda()) for k, v in w.items()}
    w64 = head.ctc_proj.weight.detach().double().clone().requires_grad_(True)
    b64 = head.ctc_proj.bias.detach().double().clone().requires_grad_(True)
    enc, _ = orc.encoder(l64, xg.double(), in_len, H)
    dec, _, _ = orc.decoder(l64, tg, tgt_len, in_len, enc, H)
    att64 = orc.cross_entropy(func.linear(dec, l64["tgt_word_proj.weight"]), gg)
    logp = func.log_softmax(func.linear(enc, w64, b64), -1).transpose(0, 1)
    ctc64 = func.ctc_loss(logp, gg, in_len, tgt_len, blank=0, reduction="mean", zero_infinity=True)
    truth = 0.3 * ctc64 + 0.7 * att64
    g64 = torch.autograd.grad(truth, [l64[k] for k in names] + [w64, b64], allow_unused=True, retain_graph=True)
    allnames = names + ["ctc_proj.weight", "ctc_proj.bias"]
    tgd = dict(zip(allnames, g64))
    # the noise floor of THIS objective: ctc_loss in float32 - what PyTorch computes for the reference and for the product alike
    # (the alpha / beta recursions run 1,000 frames deep on log-likelihoods of ~ -6,000: float32 leaves ~0.03 .. 0.1 absolute
    # error in the exponents of the occupancies) - with everything else still in float64
    ctc32 = func.ctc_loss(logp.float(), gg, in_len, tgt_len, blank=0, reduction="mean", zero_infinity=True)
    g32 = torch.autograd.grad(0.3 * ctc32.double() + 0.7 * att64, [l64[k] for k in names] + [w64, b64], allow_unused=True)
    floor = {k: (rel(a, t) if t is not None else 0.0) for k, a, t in zip(allnames, g32, g64)}
    del enc, dec, logp, g32
    torch.cuda.empty_cache()

    opt = ScheduledOptim(model, cfg["d_model"], U.AttrDict(n_warmup_steps=10 ** 9))      # lr ~ 1e-15: the weights stay put
    step = JointTrainStep(model, opt, head, max_grad_norm=1e9, use_graph=True, graph_warmup=1)
    for _ in range(3):                                     # eager, capture + replay, replay
        loss, att, ctc, gnorm = step(xg, in_len, tg, tgt_len, gg)
    torch.cuda.synchronize()
    assert step._cap is not None
    assert abs(float(loss) - truth.item()) < 2e-2 * abs(truth.item()), (float(loss), truth.item())
    assert abs(float(ctc) - ctc64.item()) < 2e-2 * abs(ctc64.item()), (float(ctc), ctc64.item())
    assert abs(float(att) - att64.item()) < 2e-2 * abs(att64.item())
    arena = arena_of(model)
    rows = []
    for nme, q in model.named_parameters():
        if "linear_k.bias" in nme or tgd[nme] is None:
            continue
        g = arena.grad_view(q).detach().double()
        assert torch.isfinite(g).all(), nme
        rows.append((rel(g, tgd[nme]), floor[nme], nme, tgd[nme].norm().item()))
    for nme, q in (("ctc_proj.weight", head.ctc_proj.weight), ("ctc_proj.bias", head.ctc_proj.bias)):
        rows.append((rel(q.grad.detach().double(), tgd[nme]), floor[nme], nme, tgd[nme].norm().item()))
    rows.sort(reverse=True)
    flat_g = torch.cat([arena.grad_view(q).detach().double().reshape(-1) for nme, q in model.named_parameters() if "linear_k.bias" not in nme and tgd[nme] is not None])
    flat_t = torch.cat([tgd[nme].reshape(-1) for nme, q in model.named_parameters() if "linear_k.bias" not in nme and tgd[nme] is not None])
    glob = rel(flat_g, flat_t)
    lines = ["# c4_b32: JointTrainStep (graph replay), joint 0.3 CTC + 0.7 attention, 6+6 / d256, B = 32: loss %.5f (oracle %.5f), ctc %.4f (%.4f), att %.4f (%.4f)"
             % (float(loss), truth.item(), float(ctc), ctc64.item(), float(att), att64.item()),
             "gradients: global rel-L2 %.3e; per-tensor rel-L2 (worst first):  HIP path | the fp64 oracle with ctc_loss in fp32 | tensor | |g|" % glob]
    lines += ["  %.3e  %.3e  %-58s %.3e" % r for r in rows]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_c4_b32.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    # encoder + head tensors (everything the CTC branch reaches): 8e-2 or 1.5 x the float32-CTC floor; the decoder's tensors
    # do not see the CTC branch at all - they are the ones test_config2_trainstep_* holds against the bf16 reference (the
    # ill-conditioned q / k projection gradients of the decoder's self-attention sit at 1.3e-1 there and here): 2e-1
    bad = [r for r in rows if r[0] > (2e-1 if r[2].startswith("decoder.") else max(GRAD_TOL_TENSOR, 1.5 * r[1]))]
    assert not bad, "\n".join(lines[:2] + ["outside the bound:"] + ["  %.3e  %.3e  %s" % r[:3] for r in bad])
    med, fmed = sorted(r[0] for r in rows)[len(rows) // 2], sorted(r[1] for r in rows)[len(rows) // 2]
    assert med < max(7e-2, 1.5 * fmed), "\n".join(lines[:12])      # (half the tensors are the decoder's: the c2 tests' median is 5.5e-2)


def test_decode_scoring_at_the_benchmarked_shape_vs_fp64_oracle():
    """The decode path (transformer/Decode.py: KV cache, shared encoder keys, step row chains) at the shape ``bench.py`` times
    it on - config 2's model, the seed-0 batch of 32 utterances of 500..1000 frames, vocabulary 4337, up to 50 steps - with the
    search switched off (``Decode.score_hypotheses``: the tokens are fed, not chosen, so there is no selection among noisy
    scores): the teacher-forced log-probability of one 25..50-token hypothesis per utterance against the oracle's encoder /
    decoder evaluated in float64 on the GPU, and against the same oracle arithmetic under bf16 autocast as the noise floor.
    Bound: every score within 0.1 (scores are ~ -300; measured: rms 0.014, max 0.032 - the bf16 reference shows 0.013 / 0.033)
    and the batch rms within 2x the bf16 reference's."""
    import transformer.Models as M
    import transformer.Utils as U
    from oracle import beam_oracle as bo
    from st_amd import synthetic
    from transformer.Decode import Decode

    cfg = C2
    torch.manual_seed(0)
    model = M.Transformer(U.AttrDict(cfg))
    U.init_parameters(model)
    w = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.eval().cuda()
    x, _, in_len, tgt_len, _ = synthetic.make_batch(32, 1000, 50, cfg["feature_dim"], cfg["vocab_size"], seed=0, t_min=500, l_min=25)
    g = torch.Generator().manual_seed(11)
    hyps = [torch.randint(4, cfg["vocab_size"], (int(n),), generator=g).tolist() for n in tgt_len]
    xg = x.cuda()
    dec = Decode(U.AttrDict(beam_size=10, n_best=1, max_steps=50), "cuda", model=model)
    got = dec.score_hypotheses((xg, in_len), hyps).double().cpu()

    L = max(len(h) for h in hyps)
    prefix = torch.zeros(32, L, dtype=torch.long)
    for b, h in enumerate(hyps):
        prefix[b, :len(h)] = torch.tensor([bo.BOS] + h[:-1])
    lens = torch.tensor([len(h) for h in hyps])
    fed = torch.zeros(32, L, dtype=torch.long)
    for b, h in enumerate(hyps):
        fed[b, :len(h)] = torch.tensor(h)
    live = (torch.arange(L).view(1, -1) < lens.view(-1, 1)).cuda()

    def score(p, xin):
        enc, _ = orc.encoder(p, xin, in_len, cfg["n_heads"])
        out, _, _ = orc.decoder(p, prefix.cuda(), lens, in_len, enc, cfg["n_heads"])
        lp = torch.log_softmax(torch.nn.functional.linear(out, p["tgt_word_proj.weight"]).double(), -1)
        return (lp.gather(2, fed.cuda().unsqueeze(2)).squeeze(2) * live).sum(1).cpu()

    with torch.no_grad():
        truth = score({k: v.double().cuda() for k, v in w.items()}, xg.double())
        with torch.autocast("cuda", dtype=torch.bfloat16):
            floor = score({k: v.float().cuda() for k, v in w.items()}, xg.float())
    err, ferr = got - truth, floor - truth
    rms = lambda v: float((v * v).mean().sqrt())
    report = ["# Decode.score_hypotheses at B = 32, T <= 1000, V = 4337, 25..50 fed tokens per utterance, vs the fp64 oracle on the GPU",
              "scores %.1f .. %.1f   product error: rms %.4f max %.4f   reference under bf16 autocast: rms %.4f max %.4f"
              % (float(truth.min()), float(truth.max()), rms(err), float(err.abs().max()), rms(ferr), float(ferr.abs().max())),
              "per utterance (product | bf16 reference): " + " ".join("%+.3f|%+.3f" % (float(a), float(b)) for a, b in zip(err, ferr))]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_decode_score_b32.txt"), "w") as f:
        f.write("\n".join(report) + "\n")
    assert float(err.abs().max()) < 0.1, report
    assert rms(err) <= max(0.08, 2.0 * rms(ferr)), report


def test_config3_dp_step_with_captured_collectives_matches_the_plain_step():
    """BASELINE config 3 (train_multi.py's model: 12+6 layers, d_model 512 - 152 MB of fp32 gradients) through the data-parallel
    graph step on a one-rank RCCL group with 12 MiB buckets (13 collectives captured in the step graph) against the plain
    single-graph step, same weights and batch, six steps: loss and clip norm must follow each other.  Round 4 found the replays
    of this configuration returning a gradient norm of 3e7 instead of 1.8 with ``ReduceOp.AVG`` (RCCL's pre-multiplied sum keeps
    its scalar in a recycled pool; more than ~8 captured collectives per step and a replay reads a stale one) - the reducer sums
    and divides once instead (st_amd/dp.py)."""
    import copy
    import socket
    import torch.distributed as dist
    import transformer.Models as M
    import transformer.Utils as U
    from st_amd import dp, synthetic
    from st_amd.arena import arena_of
    from st_amd.trainer import TrainStep
    from transformer.Optim import ScheduledOptim

    x, tokens, in_len, tgt_len, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
    n = 8
    xg, tg, gg, in_len, tgt_len = x[:n].cuda(), tokens[:n].cuda(), gt[:n].cuda(), in_len[:n], tgt_len[:n]
    torch.manual_seed(0)
    m0 = M.Transformer(U.AttrDict(C3))
    U.init_parameters(m0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "512")      # (the flight recorder: trainer.drain_collective_watchdog)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        out = {}
        for mode in ("plain", "dp"):
            m = copy.deepcopy(m0).eval().cuda()
            opt = ScheduledOptim(m, C3["d_model"], U.AttrDict(n_warmup_steps=12000))
            red = dp.GradReducer(arena_of(m), bucket_bytes=12 << 20, force=True) if mode == "dp" else None
            step = TrainStep(m, opt, C3["vocab_size"], max_grad_norm=5.0, reducer=red, use_graph=True)
            out[mode] = [tuple(float(v) for v in step(xg, in_len, tg, tgt_len, gg)) for _ in range(6)]
            if red is not None:
                assert step.dp_mode == "in-graph" and len(red.buckets) >= 12, (step.dp_mode, len(red.buckets))
        for (la, ga), (lb, gb) in zip(out["plain"], out["dp"]):
            assert abs(la - lb) <= 1e-3 * abs(la) and abs(ga - gb) <= 3e-2 * abs(ga), (out["plain"], out["dp"])
    finally:
        dist.destroy_process_group()

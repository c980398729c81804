"""transformer/Loss.py drop-in (PyTorch-ROCm per the north star) against the import-generated golden
(tests/golden/loss_optim.npz: the reference's LabelSmoothingLoss evaluated as shipped) and the oracle."""
import os

import numpy as np
import pytest
import torch

import oracle as orc
from transformer.Loss import CTCAttentionLoss, CrossEntropyLoss, LabelSmoothingLoss


def _fx(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "loss_optim.npz")))


@pytest.mark.parametrize("ign", [0, -1, 5])
def test_label_smoothing_matches_reference_golden(golden_dir, ign):
    fx = _fx(golden_dir)
    logits, target = torch.from_numpy(fx["logits"]), torch.from_numpy(fx["target"])
    for weight in (torch.ones(1, 30), None):          # the reference needs a weight tensor; None = no weights
        crit = LabelSmoothingLoss(0.1, 30, weight=weight, ignore_index=ign)
        got = crit(logits, target)
        want = float(fx["loss_ign%d" % ign])
        assert abs(got.item() - want) <= 2e-6 * abs(want)
    # gradient = the oracle restatement's gradient
    a = logits.double().clone().requires_grad_(True)
    b = logits.double().clone().requires_grad_(True)
    LabelSmoothingLoss(0.1, 30, ignore_index=ign)(a, target).backward()
    orc.label_smoothing_loss(b, target, 0.1, ign).backward()
    assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-9)      # the smoothing row is an fp32 buffer (as in the reference)


def test_dense_and_gathered_forms_agree_with_class_weights():
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(40, 17, generator=g, dtype=torch.float64)
    target = torch.randint(0, 17, (40,), generator=g)
    target[::7] = 0
    w = torch.rand(1, 17, generator=g, dtype=torch.float64) + 0.5
    for ign in (0, -1, 4):
        crit = LabelSmoothingLoss(0.2, 17, weight=w, ignore_index=ign)
        prob = crit.one_hot.double().repeat(40, 1)                       # Loss.py:33-37, built densely
        prob.scatter_(1, target.unsqueeze(1), crit.confidence)
        if ign >= 0:
            prob.masked_fill_((target == ign).unsqueeze(1), 0)
        dense = CrossEntropyLoss(w)(logits, prob)
        assert abs(crit(logits, target).item() - dense.item()) < 1e-12
        assert abs(CrossEntropyLoss(w, size_average=False)(logits, prob).item() - dense.item() * 40) < 1e-10


def test_joint_ctc_attention_head():
    g = torch.Generator().manual_seed(5)
    B, T, L, d, V = 3, 40, 6, 16, 12
    enc = torch.randn(B, T, d, generator=g, requires_grad=True)
    dec = torch.randn(B, L, V, generator=g, requires_grad=True)
    tgt = torch.randint(1, V, (B, L), generator=g)
    head = CTCAttentionLoss(d, V, ctc_weight=0.3)
    loss, att, ctc = head(enc, torch.tensor([40, 33, 21]), dec, tgt, torch.tensor([6, 4, 5]), tgt)
    assert torch.isfinite(loss) and abs(loss.item() - (0.3 * ctc.item() + 0.7 * att.item())) < 1e-5
    loss.backward()
    assert enc.grad.abs().sum() > 0 and dec.grad.abs().sum() > 0 and head.ctc_proj.weight.grad.abs().sum() > 0


def test_ctc_head_over_ragged_rows_small_alphabet_vs_dense_reference():
    """The config-4 fast path (CTCAttentionLoss.plan / project_rows / ctc_rows, functional.CtcPlan / CtcProjFn: the projection
    over the ragged encoder rows, ctc_loss on a per-utterance alphabet of blank + own labels, the dense logits gradient rebuilt
    from the small one) under the kernel emulation, against the module's own dense form (Loss.py forward: [T, B, V] log-softmax +
    ctc_loss): value, and the gradients with respect to the encoder rows and the projection's weight / bias.  The batch holds
    what the class mapping has to get right: repeated labels, adjacent repeats, a label equal to the blank id inside a target,
    padded target positions, and an utterance whose frames cannot spell its labels (zero_infinity drops it)."""
    import torch.nn.functional as func
    from tests._emul import emulated_kernels
    from st_amd import functional as F_
    torch.manual_seed(0)
    d, V, blank = 32, 23, 0
    in_len = torch.tensor([30, 17, 25, 6, 22])
    tgt_len = torch.tensor([7, 5, 9, 6, 4])
    L = int(tgt_len.max())
    tgt = torch.tensor([[3, 5, 5, 9, 3, 0, 11, 0, 0],          # adjacent repeat, a repeat further on, the blank id as a label
                        [8, 8, 8, 2, 1, 0, 0, 0, 0],           # a run of three
                        [4, 7, 4, 7, 4, 7, 22, 1, 1],          # alternation; the last two equal
                        [1, 2, 3, 4, 5, 6, 0, 0, 0],           # 6 labels on 6 frames: feasible, no slack
                        [9, 9, 9, 9, 0, 0, 0, 0, 0]])          # 4 equal labels need 7 frames of 22: fine
    tgt_len[3] = 6
    in_len[3] = 6
    tgt2 = tgt.clone()
    tgt2[1, :5] = torch.tensor([8, 8, 8, 8, 8])                 # ... and 5 equal labels need 9 frames of 17: fine; make one infeasible:
    in_len2 = in_len.clone()
    in_len2[1] = 8                                              # 5 equal labels on 8 frames: infinite loss
    for targets, ilen in ((tgt, in_len), (tgt2, in_len2)):
        R = int(ilen.sum())
        enc0 = (torch.randn(R, d) * 0.7).to(torch.bfloat16)
        with emulated_kernels():
            head = CTCAttentionLoss(d, V, ctc_weight=0.3, blank=blank)
            rows = F_.Rows.packed(ilen, "cpu")
            plan = head.plan(targets, tgt_len, ilen, rows)
            head.zero_grad_buffers()
            enc = enc0.clone().requires_grad_(True)
            lp = head.project_rows(enc, plan)
            ctc, g = head.ctc_rows(lp, plan)
            # what JointTrainStep does with the small gradient: stage it, weight the softmax term per utterance
            plan.g_lp.copy_(g)
            torch.mul(plan.finite.to(plan.roww.dtype), 1.0 / plan.B, out=plan.roww)
            plan.roww.div_(plan.tl.to(plan.roww.dtype))
            torch.autograd.backward([lp], [plan.g_lp])
            got = (float(ctc), enc.grad.float().clone(), head.ctc_proj.weight.grad.clone(), head.ctc_proj.bias.grad.clone())
        # dense reference on the same bf16-rounded operands (padded [B, T, d] layout, fp32)
        W = head.ctc_proj.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
        b = head.ctc_proj.bias.detach().clone().requires_grad_(True)
        x = enc0.float().requires_grad_(True)
        T = int(ilen.max())
        pad = torch.zeros(len(ilen), T, d)
        off = 0
        segs = []
        for i, n in enumerate(ilen.tolist()):
            segs.append((i, off, n))
            off += n
        xp = torch.stack([torch.cat([x[o:o + n], torch.zeros(T - n, d)]) for _, o, n in segs])
        logp = func.log_softmax(func.linear(xp, W, b), dim=-1).transpose(0, 1)
        want = func.ctc_loss(logp, targets, ilen, tgt_len, blank=blank, reduction='mean', zero_infinity=True)
        want.backward()
        assert abs(got[0] - want.item()) <= 2e-3 * max(1.0, abs(want.item())), (got[0], want.item())
        for name, a_, r_ in (("d enc", got[1], x.grad), ("dW", got[2], W.grad), ("db", got[3], b.grad)):
            err = float((a_ - r_).norm() / r_.norm().clamp_min(1e-12))
            assert err < 2e-2, (name, err)         # bf16 logits gradient (the kernels' dl is bf16)
        if ilen is in_len2:                         # the infeasible utterance contributes nothing, to the value or to any gradient
            o, n = segs[1][1], segs[1][2]
            assert float(got[1][o:o + n].abs().max()) == 0.0


def test_ctc_plan_refresh_labels_equals_a_fresh_plan():
    """functional.CtcPlan.refresh_labels (called by JointTrainStep on every step: a loader may refill the label buffer in
    place) must leave exactly what a plan built from the new labels holds - in the SAME tensors (a captured step reads them by
    address)."""
    import torch
    from st_amd import functional as F_
    torch.manual_seed(1)
    B, L, V = 5, 9, 12
    tgt_len = torch.tensor([9, 4, 7, 1, 6])
    in_len = torch.tensor([30, 12, 20, 7, 11])
    rows = F_.Rows.packed(in_len, "cpu")
    valid = torch.arange(L).view(1, -1) < tgt_len.view(-1, 1)
    lab_a = torch.where(valid, torch.randint(1, V, (B, L)), torch.zeros(B, L, dtype=torch.int64))
    lab_b = torch.where(valid, torch.randint(1, V, (B, L)), torch.zeros(B, L, dtype=torch.int64))
    lab_b[0, :5] = 3                                     # repeated labels: the feasibility flag depends on the VALUES
    plan = F_.CtcPlan(lab_a, tgt_len, in_len, rows, 0, 16)
    ptrs = [t.data_ptr() for t in (plan.classes, plan.cols, plan.scat, plan.finite)]
    plan.refresh_labels(lab_b)
    fresh = F_.CtcPlan(lab_b, tgt_len, in_len, rows, 0, 16)
    assert ptrs == [t.data_ptr() for t in (plan.classes, plan.cols, plan.scat, plan.finite)]
    for name in ("classes", "cols", "scat", "finite"):
        assert torch.equal(getattr(plan, name), getattr(fresh, name)), name
    assert not torch.equal(plan.classes, F_.CtcPlan(lab_a, tgt_len, in_len, rows, 0, 16).classes)

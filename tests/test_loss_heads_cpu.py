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

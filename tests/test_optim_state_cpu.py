"""-m "not gpu": ScheduledOptim checkpoint format and resume (train.py:110-114: ``optimizer.load_state_dict(
checkpoint['optimizer'])`` then ``optimizer.step(global_step)``).

The HIP path keeps ONE flat Adam state over the parameter arena with the learning rate in a device scalar; its
``state_dict`` must still be the reference's (one entry per parameter, float rate), a reference-format checkpoint
must resume on the arena path, and after a resume the applied rate must follow the Noam schedule (the
rate tensor was orphaned by ``torch.optim.Optimizer.load_state_dict``)."""
import copy

import torch

from tests._emul import emulated_kernels
from tests.test_composition_cpu import _build, _load_c1


def _cfg():
    import transformer.Utils as U
    return U.AttrDict(n_warmup_steps=100)


def _one_step(m, opt, step_no, scale=1.0):
    """A deterministic pseudo-gradient, then ``opt.step(step_no)`` (train.py:46)."""
    opt.zero_grad()
    g = torch.Generator().manual_seed(step_no)
    for p in m.parameters():
        grad = torch.randn(p.shape, generator=g) * scale
        if p.grad is None:
            p.grad = grad
        else:
            p.grad.copy_(grad)
    opt.step(step_no)


def test_state_dict_is_the_reference_format_and_round_trips(golden_dir):
    from transformer.Optim import ScheduledOptim
    from transformer.Utils import learn_rate
    _, w, _ = _load_c1(golden_dir)
    with emulated_kernels():
        m = _build(w)
        opt = ScheduledOptim(m, 128, _cfg())
        assert opt.arena is not None
        for s in (1, 2, 3):
            _one_step(m, opt, s)
        sd = opt.state_dict()
        params = list(m.parameters())
        assert len(sd["state"]) == len(params) == 90 and sd["param_groups"][0]["params"] == list(range(90))
        assert isinstance(sd["param_groups"][0]["lr"], float)
        assert abs(sd["param_groups"][0]["lr"] - learn_rate(128, 100, 3)) < 1e-12
        for i, p in enumerate(params):
            assert sd["state"][i]["exp_avg"].shape == p.shape and float(sd["state"][i]["step"]) == 3.0
        # a plain per-parameter Adam (what the reference builds, Optim.py:11-16) accepts it as is
        ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in params], lr=0.0, betas=(0.9, 0.98), eps=1e-9)
        ref.load_state_dict(copy.deepcopy(sd))
        assert torch.equal(ref.state[ref.param_groups[0]["params"][5]]["exp_avg"], sd["state"][5]["exp_avg"])

        # resume in a fresh model / optimiser: same trajectory as the uninterrupted run, and the rate moves on
        m2 = _build({k: v.detach().clone() for k, v in m.state_dict().items()})
        opt2 = ScheduledOptim(m2, 128, _cfg())
        opt2.load_state_dict(copy.deepcopy(sd))
        assert opt2.optimizer.param_groups[0]["lr"] is opt2.lr_tensor
        for s in (4, 5):
            _one_step(m, opt, s)
            _one_step(m2, opt2, s)
            assert abs(float(opt2.lr_tensor) - learn_rate(128, 100, s)) < 1e-6 * learn_rate(128, 100, s)   # fp32 device scalar
        for (n, a), (_, b) in zip(m.named_parameters(), m2.named_parameters()):
            assert torch.allclose(a, b, rtol=0, atol=1e-7), n


def test_resume_applies_the_scheduled_rate_not_the_checkpointed_one(golden_dir):
    """Adam's first-ish updates move a weight by ~lr: after a resume at step 50 the move must be lr(50), not lr(1)."""
    from transformer.Optim import ScheduledOptim
    from transformer.Utils import learn_rate
    _, w, _ = _load_c1(golden_dir)
    with emulated_kernels():
        m = _build(w)
        opt = ScheduledOptim(m, 128, _cfg())
        _one_step(m, opt, 1)
        sd = opt.state_dict()
        m2 = _build({k: v.detach().clone() for k, v in m.state_dict().items()})
        opt2 = ScheduledOptim(m2, 128, _cfg())
        opt2.load_state_dict(sd)
        before = [p.detach().clone() for p in m2.parameters()]
        _one_step(m2, opt2, 50)
        moved = max((p.detach() - b).abs().max().item() for p, b in zip(m2.parameters(), before))
        assert 0.3 * learn_rate(128, 100, 50) < moved < 3.0 * learn_rate(128, 100, 50)
        assert moved > 10 * learn_rate(128, 100, 1)


def test_cpu_path_checkpoint_loads_on_the_arena_path_and_back(golden_dir):
    from transformer.Optim import ScheduledOptim
    _, w, _ = _load_c1(golden_dir)
    m_cpu = _build(w)
    opt_cpu = ScheduledOptim(m_cpu, 128, _cfg())          # CPU tensors, no emulation: the reference's per-tensor Adam
    assert opt_cpu.arena is None
    _one_step(m_cpu, opt_cpu, 1)
    _one_step(m_cpu, opt_cpu, 2)
    sd = opt_cpu.state_dict()
    with emulated_kernels():
        m = _build({k: v.detach().clone() for k, v in m_cpu.state_dict().items()})
        opt = ScheduledOptim(m, 128, _cfg())
        opt.load_state_dict(copy.deepcopy(sd))
        _one_step(m, opt, 3)
        back = opt.state_dict()
    _one_step(m_cpu, opt_cpu, 3)
    for (n, a), (_, b) in zip(m_cpu.named_parameters(), m.named_parameters()):
        assert torch.allclose(a, b, rtol=0, atol=2e-7), n
    opt_cpu.load_state_dict(back)                          # and the arena path's checkpoint goes back into the per-tensor Adam
    assert float(opt_cpu.optimizer.state[list(m_cpu.parameters())[0]]["step"]) == 3.0


def test_step_captured_clips_when_torch_adam_does_the_update(golden_dir):
    """with weight decay (or amsgrad / maximize) the update is torch's own Adam - the clipping of train.py:45 must
    still happen and the norm must still be returned (it used to fall through to optimizer.step() and return None)."""
    from transformer.Optim import ScheduledOptim
    _, w, _ = _load_c1(golden_dir)
    with emulated_kernels():
        m = _build(w)
        opt = ScheduledOptim(m, 128, _cfg())
        opt.optimizer.param_groups[0]["weight_decay"] = 1e-2
        opt.update_learning_rate(1)
        flat = opt._flat_state()[0]          # the flat parameter over the arena; .grad is the arena's gradient buffer
        g = torch.Generator().manual_seed(3)
        flat.grad.copy_(torch.randn(flat.shape, generator=g))
        norm_before = float(flat.grad.norm())
        assert norm_before > 5.0
        before = flat.detach().clone()
        gnorm = opt.step_captured(grad_norm=True, max_norm=5.0)
        assert gnorm is not None and abs(float(gnorm) - norm_before) < 1e-3 * norm_before
        assert abs(float(flat.grad.norm()) - 5.0) < 1e-3          # clipped in place
        assert float((flat.detach() - before).abs().max()) > 0     # and the update ran
        # below the bound nothing is scaled
        flat.grad.copy_(torch.randn(flat.shape, generator=g) * 1e-4)
        small = flat.grad.detach().clone()
        opt.step_captured(grad_norm=True, max_norm=5.0)
        assert torch.equal(flat.grad, small)

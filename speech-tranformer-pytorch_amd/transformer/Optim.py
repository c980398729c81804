"""Noam-scheduled Adam (drop-in for reference transformer/Optim.py:6-45)."""
import torch
import torch.optim as optim

from transformer.Utils import learn_rate


class ScheduledOptim(object):
    """Adam(betas=(0.9, 0.98), eps=1e-9) whose rate is set from the global step
    before every update: ``lr = d_model^-0.5 * min(step^-0.5, step * warmup^-1.5)``."""

    def __init__(self, model, d_model, config):
        self.lr = 0
        params = list(model.parameters())
        self.arena = None
        if params and all(p.is_cuda for p in params):
            # HIP path: every parameter is a view of one flat buffer (st_amd.arena), so Adam runs as a
            # single fused kernel over it instead of one multi-tensor launch chain over 258 tensors.
            # Same per-element arithmetic as the reference's per-tensor Adam; alignment gaps carry
            # zero gradients and therefore never move.
            from st_amd.arena import arena_of
            self.arena = arena_of(model)
            # the rate lives in a device scalar and the step counter on the device (capturable), so the
            # whole update can be replayed from a HIP graph while the Noam rate still changes per step
            self.lr_tensor = torch.zeros((), dtype=torch.float32, device=self.arena.device)
            self.optimizer = optim.Adam([self.arena.flat_parameter()], lr=self.lr_tensor, betas=(0.9, 0.98), eps=1e-9,
                                        fused=True, capturable=True)
        else:
            self.optimizer = optim.Adam(params, lr=self.lr, betas=(0.9, 0.98), eps=1e-9)
        self.d_model = d_model
        self.n_warmup_steps = config.n_warmup_steps

    def step(self, global_step):
        self.update_learning_rate(global_step)
        self.optimizer.step()

    def step_captured(self):
        """The update alone (rate already set with update_learning_rate) - what a HIP graph captures."""
        self.optimizer.step()

    def zero_grad(self):
        if self.arena is not None:
            self.arena.zero_grads()
        else:
            self.optimizer.zero_grad()

    def state_dict(self):
        return self.optimizer.state_dict()

    def load_state_dict(self, optimizer_state_dict):
        return self.optimizer.load_state_dict(optimizer_state_dict)

    def update_learning_rate(self, global_step):
        self.lr = learn_rate(self.d_model, self.n_warmup_steps, global_step)
        if self.arena is not None:
            self.lr_tensor.fill_(self.lr)
            return
        for group in self.optimizer.param_groups:
            group['lr'] = self.lr

"""Noam-scheduled Adam (drop-in for reference transformer/Optim.py:6-45)."""
import torch.optim as optim

from transformer.Utils import learn_rate


class ScheduledOptim(object):
    """Adam(betas=(0.9, 0.98), eps=1e-9) whose rate is set from the global step
    before every update: ``lr = d_model^-0.5 * min(step^-0.5, step * warmup^-1.5)``."""

    def __init__(self, model, d_model, config):
        self.lr = 0
        params = list(model.parameters())
        fused = all(p.is_cuda for p in params)
        self.optimizer = optim.Adam(params, lr=self.lr, betas=(0.9, 0.98), eps=1e-9,
                                    **({'fused': True} if fused else {}))
        self.d_model = d_model
        self.n_warmup_steps = config.n_warmup_steps

    def step(self, global_step):
        self.update_learning_rate(global_step)
        self.optimizer.step()

    def zero_grad(self):
        self.optimizer.zero_grad()

    def state_dict(self):
        return self.optimizer.state_dict()

    def load_state_dict(self, optimizer_state_dict):
        return self.optimizer.load_state_dict(optimizer_state_dict)

    def update_learning_rate(self, global_step):
        self.lr = learn_rate(self.d_model, self.n_warmup_steps, global_step)
        for group in self.optimizer.param_groups:
            group['lr'] = self.lr

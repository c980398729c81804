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

    def step_captured(self, grad_norm=None, max_norm=None):
        """The update alone (rate already set with update_learning_rate) - what a HIP graph captures.
        With ``grad_norm`` (device scalar: the global gradient norm) and ``max_norm`` on the flat-arena path, gradient
        clipping (train.py:45) and the Adam update are ONE pass over the buffers (``st_adam_clip``: the arithmetic of
        torch's fused Adam, on this optimizer's own state tensors - ``state_dict`` is unchanged)."""
        group = self.optimizer.param_groups[0]
        plain = not (group["weight_decay"] or group["amsgrad"] or group["maximize"])
        if self.arena is None or grad_norm is None or not plain:
            self.optimizer.step()
            return
        from st_amd import native as nv
        p = group["params"][0]
        st = self.optimizer.state[p]
        if len(st) == 0:                 # what torch.optim.Adam creates lazily on its first step (capturable / fused)
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        st["step"].add_(1)
        beta1, beta2 = group["betas"]
        nv.adam_clip(p.data, p.grad, st["exp_avg"], st["exp_avg_sq"], self.lr_tensor, st["step"], grad_norm, max_norm,
                     beta1, beta2, group["eps"])

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

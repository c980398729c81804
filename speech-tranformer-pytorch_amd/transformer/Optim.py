"""Noam-scheduled Adam (drop-in for reference transformer/Optim.py:6-45)."""
import torch
import torch.optim as optim

from transformer.Utils import learn_rate


class ScheduledOptim(object):
    """Adam(betas=(0.9, 0.98), eps=1e-9) whose rate is set from the global step
    before every update: ``lr = d_model^-0.5 * min(step^-0.5, step * warmup^-1.5)``.

    ``state_dict()`` / ``load_state_dict()`` speak the REFERENCE's checkpoint format (Optim.py:26-30 saves the
    state of ``Adam(model.parameters())``: one ``exp_avg`` / ``exp_avg_sq`` / ``step`` entry per parameter in
    ``model.parameters()`` order, a float learning rate) although the HIP path keeps ONE flat state over the
    parameter arena: the arena offsets give the mapping, so a checkpoint written by the reference (or by this
    class on the CPU path) resumes here and vice versa (train.py:110-114)."""

    _allow_cpu_arena = False     # tests/_emul.py: exercise the arena path with emulated kernels on the CPU

    def __init__(self, model, d_model, config):
        self.lr = 0
        params = list(model.parameters())
        self._params = params
        self.arena = None
        self._norm_scratch = None
        if params and (all(p.is_cuda for p in params) or ScheduledOptim._allow_cpu_arena):
            # HIP path: every parameter is a view of one flat buffer (st_amd.arena), so Adam runs as a
            # single fused kernel over it instead of one multi-tensor launch chain over 258 tensors.
            # Same per-element arithmetic as the reference's per-tensor Adam; alignment gaps carry
            # zero gradients and therefore never move.
            from st_amd.arena import arena_of
            self.arena = arena_of(model)
            # the rate lives in a device scalar and the step counter on the device (capturable), so the
            # whole update can be replayed from a HIP graph while the Noam rate still changes per step
            self.lr_tensor = torch.zeros((), dtype=torch.float32, device=self.arena.device)
            on_gpu = self.arena.device.type == "cuda"
            self.optimizer = optim.Adam([self.arena.flat_parameter()], lr=self.lr_tensor, betas=(0.9, 0.98), eps=1e-9,
                                        fused=on_gpu, capturable=on_gpu, foreach=False if not on_gpu else None)
        else:
            self.optimizer = optim.Adam(params, lr=self.lr, betas=(0.9, 0.98), eps=1e-9)
        self.d_model = d_model
        self.n_warmup_steps = config.n_warmup_steps

    def step(self, global_step):
        self.update_learning_rate(global_step)
        self.optimizer.step()

    def _flat_state(self):
        """The flat Adam state over the arena, created the way torch.optim.Adam does lazily on its first step."""
        p = self.optimizer.param_groups[0]["params"][0]
        st = self.optimizer.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return p, st

    def norm_scratch(self, device):
        """st_grad_norm's zeroed block partials + ticket, one per optimizer; TrainStep asks for it before a graph capture."""
        if self._norm_scratch is None or self._norm_scratch.device != torch.device(device):
            from st_amd import native as nv
            self._norm_scratch = nv.grad_norm_scratch(device)
        return self._norm_scratch

    def step_captured(self, grad_norm=None, max_norm=None, grad_scale=1.0):
        """The update alone (rate already set with update_learning_rate) - what a HIP graph captures.
        With ``grad_norm`` and ``max_norm`` on the flat-arena path, gradient clipping (train.py:45) and the Adam update are
        two launches over the buffers: ``st_grad_norm`` (the global norm; also advances the step count) and
        ``st_adam_clip`` (clip + the arithmetic of torch's fused Adam, on this optimizer's own state tensors).
        grad_norm: True = compute it here (returned as a device scalar), or a device scalar already computed.
        grad_scale: the gradient buffer still has to be multiplied by this (st_amd.dp.GradReducer.synchronize(divide=False)
        leaves the rank SUM and returns 1 / world): folded into the norm and the clip coefficient - no pass of its own."""
        group = self.optimizer.param_groups[0]
        plain = not (group["weight_decay"] or group["amsgrad"] or group["maximize"])
        if self.arena is None or grad_norm is None:
            if grad_scale != 1.0:
                for q in self.optimizer.param_groups[0]["params"]:
                    if q.grad is not None:
                        q.grad.mul_(grad_scale)
            self.optimizer.step()
            return None
        if not plain:
            # weight decay / amsgrad / maximize: torch's own Adam does the update, the clipping still happens here (train.py:45
            # clips whatever optimizer follows) - the norm over the flat gradient, the gradient scaled in place, no host sync
            g = self._flat_state()[0].grad
            if grad_scale != 1.0:
                g.mul_(grad_scale)
            if grad_norm is True:
                grad_norm = torch.linalg.vector_norm(g.float())
            if max_norm is not None:
                g.mul_(torch.clamp(float(max_norm) / (grad_norm + 1e-6), max=1.0))
            self.optimizer.step()
            return grad_norm
        from st_amd import native as nv
        p, st = self._flat_state()
        if grad_norm is True:
            self.norm_scratch(p.device)
            grad_norm = nv.grad_norm(p.grad, self._norm_scratch, torch.empty((), dtype=torch.float32, device=p.device), step=st["step"],
                                     grad_scale=grad_scale)
        else:
            st["step"].add_(1)
        beta1, beta2 = group["betas"]
        nv.adam_clip(p.data, p.grad, st["exp_avg"], st["exp_avg_sq"], self.lr_tensor, st["step"], grad_norm, max_norm,
                     beta1, beta2, group["eps"], grad_scale=grad_scale)
        return grad_norm

    def zero_grad(self):
        if self.arena is not None:
            self.arena.zero_grads()
        else:
            self.optimizer.zero_grad()

    # ---- checkpoint format: the reference's (one entry per parameter, float rate) ----------------------------
    def state_dict(self):
        if self.arena is None:
            return self.optimizer.state_dict()
        group = self.optimizer.param_groups[0]
        flat_p = group["params"][0]
        st = self.optimizer.state.get(flat_p, {})
        state = {}
        if len(st):
            step = st["step"].detach().to("cpu", torch.float32).reshape(())
            for i, p in enumerate(self._params):
                off, n = self.arena.offset[id(p)], p.numel()
                state[i] = {"step": step.clone(),
                            "exp_avg": st["exp_avg"][off:off + n].detach().view(p.shape).clone(),
                            "exp_avg_sq": st["exp_avg_sq"][off:off + n].detach().view(p.shape).clone()}
        # the hyper-parameters a plain ``optim.Adam(model.parameters(), ...)`` saves (Optim.py:11-16)
        g = {k: v for k, v in group.items() if k != "params"}
        g.update(lr=float(self.lr), fused=None, capturable=False, foreach=None, params=list(range(len(self._params))))
        return {"state": state, "param_groups": [g]}

    def load_state_dict(self, optimizer_state_dict):
        if self.arena is None:
            return self.optimizer.load_state_dict(optimizer_state_dict)
        groups = optimizer_state_dict["param_groups"]
        n_saved = sum(len(g["params"]) for g in groups)
        if n_saved == 1 and len(self._params) != 1:
            # round-1 format of this class: one flat tensor over the same arena layout
            self.optimizer.load_state_dict(optimizer_state_dict)
            lr = self.optimizer.param_groups[0]["lr"]
            self.lr = float(lr)
        elif n_saved == len(self._params) and len(groups) == 1:
            p, st = self._flat_state()
            saved = optimizer_state_dict["state"]
            keys = groups[0]["params"]
            step = None
            with torch.no_grad():
                st["exp_avg"].zero_()
                st["exp_avg_sq"].zero_()
                for key, q in zip(keys, self._params):
                    ent = saved.get(key)
                    if not ent:
                        continue
                    off, n = self.arena.offset[id(q)], q.numel()
                    if ent["exp_avg"].numel() != n:
                        raise ValueError("ScheduledOptim.load_state_dict: state %r does not match parameter shape %s"
                                         % (key, tuple(q.shape)))
                    st["exp_avg"][off:off + n].copy_(ent["exp_avg"].reshape(-1))
                    st["exp_avg_sq"][off:off + n].copy_(ent["exp_avg_sq"].reshape(-1))
                    step = float(ent["step"]) if step is None else max(step, float(ent["step"]))
                st["step"].fill_(0.0 if step is None else step)
            g = self.optimizer.param_groups[0]
            for k in ("betas", "eps", "weight_decay", "amsgrad", "maximize"):
                if k in groups[0]:
                    g[k] = groups[0][k]
            self.lr = float(groups[0]["lr"])
        else:
            raise ValueError("ScheduledOptim.load_state_dict: checkpoint holds %d parameter states in %d groups; "
                             "this model has %d parameters" % (n_saved, len(groups), len(self._params)))
        # torch's load_state_dict rebuilds param_groups from the saved dict: re-attach the device-resident rate,
        # or every later update would keep the checkpoint's last learning rate (the Noam schedule frozen on resume)
        self.lr_tensor.fill_(self.lr)
        for g in self.optimizer.param_groups:
            g["lr"] = self.lr_tensor

    def update_learning_rate(self, global_step):
        self.lr = learn_rate(self.d_model, self.n_warmup_steps, global_step)
        if self.arena is not None:
            self.lr_tensor.fill_(self.lr)
            return
        for group in self.optimizer.param_groups:
            group['lr'] = self.lr

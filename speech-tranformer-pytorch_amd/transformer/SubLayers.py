"""Position-wise feed-forward sub-layer (drop-in for reference transformer/SubLayers.py:9-28)."""
import torch
import torch.nn as nn
import torch.nn.init as init

from st_amd import functional as F_
from st_amd import rng
from st_amd.arena import arena_of, bundle


class PositionwiseFeedForward(nn.Module):
    """``LN(x + fc2(relu(fc1(x))))`` - fc1+bias+ReLU is one MFMA GEMM with a fused
    epilogue, fc2+bias+residual+LayerNorm another (st_gemm / st_gemm_ln)."""

    def __init__(self, d_model, d_ff, dropout=0.1):
        super(PositionwiseFeedForward, self).__init__()
        self.fc1 = nn.Linear(d_model, d_ff, bias=True)
        self.fc2 = nn.Linear(d_ff, d_model, bias=True)
        self.relu = nn.ReLU()
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.layernorm = nn.LayerNorm(d_model, eps=1e-6)
        init.xavier_normal_(self.fc1.weight.data)
        init.xavier_normal_(self.fc2.weight.data)

    def _st_param_order(self):
        return [self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias,
                self.layernorm.weight, self.layernorm.bias]

    def _st_bind(self, a):
        f1, f2, ln = self.fc1, self.fc2, self.layernorm
        params = self._st_param_order()
        lo, hi = a.span(params)
        return bundle(d_ff=f1.out_features, params=params, lo=lo, hi=hi,
                      w1=a.bf16(f1.weight), b1=a.master(f1.bias), w2=a.bf16(f2.weight), b2=a.master(f2.bias),
                      gamma=a.master(ln.weight), beta=a.master(ln.bias),
                      g_w1=a.grad_view(f1.weight), g_b1=a.grad_view(f1.bias), g_w2=a.grad_view(f2.weight),
                      g_b2=a.grad_view(f2.bias), g_gamma=a.grad_view(ln.weight), g_beta=a.grad_view(ln.bias))

    def _drops(self, device):
        """The two dropout sites of one call (None in eval mode / p = 0)."""
        d1 = rng.site(device, self.dropout1.p) if self.training else None      # SubLayers.py:25
        d2 = rng.site(device, self.dropout2.p) if self.training else None      # SubLayers.py:27 (after the LN)
        return d1, d2

    def forward_rows(self, x, up=None, down=None, pre=None):
        """pre (st_amd.chains.SubPre): the forward values were already computed by a fused launch - only record the
        autograd node."""
        d1, d2 = (pre.drop1, pre.drop2) if pre is not None else self._drops(x.device)
        arena = arena_of(self)
        with arena.scope():
            return F_.FfnFn.apply(x, self.fc1.weight, self, d1, d2, up, down, pre)

    def forward(self, inputs):
        shape = inputs.shape
        x = inputs.reshape(-1, shape[-1]).to(torch.bfloat16)
        return self.forward_rows(x).to(inputs.dtype).view(shape)

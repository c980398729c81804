"""Sinusoidal positional encoding (reference transformer/Embedding.py:7-29)."""
import math

import torch
import torch.nn as nn


class PositionalEncoding(nn.Module):
    """Holds the persistent buffer ``pe[1, max_len, dim]``; ``forward(lengths)``
    returns the first ``max(lengths)`` rows repeated over the batch - the CALLER
    adds them (the fused kernels add ``pe[position]`` inside their epilogues and
    never call this).  ``dropout`` is accepted and, as in the reference, unused."""

    def __init__(self, dropout, dim, max_len=600):
        super(PositionalEncoding, self).__init__()
        pos = torch.arange(0, max_len).unsqueeze(1).float()
        inv = torch.exp(torch.arange(0, dim, 2, dtype=torch.float) * -(math.log(10000.0) / dim))
        table = torch.zeros(max_len, dim)
        table[:, 0::2] = torch.sin(pos * inv)
        table[:, 1::2] = torch.cos(pos * inv)
        self.register_buffer('pe', table.unsqueeze(0))
        self.dropout = nn.Dropout(p=dropout)
        self.dim = dim

    def forward(self, inputs_length, step=None):
        bsz = inputs_length.size(0)
        if step is None:
            return self.pe[:, :int(inputs_length.max())].repeat(bsz, 1, 1)
        return self.pe[:, step].repeat(bsz, 1, 1)

"""The training step of the reference's ``train()`` loops, as one callable.

Sequence (train.py:25-46 / train_multi.py:47-68): trim the batch to its longest
utterance, zero_grad, forward, ``CrossEntropyLoss(ignore_index=0)`` over
``logits.view(-1, V)`` vs ``ground_truth.view(-1)``, backward, [gradient average
over ranks], global-norm clip, Noam-Adam update.  The per-step ``.item()`` syncs of
the reference (train.py:31-32,42) are not reproduced: lengths are taken on the host
(where the loader produced them) and the loss / grad-norm come back as device tensors.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .arena import arena_of
from .dp import GradReducer


def clip_grad_norm_flat(arena, max_norm: float) -> torch.Tensor:
    """``nn.utils.clip_grad_norm_`` (train.py:45) over the flat gradient buffer: one
    reduction + one scale instead of a pass per tensor; padding elements are zero."""
    total = torch.linalg.vector_norm(arena.grad)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    arena.grad.mul_(coef)
    return total


class TrainStep:
    def __init__(self, model: nn.Module, optimizer, vocab_size: int, max_grad_norm: float,
                 reducer: Optional[GradReducer] = None):
        self.model, self.optimizer = model, optimizer
        self.vocab_size, self.max_grad_norm = vocab_size, max_grad_norm
        self.crit = nn.CrossEntropyLoss(ignore_index=0)          # train.py:120
        self.reducer = reducer
        self.global_step = 0

    def __call__(self, inputs, input_lengths, targets, target_lengths, ground_truth):
        """inputs [B, T, F] / targets, ground_truth [B, L] on the GPU; lengths on host or GPU."""
        self.global_step += 1
        t_max, l_max = int(input_lengths.max()), int(target_lengths.max())     # host ints when lengths are CPU tensors
        inputs, targets, ground_truth = inputs[:, :t_max], targets[:, :l_max], ground_truth[:, :l_max]
        self.optimizer.zero_grad()
        logits, _ = self.model(inputs, input_lengths, targets, target_lengths)
        loss = self.crit(logits.contiguous().view(-1, self.vocab_size), ground_truth.contiguous().view(-1))
        loss.backward()
        if self.reducer is not None:
            self.reducer.synchronize()
        arena = arena_of(self.model)
        grad_norm = clip_grad_norm_flat(arena, self.max_grad_norm)
        self.optimizer.step(self.global_step)
        return loss.detach(), grad_norm

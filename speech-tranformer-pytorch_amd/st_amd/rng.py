"""Dropout randomness of the HIP path (training mode of the reference's nn.Dropout sites: Attention.py:89,
SubLayers.py:25,27, Models.py:31).

The kernels draw counter-based masks: a mask is a pure function of (device seed, call-site salt, element
index), see csrc/st_common.cuh.  This module owns the two host-visible pieces:

* the SEED - one int32 element per device, in device memory.  Kernels read it at run time, so a captured HIP
  graph draws fresh masks on every replay as long as ``advance()`` (itself capturable: an in-place add) runs
  once per step; ``trainer.TrainStep`` does that.  Initialised from ``torch.initial_seed()``, so
  ``torch.manual_seed`` makes runs reproducible; ``manual_seed`` here re-seeds explicitly.
* the SALT - a host counter handed out per dropout call, so two sites (or two eager calls of one site) never
  share a mask even between ``advance()`` calls.  A backward pass reuses the forward's (seed tensor, salt).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import native as nv

_seeds: Dict[str, torch.Tensor] = {}
_salt = 0
_manual = None        # seed set by manual_seed(): also used by seed tensors created afterwards
_rank_offset = 0      # data-parallel ranks draw different masks from the same torch seed (set_rank)


def seed_tensor(device) -> torch.Tensor:
    key = str(torch.device(device))
    t = _seeds.get(key)
    if t is None:
        base = torch.initial_seed() if _manual is None else _manual
        t = torch.tensor([(base + _rank_offset) & 0x7FFFFFFF], dtype=torch.int32, device=device)
        _seeds[key] = t
    return t


def set_rank(rank: int) -> None:
    """Called by ``dp.init_from_env``: rank r's stream starts r * 0x9E3779B1 away from rank 0's, so the shards of a
    data-parallel batch do not share dropout masks even when every rank called ``torch.manual_seed`` alike."""
    global _rank_offset
    delta = (int(rank) * 0x9E3779B1 - _rank_offset) & 0x7FFFFFFF
    _rank_offset = (int(rank) * 0x9E3779B1) & 0x7FFFFFFF
    for t in _seeds.values():
        t.add_(delta)


def manual_seed(seed: int) -> None:
    """Re-seed every device's dropout stream (and restart the salt counter)."""
    global _salt, _manual
    _salt, _manual = 0, int(seed)
    for t in _seeds.values():
        t.fill_((int(seed) + _rank_offset) & 0x7FFFFFFF)


def advance(device=None) -> None:
    """Move to the next step's masks (in-place on the device: safe inside a HIP-graph capture)."""
    for key, t in _seeds.items():
        if device is None or key == str(torch.device(device)):
            t.add_(1)


def site(device, p: float) -> Optional["nv.Drop"]:
    """A dropout call site for one forward call (None when p == 0): pass it to the forward kernel wrapper
    and keep it for the backward."""
    global _salt
    if p <= 0.0:
        return None
    _salt = (_salt + 1) & 0xFFFFFFFF
    return nv.Drop(seed_tensor(device), _salt, p)

"""Flat parameter storage for the HIP path.

All parameters of a module tree live in ONE fp32 buffer (`flat`), their
gradients in one fp32 buffer (`grad`), and a bf16 copy the MFMA kernels read in
one bf16 buffer (`shadow`).  ``nn.Parameter.data`` / ``.grad`` become views into
those buffers, so

* the reference's ``state_dict`` keys and shapes are untouched (Models.py:117-145);
* q/k/v projection weights are adjacent -> one fused [3d, d] GEMM operand;
* the fp32 -> bf16 refresh after an optimiser step is one kernel over `flat`;
* data-parallel all-reduce runs on contiguous slices of `grad` with no copies
  (the tensor-fusion buffer Horovod builds for train_multi.py:161-163).

Sized for 288 GB of HBM: nothing is ever freed or re-packed.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn

from . import native

ALIGN = 64  # elements; keeps every slot 16-byte aligned in bf16 and 256-byte aligned in fp32


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class ParamArena:
    def __init__(self, root: nn.Module):
        self.root = root
        order: List[nn.Parameter] = []
        seen = set()
        pad_rows: Dict[int, int] = {}
        for mod in root.modules():
            plist = mod._st_param_order() if hasattr(mod, "_st_param_order") else list(mod._parameters.values())
            for p in plist:
                if p is not None and id(p) not in seen:
                    seen.add(id(p))
                    order.append(p)
            for p, rows in getattr(mod, "_st_row_padding", lambda: [])():
                pad_rows[id(p)] = rows
        if not order:
            raise ValueError("ParamArena: module has no parameters")
        dev = order[0].device
        self._require_gpu(dev)
        self.device = dev
        self.params = order
        self.offset: Dict[int, int] = {}
        self.size: Dict[int, int] = {}
        total = 0
        for p in order:
            n = p.numel()
            if id(p) in pad_rows:  # e.g. vocabulary rows padded to a multiple of 8
                n = pad_rows[id(p)] * (p.numel() // p.shape[0])
            self.offset[id(p)] = total
            self.size[id(p)] = n
            total += _round_up(n, ALIGN)
        self.total = total
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(total, dtype=torch.bfloat16, device=dev)
        with torch.no_grad():
            for p in order:
                v = self.master(p)
                v.copy_(p.data)
                p.data = v
                if p.grad is not None:
                    self.grad_view(p).copy_(p.grad)
                    p.grad = self.grad_view(p)
        self._depth = 0
        self._derived = []      # objects with .refresh(): layouts derived from the bf16 shadow (st_amd.chains)
        self._grad_ready_cb: Optional[Callable[[int, int], None]] = None
        for mod in root.modules():
            if hasattr(mod, "_st_bind"):
                mod._st_arena = self
                mod._st = mod._st_bind(self)

    @staticmethod
    def _require_gpu(dev) -> None:
        if dev.type != "cuda":
            raise RuntimeError("the HIP path needs the module on a GPU (call .cuda() first); there is no CPU fallback")

    # ---- views ---------------------------------------------------------------------------------
    def _view(self, buf, p, rows=None):
        off = self.offset[id(p)]
        if rows is None:
            return buf[off:off + p.numel()].view(p.shape)
        cols = p.numel() // p.shape[0] if p.dim() > 1 else 1
        shape = (rows, cols) if p.dim() > 1 else (rows,)
        return buf[off:off + rows * cols].view(shape)

    def master(self, p, rows=None):
        return self._view(self.flat, p, rows)

    def grad_view(self, p, rows=None):
        return self._view(self.grad, p, rows)

    def bf16(self, p, rows=None):
        return self._view(self.shadow, p, rows)

    def span(self, params) -> tuple:
        lo = min(self.offset[id(p)] for p in params)
        hi = max(self.offset[id(p)] + _round_up(self.size[id(p)], ALIGN) for p in params)
        return lo, hi

    def valid(self) -> bool:
        p = self.params[0]
        return p.data_ptr() == self.flat.data_ptr() + 4 * self.offset[id(p)] and \
            self.params[-1].data_ptr() == self.flat.data_ptr() + 4 * self.offset[id(self.params[-1])]

    # ---- bf16 shadow ---------------------------------------------------------------------------
    def refresh(self) -> None:
        """fp32 master -> bf16 shadow (one streaming kernel, ~80 MB at config 2)."""
        native.cast_bf16(self.flat, self.shadow)
        for d in self._derived:
            d.refresh()

    class _Scope:
        def __init__(self, arena):
            self.arena = arena

        def __enter__(self):
            if self.arena._depth == 0:
                self.arena.refresh()
            self.arena._depth += 1

        def __exit__(self, *exc):
            self.arena._depth -= 1

    def scope(self):
        """Outermost module forward refreshes the shadow once; nested calls reuse it."""
        return ParamArena._Scope(self)

    # ---- gradients -----------------------------------------------------------------------------
    def attach_grads(self, params, lo: int, hi: int) -> None:
        """Make ``p.grad`` the arena views, zero-filled if the slot held no gradient
        (``zero_grad(set_to_none=True)`` semantics); existing arena grads accumulate."""
        fresh = [p for p in params if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * self.offset[id(p)]]
        if not fresh:
            return
        if len(fresh) == len(params):
            self.grad[lo:hi].zero_()
        else:
            for p in fresh:
                self.grad_view(p, None).zero_()
        for p in fresh:
            p.grad = self.grad_view(p)

    def zero_grads(self) -> None:
        """One memset over the whole gradient buffer, with every ``p.grad`` (re)attached -
        the arena's ``optimizer.zero_grad()`` (train.py:37)."""
        if self.grad.is_cuda:
            native.zero_(self.grad)
        else:
            self.grad.zero_()
        base = self.grad.data_ptr()
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != base + 4 * self.offset[id(p)]:
                p.grad = self.grad_view(p)

    def flat_parameter(self) -> nn.Parameter:
        """The whole arena as ONE parameter (grad = the flat gradient buffer): lets the
        optimiser update every weight with a single fused kernel."""
        fp = nn.Parameter(self.flat, requires_grad=False)
        fp.grad = self.grad
        return fp

    def set_grad_ready_callback(self, cb) -> None:
        self._grad_ready_cb = cb

    def grads_ready(self, lo: int, hi: int) -> None:
        if self._grad_ready_cb is not None:
            self._grad_ready_cb(lo, hi)


def arena_of(mod: nn.Module) -> ParamArena:
    """The arena that owns ``mod``'s parameters, building one rooted at ``mod`` if needed."""
    a = getattr(mod, "_st_arena", None)
    if a is None or not a.valid():
        a = ParamArena(mod)
    return a


def bundle(**kw) -> SimpleNamespace:
    return SimpleNamespace(**kw)

"""ctypes binding of libst_hip.so - the C-ABI declared in include/st_hip.h.

Each Python wrapper validates tensor dtype / layout on the host (the kernels
themselves only see raw pointers), passes ``torch.cuda.current_stream()`` and
raises on a non-zero status.  There is deliberately NO fallback: if the library
is missing, or a tensor is not on a GPU, the call raises.

Tensor conventions: activations are 2-D row matrices with ``stride(1) == 1``;
column slices of a wider matrix (``qkv[:, :d]``) are fine - the leading
dimension is taken from ``stride(0)``.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import build as _build

EPI_BF16, EPI_BF16_RELU, EPI_F32, EPI_BF16_MASK, EPI_BF16_ADD, EPI_F32_ATOMIC, EPI_F32_ATOMIC_T, EPI_BF16_DELTA = range(8)

_c_int, _c_float, _c_void_p, _c_ll = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_longlong
_c_uint = ctypes.c_uint
_c_long = ctypes.c_long

# name -> argtypes (restype is int everywhere); must mirror include/st_hip.h exactly.
SIGNATURES = {
    "st_version": [],
    "st_env_refresh": [],
    "st_clock_probe": [_c_void_p, _c_void_p, _c_int, _c_int],
    "st_gemm": [_c_void_p, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int,
                _c_int, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_uint, _c_int, _c_float, _c_void_p],
    "st_gemm_stacked": [_c_void_p, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_int,
                        _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_uint, _c_int,
                        _c_float, _c_int, _c_long, _c_long],
    "st_gemm_ws": [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p,
                   _c_int, _c_void_p, _c_uint, _c_int, _c_float, _c_int, _c_long, _c_long],
    "st_wgrad_group": [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                       _c_void_p, _c_void_p, _c_void_p, _c_void_p],
    "st_wgrad_wide": [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                      _c_void_p, _c_void_p, _c_void_p, _c_void_p],
    "st_gemm_ln": [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_int,
                   _c_void_p, _c_void_p, _c_float, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p,
                   _c_void_p, _c_void_p, _c_void_p, _c_uint, _c_int, _c_float, _c_int],
    "st_wfrag_depth": [],
    "st_wfrag_build": [_c_void_p, _c_void_p, _c_int],
    "st_row_chain": [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_float, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p,
                     _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                     _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_uint, _c_int, _c_float, _c_uint, _c_int, _c_float,
                     _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_ll, _c_float],
    "st_row_chain_mask_words": [_c_int, _c_int],
    "st_row_chain512": [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_float, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p,
                        _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                        _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_uint, _c_int, _c_float, _c_uint, _c_int, _c_float,
                        _c_int, _c_void_p, _c_void_p, _c_int, _c_float],
    "st_row_chain512_mask_words": [_c_int, _c_int],
    "st_gemm_kscale": [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_int,
                       _c_float],
    "st_gemm_splitk": [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                       _c_void_p, _c_ll],
    "st_row_chain_bwd": [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p,
                         _c_void_p, _c_void_p, _c_void_p, _c_uint, _c_int, _c_float, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                         _c_void_p, _c_int, _c_void_p, _c_float, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                         _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_ll, _c_void_p, _c_ll],
    "st_zero": [_c_void_p, _c_void_p, _c_ll],
    "st_row_chain_bwd_colsum_rows": [_c_int, _c_int, _c_int, _c_int],
    "st_colsum_fold": [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p],
    "st_row_chain512_bwd": [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p,
                            _c_void_p, _c_void_p, _c_void_p, _c_uint, _c_int, _c_float, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                            _c_int, _c_void_p, _c_float, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                            _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p],
    "st_gemm_lnbwd": [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_int,
                      _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p,
                      _c_void_p, _c_uint, _c_int, _c_float],
    "st_ln_bwd": [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int,
                  _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_uint, _c_int, _c_float, _c_float],
    "st_attn_tile_rows": [_c_int, _c_int, _c_int, _c_int, _c_int],
    "st_attn_fwd": [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p,
                    _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                    _c_float, _c_void_p, _c_int, _c_void_p, _c_uint, _c_int, _c_float, _c_int],
    "st_attn_f1_applicable": [_c_int, _c_int, _c_int, _c_int],
    "st_attn_f1_fwd": [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_float, _c_void_p, _c_void_p,
                       _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int,
                       _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                       _c_int, _c_int, _c_int, _c_float, _c_void_p, _c_int, _c_void_p, _c_uint, _c_int, _c_float],
    "st_attn_sf1_fwd": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_uint, _c_int,
                        _c_float, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_float, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                        _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int,
                        _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                        _c_int, _c_float, _c_void_p, _c_int, _c_void_p, _c_uint, _c_int, _c_float],
    "st_attn_bwd": [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p,
                    _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p,
                    _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                    _c_float, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_uint, _c_int, _c_float, _c_int, _c_void_p, _c_ll],
    "st_attn_bwd_split_kib": [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int],
    "st_row_index": [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p],
    "st_pack_rows": [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p],
    "st_ctc_gather": [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p],
    "st_ctc_dlogits": [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int,
                       _c_void_p, _c_void_p, _c_void_p, _c_int],
    "st_attn_probs": [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                      _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int],
    "st_attn_dense_fwd": [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p,
                          _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _c_void_p, _c_uint, _c_int, _c_float],
    "st_attn_dense_bwd": [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p,
                          _c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                          _c_float, _c_void_p, _c_uint, _c_int, _c_float],
    "st_feat_stack": [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                      _c_void_p, _c_void_p, _c_int, _c_void_p, _c_int],
    "st_unpack_rows": [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p],
    "st_pack_grad": [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int],
    "st_embed_pe_fwd": [_c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p,
                        _c_void_p, _c_void_p],
    "st_embed_bwd": [_c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_int,
                     _c_void_p, _c_int],
    "st_cast_bf16": [_c_void_p, _c_void_p, _c_void_p, _c_ll],
    "st_embed_step": [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int],
    "st_decode_self_attn": [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                            _c_int, _c_float],
    "st_beam_advance": [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p,
                        _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p,
                        _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int],
    "st_ce_fwd": [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p],
    "st_ce_bwd": [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p,
                  _c_void_p, _c_int],
    "st_zero_tails": [_c_void_p, _c_void_p, _c_int],
    "st_grad_norm_blocks": [],
    "st_grad_norm": [_c_void_p, _c_void_p, _c_ll, _c_void_p, _c_void_p, _c_void_p, _c_float],
    "st_cache_reorder": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int],
    "st_adam_clip": [_c_void_p, _c_ll, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                     _c_float, _c_float, _c_float, _c_float, _c_float],
    "st_probe_tr16": [_c_void_p, _c_void_p, _c_void_p],
    "st_probe_mfma": [_c_void_p, _c_void_p, _c_void_p, _c_void_p],
}

_LIB = None
_TIMING = None   # list of (kernel name, tag, algorithmic bytes, start event, end event) while bench.py profiles a step
_TAG = None
_TAG_BYTES = None


class _Timed:
    """Per-launch HIP-event bracket on torch's current stream (the stream the kernels are
    launched on); off unless ``timing_start()`` was called - bench.py's roofline pass."""

    def __init__(self, name, fn):
        self.name, self.fn = name, fn

    def __call__(self, *args):
        global _TAG, _TAG_BYTES
        if _TIMING is None:
            return self.fn(*args)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = self.fn(*args)
        e.record()
        _TIMING.append((self.name, _TAG, _TAG_BYTES, s, e))
        _TAG = _TAG_BYTES = None
        return rc


class _Lib:
    pass


def timing_start() -> None:
    global _TIMING
    _TIMING = []


def timing_stop():
    """-> list of (name, tag, milliseconds, algorithmic HBM bytes of the launch or None); synchronises the device."""
    global _TIMING
    torch.cuda.synchronize()
    out = [(n, t, s.elapsed_time(e), nb) for n, t, nb, s, e in (_TIMING or [])]
    _TIMING = None
    return out


def _io_bytes(io) -> float:
    """Bytes of a launch's operands, each counted once: items are tensors (whole), (tensor, rows) (the first ``rows`` rows
    of a row matrix / elements of a vector), plain byte counts, or None."""
    n = 0.0
    for item in io:
        if item is None:
            continue
        if isinstance(item, (int, float)):
            n += item
            continue
        t, rows = item if isinstance(item, tuple) else (item, None)
        if t is None:
            continue
        if t.dim() == 2:
            n += (t.shape[0] if rows is None else min(int(rows), t.shape[0])) * t.shape[1] * t.element_size()
        else:
            n += (t.numel() if rows is None else min(int(rows), t.numel())) * t.element_size()
    return n


def _tag(*info, io=None) -> None:
    """Label the next launch for bench.py's per-launch timing pass: ``info`` = (class, shape numbers ...), ``io`` = the
    operands the launch must move through HBM once (its ALGORITHMIC traffic - see _io_bytes)."""
    global _TAG, _TAG_BYTES
    if _TIMING is not None:
        _TAG = info
        _TAG_BYTES = _io_bytes(io) if io is not None else None


def lib_path() -> str:
    return _build.LIB


ABI_VERSION = 4      # == ST_ABI_VERSION in include/st_hip.h == st_version() of the library this binding was written against


def load(build_if_missing: bool = True):
    """dlopen libst_hip.so (building it first if hipcc is available)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("ST_HIP_LIB") or lib_path()      # (development: A/B a library built from variant sources)
    if not os.path.exists(path):
        if not build_if_missing:
            raise RuntimeError("libst_hip.so not built: run `python __graft_entry__.py` (build())")
        _build.build_lib()
    cdll = ctypes.CDLL(path)
    # the ABI version first: a library built from other sources would be called with shifted arguments
    try:
        cdll.st_version.restype = _c_int
        ver = int(cdll.st_version())
    except AttributeError:
        ver = -1
    if ver != ABI_VERSION:
        raise RuntimeError("libst_hip.so at %s has ABI version %d, this binding needs %d: rebuild it (python __graft_entry__.py)"
                           % (path, ver, ABI_VERSION))
    missing = [name for name in SIGNATURES if not hasattr(cdll, name)]
    if missing:
        raise RuntimeError("libst_hip.so at %s lacks %s: rebuild it (python __graft_entry__.py)" % (path, ", ".join(missing)))
    lib = _Lib()
    for name, argtypes in SIGNATURES.items():
        fn = getattr(cdll, name)
        fn.argtypes = argtypes
        fn.restype = _c_int
        setattr(lib, name, _Timed(name, fn))
    lib._cdll = cdll
    _LIB = lib
    return lib


def _stream() -> int:
    """Raw handle of torch's current stream on the current device (torch.cuda.current_stream() builds a Stream object
    per call: ~8 us of host time, measurable in the eager decode loop's ~100 launches per step)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def sustained_clock_mhz(device=None, iters: int = 30000) -> float:
    """The shader clock (MHz) the device holds under ~2 ms of chip-wide dense MFMA work (st_clock_probe): median over the
    compute units' own s_memtime / s_memrealtime ratios.  bench.py records it next to its roofline - the same commit measures
    4-8 % apart on different boxes, and most of that is this number."""
    dev = torch.device("cuda" if device is None else device)
    n_wg = int(torch.cuda.get_device_properties(dev).multi_processor_count)
    out = torch.zeros(2 * n_wg, dtype=torch.int64, device=dev)
    for _ in range(2):       # (the first launch also pays the clock's ramp)
        _check(load().st_clock_probe(_stream(), out.data_ptr(), n_wg, int(iters)), "st_clock_probe")
    torch.cuda.synchronize(dev)
    t = out.view(n_wg, 2).double().cpu()
    return float((100.0 * t[:, 0] / t[:, 1].clamp_min(1)).median())


def env_refresh() -> None:
    """After changing ST_ATTN_FWD64 / ST_ATTN_XS / ST_ATTN_BWD64 in os.environ: make the library re-read them (it caches the
    switches at its first attention call; the production dispatch never calls getenv)."""
    load().st_env_refresh()


def _check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        raise ValueError("%s: unsupported argument (code %d) - see include/st_hip.h" % (what, rc))
    raise RuntimeError("%s: HIP launch failed (hipError_t %d)" % (what, rc))


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _mat(t: torch.Tensor, dtype, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: the HIP path has no CPU fallback" % name)
    if t.dtype != dtype:
        raise TypeError("%s: expected %s, got %s" % (name, dtype, t.dtype))
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("%s: expected a 2-D row matrix with unit column stride, got shape %s stride %s"
                         % (name, tuple(t.shape), t.stride()))


def _vec(t: Optional[torch.Tensor], dtype, n: int, name: str) -> None:
    if t is None:
        return
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous() or t.numel() < n:
        raise ValueError("%s: expected contiguous %s[%d] on the GPU" % (name, dtype, n))


BF16, F32, I32, I64 = torch.bfloat16, torch.float32, torch.int32, torch.int64


class Drop:
    """One dropout call site (nn.Dropout in training mode): ``seed`` is a 1-element int32 DEVICE tensor (the
    kernels read it, so a captured HIP graph draws new masks once it is advanced), ``salt`` a per-site
    constant, ``p`` the drop probability - quantised to 8 bits: keep iff draw >= round(256 p), survivors scaled
    by 256 / (256 - thresh).  The backward of an op takes the same Drop object as its forward."""
    __slots__ = ("seed", "salt", "thresh", "scale")

    def __init__(self, seed: torch.Tensor, salt: int, p: float):
        if not (seed.is_cuda and seed.dtype == torch.int32 and seed.numel() == 1):
            raise ValueError("Drop: seed must be a 1-element int32 tensor on the GPU")
        if not 0.0 <= p < 1.0:
            raise ValueError("Drop: p must be in [0, 1)")
        self.seed, self.salt = seed, int(salt) & 0xFFFFFFFF
        self.thresh = min(255, int(round(256.0 * p)))
        self.scale = 256.0 / (256 - self.thresh)

    @property
    def p_effective(self) -> float:
        return self.thresh / 256.0


def _drop(d: Optional["Drop"]):
    if d is None or d.thresh == 0:
        return None, 0, 0, 1.0
    return d.seed.data_ptr(), d.salt, d.thresh, d.scale


# ------------------------------------------------------------------------------------------------
def gemm(X, Y, out, bias=None, aux=None, epi=EPI_BF16, x_cmajor=False, y_cmajor=False, splits=1,
         m=None, n=None, kc=None, drop=None, delta=None, head_dim=0, stack=None, aux2=None):
    """out[i][j] (+)= sum_c X(i,c) Y(j,c).  ``*_cmajor``: that tensor is stored [c, rows].
    ``stack = (blocks, stride, bias_stride)``: Y (and bias) is the FIRST of ``blocks`` equally shaped blocks lying
    ``stride`` (``bias_stride``) elements apart - the same weight of consecutive identical layers in the parameter
    arena; the operand is their vertical stack (more output columns forward, a longer contraction for dgrad).
    EPI_BF16_DELTA: additionally delta[h][i] = sum over head h's ``head_dim`` columns of out(i, .) * (aux + aux2)(i, .)
    (fp32 [N / head_dim, M]) - the attention backward's rowsum(dO * O), produced by the GEMM that produces dO;
    ``aux2`` (optional) = attn_fwd's ``ores``, the part of O its bf16 rounding dropped."""
    _mat(X, BF16, "X"), _mat(Y, BF16, "Y")
    _mat(out, F32 if epi in (EPI_F32, EPI_F32_ATOMIC, EPI_F32_ATOMIC_T) else BF16, "out")
    M = m if m is not None else (X.shape[1] if x_cmajor else X.shape[0])
    N = n if n is not None else (Y.shape[1] if y_cmajor else Y.shape[0])
    Kc = kc if kc is not None else (X.shape[0] if x_cmajor else X.shape[1])
    blocks, block_rows, y_stride, b_stride = 1, 0, 0, 0
    if stack is not None:
        blocks, y_stride, b_stride = stack
        block_rows = Y.shape[0]
        if y_cmajor:
            Kc = blocks * block_rows
        else:
            N = blocks * block_rows
    need = (N, M) if epi == EPI_F32_ATOMIC_T else (M, N)
    if out.shape[0] < need[0] or out.shape[1] < need[1]:
        raise ValueError("gemm: out %s too small for %dx%d" % (tuple(out.shape), need[0], need[1]))
    _vec(bias, F32, N // blocks if not y_cmajor else N, "bias")
    if epi == EPI_BF16_DELTA:
        if head_dim not in (32, 64, 128) or N % head_dim:
            raise ValueError("gemm: EPI_BF16_DELTA needs head_dim in {32, 64, 128} dividing N")
        _vec(delta, F32, (N // head_dim) * M, "delta")
        bias, splits = delta, head_dim          # the C-ABI passes them in the bias / splits slots of this epilogue
    ldaux = 0
    if epi in (EPI_BF16_MASK, EPI_BF16_ADD, EPI_BF16_DELTA):
        _mat(aux, BF16, "aux")
        ldaux = aux.stride(0)
        if aux2 is not None:
            _mat(aux2, BF16, "aux2")
            if epi != EPI_BF16_DELTA or aux2.stride(0) != ldaux or aux2.shape != aux.shape:
                raise ValueError("gemm: aux2 goes with EPI_BF16_DELTA and must have aux's shape and stride")
    esz = 4 if epi in (EPI_F32, EPI_F32_ATOMIC, EPI_F32_ATOMIC_T) else 2
    _tag("gemm", int(x_cmajor), int(y_cmajor), M, N, Kc, epi,
         io=(2.0 * M * Kc, 2.0 * N * Kc, float(esz) * M * N, 2.0 * M * N if aux is not None else 0, 2.0 * M * N if aux2 is not None else 0))
    dropargs = _drop(drop) if epi in (EPI_BF16_RELU, EPI_BF16_MASK) else _drop(None)
    if stack is None:
        rc = load().st_gemm(_stream(), int(x_cmajor), int(y_cmajor), X.data_ptr(), X.stride(0), Y.data_ptr(),
                            Y.stride(0), out.data_ptr(), out.stride(0), M, N, Kc, _p(bias), _p(aux), ldaux, epi, splits,
                            *dropargs, _p(aux2))
    else:
        rc = load().st_gemm_stacked(_stream(), int(x_cmajor), int(y_cmajor), X.data_ptr(), X.stride(0), Y.data_ptr(),
                                    Y.stride(0), out.data_ptr(), out.stride(0), M, N, Kc, _p(bias), _p(aux), ldaux,
                                    epi, splits, *dropargs, block_rows, y_stride, b_stride)
    _check(rc, "st_gemm")
    return out


def gemm_ws(X, W, out, bias=None, relu=False, drop=None, stack=None):
    """out = act(X W^T + bias) through the weight-stationary streaming kernel (K = 256, N % 256 == 0); stack as in gemm()."""
    _mat(X, BF16, "X"), _mat(W, BF16, "W"), _mat(out, BF16, "out")
    M, K = X.shape
    N = W.shape[0]
    blocks, block_rows, w_stride, b_stride = 1, 0, 0, 0
    if stack is not None:
        blocks, w_stride, b_stride = stack
        block_rows = W.shape[0]
        N = blocks * block_rows
    if out.shape[0] < M or out.shape[1] < N:
        raise ValueError("gemm_ws: out %s too small for %dx%d" % (tuple(out.shape), M, N))
    _vec(bias, F32, N // blocks, "bias")
    _tag("gemm", 0, 0, M, N, K, EPI_BF16_RELU if relu else EPI_BF16, io=(2.0 * M * K, 2.0 * N * K, 2.0 * M * N))
    rc = load().st_gemm_ws(_stream(), X.data_ptr(), X.stride(0), W.data_ptr(), W.stride(0), out.data_ptr(), out.stride(0),
                           M, N, K, _p(bias), int(relu), *(_drop(drop) if relu else _drop(None)), block_rows, w_stride, b_stride)
    _check(rc, "st_gemm_ws")
    return out


def gemm_kscale(X, W, out, bias, col_lo, col_hi, scale):
    """out = X W^T + bias with columns [col_lo, col_hi) scaled by `scale` in fp32 before the rounding (st_gemm_kscale): a
    q | k | v projection whose key block leaves pre-scaled (attn_fwd's k_prescaled)."""
    _mat(X, BF16, "X"), _mat(W, BF16, "W"), _mat(out, BF16, "out")
    M, K = X.shape
    N = W.shape[0]
    _vec(bias, F32, N, "bias")
    _tag("gemm", 0, 0, M, N, K, EPI_BF16, io=(2.0 * M * K, 2.0 * N * K, 2.0 * M * N, 0, 0))
    _check(load().st_gemm_kscale(_stream(), X.data_ptr(), X.stride(0), W.data_ptr(), W.stride(0), out.data_ptr(), out.stride(0),
                                 M, N, K, _p(bias), int(col_lo), int(col_hi), float(scale)), "st_gemm_kscale")
    return out


def ws_ok(M, N, K):
    """Shapes the weight-stationary kernel takes, and where it pays (measured on MI355X, tools/dev/ws_bench.py: 1.2-1.4x
    over the tiled kernel from ~16 k rows up, on par at 7 k, so only encoder-sized row counts are routed to it)."""
    return K == 256 and N % 256 == 0 and M >= 12288


def wgrad_group(problems, wide=False):
    """One launch for several weight gradients.  problems: iterable of (X [tokens, K_in] bf16, dY [tokens, N_out]
    bf16, gW f32 [>= N_out, K_in], gB f32 [N_out] or None, splits, N_out) - the argument tuple of
    ``functional.wgrad``; each is gW[n][k] += sum_m dY[m][n] X[m][k] (and gB[n] += sum_m dY[m][n]).
    wide: the 256 x 256-tile kernel for encoder-sized token counts (st_wgrad_wide) instead of the 128 x 128 one."""
    problems = list(problems)
    n = len(problems)
    if n == 0:
        return
    PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
    X, dY, dW, dB = PA(), PA(), PA(), PA()
    ldx, lddy, lddw, tokens, k_in, n_out, splits = IA(), IA(), IA(), IA(), IA(), IA(), IA()
    for q, (x, dy, gw, gb, sp, rows) in enumerate(problems):
        _mat(x, BF16, "X"), _mat(dy, BF16, "dY"), _mat(gw, F32, "gW")
        if x.shape[0] != dy.shape[0] or gw.shape[0] < rows or gw.shape[1] < x.shape[1]:
            raise ValueError("wgrad_group: problem %d has inconsistent shapes" % q)
        _vec(gb, F32, rows, "gB")
        X[q], dY[q], dW[q], dB[q] = x.data_ptr(), dy.data_ptr(), gw.data_ptr(), _p(gb)
        ldx[q], lddy[q], lddw[q] = x.stride(0), dy.stride(0), gw.stride(0)
        tokens[q], k_in[q], n_out[q], splits[q] = x.shape[0], x.shape[1], rows, sp
    name = "st_wgrad_wide" if wide else "st_wgrad_group"
    _tag(name[3:], n, sum(2.0 * pr[0].shape[0] * pr[0].shape[1] * pr[5] for pr in problems),
         io=[2.0 * pr[0].shape[0] * (pr[0].shape[1] + pr[5]) + 8.0 * pr[5] * pr[0].shape[1] for pr in problems])   # X, dY in; gW read + written (fp32)
    rc = getattr(load(), name)(_stream(), n, X, ldx, dY, lddy, dW, lddw, dB, tokens, k_in, n_out, splits)
    _check(rc, name)


def gemm_ln(X, W, bias, res, gamma, beta, out, xhat, rstd, eps=1e-6, relu=False, pe=None, pos=None, pre=None,
            drop=None, drop_where=0):
    """drop_where: 1 = dropout before the LayerNorm (after bias / ReLU), 2 = on the LayerNorm output."""
    _mat(X, BF16, "X"), _mat(W, BF16, "W"), _mat(out, BF16, "out")
    M, K = X.shape
    N = W.shape[0]
    if W.stride(0) != K or W.shape[1] != K:
        raise ValueError("gemm_ln: W must be a contiguous [N, K] matrix")
    _vec(bias, F32, N, "bias"), _vec(gamma, F32, N, "gamma"), _vec(beta, F32, N, "beta")
    if res is not None:
        _mat(res, BF16, "res")
    if xhat is not None:
        _mat(xhat, BF16, "xhat")
        assert xhat.stride(0) == N
    if pre is not None:
        _mat(pre, BF16, "pre")
        assert pre.stride(0) == N
    _vec(rstd, F32, M, "rstd")
    if pe is not None:
        _mat(pe, F32, "pe")
        assert pe.stride(0) == N
        _vec(pos, I32, M, "pos")
    _tag("gemm_ln", M, N, K, io=((X, M), W, (res, M), (out, M), (xhat, M), (pre, M), (rstd, M)))
    rc = load().st_gemm_ln(_stream(), X.data_ptr(), X.stride(0), W.data_ptr(), M, N, K, bias.data_ptr(), _p(res),
                           0 if res is None else res.stride(0), gamma.data_ptr(), beta.data_ptr(), float(eps),
                           int(relu), _p(pe), _p(pos), out.data_ptr(), out.stride(0), _p(xhat), _p(rstd), _p(pre),
                           *_drop(drop if drop_where in (1, 2) else None), int(drop_where))
    _check(rc, "st_gemm_ln")
    return out


_WFRAG_DEPTH = None


def wfrag_depth() -> int:
    """Padding fragments at the end of every wave stream (the chain kernel prefetches this far ahead)."""
    global _WFRAG_DEPTH
    if _WFRAG_DEPTH is None:
        _WFRAG_DEPTH = load().st_wfrag_depth()
    return _WFRAG_DEPTH


def wfrag_build(table):
    """(Re)build weight-fragment streams (csrc/st_rowchain.hip): ``table`` int64 [n_blocks, 4] on the device, one row per
    256 x 256 weight block: (address of its first element, leading dimension | transposed << 32, fragment index inside a
    wave stream | wave stride in fragments << 32, address of the chain's buffer) - see st_amd/chains.py."""
    if table.dtype != torch.int64 or table.dim() != 2 or table.shape[1] != 4 or not table.is_contiguous() or not table.is_cuda:
        raise ValueError("wfrag_build: table must be a contiguous int64 [n, 4] device tensor")
    _tag("wfrag_build", table.shape[0], 0, 0, io=(4.0 * 256 * 256 * table.shape[0],))
    _check(load().st_wfrag_build(_stream(), table.data_ptr(), table.shape[0]), "st_wfrag_build")


K_LOG2_SCALE = 1.4426950408889634      # log2(e): a pre-scaled key projection holds scale * K_LOG2_SCALE * k (attn_fwd's k_prescaled)


def row_chain(A, chain, pre=None, ffn=None, post=None, eps=1e-6, post_kscale=0.0):
    """One launch for a chain of row-wise layers over decoder-sized row counts (csrc/st_rowchain.hip).
    ``chain``: st_amd.chains.Chain (the fragment streams of this chain's weight blocks); A [M, 256] bf16.
    pre  = (R, bo, gamma, beta, out, xhat, rstd):                 cur = LN(A Wo^T + bo + R)
    ffn  = (d_ff, b1, b2, gamma, beta, H, out, xhat, rstd, drop1, drop2[, relu_bits]): cur = drop2(LN(drop1(relu(cur W1^T + b1)) W2^T + b2 + cur));
           H (the hidden activation, the weight gradient's operand) may be None when no backward follows
           relu_bits (int64 [chain_mask_words(M, d_ff)]) receives the H > 0 mask row_chain_bwd reads
    post = (n_blocks_out, bias, P):                               P = cur Wp^T + bias, Wp [256 n_blocks_out, 256]
    post_kscale (n_blocks_out == 3 only; 0 = plain): the KEY block of a q | k | v projection leaves multiplied by it (in fp32,
    before its one rounding) - scale * K_LOG2_SCALE for attention kernels called with k_prescaled=True.
    xhat / rstd may be None when no backward follows."""
    _mat(A, BF16, "A")
    M, d = A.shape
    if d not in (256, 512):
        raise ValueError("row_chain: d_model 256 or 512")
    wfrag, n_blocks = chain.stream, chain.n_blocks
    if d == 512:      # csrc/st_rowchain_pipe512.cuh: PRE + FFN [+ a q | k | v projection], 256 x 256 blocks of 512-wide matrices
        if not (pre and ffn) or (post and post[0] != 6):
            raise ValueError("row_chain (d_model 512): PRE and FFN are required, POST is a 1,536-column projection (6 blocks)")
        nb = 4 + 4 * (ffn[0] // 256) + (12 if post else 0)
    else:
        nb = (1 if pre else 0) + (2 * (ffn[0] // 256) if ffn else 0) + (post[0] if post else 0)
    if nb != n_blocks or wfrag.numel() != 8 * (n_blocks * 16 + wfrag_depth()) * 512 or wfrag.dtype != BF16:
        raise ValueError("row_chain: the fragment stream does not match the chain")
    z = (None,) * 11
    R, bo, g0, be0, out0, xhat0, rstd0 = pre if pre else z[:7]
    relu_bits = None
    if ffn and len(ffn) == 12:        # (.., relu_bits): int64 [chain_mask_words(M, d_ff)] for row_chain_bwd
        relu_bits, ffn = ffn[11], ffn[:11]
    d_ff, b1, b2, g1, be1, H, out1, xhat1, rstd1, drop1, drop2 = ffn if ffn else (0,) + z[:10]
    pb, bp, P = post if post else (0, None, None)
    if pre:
        _mat(R, BF16, "R"), _vec(bo, F32, d, "bo"), _vec(g0, F32, d, "g0"), _vec(be0, F32, d, "be0")
        if out0 is not None:       # (None: the sublayer's output is only the chain's own running activation - inference)
            _mat(out0, BF16, "out0")
            assert out0.stride(0) == d
        assert (xhat0 is None or xhat0.stride(0) == d) and R.shape[0] >= M
        assert rstd0 is None or (rstd0.dtype == F32 and rstd0.numel() >= M)
    if ffn:
        _mat(out1, BF16, "out1"), _vec(b1, F32, d_ff, "b1"), _vec(b2, F32, d, "b2")
        _vec(g1, F32, d, "g1"), _vec(be1, F32, d, "be1")
        if H is not None:          # (None: inference - the hidden activation stays on the chip)
            _mat(H, BF16, "H")
            assert H.stride(0) == d_ff and H.shape == (M, d_ff)
        assert out1.stride(0) == d and (xhat1 is None or xhat1.stride(0) == d)
        assert rstd1 is None or (rstd1.dtype == F32 and rstd1.numel() >= M)
        assert relu_bits is None or (relu_bits.dtype == torch.int64 and relu_bits.is_contiguous() and relu_bits.numel() >= chain_mask_words(M, d_ff, d))
    if post:
        _mat(P, BF16, "P"), _vec(bp, F32, 256 * pb, "bp")
        assert P.shape == (M, 256 * pb)
    seed = None
    for dr in (drop1, drop2):
        if dr is not None and dr.thresh:
            if seed is not None and seed.data_ptr() != dr.seed.data_ptr():
                raise ValueError("row_chain: both dropout sites must read the same device seed")
            seed = dr.seed
    s1, s2 = _drop(drop1), _drop(drop2)
    work = getattr(chain, "split_work", None) if ffn else None
    if work is not None and not (work.is_cuda and work.is_contiguous() and work.dtype == torch.int32):
        raise ValueError("row_chain: chain.split_work must be a contiguous int32 tensor on the GPU")
    _tag("row_chain", M, n_blocks, d_ff, io=((A, M), (R, M), (out0, M), (xhat0, M), (H, M), relu_bits, (out1, M), (xhat1, M), (P, M),
                                             (rstd0, M), (rstd1, M), 2.0 * 256 * 256 * n_blocks))
    if d == 512:
        rc = load().st_row_chain512(_stream(), M, wfrag.data_ptr(), n_blocks, int(chain.next_blocks), float(eps), A.data_ptr(), A.stride(0),
                                    _p(R), R.stride(0), _p(bo), _p(g0), _p(be0), _p(out0), _p(xhat0), _p(rstd0), int(d_ff), _p(b1), _p(b2),
                                    _p(g1), _p(be1), _p(H), _p(relu_bits), _p(out1), _p(xhat1), _p(rstd1), _p(seed), s1[1], s1[2], s1[3],
                                    s2[1], s2[2], s2[3], int(pb), _p(bp), _p(P), 0 if P is None else P.stride(0), float(post_kscale))
        _check(rc, "st_row_chain512")
        return
    rc = load().st_row_chain(_stream(), M, wfrag.data_ptr(), n_blocks, int(chain.next_blocks), float(eps), A.data_ptr(), A.stride(0), _p(R),
                             0 if R is None else R.stride(0), _p(bo), _p(g0), _p(be0), _p(out0), _p(xhat0), _p(rstd0),
                             int(d_ff), _p(b1), _p(b2), _p(g1), _p(be1), _p(H), _p(relu_bits), _p(out1), _p(xhat1), _p(rstd1), _p(seed),
                             s1[1], s1[2], s1[3], s2[1], s2[2], s2[3], int(pb), _p(bp), _p(P), 0 if P is None else P.stride(0),
                             _p(work), 0 if work is None else work.numel() * work.element_size(), float(post_kscale))
    _check(rc, "st_row_chain")


def split_work_words() -> int:
    """int32 elements of the scratch a Chain may carry as ``split_work`` (zero-initialised; the kernel leaves it zeroed where
    it matters): the 32 x 256 fp32 partial sums of up to 256 workgroups, and one ticket per row block."""
    return 256 * 512 * 16 + 256          # 256 workgroups' partials (the launch never uses more) + tickets


_splitk_work = {}
_SPLITK_BYTES = 4096 + 256 * 65536


def splitk_scratch(device):
    """The per-device scratch of st_gemm_splitk (tickets + 256 fp32 tile partials), zeroed once; the kernel restores its
    tickets.  TrainStep creates it before a capture (a tensor born inside one belongs to that graph's pool)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    work = _splitk_work.get(device)
    if work is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("splitk_scratch: first use inside a stream capture; call st_amd.native.splitk_scratch(device) before it")
        work = _splitk_work[device] = torch.zeros(_SPLITK_BYTES // 4, dtype=torch.int32, device=device)
    return work


def gemm_splitk(X, Y, out, splits, y_cmajor=False):
    """out (bf16 [M, N]) = X Y^T with the contraction cut ``splits`` ways over workgroups and merged by the last one to finish
    each output tile (st_gemm_splitk: few output tiles, long contraction).  Y: [N, Kc], or [Kc, N] with y_cmajor."""
    _mat(X, BF16, "X"), _mat(Y, BF16, "Y"), _mat(out, BF16, "out")
    M, Kc = X.shape
    N = Y.shape[1] if y_cmajor else Y.shape[0]
    if (Y.shape[0] if y_cmajor else Y.shape[1]) != Kc or tuple(out.shape) != (M, N):
        raise ValueError("gemm_splitk: shapes")
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles * int(splits) > 256:
        raise ValueError("gemm_splitk: at most 256 (output tile, split) pairs")
    # ONE scratch per device, sized for the largest launch (16.8 MB): a size that followed the shape would be allocated anew -
    # possibly inside a graph capture, from the graph's private pool - whenever a batch brings another row count
    need = _SPLITK_BYTES
    work = splitk_scratch(X.device)
    _tag("gemm", 0, int(y_cmajor), M, N, Kc, 0, io=(X, Y, out, 2.0 * tiles * int(splits) * 65536))
    _check(load().st_gemm_splitk(_stream(), int(y_cmajor), X.data_ptr(), X.stride(0), Y.data_ptr(), Y.stride(0), out.data_ptr(),
                                 out.stride(0), M, N, Kc, int(splits), work.data_ptr(), need), "st_gemm_splitk")
    return out


def chain_mask_words(M: int, d_ff: int, d_model: int = 256) -> int:
    """int64 words of the ReLU-mask buffer a feed-forward row chain over M rows writes (row_chain) and reads (row_chain_bwd)."""
    if d_model == 512:
        return int(load()._cdll.st_row_chain512_mask_words(int(M), int(d_ff)))
    return int(load()._cdll.st_row_chain_mask_words(int(M), int(d_ff)))     # host-only: not a launch (bypasses the per-launch timer)


def relu_bits_from(H, d_model: int = 256):
    """The relu_bits buffer of a hidden activation H [M, d_ff] that did NOT come out of row_chain (a separate forward, a
    test): word ((workgroup * d_ff/256 + chunk) * 8 + wave) * 64 + lane, bit 16 mt + 4 g + e  <->  row
    workgroup * 32 MT + 32 mt + (lane & 31), column 256 chunk + 32 wave + 8 g + 4 (lane >> 5) + e - the accumulator layout
    of csrc/st_rowchain.hip (MT row tiles per workgroup as st_row_chain picks them for M)."""
    M, d_ff = H.shape
    words = chain_mask_words(M, d_ff, d_model)
    nc = d_ff // 256
    n_wg = words // (nc * 512)
    mt_n = -(-M // (32 * n_wg))                       # rows per workgroup / 32
    rows = n_wg * 32 * mt_n
    pos = torch.zeros(rows, d_ff, dtype=torch.int64, device=H.device)
    pos[:M] = (H.float() > 0).to(torch.int64)
    # [wg, mt, r, ch, wave, g, hi, e] -> [wg, ch, wave, hi, r, mt, g, e]
    v = pos.view(n_wg, mt_n, 32, nc, 8, 4, 2, 4).permute(0, 3, 4, 6, 2, 1, 5, 7).reshape(n_wg * nc * 8 * 64, mt_n * 16)
    shifts = torch.arange(mt_n * 16, device=H.device, dtype=torch.int64)
    return (v << shifts).sum(1).contiguous()


def row_chain_bwd(chain, M, head=None, ds_in=None, ffn=None, tail=None):
    """One launch for the backward of a row chain (csrc/st_rowchain.hip; ``chain`` holds the TRANSPOSED weight blocks in
    the order HEAD | FFN | TAIL).
    head = (n_blocks, dP, G, xhat, rstd, gamma, drop, ds_out, dgamma, dbeta, dbias):
           dy = dP Wp + G;  ds_out = LayerNorm-backward(dropout-backward(dy)); column sums accumulated atomically
           (n_blocks = 0, dP = None: dy = G - the bare LayerNorm backward of a gradient arriving from outside the stack)
    ds_in: without head, the running gradient [M, 256] the chain starts from
    ffn  = (d_ff, relu_bits, mask_scale, dH, xhat, rstd, gamma, ds_out, dgamma, dbeta, dbias):
           dH = (ds W2) masked by H > 0 (the bits the forward chain wrote), scaled;  ds_out = LayerNorm-backward(dH W1 + ds)
    tail = (O, Ores, dctx, delta): dctx = ds Wo, delta[h][i] = sum over head h (64 columns) of dctx (O + Ores)"""
    z = (None,) * 11
    nb, dP, G, xa, ra, ga, drop, dsa, dga, dba, dbia = head if head else (0,) + z[:10]
    d_ff, bits, msc, dH, xb, rb, gb, dsb, dgb, dbb, dbib = ffn if ffn else (0, None, 1.0) + z[:8]
    O, Ores, dctx, delta = tail if tail else z[:4]
    d = 512 if (xb is not None and xb.shape[1] == 512) else 256      # (d_model 512: csrc/st_rowchain_pipe512_bwd.cuh - HEAD + FFN + TAIL)
    if d == 512:
        if not (head and ffn and tail) or nb not in (0, 6):
            raise ValueError("row_chain_bwd (d_model 512): HEAD, FFN and TAIL are required, head blocks 0 or 6")
        n_blocks = 2 * nb + 4 * (d_ff // 256) + 4
    else:
        n_blocks = nb + (2 * (d_ff // 256) if ffn else 0) + (1 if tail else 0)
    if n_blocks != chain.n_blocks or chain.stream.numel() != 8 * (n_blocks * 16 + wfrag_depth()) * 512:
        raise ValueError("row_chain_bwd: the fragment stream does not match the chain")
    for t, cols, name in ((dP, 256 * nb, "dP"), (G, d, "G"), (xa, d, "xhat_a"), (dsa, d, "ds_a"), (ds_in, d, "ds_in"),
                          (dH, d_ff, "dH"), (xb, d, "xhat_b"), (dsb, d, "ds_b"), (O, d, "O"),
                          (Ores, d, "Ores"), (dctx, d, "dctx")):
        if t is not None:
            _mat(t, BF16, name)
            if t.shape[0] < M or t.shape[1] != cols:
                raise ValueError("row_chain_bwd: %s has shape %s, expected [>= %d, %d]" % (name, tuple(t.shape), M, cols))
    for t, name in ((xa, "xhat_a"), (dsa, "ds_a"), (ds_in, "ds_in"), (xb, "xhat_b"), (dsb, "ds_b")):
        assert t is None or t.stride(0) == d, name
    assert dH is None or dH.stride(0) == d_ff
    if ffn and (bits is None or bits.dtype != torch.int64 or not bits.is_contiguous() or bits.numel() < chain_mask_words(M, d_ff, d)):
        raise ValueError("row_chain_bwd: relu_bits must be the int64 buffer the forward chain wrote (chain_mask_words(M, d_ff) words)")
    assert O is None or Ores is None or Ores.stride(0) == O.stride(0)
    for v, n, name in ((ra, M, "rstd_a"), (ga, d, "gamma_a"), (dga, d, "dgamma_a"), (dba, d, "dbeta_a"), (dbia, d, "dbias_a"),
                       (rb, M, "rstd_b"), (gb, d, "gamma_b"), (dgb, d, "dgamma_b"), (dbb, d, "dbeta_b"), (dbib, d, "dbias_b"),
                       (delta, (d // 64) * M, "delta")):
        _vec(v, F32, n, name)
    if head is None and ds_in is None:
        raise ValueError("row_chain_bwd: without head the chain needs ds_in")
    sd = _drop(drop)
    work = getattr(chain, "split_work", None) if ffn else None
    if work is not None and not (work.is_cuda and work.is_contiguous() and work.dtype == torch.int32):
        raise ValueError("row_chain_bwd: chain.split_work must be a contiguous int32 tensor on the GPU")
    _tag("row_chain_bwd", M, n_blocks, d_ff, io=((dP, M), (G, M), (xa, M), (dsa, M), (ds_in, M), bits, (dH, M), (xb, M), (dsb, M), (O, M),
                                                 (Ores, M), (dctx, M), (delta, 4 * M), (ra, M), (rb, M), 2.0 * 256 * 256 * n_blocks))
    # encoder-sized HEAD + FFN + TAIL launches leave their LayerNorm column sums in a per-workgroup workspace (no atomics: 251 workgroups
    # adding to the same 768 floats queue for ~8 us per launch); st_colsum_fold adds it to the gradients - at once, or with the deferred
    # weight gradients (flush_colsum_folds)
    ws_rows = 0 if d == 512 else load()._cdll.st_row_chain_bwd_colsum_rows(int(M), int(head is not None), int(d_ff), int(tail is not None))
    cws = torch.empty(ws_rows * 1536, dtype=F32, device=xb.device) if ws_rows > 0 else None
    if d == 512:
        rc = load().st_row_chain512_bwd(
            _stream(), M, chain.stream.data_ptr(), n_blocks, int(chain.next_blocks), int(nb), _p(dP), 0 if dP is None else dP.stride(0),
            _p(G), 0 if G is None else G.stride(0), _p(xa), _p(ra), _p(ga), sd[0], sd[1], sd[2], sd[3], _p(dsa), _p(dga), _p(dba),
            _p(dbia), int(d_ff), _p(bits), float(msc), _p(dH), _p(xb), _p(rb), _p(gb), _p(dsb), _p(dgb), _p(dbb), _p(dbib),
            _p(O), _p(Ores), O.stride(0), _p(dctx), dctx.stride(0), _p(delta))
        _check(rc, "st_row_chain512_bwd")
        return
    rc = load().st_row_chain_bwd(
        _stream(), M, chain.stream.data_ptr(), n_blocks, int(chain.next_blocks), int(nb), _p(dP), 0 if dP is None else dP.stride(0),
        _p(G), 0 if G is None else G.stride(0), _p(xa), _p(ra), _p(ga), sd[0], sd[1], sd[2], sd[3], _p(dsa), _p(dga), _p(dba),
        _p(dbia), _p(ds_in), int(d_ff), _p(bits), float(msc), _p(dH), _p(xb), _p(rb), _p(gb), _p(dsb), _p(dgb), _p(dbb), _p(dbib),
        _p(O), _p(Ores), 0 if O is None else O.stride(0), _p(dctx), 0 if dctx is None else dctx.stride(0), _p(delta),
        _p(work), 0 if work is None else work.numel() * work.element_size(), _p(cws), 0 if cws is None else cws.numel() * 4)
    _check(rc, "st_row_chain_bwd")
    if cws is not None:
        _fold_pending.append((cws, ws_rows, (dga, dba, dbia, dgb, dbb, dbib)))
        if not fold_deferred:
            flush_colsum_folds()


_fold_pending = []        # (workspace, rows, six destination vectors) of row_chain_bwd launches whose column sums are not folded yet
fold_deferred = False     # functional.deferred_wgrads: folds wait for flush_deferred_wgrads (they feed nothing but parameter gradients)


def colsum_fold(items):
    """dst[v] += column sums over the workspace rows, for every (workspace [rows, 6, 256] fp32, rows, (6 fp32 [256] vectors or None))."""
    n = len(items)
    if n == 0:
        return
    ws = (ctypes.c_void_p * n)(*[it[0].data_ptr() for it in items])
    rows = (ctypes.c_int * n)(*[int(it[1]) for it in items])
    for it in items:
        for v in it[2]:
            _vec(v, F32, 256, "colsum_fold dst")
    dst = (ctypes.c_void_p * (6 * n))(*[_p(v) for it in items for v in it[2]])
    _tag("colsum_fold", n, 0, 0, io=tuple((it[0], it[1] * 1536) for it in items))
    _check(load().st_colsum_fold(_stream(), n, ws, rows, dst), "st_colsum_fold")


def flush_colsum_folds():
    if _fold_pending:
        items = list(_fold_pending)
        _fold_pending.clear()
        colsum_fold(items)


def gemm_lnbwd(dY, W, aux, xhat, rstd, gamma, dx, dgamma=None, dbeta=None, dbias=None, drop=None):
    """dx = LayerNorm-backward(bf16(dY W (+ aux)); xhat, rstd, gamma) in one launch; W is the nn.Linear weight
    [Kc, N] as stored (contraction-major for this product).  == gemm(dY, W, tmp, y_cmajor=True[, ADD aux]) + ln_bwd.
    drop: the forward dropped the LayerNorm output (gemm_ln drop_where = 2)."""
    _mat(dY, BF16, "dY"), _mat(W, BF16, "W"), _mat(xhat, BF16, "xhat"), _mat(dx, BF16, "dx")
    M, Kc = dY.shape
    N = W.shape[1]
    if W.shape[0] != Kc or xhat.stride(0) != N or xhat.shape[0] < M:
        raise ValueError("gemm_lnbwd: inconsistent shapes")
    if aux is not None:
        _mat(aux, BF16, "aux")
    _vec(rstd, F32, M, "rstd"), _vec(gamma, F32, N, "gamma")
    _vec(dgamma, F32, N, "dgamma"), _vec(dbeta, F32, N, "dbeta"), _vec(dbias, F32, N, "dbias")
    _tag("gemm_lnbwd", M, N, Kc, io=((dY, M), 2.0 * N * Kc, (aux, M), (xhat, M), (dx, M), (rstd, M)))
    rc = load().st_gemm_lnbwd(_stream(), dY.data_ptr(), dY.stride(0), W.data_ptr(), W.stride(0), M, N, Kc, _p(aux),
                              0 if aux is None else aux.stride(0), xhat.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                              dx.data_ptr(), dx.stride(0), _p(dgamma), _p(dbeta), _p(dbias), *_drop(drop))
    _check(rc, "st_gemm_lnbwd")
    return dx


def ln_bwd(dy, xhat, rstd, gamma, dx, dgamma=None, dbeta=None, dbias=None, mask=None, drop=None, mask_scale=1.0):
    """drop: the forward dropped the LayerNorm output; mask_scale: 1/(1-p) of a dropout between `mask`'s ReLU and the LN."""
    _mat(dy, BF16, "dy"), _mat(xhat, BF16, "xhat"), _mat(dx, BF16, "dx")
    M, N = dy.shape
    assert xhat.stride(0) == N
    if mask is not None:
        _mat(mask, BF16, "mask")
        assert mask.stride(0) == N
    _vec(rstd, F32, M, "rstd"), _vec(gamma, F32, N, "gamma")
    _vec(dgamma, F32, N, "dgamma"), _vec(dbeta, F32, N, "dbeta"), _vec(dbias, F32, N, "dbias")
    _tag("ln_bwd", M, N, io=((dy, M), (xhat, M), (dx, M), (mask, M), (rstd, M)))
    rc = load().st_ln_bwd(_stream(), dy.data_ptr(), dy.stride(0), xhat.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                          _p(mask), dx.data_ptr(), dx.stride(0), _p(dgamma), _p(dbeta), _p(dbias), M, N, *_drop(drop),
                          float(mask_scale))
    _check(rc, "st_ln_bwd")
    return dx


def _work(w):
    if w is None:
        return None, 0
    _vec(w, I32, w.numel(), "work")
    return w.data_ptr(), w.numel()


def attn_tile_rows(which: int, d_k: int, max_q: int, max_k: int, causal: bool) -> int:
    """Rows per work-list tile of the kernel st_attn_fwd (which = 0) / st_attn_bwd's dQ (1) and dK/dV (2) parts will
    run for a problem of this shape (host-side query, no launch)."""
    return int(load().st_attn_tile_rows(int(which), int(d_k), int(max_q), int(max_k), int(bool(causal))))


def attn_fwd(Q, K, V, O, lse, q_off, q_len, k_off, k_len, n_head, max_q, causal, scale, work=None, drop=None,
             max_k=0, ores=None, k_prescaled=False):
    """max_k (longest key sequence) only selects the kernel variant: <= 64 queries against >= 256 keys take
    the key-split path."""
    """work: optional int32 device list of (b << 16) | q_tile, heaviest first (functional.attn_work).
    ores (optional, bf16, O's shape and stride): receives bf16(O_fp32 - bf16(O_fp32)).
    k_prescaled: K holds scale * log2(e) * (key projection) (st_attn_fwd; row_chain's post_kscale) - the backward and
    attn_probs of the same sublayer must be told the same."""
    for t, nm in ((Q, "Q"), (K, "K"), (V, "V"), (O, "O")):
        _mat(t, BF16, nm)
    if ores is not None:
        _mat(ores, BF16, "ores")
        if ores.stride(0) != O.stride(0) or ores.shape != O.shape:
            raise ValueError("attn_fwd: ores must have O's shape and stride")
    B = q_off.numel()
    d_k = Q.shape[1] // n_head
    for t, nm in ((q_off, "q_off"), (q_len, "q_len"), (k_off, "k_off"), (k_len, "k_len")):
        _vec(t, I32, B, nm)
    rows = Q.shape[0]
    _vec(lse, F32, n_head * rows, "lse")
    _tag("attn_fwd", n_head, d_k, int(causal), q_len, k_len, io=(Q, K, V, O, ores, lse))
    rc = load().st_attn_fwd(_stream(), Q.data_ptr(), Q.stride(0), K.data_ptr(), K.stride(0), V.data_ptr(), V.stride(0),
                            O.data_ptr(), O.stride(0), _p(ores), lse.data_ptr(), q_off.data_ptr(), q_len.data_ptr(),
                            k_off.data_ptr(), k_len.data_ptr(), B, n_head, d_k, int(max_q), int(max_k), rows, int(causal),
                            float(scale), *_work(work), *_drop(drop), int(bool(k_prescaled)))
    _check(rc, "st_attn_fwd")
    return O


_F1_OK = {}


def _f1_applicable(d, d_k, max_q, max_k):
    """Host-only query (not a launch: bypasses the per-launch timer), cached per shape."""
    key = (d, d_k, max_q, max_k)
    if key not in _F1_OK:
        _F1_OK[key] = bool(load()._cdll.st_attn_f1_applicable(d, d_k, max_q, max_k))
    return _F1_OK[key]


def attn_f1_fwd(A, chain, pre, post, K, V, O, lse, q_off, q_len, k_off, k_len, n_head, max_q, scale, work=None, drop=None,
                max_k=0, ores=None, eps=1e-6):
    """The decoder-encoder attention together with the chain stage in front of it, as ONE launch where the few-queries
    kernel serves the shape (d_model 256, 64-wide heads, <= 64 queries, >= 256 keys; csrc/st_attn_xs.hip), else as the two
    launches it stands for:
        row_chain(A, chain, pre=pre, post=post)        cur = LN(A Wo^T + bo + R) -> out / xhat / rstd;  q = cur Wq^T + bq -> post[2]
        attn_fwd(post[2], K, V, O, lse, ..., causal=False)
    pre = (R, bo, gamma, beta, out, xhat, rstd), post = (1, bq, Qout) exactly as row_chain takes them."""
    R, bo, g0, be0, out0, xhat0, rstd0 = pre
    pb, bq, Qout = post
    M, d = A.shape
    d_k = d // n_head
    if pb != 1 or chain.n_blocks != 2 or not _f1_applicable(int(d), int(d_k), int(max_q), int(max_k)):
        row_chain(A, chain, pre=pre, post=post, eps=eps)
        return attn_fwd(Qout, K, V, O, lse, q_off, q_len, k_off, k_len, n_head, max_q, False, scale, work=work, drop=drop,
                        max_k=max_k, ores=ores)
    for t, nm in ((A, "A"), (R, "R"), (out0, "out0"), (Qout, "Qout"), (K, "K"), (V, "V"), (O, "O")):
        _mat(t, BF16, nm)
    _vec(bo, F32, d, "bo"), _vec(g0, F32, d, "g0"), _vec(be0, F32, d, "be0"), _vec(bq, F32, d, "bq")
    assert out0.stride(0) == d and (xhat0 is None or xhat0.stride(0) == d) and R.shape[0] >= M and Qout.shape == (M, d)
    assert rstd0 is None or (rstd0.dtype == F32 and rstd0.numel() >= M)
    if chain.stream.numel() != 8 * (2 * 16 + wfrag_depth()) * 512 or chain.stream.dtype != BF16:
        raise ValueError("attn_f1_fwd: the fragment stream does not match a two-block chain")
    if ores is not None:
        _mat(ores, BF16, "ores")
        if ores.stride(0) != O.stride(0) or ores.shape != O.shape:
            raise ValueError("attn_f1_fwd: ores must have O's shape and stride")
    B = q_off.numel()
    for t, nm in ((q_off, "q_off"), (q_len, "q_len"), (k_off, "k_off"), (k_len, "k_len")):
        _vec(t, I32, B, nm)
    _vec(lse, F32, n_head * M, "lse")
    _tag("attn_fwd", n_head, d_k, 0, q_len, k_len, io=(A, R, out0, xhat0, rstd0, Qout, K, V, O, ores, lse))
    rc = load().st_attn_f1_fwd(_stream(), A.data_ptr(), A.stride(0), R.data_ptr(), R.stride(0), chain.stream.data_ptr(), 2,
                               int(chain.next_blocks), float(eps), bo.data_ptr(), g0.data_ptr(), be0.data_ptr(), out0.data_ptr(),
                               _p(xhat0), _p(rstd0), bq.data_ptr(), Qout.data_ptr(), Qout.stride(0), K.data_ptr(), K.stride(0),
                               V.data_ptr(), V.stride(0), O.data_ptr(), O.stride(0), _p(ores), lse.data_ptr(), q_off.data_ptr(),
                               q_len.data_ptr(), k_off.data_ptr(), k_len.data_ptr(), B, n_head, d_k, int(max_q), int(max_k), M,
                               float(scale), *_work(work), *_drop(drop))
    _check(rc, "st_attn_f1_fwd")
    return O


def attn_sf1_fwd(qkv, Os, lses, pre, post, chain, K, V, O, lse, q_off, q_len, k_off, k_len, n_head, max_q, scale, work_self=None,
                 work=None, drop_self=None, drop=None, max_k=0, ores_self=None, ores=None, eps=1e-6):
    """A decoder layer's causal self-attention, the chain stage behind it and its decoder-encoder attention as ONE launch where the
    few-queries kernel serves the shape (st_attn_sf1_fwd), else as the three launches it stands for:
        attn_fwd(q, k, v of qkv, Os, lses, ..., causal=True)                          (keys = the utterance's own target positions)
        row_chain(Os, chain, pre=pre, post=post)
        attn_fwd(post[2], K, V, O, lse, ..., causal=False)
    qkv [M, 3 d]: the layer's q | k | v projection; pre / post as attn_f1_fwd."""
    R, bo, g0, be0, out0, xhat0, rstd0 = pre
    pb, bq, Qout = post
    M, d3 = qkv.shape
    d = d3 // 3
    d_k = d // n_head
    if (pb != 1 or chain.n_blocks != 2 or n_head > 8 or not _f1_applicable(int(d), int(d_k), int(max_q), int(max_k))
            or (drop_self is not None and drop is not None and drop_self.thresh and drop.thresh
                and drop_self.seed.data_ptr() != drop.seed.data_ptr())):
        attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], Os, lses, q_off, q_len, q_off, q_len, n_head, max_q, True, scale,
                 work=work_self, drop=drop_self, max_k=max_q, ores=ores_self)
        return attn_f1_fwd(Os, chain, pre, post, K, V, O, lse, q_off, q_len, k_off, k_len, n_head, max_q, scale, work=work, drop=drop,
                           max_k=max_k, ores=ores, eps=eps)
    for t, nm in ((qkv, "qkv"), (Os, "Os"), (R, "R"), (out0, "out0"), (Qout, "Qout"), (K, "K"), (V, "V"), (O, "O")):
        _mat(t, BF16, nm)
    _vec(bo, F32, d, "bo"), _vec(g0, F32, d, "g0"), _vec(be0, F32, d, "be0"), _vec(bq, F32, d, "bq")
    assert out0.stride(0) == d and (xhat0 is None or xhat0.stride(0) == d) and R.shape[0] >= M and Qout.shape == (M, d) and Os.shape == (M, d)
    assert rstd0 is None or (rstd0.dtype == F32 and rstd0.numel() >= M)
    if chain.stream.numel() != 8 * (2 * 16 + wfrag_depth()) * 512 or chain.stream.dtype != BF16:
        raise ValueError("attn_sf1_fwd: the fragment stream does not match a two-block chain")
    for o_, r_, nm in ((O, ores, "ores"), (Os, ores_self, "ores_self")):
        if r_ is not None:
            _mat(r_, BF16, nm)
            if r_.stride(0) != o_.stride(0) or r_.shape != o_.shape:
                raise ValueError("attn_sf1_fwd: %s must have its output's shape and stride" % nm)
    B = q_off.numel()
    for t, nm in ((q_off, "q_off"), (q_len, "q_len"), (k_off, "k_off"), (k_len, "k_len")):
        _vec(t, I32, B, nm)
    _vec(lse, F32, n_head * M, "lse"), _vec(lses, F32, n_head * M, "lses")
    sd, cd = _drop(drop_self), _drop(drop)
    seed = sd[0] if sd[0] is not None else cd[0]
    _tag("attn_fwd", n_head, d_k, 0, q_len, k_len, io=(qkv, Os, ores_self, lses, R, out0, xhat0, rstd0, Qout, K, V, O, ores, lse))
    q_, k_, v_ = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    rc = load().st_attn_sf1_fwd(_stream(), q_.data_ptr(), k_.data_ptr(), v_.data_ptr(), qkv.stride(0), Os.data_ptr(), _p(ores_self),
                                Os.stride(0), lses.data_ptr(), sd[1], sd[2], sd[3], R.data_ptr(), R.stride(0), chain.stream.data_ptr(), 2,
                                int(chain.next_blocks), float(eps), bo.data_ptr(), g0.data_ptr(), be0.data_ptr(), out0.data_ptr(),
                                _p(xhat0), _p(rstd0), bq.data_ptr(), Qout.data_ptr(), Qout.stride(0), K.data_ptr(), K.stride(0),
                                V.data_ptr(), V.stride(0), O.data_ptr(), O.stride(0), _p(ores), lse.data_ptr(), q_off.data_ptr(),
                                q_len.data_ptr(), k_off.data_ptr(), k_len.data_ptr(), B, n_head, d_k, int(max_q), int(max_k), M,
                                float(scale), *_work(work), seed, cd[1], cd[2], cd[3])
    _check(rc, "st_attn_sf1_fwd")
    return O


def attn_bwd(Q, K, V, O, dO, lse, delta, dQ, dK, dV, q_off, q_len, k_off, k_len, n_head, max_q, max_k, causal, scale,
             parts=3, work_q=None, work_k=None, drop=None, k_prescaled=False):
    """parts: 1 = dQ (+delta) kernel, 2 = dK/dV kernel (needs delta from part 1), 3 = both.
    k_prescaled: as attn_fwd; dK is the gradient of the UNSCALED key projection either way."""
    for t, nm in ((Q, "Q"), (K, "K"), (V, "V"), (dO, "dO"), (dQ, "dQ"), (dK, "dK"), (dV, "dV")):
        _mat(t, BF16, nm)
    if O is not None:      # O = None: delta is an INPUT (produced with dO by gemm(..., epi=EPI_BF16_DELTA)); one launch
        _mat(O, BF16, "O")
    B = q_off.numel()
    d_k = Q.shape[1] // n_head
    rows = Q.shape[0]
    _vec(lse, F32, n_head * rows, "lse"), _vec(delta, F32, n_head * rows, "delta")
    if _TIMING is not None and parts == 3 and O is not None:   # profile the two kernels of the call separately
        for part in (1, 2):
            attn_bwd(Q, K, V, O, dO, lse, delta, dQ, dK, dV, q_off, q_len, k_off, k_len, n_head, max_q, max_k, causal,
                     scale, parts=part, work_q=work_q, work_k=work_k, drop=drop, k_prescaled=k_prescaled)
        return
    # few queries against many keys (the decoder-encoder attention): scratch for the dQ items' key split across workgroups
    kib = load()._cdll.st_attn_bwd_split_kib(B, n_head, d_k, int(max_q), int(max_k), int(causal)) if (parts == 3 and O is None) else 0
    swork = _attn_split_work(Q.device, kib) if kib > 0 else None
    _tag("attn_bwd", n_head, d_k, int(causal), q_len, k_len, parts, io=(Q, K, V, O, dO, dQ, dK, dV, lse, delta))
    rc = load().st_attn_bwd(_stream(), Q.data_ptr(), Q.stride(0), K.data_ptr(), K.stride(0), V.data_ptr(), V.stride(0),
                            _p(O), 0 if O is None else O.stride(0), dO.data_ptr(), dO.stride(0), lse.data_ptr(),
                            delta.data_ptr(),
                            dQ.data_ptr(), dQ.stride(0), dK.data_ptr(), dK.stride(0), dV.data_ptr(), dV.stride(0),
                            q_off.data_ptr(), q_len.data_ptr(), k_off.data_ptr(), k_len.data_ptr(), B, n_head, d_k,
                            int(max_q), int(max_k), rows, int(causal), float(scale), int(parts), *_work(work_q),
                            *_work(work_k), *_drop(drop), int(bool(k_prescaled)), _p(swork),
                            0 if swork is None else swork.numel() * 4)
    _check(rc, "st_attn_bwd")


_ATTN_SPLIT_WORK = {}


def _attn_split_work(device, kib):
    """The attention backward's key-split scratch (tickets + fp32 partials), one per device, grown on demand: launches are
    stream-ordered, the kernel leaves the tickets zero, so every launch may use the same buffer."""
    key = torch.device(device).index
    t = _ATTN_SPLIT_WORK.get(key)
    if t is None or t.numel() * 4 < kib * 1024:
        t = torch.zeros(kib * 256, dtype=torch.int32, device=device)
        _ATTN_SPLIT_WORK[key] = t
    return t


def ctc_gather(logits, rowmap, T, cols, lse, lp, V=None):
    """lse[r] = logsumexp(logits[r, :V]); lp[b][t][k] = logits[row(b, t)][cols[b][k]] - lse - see st_ctc_gather."""
    if not (logits.is_cuda and logits.dtype == F32 and logits.dim() == 2 and logits.stride(1) == 1):
        raise ValueError("ctc_gather: logits must be an fp32 row matrix on the GPU")
    R = logits.shape[0]
    V = logits.shape[1] if V is None else V
    B, C = cols.shape
    if not (rowmap.dtype == torch.int64 and rowmap.numel() == R and cols.dtype == I32 and cols.is_contiguous()
            and lp.dtype == F32 and lp.is_contiguous() and tuple(lp.shape) == (B, int(T), C)):
        raise ValueError("ctc_gather: rowmap i64 [R], cols i32 [B, C], lp f32 [B, T, C] expected")
    _vec(lse, F32, R, "lse")
    _tag("ctc_gather", R, V, C, io=(4.0 * R * V, 4.0 * R * C))
    _check(load().st_ctc_gather(_stream(), logits.data_ptr(), logits.stride(0), R, V, rowmap.data_ptr(), int(T), cols.data_ptr(), C,
                                lse.data_ptr(), lp.data_ptr()), "st_ctc_gather")


def ctc_dlogits(logits, lse, rowmap, T, roww, scat, gsmall, grad_out, dlogits, V=None):
    """dlogits (bf16) = grad_out * (roww[b] * softmax(logits), label columns overwritten with gsmall) - see st_ctc_dlogits."""
    R = logits.shape[0]
    V = logits.shape[1] if V is None else V
    B, C = scat.shape
    _mat(dlogits, BF16, "dlogits")
    if dlogits.shape[0] != R or dlogits.shape[1] < V or dlogits.stride(0) % 8 or dlogits.shape[1] != dlogits.stride(0):
        raise ValueError("ctc_dlogits: dlogits must be a contiguous bf16 [R, >= V] matrix with a row length that is a multiple of 8")
    if not (scat.dtype == I32 and scat.is_contiguous() and gsmall.dtype == F32 and gsmall.is_contiguous()
            and tuple(gsmall.shape) == (B, int(T), C) and roww.dtype == F32 and roww.numel() == B):
        raise ValueError("ctc_dlogits: scat i32 [B, C], gsmall f32 [B, T, C], roww f32 [B] expected")
    _vec(grad_out, F32, 1, "grad_out")
    _tag("ctc_dlogits", R, V, C, io=(4.0 * R * V, 2.0 * R * dlogits.shape[1]))
    _check(load().st_ctc_dlogits(_stream(), logits.data_ptr(), logits.stride(0), R, V, lse.data_ptr(), rowmap.data_ptr(), int(T),
                                 roww.data_ptr(), scat.data_ptr(), C, gsmall.data_ptr(), grad_out.data_ptr(), dlogits.data_ptr(),
                                 dlogits.stride(0)), "st_ctc_dlogits")


def attn_probs(Q, K, q_off, q_len, k_off, k_len, n_head, Lq, Lk, causal, scale, k_prescaled=False):
    """The attention probabilities of one sublayer, materialised: f32 [B, n_head, Lq, Lk] (zeros at masked keys and in the
    rows of padding positions).  Q, K: bf16 row matrices (column slices of the q | k | v projection are fine)."""
    for t, nm in ((Q, "Q"), (K, "K")):
        _mat(t, BF16, nm)
    B = q_off.numel()
    d_k = Q.shape[1] // n_head
    P = torch.empty(B, n_head, int(Lq), int(Lk), dtype=F32, device=Q.device)
    _tag("attn_probs", B, n_head, d_k, int(Lq), int(Lk))
    rc = load().st_attn_probs(_stream(), Q.data_ptr(), Q.stride(0), K.data_ptr(), K.stride(0), P.data_ptr(), q_off.data_ptr(),
                              q_len.data_ptr(), k_off.data_ptr(), k_len.data_ptr(), B, n_head, d_k, int(Lq), int(Lk), int(causal),
                              float(scale), int(bool(k_prescaled)))
    _check(rc, "st_attn_probs")
    return P


def attn_dense_fwd(Q, K, V, mask, O, lse, B, n_head, Lq, Lk, scale, drop=None, want_probs=False):
    """Attention under an arbitrary dense mask (uint8 [B, Lq, Lk], nonzero = masked; None = no mask) over padded row matrices
    Q [B Lq, H d_k], K / V [B Lk, H d_k] (see st_attn_dense_fwd: the slow general path).  -> P (f32 [B, H, Lq, Lk]) or None."""
    for t, nm in ((Q, "Q"), (K, "K"), (V, "V"), (O, "O")):
        _mat(t, BF16, nm)
    d_k = Q.shape[1] // n_head
    if mask is not None and not (mask.is_cuda and mask.dtype == torch.uint8 and mask.is_contiguous() and tuple(mask.shape) == (B, Lq, Lk)):
        raise ValueError("attn_dense_fwd: mask must be a contiguous uint8 [B, Lq, Lk] tensor on the GPU")
    if Q.shape[0] != B * Lq or K.shape[0] != B * Lk or V.shape[0] != B * Lk:
        raise ValueError("attn_dense_fwd: padded layouts expected (B Lq query rows, B Lk key rows)")
    _vec(lse, F32, n_head * B * Lq, "lse")
    P = torch.empty(B, n_head, Lq, Lk, dtype=F32, device=Q.device) if want_probs else None
    _tag("attn_dense", B, n_head, d_k, Lq, Lk)
    rc = load().st_attn_dense_fwd(_stream(), Q.data_ptr(), Q.stride(0), K.data_ptr(), K.stride(0), V.data_ptr(), V.stride(0), _p(mask),
                                  O.data_ptr(), O.stride(0), lse.data_ptr(), _p(P), B, n_head, d_k, int(Lq), int(Lk), float(scale),
                                  *_drop(drop))
    _check(rc, "st_attn_dense_fwd")
    return P


def attn_dense_bwd(Q, K, V, mask, dO, lse, delta, dQ, dK, dV, B, n_head, Lq, Lk, scale, drop=None):
    for t, nm in ((Q, "Q"), (K, "K"), (V, "V"), (dO, "dO"), (dQ, "dQ"), (dK, "dK"), (dV, "dV")):
        _mat(t, BF16, nm)
    d_k = Q.shape[1] // n_head
    _vec(lse, F32, n_head * B * Lq, "lse"), _vec(delta, F32, n_head * B * Lq, "delta")
    _tag("attn_dense_bwd", B, n_head, d_k, Lq, Lk)
    rc = load().st_attn_dense_bwd(_stream(), Q.data_ptr(), Q.stride(0), K.data_ptr(), K.stride(0), V.data_ptr(), V.stride(0), _p(mask),
                                  dO.data_ptr(), dO.stride(0), lse.data_ptr(), delta.data_ptr(), dQ.data_ptr(), dQ.stride(0),
                                  dK.data_ptr(), dK.stride(0), dV.data_ptr(), dV.stride(0), B, n_head, d_k, int(Lq), int(Lk),
                                  float(scale), *_drop(drop))
    _check(rc, "st_attn_dense_bwd")


def feat_stack(x, in_len, stats, left, right, interval, out_off, out_len, max_out_len, out):
    """Raw padded features x fp32 [B, T, F] -> stacked / subsampled / normalised bf16 rows (Dataset.py front-end)."""
    if not (x.is_cuda and x.dtype == F32 and x.is_contiguous() and x.dim() == 3):
        raise ValueError("feat_stack: x must be a contiguous fp32 [B, T, F] tensor on the GPU")
    B, T, F = x.shape
    _vec(in_len, I32, B, "in_len"), _vec(out_off, I32, B, "out_off"), _vec(out_len, I32, B, "out_len")
    _mat(out, BF16, "out")
    if stats is not None and not (stats.is_cuda and stats.dtype == F32 and stats.is_contiguous()
                                  and tuple(stats.shape) == (B, 2, F + 1)):
        raise ValueError("feat_stack: stats must be fp32 [B, 2, F+1] on the GPU")
    _tag("feat_stack", B, T, F)
    rc = load().st_feat_stack(_stream(), x.data_ptr(), B, T, F, in_len.data_ptr(), _p(stats), int(left), int(right),
                              int(interval), out_off.data_ptr(), out_len.data_ptr(), int(max_out_len), out.data_ptr(),
                              out.stride(0))
    _check(rc, "st_feat_stack")
    return out


def row_index(off, length, max_len, row_pos, row_seq=None):
    B = off.numel()
    _vec(off, I32, B, "off"), _vec(length, I32, B, "len")
    _check(load().st_row_index(_stream(), off.data_ptr(), length.data_ptr(), B, int(max_len), row_pos.data_ptr(),
                               _p(row_seq)), "st_row_index")
    return row_pos


def pack_rows(x, off, length, out):
    if not (x.is_cuda and x.dtype == F32 and x.is_contiguous() and x.dim() == 3):
        raise ValueError("pack_rows: x must be a contiguous fp32 [B, T, F] GPU tensor")
    B, T, Fd = x.shape
    _mat(out, BF16, "out")
    assert out.stride(0) == Fd
    _tag("pack_rows", out.shape[0], Fd, io=(float(out.shape[0]) * Fd * x.element_size(), out))
    _check(load().st_pack_rows(_stream(), x.data_ptr(), B, T, Fd, off.data_ptr(), length.data_ptr(), out.data_ptr()),
           "st_pack_rows")
    return out


def unpack_rows(x, off, length, out):
    _mat(x, BF16, "x")
    B, T, D = out.shape
    assert out.dtype == F32 and out.is_contiguous() and out.is_cuda
    _check(load().st_unpack_rows(_stream(), x.data_ptr(), x.stride(0), B, T, D, off.data_ptr(), length.data_ptr(),
                                 out.data_ptr()), "st_unpack_rows")
    return out


def pack_grad(g, off, length, out):
    assert g.is_cuda and g.dtype == F32 and g.is_contiguous() and g.dim() == 3
    B, T, D = g.shape
    _mat(out, BF16, "out")
    _check(load().st_pack_grad(_stream(), g.data_ptr(), B, T, D, off.data_ptr(), length.data_ptr(), out.data_ptr(),
                               out.stride(0)), "st_pack_grad")
    return out


def embed_pe_fwd(tok, emb, pe, off, length, out):
    assert tok.is_cuda and tok.dtype == I64 and tok.is_contiguous() and tok.dim() == 2
    B, L = tok.shape
    _mat(emb, F32, "emb"), _mat(pe, F32, "pe"), _mat(out, BF16, "out")
    D = emb.shape[1]
    assert emb.stride(0) == D and pe.stride(0) == D and out.stride(0) == D and pe.shape[0] >= L
    _tag("embed_pe_fwd", out.shape[0], D, io=(8.0 * out.shape[0] * D, out))
    _check(load().st_embed_pe_fwd(_stream(), tok.data_ptr(), B, L, emb.data_ptr(), emb.shape[0], pe.data_ptr(), D,
                                  off.data_ptr(), length.data_ptr(), out.data_ptr()), "st_embed_pe_fwd")
    return out


def embed_bwd(tok, dy, off, length, pad_idx, demb):
    assert tok.is_cuda and tok.dtype == I64 and tok.is_contiguous()
    B, L = tok.shape
    _mat(dy, BF16, "dy"), _mat(demb, F32, "demb")
    D = demb.shape[1]
    assert demb.stride(0) == D
    _tag("embed_bwd", dy.shape[0], D, io=(dy, 8.0 * dy.shape[0] * D))
    _check(load().st_embed_bwd(_stream(), tok.data_ptr(), B, L, dy.data_ptr(), dy.stride(0), D, off.data_ptr(),
                               length.data_ptr(), int(pad_idx), demb.data_ptr(), demb.shape[0]), "st_embed_bwd")
    return demb


def embed_step(tokens, emb, pe, step, out):
    """out bf16 [n, D] = emb[tokens] + pe[step] (step: i64 [1] on the device) - one beam-search step's decoder input."""
    n, D = out.shape
    _mat(out, BF16, "out"), _vec(tokens, I64, n, "tokens"), _vec(step, I64, 1, "step")
    if not (emb.is_cuda and emb.dtype == F32 and emb.dim() == 2 and emb.is_contiguous() and emb.shape[1] == D and
            pe.is_cuda and pe.dtype == F32 and pe.dim() == 2 and pe.is_contiguous() and pe.shape[1] == D and out.stride(0) == D):
        raise ValueError("embed_step: emb [V, D] / pe [S, D] contiguous fp32, out contiguous [n, D]")
    _check(load().st_embed_step(_stream(), tokens.data_ptr(), emb.data_ptr(), emb.shape[0], pe.data_ptr(), step.data_ptr(),
                                out.data_ptr(), n, D), "st_embed_step")
    return out


def _lineage(anc, n, S, name):
    if anc is not None and not (anc.is_cuda and anc.dtype == I32 and anc.is_contiguous() and tuple(anc.shape) == (n, S)):
        raise ValueError("%s: anc must be a contiguous int32 [n, S] tensor on the GPU" % name)
    return anc.data_ptr() if anc is not None else None


def decode_self_attn(qkv, cache, step, ctx, n_head, scale, anc=None):
    """One beam-search step's self-attention (one query per hypothesis): appends this step's k | v (columns [d, 3d) of qkv
    [n, 3d]) to ``cache`` [n, S, 2d] (one layer, contiguous) at position ``step`` (i64 [1], device) and writes the context
    [n, d] over positions 0 .. step.  d / n_head = 64, S <= 128.  ``anc``: optional lineage table i32 [n, S] (see
    st_decode_self_attn): earlier positions are read from the cache rows it names."""
    _mat(qkv, BF16, "qkv"), _mat(ctx, BF16, "ctx")
    n, S, w = cache.shape
    d = w // 2
    if not (cache.is_cuda and cache.dtype == BF16 and cache.is_contiguous()) or qkv.shape != (n, 3 * d) or ctx.shape != (n, d):
        raise ValueError("decode_self_attn: qkv [n, 3d], cache [n, S, 2d] (contiguous bf16), ctx [n, d]")
    if d % n_head or d // n_head != 64 or S > 128:
        raise ValueError("decode_self_attn: head width 64 and at most 128 cached positions")
    _vec(step, I64, 1, "step")
    _tag("decode_self_attn", n, n_head, S)
    _check(load().st_decode_self_attn(_stream(), qkv.data_ptr(), qkv.stride(0), cache.data_ptr(), step.data_ptr(),
                                      _lineage(anc, n, S, "decode_self_attn"), ctx.data_ptr(), ctx.stride(0), n, S, int(n_head),
                                      d // n_head, float(scale)), "st_decode_self_attn")


def beam_advance(logits, V, beam, step, eos, scores, tokens, done, lengths, hist_scores, back, toks, order, work=None, anc=None,
                 advance_step=False, embed=None):
    """Beam.advance for all utterances on the device (see st_beam_advance): logits f32 [B * beam, >= V]; the state tensors
    are updated in place (scores f32 [B, beam], tokens i64 [B * beam], done bool [B], lengths i64 [B], hist_scores f32 /
    back i64 / toks i64 [S, B, beam] at row ``step`` (i64 [1], device)); order i64 [B * beam] receives the cache rows.
    ``work``: optional zeroed i64 [>= beam_work_words(B, beam)] scratch - with it the step runs over B * beam workgroups.
    ``anc``: optional lineage table i32 [B * beam, S'] of decode_self_attn, updated for the new hypotheses (needs ``work``).
    ``advance_step``: the launch also does ``step += 1`` (needs ``work``).
    ``embed`` = (emb f32 [V', D], pe f32 [P, D], x_next bf16 [B * beam, D]): the launch also writes the next step's decoder
    input for the chosen tokens (embed_step's arithmetic at position step + 1; needs ``work``)."""
    B = scores.shape[0]
    if not (logits.is_cuda and logits.dtype == F32 and logits.dim() == 2 and logits.stride(1) == 1 and logits.shape[0] == B * beam):
        raise ValueError("beam_advance: logits must be an fp32 [B * beam, >= V] row matrix on the GPU")
    for t, dt, n, name in ((scores, F32, B * beam, "scores"), (tokens, I64, B * beam, "tokens"), (lengths, I64, B, "lengths"),
                           (order, I64, B * beam, "order"), (step, I64, 1, "step")):
        _vec(t, dt, n, name)
    if done.dtype != torch.bool or not done.is_contiguous() or done.numel() != B or not done.is_cuda:
        raise ValueError("beam_advance: done must be a contiguous bool [B] tensor on the GPU")
    S = hist_scores.shape[0]
    for t, dt, name in ((hist_scores, F32, "hist_scores"), (back, I64, "back"), (toks, I64, "toks")):
        if tuple(t.shape) != (S, B, beam) or t.dtype != dt or not t.is_contiguous() or not t.is_cuda:
            raise ValueError("beam_advance: %s must be a contiguous [S, B, beam] tensor" % name)
    if work is not None and not (work.is_cuda and work.dtype == I64 and work.is_contiguous() and work.numel() >= beam_work_words(B, beam)):
        raise ValueError("beam_advance: work must be a contiguous int64 tensor of >= beam_work_words(B, beam) elements on the GPU")
    if (advance_step or embed is not None) and work is None:
        raise ValueError("beam_advance: advance_step / embed need the work buffer")
    emb = pe = x_next = None
    if embed is not None:
        emb, pe, x_next = embed
        if not (emb.is_cuda and emb.dtype == F32 and emb.is_contiguous() and pe.is_cuda and pe.dtype == F32 and pe.is_contiguous()
                and emb.dim() == 2 and pe.dim() == 2 and emb.shape[1] == pe.shape[1] and x_next.is_cuda and x_next.dtype == BF16
                and x_next.is_contiguous() and tuple(x_next.shape) == (B * beam, emb.shape[1])):
            raise ValueError("beam_advance: embed = (emb f32 [V', D], pe f32 [P, D], x_next bf16 [B * beam, D]), contiguous, on the GPU")
    _tag("beam_advance", B, beam, V)
    _check(load().st_beam_advance(_stream(), logits.data_ptr(), logits.stride(0), int(V), int(beam), B, step.data_ptr(), int(eos),
                                  scores.data_ptr(), tokens.data_ptr(), done.data_ptr(), lengths.data_ptr(),
                                  hist_scores.data_ptr(), back.data_ptr(), toks.data_ptr(), order.data_ptr(),
                                  work.data_ptr() if work is not None else None,
                                  _lineage(anc, B * beam, anc.shape[1] if anc is not None and anc.dim() == 2 else 0, "beam_advance"),
                                  int(anc.shape[1]) if anc is not None else 0,
                                  step.data_ptr() if advance_step else None, _p(emb), 0 if emb is None else emb.shape[0], _p(pe),
                                  0 if pe is None else pe.shape[0], _p(x_next), 0 if emb is None else emb.shape[1]), "st_beam_advance")


def beam_work_words(B: int, beam: int) -> int:
    """int64 elements of beam_advance's ``work`` scratch (zero-initialised): beam keys per hypothesis row, the step ticket, a
    ticket per utterance."""
    return B * beam * beam + 1 + B


def cache_reorder(cache, order, step, beam):
    """cache bf16 [L, n, S, W] (contiguous): rows of every utterance re-gathered by ``order`` (int64 [n], device) for the
    positions <= step (int64 [1], device) - see st_cache_reorder."""
    if not (cache.is_cuda and cache.dtype == BF16 and cache.is_contiguous() and cache.dim() == 4):
        raise ValueError("cache_reorder: cache must be a contiguous bf16 [L, n, S, W] tensor on the GPU")
    L, n, S, W = cache.shape
    _vec(order, I64, n, "order"), _vec(step, I64, 1, "step")
    _check(load().st_cache_reorder(_stream(), cache.data_ptr(), order.data_ptr(), step.data_ptr(), L, n, S, W, int(beam)),
           "st_cache_reorder")
    return cache


def _ce_target(target, index, R, who):
    """target i64 [R], or any i64 vector addressed through index (i64 [R])."""
    if index is None:
        _vec(target, I64, R, "target")
    else:
        _vec(index, I64, R, "target_index")
        if not (target.is_cuda and target.dtype == I64 and target.is_contiguous()):
            raise ValueError("%s: target must be a contiguous int64 tensor on the GPU" % who)


def ce_fwd(logits, target, ignore_index, lse, sums, V=None, index=None):
    """lse[r] = logsumexp(logits[r, :V]); sums = (sum of the non-ignored rows' losses, their count, the mean = the loss)
    - see st_ce_fwd.  index: row r's target is target.view(-1)[index[r]]."""
    if not (logits.is_cuda and logits.dtype == F32 and logits.dim() == 2 and logits.stride(1) == 1):
        raise ValueError("ce_fwd: logits must be an fp32 row matrix on the GPU")
    R = logits.shape[0]
    V = logits.shape[1] if V is None else V
    _ce_target(target, index, R, "ce_fwd")
    _vec(lse, F32, R, "lse"), _vec(sums, F32, 3, "sums")
    row_loss = torch.empty(R, dtype=F32, device=logits.device)       # scratch: summed by the call's second launch
    _tag("ce_fwd", R, V, 0, io=(2.0 * R * V, 8.0 * R))
    _check(load().st_ce_fwd(_stream(), logits.data_ptr(), logits.stride(0), R, V, target.data_ptr(), _p(index), int(ignore_index),
                            lse.data_ptr(), row_loss.data_ptr(), sums.data_ptr()), "st_ce_fwd")


def ce_bwd(logits, target, ignore_index, lse, sums, grad_out, dlogits, V=None, index=None):
    """dlogits (bf16, same shape as logits) = d(mean loss) / d(logits) * grad_out - see st_ce_bwd."""
    R = logits.shape[0]
    V = logits.shape[1] if V is None else V
    _mat(dlogits, BF16, "dlogits")
    if dlogits.shape[0] != R or dlogits.shape[1] < V or dlogits.stride(0) % 8 or dlogits.shape[1] != dlogits.stride(0):
        raise ValueError("ce_bwd: dlogits must be a contiguous bf16 [R, >= V] matrix with a row length that is a multiple of 8")
    _ce_target(target, index, R, "ce_bwd")
    _vec(grad_out, F32, 1, "grad_out")
    _tag("ce_bwd", R, V, 0, io=(4.0 * R * V, 8.0 * R))
    _check(load().st_ce_bwd(_stream(), logits.data_ptr(), logits.stride(0), R, V, target.data_ptr(), _p(index), int(ignore_index),
                            lse.data_ptr(), sums.data_ptr(), grad_out.data_ptr(), dlogits.data_ptr(), dlogits.stride(0)),
           "st_ce_bwd")


def zero_tails(table, n_max):
    """table: int64 device tensor [n_max * 4] of (address, bytes per row, capacity rows, address of the valid-row count) - see
    st_zero_tails."""
    if not (table.is_cuda and table.dtype == I64 and table.is_contiguous() and table.numel() >= 4 * n_max):
        raise ValueError("zero_tails: table must be a contiguous int64 GPU tensor of 4 * n_max elements")
    _tag("zero_tails", n_max)
    _check(load().st_zero_tails(_stream(), table.data_ptr(), int(n_max)), "st_zero_tails")


def zero_(t):
    """t.zero_() for a contiguous fp32 / bf16 GPU buffer whose byte count and address are multiples of 16 (the flat gradient buffer);
    anything else goes to torch."""
    nbytes = t.numel() * t.element_size()
    if not (t.is_cuda and t.is_contiguous()) or nbytes % 16 or t.data_ptr() % 16:
        return t.zero_()
    _tag("zero", nbytes, io=(float(nbytes),))
    _check(load().st_zero(_stream(), t.data_ptr(), nbytes), "st_zero")
    return t


_NORM_BLOCKS = None


def _norm_blocks() -> int:
    global _NORM_BLOCKS
    if _NORM_BLOCKS is None:      # (a host-side query: asked once, through the untimed handle)
        _NORM_BLOCKS = int(load()._cdll.st_grad_norm_blocks())
    return _NORM_BLOCKS


def grad_norm_scratch(device):
    """Zeroed scratch of st_grad_norm (block partials + the ticket): allocate once per gradient buffer."""
    return torch.zeros(_norm_blocks() + 1, dtype=F32, device=device)


def grad_norm(g, scratch, out, step=None, grad_scale=1.0):
    """out (fp32 scalar tensor) = grad_scale * ||g||_2 over the flat fp32 buffer g; step (fp32 scalar tensor, optional) += 1 -
    see st_grad_norm."""
    if not (g.is_cuda and g.dtype == F32 and g.is_contiguous() and g.numel() % 4 == 0):
        raise ValueError("grad_norm: g must be a contiguous fp32 GPU buffer of a multiple of 4 elements")
    _vec(scratch, F32, _norm_blocks() + 1, "scratch")
    for t, nm in ((out, "out"), (step, "step")):
        if t is not None and not (t.is_cuda and t.dtype == F32 and t.numel() == 1):
            raise ValueError("grad_norm: %s must be an fp32 scalar on the GPU" % nm)
    _tag("grad_norm", g.numel(), io=(4.0 * g.numel(),))
    _check(load().st_grad_norm(_stream(), g.data_ptr(), g.numel(), scratch.data_ptr(), out.data_ptr(), _p(step), float(grad_scale)),
           "st_grad_norm")
    return out


def cast_bf16(src, dst):
    assert src.is_cuda and dst.is_cuda and src.dtype == F32 and dst.dtype == BF16
    assert src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
    _tag("cast_bf16", src.numel(), io=(src, dst))
    _check(load().st_cast_bf16(_stream(), src.data_ptr(), dst.data_ptr(), src.numel()), "st_cast_bf16")
    return dst


def adam_clip(p, g, m, v, lr, step, gnorm, max_norm, beta1, beta2, eps, grad_scale=1.0):
    """In place: g *= grad_scale * min(1, max_norm / (gnorm + 1e-6)); (p, m, v) <- Adam(p, g, m, v; lr, step).  lr / step /
    gnorm are 0-dim fp32 device tensors (gnorm None: no clipping); grad_scale: see st_adam_clip (1 / world behind a summing
    all-reduce)."""
    for t in (p, g, m, v):
        assert t.is_cuda and t.dtype == F32 and t.is_contiguous() and t.numel() == p.numel()
    for t in (lr, step) + ((gnorm,) if gnorm is not None else ()):
        assert t.is_cuda and t.dtype == F32 and t.numel() == 1
    _tag("adam_clip", p.numel(), io=(32.0 * p.numel(),))      # p, g, m, v read; p, m, v written; g zeroed (fp32)
    _check(load().st_adam_clip(_stream(), p.numel(), p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), lr.data_ptr(),
                               step.data_ptr(), _p(gnorm), float(max_norm), float(beta1), float(beta2), float(eps),
                               float(grad_scale)), "st_adam_clip")


def probe_tr16(inp, out):
    _check(load().st_probe_tr16(_stream(), inp.data_ptr(), out.data_ptr()), "st_probe_tr16")
    return out


def probe_mfma(A, Bt, D):
    _check(load().st_probe_mfma(_stream(), A.data_ptr(), Bt.data_ptr(), D.data_ptr()), "st_probe_mfma")
    return D

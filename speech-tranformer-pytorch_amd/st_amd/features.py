"""GPU feature front-end: the reference's per-utterance numpy pipeline (Dataset.py: cmvn :89-92, concat_frame
:121-143, subsampling :145-153) as one kernel that writes straight into the ragged bf16 row layout the encoder
consumes (``Encoder.forward_rows(..., packed=(rows_matrix, rows))``).  Kaldi I/O stays on the host (out of scope)."""
from typing import Optional, Tuple

import torch

from . import native as nv
from .functional import Rows


def stack_frames(x: torch.Tensor, lengths: torch.Tensor, left: int, right: int, frame_rate: int = 10,
                 stats: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Rows]:
    """x fp32 [B, T, F] raw padded features on the GPU, lengths [B] valid frames (host or device),
    stats fp32 [B, 2, F+1] per-utterance Kaldi CMVN statistics (or None)
    -> (bf16 row matrix [sum(out_len), F*(1+left+right)], Rows of the subsampled utterances)."""
    if right > left:
        raise ValueError("stack_frames: right context > left context is a shape error in the reference (Dataset.py:139)")
    interval = 1 if frame_rate == 10 else int(frame_rate / 10)
    host = lengths.detach().to("cpu", torch.int64)
    out_len = (host + interval - 1) // interval
    rows = Rows.packed(out_len, x.device)
    B, T, F = x.shape
    width = F * (1 + left + right)
    ld = (width + 7) // 8 * 8
    out = torch.zeros(rows.total, ld, dtype=torch.bfloat16, device=x.device)
    nv.feat_stack(x.contiguous().float(), host.to(x.device, torch.int32), None if stats is None else stats.float().contiguous(),
                  left, right, interval, rows.off, rows.len, rows.max_len, out)
    return out[:, :width], rows

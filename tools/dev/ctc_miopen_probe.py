#!/usr/bin/env python3
"""Dev: does torch.nn.functional.ctc_loss reach MIOpen's CTC on this build, and how long do the two implementations take at the
config-4 shape (T = 1000, B = 32, small alphabet C = 51)?  MIOpen's path (torch's `_use_cudnn_ctc_loss`) wants blank 0, int32
targets on the host, equal input lengths - shorter utterances can be padded with frames whose blank log-probability is 0 and
every other -inf: the path stays on blank, the likelihood is unchanged."""
import sys
import time

import torch
import torch.nn.functional as F

dev = "cuda"
T, B, C, L = 1000, 32, 51, 50
torch.manual_seed(0)
g = torch.Generator().manual_seed(0)
in_len = torch.randint(500, T + 1, (B,), generator=g)
tgt_len = torch.randint(25, L + 1, (B,), generator=g)
tg = [torch.randint(1, C, (int(n),), generator=g) for n in tgt_len]
lp = torch.log_softmax(torch.randn(T, B, C, device=dev), -1)
# padded form: frames past an utterance's length put everything on blank
pad = lp.clone()
for b in range(B):
    n = int(in_len[b])
    pad[n:, b, :] = float("-inf")
    pad[n:, b, 0] = 0.0


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def native():
    x = lp.detach().requires_grad_(True)
    tp = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        tp[b, :len(tg[b])] = tg[b]
    nll = F.ctc_loss(x, tp.to(dev), in_len.tolist(), tgt_len.tolist(), blank=0, reduction="none", zero_infinity=True)
    (gr,) = torch.autograd.grad(nll.sum(), x)
    return nll.detach(), gr


def miopen():
    x = pad.detach().requires_grad_(True)
    flat = torch.cat(tg).to(torch.int32)                    # host, int32, concatenated
    nll = F.ctc_loss(x, flat, [T] * B, [int(n) for n in tgt_len], blank=0, reduction="none", zero_infinity=False)
    (gr,) = torch.autograd.grad(nll.sum(), x)
    return nll.detach(), gr


print("cudnn.enabled", torch.backends.cudnn.enabled, "is_available", torch.backends.cudnn.is_available())
a, ga = native()
print("native: %.3f ms per fwd+bwd" % timed(native))
try:
    with torch.backends.cudnn.flags(enabled=True, deterministic=True):
        b_, gb = miopen()
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            miopen()
        names = sorted({e.key for e in prof.key_averages()})
        print("kernels on the MIOpen-eligible call:", [n[:60] for n in names][:12])
        print("eligible form: %.3f ms per fwd+bwd; nll max |diff| vs native %.3e" % (timed(miopen), float((a - b_).abs().max())))
except Exception as e:  # noqa: BLE001
    print("MIOpen-eligible call failed:", type(e).__name__, e)

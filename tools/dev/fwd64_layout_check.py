#!/usr/bin/env python3
"""Dev: the long attention forward on the same utterances in a packed layout and in a padded one whose padding rows hold
garbage (large finite values): O, Ores and the LSE of the utterance rows must be bit-identical."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from st_amd import native as nv  # noqa: E402

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32
dev, H, dk = "cuda", 4, 64
d = H * dk
scale = 1 / math.sqrt(dk)
lens = [400, 370, 298, 389, 131, 257]
T = 400
torch.manual_seed(0)
B, M = len(lens), sum(lens)
qkv = (torch.randn(M, 3 * d, device=dev) * 0.7).to(BF16)
off = torch.tensor([sum(lens[:i]) for i in range(B)], dtype=I32, device=dev)
ln = torch.tensor(lens, dtype=I32, device=dev)
O1 = torch.zeros(M, d, dtype=BF16, device=dev)
R1 = torch.zeros(M, d, dtype=BF16, device=dev)
l1 = torch.zeros(H * M, dtype=F32, device=dev)
nv.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], O1, l1, off, ln, off, ln, H, T, False, scale, max_k=T, ores=R1)
# padded: utterance b at rows [b T, b T + len); the rest = garbage
Mp = B * T
pad = (torch.randn(Mp, 3 * d, device=dev) * 50).to(BF16)
offp = torch.arange(B, dtype=I32, device=dev) * T
for b in range(B):
    pad[b * T:b * T + lens[b]] = qkv[int(off[b]):int(off[b]) + lens[b]]
O2 = torch.zeros(Mp, d, dtype=BF16, device=dev)
R2 = torch.zeros(Mp, d, dtype=BF16, device=dev)
l2 = torch.zeros(H * Mp, dtype=F32, device=dev)
nv.attn_fwd(pad[:, :d], pad[:, d:2 * d], pad[:, 2 * d:], O2, l2, offp, ln, offp, ln, H, T, False, scale, max_k=T, ores=R2)
torch.cuda.synchronize()
ok = True
for b in range(B):
    a0, p0 = int(off[b]), b * T
    for nm, x, y in (("O", O1[a0:a0 + lens[b]], O2[p0:p0 + lens[b]]), ("Ores", R1[a0:a0 + lens[b]], R2[p0:p0 + lens[b]])):
        if not torch.equal(x, y):
            ok = False
            dd = (x.float() - y.float()).abs()
            rows = dd.amax(1).nonzero().flatten()
            print("utterance %d (len %d) %s differs: max %.3e, %d rows, first %s last %s" % (b, lens[b], nm, float(dd.max()), rows.numel(), rows[:5].tolist(), rows[-3:].tolist()))
    la = l1.view(H, M)[:, a0:a0 + lens[b]]
    lb = l2.view(H, Mp)[:, p0:p0 + lens[b]]
    if not torch.equal(la, lb):
        ok = False
        print("utterance %d lse differs: max %.3e" % (b, float((la - lb).abs().max())))
print("IDENTICAL" if ok else "DIFFERENT")

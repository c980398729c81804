#!/usr/bin/env python3
"""Dev: the CTC head's kernels (st_gemm + st_ctc_gather, torch ctc_loss on the small alphabet, st_ctc_dlogits, the backward GEMMs)
against plain torch autograd through a dense [B, T, V] log-softmax, at the config-2 size."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as func  # noqa: E402

from st_amd import functional as F_, native as nv, synthetic  # noqa: E402
from transformer.Loss import CTCAttentionLoss  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
nB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
_, _, in_len, tgt_len, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
in_len, tgt_len, gt = in_len[:nB], tgt_len[:nB], gt[:nB]
L, T = int(tgt_len.max()), int(in_len.max())
gg = gt[:, :L].to(dev)
head = CTCAttentionLoss(256, 4337, ctc_weight=0.3).to(dev)
rows = F_.Rows.packed(in_len, dev)
R = int(in_len.sum())
enc = (torch.randn(R, 256, device=dev) * 0.7).to(torch.bfloat16).requires_grad_(True)
plan = head.plan(gg, tgt_len, in_len, rows)
head.zero_grad_buffers()
lp = head.project_rows(enc, plan)
ctc, g = head.ctc_rows(lp, plan)
plan.g_lp.copy_(g)
plan.roww.copy_(plan.finite.float() / (nB * plan.tl.float()))
lp.backward(plan.g_lp)
torch.cuda.synchronize()
# reference (fp32 logits from the same bf16 operands, fp64 afterwards)
Wb = head._st_wb[:4337].float()
z = (enc.detach().float() @ Wb.t() + head.ctc_proj.bias.detach().float()).double().requires_grad_(True)
idx = rows.scatter_index(T)
zp = torch.zeros(nB * T, 4337, dtype=torch.float64, device=dev).index_copy(0, idx, z)
logp = func.log_softmax(zp.view(nB, T, -1), -1).transpose(0, 1)
ref = func.ctc_loss(logp, gg, in_len, tgt_len, blank=0, reduction="mean", zero_infinity=True)
(dz,) = torch.autograd.grad(ref, z)
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
print("ctc %.6f ref %.6f" % (float(ctc), float(ref)))
# rebuild dl as the backward did
dl = torch.empty(R, plan.v_pad, dtype=torch.bfloat16, device=dev)
logits = torch.empty(R, plan.v_pad, dtype=torch.float32, device=dev)
nv.gemm(enc.detach(), head._st_wb, logits, epi=nv.EPI_F32, bias=head._st_bias)
lse = torch.empty(R, dtype=torch.float32, device=dev)
lp2 = torch.zeros_like(plan.lp)
nv.ctc_gather(logits, plan.rowmap, plan.T, plan.cols, lse, lp2)
print("logits vs ref z: %.2e   lse: %.2e" % (rel(logits[:, :4337], z.detach()), rel(lse, torch.logsumexp(z.detach(), 1))))
nv.ctc_dlogits(logits, lse, plan.rowmap, plan.T, plan.roww, plan.scat, plan.g_lp, plan.one, dl, V=4337)
print("dlogits vs autograd dz: all %.3e" % rel(dl[:, :4337], dz))
lab = torch.zeros(R, 4337, dtype=torch.bool, device=dev)
bidx = torch.div(plan.rowmap, T, rounding_mode="floor")
sc = plan.scat[bidx].long()                 # [R, C]
rr = torch.arange(R, device=dev).view(-1, 1).expand_as(sc)
ok = sc >= 0
lab[rr[ok], sc[ok]] = True
print("   label columns: %.3e (|ref| %.3e)   dense columns: %.3e (|ref| %.3e)" % (rel(dl[:, :4337][lab], dz[lab]), dz[lab].norm(), rel(dl[:, :4337][~lab], dz[~lab]), dz[~lab].norm()))
gW = (dz.t() @ enc.detach().double())
gE = dz @ Wb.double()
print("dW rel %.3e   dEnc rel %.3e   db rel %.3e" % (rel(head.ctc_proj.weight.grad, gW), rel(enc.grad, gE), rel(head.ctc_proj.bias.grad, dz.sum(0))))
print("dW from kernel dl (torch matmul): %.3e ; dEnc from kernel dl: %.3e" % (rel(dl[:, :4337].double().t() @ enc.detach().double(), gW), rel(dl[:, :4337].double() @ Wb.double(), gE)))

#!/usr/bin/env python3
"""Dev: gradients of one config-2 step with the few-queries kernels (default) and without (ST_ATTN_XS=0), same weights and batch:
which parameter gradients differ, and by how much (one process per setting: `ST_ATTN_XS=1 python xs_grad_diff.py 32; ST_ATTN_XS=0 python
xs_grad_diff.py 32; python xs_grad_diff.py 32 diff`) (the forward tensors are bit-identical by test_attn_sf1_fwd_equals_the_three_launches
except the self-attention's bf16 residual)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import transformer.Models as M  # noqa: E402
import transformer.Utils as U  # noqa: E402
from st_amd import synthetic  # noqa: E402
from st_amd.arena import arena_of  # noqa: E402

C2 = dict(feature_dim=80, max_inputs_length=1000, max_target_length=50, num_enc_layer=6, num_dec_layer=6, n_heads=4,
          d_k=64, d_v=64, d_model=256, d_inner_hid=1024, dropout=0.1, vocab_size=4337)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x, tokens, in_len, tgt_len, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
x, tokens, in_len, tgt_len, gt = x[:n].cuda(), tokens[:n].cuda(), in_len[:n], tgt_len[:n], gt[:n].cuda()
L = int(tgt_len.max())
mode = os.environ.get("ST_ATTN_XS", "1")
if len(sys.argv) > 2 and sys.argv[2] == "diff":
    g1, g0 = torch.load("/tmp/xs_grads_1.pt"), torch.load("/tmp/xs_grads_0.pt")
    rows = []
    for k in g1:
        a, b = g1[k], g0[k]
        rows.append((float((a - b).norm() / b.norm().clamp_min(1e-300)), k, float(b.norm())))
    rows.sort(reverse=True)
    for r in [r for r in rows if "linear_k.bias" not in r[1]][:14]:
        print("  %.3e  %-55s |g| %.3e" % r)
    import statistics
    print("median over tensors: %.3e" % statistics.median(r[0] for r in rows))
    print("tensors that differ at all: %d of %d" % (sum(1 for r in rows if r[0] > 0), len(rows)))
    sys.exit(0)
torch.manual_seed(0)
model = M.Transformer(U.AttrDict(C2))
U.init_parameters(model)
model = model.eval().cuda()
logits, rows = model.forward_packed(x, in_len, tokens[:, :L], tgt_len)
valid = (torch.arange(L).view(1, -1) < tgt_len.view(-1, 1)).cuda()
loss = torch.nn.functional.cross_entropy(logits.float(), gt[:, :L][valid], ignore_index=0)
loss.backward()
torch.cuda.synchronize()
ar = arena_of(model)
torch.save({k: ar.grad_view(p).detach().double().cpu().clone() for k, p in model.named_parameters()}, "/tmp/xs_grads_%s.pt" % mode)
print("ST_ATTN_XS=%s loss %.6f" % (mode, float(loss.detach())))

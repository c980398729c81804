#!/usr/bin/env python3
"""Dev: ONE batch through TrainStep(bucket=..., bucket_rows=...) (it exceeds the packed capacity: padded bucket layouts) and through
the eager packed step, from identical fresh models: loss, norm, gradient and weight differences."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import oracle as orc  # noqa: E402
import transformer.Models as M  # noqa: E402
import transformer.Utils as U  # noqa: E402
from st_amd.arena import arena_of  # noqa: E402
from st_amd.trainer import TrainStep  # noqa: E402
from transformer.Optim import ScheduledOptim  # noqa: E402

T_cap, L_cap = 400, 30
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 22
torch.manual_seed(5)
cfg = U.AttrDict(dict(feature_dim=80, max_inputs_length=T_cap, max_target_length=L_cap, num_enc_layer=2, num_dec_layer=2,
                      n_heads=4, d_k=64, d_v=64, d_model=256, d_inner_hid=512, dropout=0.0, vocab_size=30))
ma = M.Transformer(cfg)
U.init_parameters(ma)
mb = copy.deepcopy(ma)
ma, mb = ma.eval().cuda(), mb.eval().cuda()
oa = ScheduledOptim(ma, 256, U.AttrDict(n_warmup_steps=50))
ob = ScheduledOptim(mb, 256, U.AttrDict(n_warmup_steps=50))
sa = TrainStep(ma, oa, 30, 5.0, use_graph=False, bucket=(T_cap, L_cap), bucket_rows=(1300, 110))
sb = TrainStep(mb, ob, 30, 5.0, use_graph=False)
b = orc.synthetic_batch(4, T_cap, L_cap, 80, 30, seed=seed, t_min=260, l_min=4)
print("lens", b["in_len"].tolist(), "rows", int(b["in_len"].sum()))
T, L = int(b["in_len"].max()), int(b["tgt_len"].max())
x, tok, gt = b["x"][:, :T].cuda(), b["tokens"][:, :L].cuda(), b["gt"][:, :L].cuda()
la, ga = sa(x, b["in_len"], tok, b["tgt_len"], gt)
lb, gb = sb(x, b["in_len"], tok, b["tgt_len"], gt)
A, B = arena_of(ma), arena_of(mb)
print("loss %.7f vs %.7f   gnorm %.7f vs %.7f   grad rel %.3e   weights rel %.3e" %
      (float(la), float(lb), float(ga), float(gb), float((A.grad - B.grad).norm() / B.grad.norm()), float((A.flat - B.flat).norm() / B.flat.norm())))
rows = []
for (n, p), q in zip(ma.named_parameters(), mb.parameters()):
    g1, g2 = A.grad_view(p).double(), B.grad_view(q).double()
    rows.append((float((g1 - g2).norm() / g2.norm().clamp_min(1e-300)), n))
rows.sort(reverse=True)
for r in rows[:6]:
    print("   %.3e %s" % r)

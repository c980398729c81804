#!/usr/bin/env python3
"""Dev: the loss / norm / weight differences between the packed-bucket graph step and the eager packed step over the six batches
of tests/test_composition_cpu.py::run_bucket_mode at the long-input shape (the test's tolerances are 2e-3 / 3e-2 / 2e-3)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import copy  # noqa: E402

import torch  # noqa: E402

import oracle as orc  # noqa: E402
import transformer.Models as M  # noqa: E402
import transformer.Utils as U  # noqa: E402
from st_amd.trainer import TrainStep  # noqa: E402
from transformer.Optim import ScheduledOptim  # noqa: E402

T_cap, L_cap, t_min, bucket_rows = 400, 30, 260, (1300, 110)
torch.manual_seed(5)
cfg = U.AttrDict(dict(feature_dim=80, max_inputs_length=T_cap, max_target_length=L_cap, num_enc_layer=2, num_dec_layer=2,
                      n_heads=4, d_k=64, d_v=64, d_model=256, d_inner_hid=512, dropout=0.0, vocab_size=30))
ma = M.Transformer(cfg)
U.init_parameters(ma)
mb = copy.deepcopy(ma)
ma, mb = ma.eval().cuda(), mb.eval().cuda()
oa = ScheduledOptim(ma, 256, U.AttrDict(n_warmup_steps=50))
ob = ScheduledOptim(mb, 256, U.AttrDict(n_warmup_steps=50))
use_graph = os.environ.get("GRAPH", "1") == "1"
sa = TrainStep(ma, oa, 30, 5.0, use_graph=use_graph, graph_warmup=1, bucket=(T_cap, L_cap), bucket_rows=bucket_rows)
sb = TrainStep(mb, ob, 30, 5.0, use_graph=False)
for i in range(6):
    full = i == 3
    b = orc.synthetic_batch(4, T_cap, L_cap, 80, 30, seed=20 + i, t_min=T_cap if full else t_min, l_min=L_cap if full else 4)
    T, L = int(b["in_len"].max()), int(b["tgt_len"].max())
    x, tok, gt = b["x"][:, :T].cuda(), b["tokens"][:, :L].cuda(), b["gt"][:, :L].cuda()
    la, ga = sa(x, b["in_len"], tok, b["tgt_len"], gt)
    lb, gb = sb(x, b["in_len"], tok, b["tgt_len"], gt)
    da = torch.cat([p.detach().reshape(-1) for p in ma.parameters()]).double()
    db = torch.cat([p.detach().reshape(-1) for p in mb.parameters()]).double()
    print("batch %d lens %s: loss %.6f vs %.6f (rel %.2e)  gnorm rel %.2e  weights rel %.2e" %
          (i, b["in_len"].tolist(), float(la), float(lb), abs(float(la) - float(lb)) / abs(float(lb)), abs(float(ga) - float(gb)) / abs(float(gb)),
           float((da - db).norm() / db.norm())))

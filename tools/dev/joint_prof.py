#!/usr/bin/env python3
"""Dev: N steps of st_amd.trainer.JointTrainStep (BASELINE config 4) at the config-2 batch - run under rocprofv3 --kernel-trace."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import transformer.Models as M  # noqa: E402
import transformer.Utils as U  # noqa: E402
from st_amd import synthetic  # noqa: E402
from st_amd.trainer import JointTrainStep  # noqa: E402
from transformer.Loss import CTCAttentionLoss  # noqa: E402
from transformer.Optim import ScheduledOptim  # noqa: E402

C2 = dict(feature_dim=80, max_inputs_length=1000, max_target_length=50, num_enc_layer=6, num_dec_layer=6, n_heads=4,
          d_k=64, d_v=64, d_model=256, d_inner_hid=1024, dropout=0.1, vocab_size=4337)
torch.manual_seed(0)
model = M.Transformer(U.AttrDict(C2))
U.init_parameters(model)
model = model.eval().cuda()
x, tokens, in_len, tgt_len, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
xg, tg, gg = x.cuda(), tokens.cuda(), gt.cuda()
head = CTCAttentionLoss(256, 4337, ctc_weight=0.3).cuda()
head._st_prepare("cuda")
opt = ScheduledOptim(model, 256, U.AttrDict(n_warmup_steps=12000))
hopt = torch.optim.Adam(head.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9, capturable=True, fused=True)
step = JointTrainStep(model, opt, head, 5.0, head_optimizer=hopt, use_graph=True)
for _ in range(4):
    step(xg, in_len, tg, tgt_len, gg)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
t0 = time.perf_counter()
for _ in range(n):
    out = step(xg, in_len, tg, tgt_len, gg)
torch.cuda.synchronize()
print("joint step %.3f ms" % ((time.perf_counter() - t0) / n * 1e3), [float(v) for v in out[:3]])

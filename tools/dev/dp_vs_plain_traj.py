#!/usr/bin/env python3
"""Dev: the loss / gradient-norm trajectory of TrainStep with a one-rank RCCL reducer (collectives captured in the step graph)
against the plain single-graph step, same model and batch, config 2 or 3 (argv[1]), a few steps each."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import copy  # noqa: E402

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import transformer.Models as M  # noqa: E402
import transformer.Utils as U  # noqa: E402
from st_amd import dp, synthetic  # noqa: E402
from st_amd.arena import arena_of  # noqa: E402
from st_amd.trainer import TrainStep  # noqa: E402
from transformer.Optim import ScheduledOptim  # noqa: E402

cfgn = int(sys.argv[1]) if len(sys.argv) > 1 else 3
C = {2: dict(num_enc_layer=6, n_heads=4, d_model=256), 3: dict(num_enc_layer=12, n_heads=8, d_model=512)}[cfgn]
cfg = dict(feature_dim=80, max_inputs_length=1000, max_target_length=50, num_dec_layer=6, d_k=64, d_v=64, d_inner_hid=1024, dropout=0.1,
           vocab_size=4337, **C)
x, tokens, in_len, tgt_len, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
xg, tg, gg = x.cuda(), tokens.cuda(), gt.cuda()
torch.manual_seed(0)
m0 = M.Transformer(U.AttrDict(cfg))
U.init_parameters(m0)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
dist.init_process_group("nccl", rank=0, world_size=1)
out = {}
for mode in (("dp",) if os.environ.get("DP_ONLY") else ("plain", "dp")):
    m = copy.deepcopy(m0).eval().cuda()
    opt = ScheduledOptim(m, cfg["d_model"], U.AttrDict(n_warmup_steps=12000))
    red = dp.GradReducer(arena_of(m), bucket_bytes=int(os.environ.get("BUCKET_MB", "8")) << 20, force=True) if mode == "dp" else None
    step = TrainStep(m, opt, 4337, max_grad_norm=5.0, reducer=red, use_graph=True)
    traj = []
    for i in range(8):
        loss, gn = step(xg, in_len, tg, tgt_len, gg)
        traj.append((float(loss), float(gn)))
    out[mode] = traj
    print(mode, getattr(step, "dp_mode", None), "buckets", len(red.buckets) if red else 0, " ".join("%.4f/%.3f" % t for t in traj))
dist.destroy_process_group()

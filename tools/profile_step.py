#!/usr/bin/env python3
"""Dev tool: torch.profiler over one eager training step of the benchmark model - which ATen ops (and how many
device copies / fills) surround the HIP kernels.  Run through gpurun."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
import transformer.Models as M  # noqa: E402
import transformer.Utils as U  # noqa: E402
from st_amd import synthetic  # noqa: E402
from st_amd.trainer import TrainStep  # noqa: E402
from transformer.Optim import ScheduledOptim  # noqa: E402

torch.manual_seed(0)
model = M.Transformer(U.AttrDict(bench.C2))
U.init_parameters(model)
model = model.eval().cuda()
optim = ScheduledOptim(model, 256, U.AttrDict(n_warmup_steps=12000))
step = TrainStep(model, optim, 4337, 5.0, use_graph=False)
x, tok, il, tl, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
xg, tg, gg = x.cuda(), tok.cuda(), gt.cuda()
for _ in range(3):
    step(xg, il, tg, tl, gg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    step(xg, il, tg, tl, gg)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))

#!/usr/bin/env python3
"""Host-side view of one training step on the GPU box: enqueue time vs device time and a
cProfile of the Python launch path.  Development aid (not part of bench.py)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
import transformer.Models as M  # noqa: E402
import transformer.Utils as U  # noqa: E402
from transformer.Optim import ScheduledOptim  # noqa: E402
from st_amd import synthetic  # noqa: E402
from st_amd.trainer import TrainStep  # noqa: E402

torch.manual_seed(0)
model = M.Transformer(U.AttrDict(bench.C2))
U.init_parameters(model)
model = model.eval().cuda()
optim = ScheduledOptim(model, 256, U.AttrDict(n_warmup_steps=12000))
step = TrainStep(model, optim, 4337, 5.0)
x, tok, il, tl, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
xg, tg, gg = x.cuda(), tok.cuda(), gt.cuda()
for _ in range(3):
    step(xg, il, tg, tl, gg)
torch.cuda.synchronize()
for tag in ("a", "b"):
    t0 = time.perf_counter()
    for _ in range(10):
        step(xg, il, tg, tl, gg)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("10 steps: enqueue %.2f ms/step, +drain %.2f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step(xg, il, tg, tl, gg)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(45)

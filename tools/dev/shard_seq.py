"""Dev: kernel sequence of ONE graph-replayed step on a 4-utterance shard (one rank of the DP = 8 partition of the global B = 32
batch) - run under `rocprofv3 --kernel-trace -d DIR -o t -- python tools/dev/shard_seq.py`, then `step_segment.py DIR/t_results.db 12`."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import bench
import transformer.Models as M, transformer.Utils as U
from transformer.Optim import ScheduledOptim
from st_amd import synthetic
from st_amd.trainer import TrainStep
torch.manual_seed(0)
model = M.Transformer(U.AttrDict(bench.C2)); U.init_parameters(model); model = model.eval().cuda()
opt = ScheduledOptim(model, 256, U.AttrDict(n_warmup_steps=12000))
x, tok, il, tl, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
xs, ts, gs, ils, tls = x[:b].cuda(), tok[:b].cuda(), gt[:b].cuda(), il[:b], tl[:b]
step = TrainStep(model, opt, 4337, 5.0, use_graph=True)
for _ in range(20):
    step(xs, ils, ts, tls, gs)
torch.cuda.synchronize()

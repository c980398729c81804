"""Dev: step time of one GPU's shard of the global B = 32 batch at N = 1, 2, 4, 8 (strong scaling without the all-reduce)."""
import os, sys, time, torch
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
for n in (1, 2, 4, 8):
    b = 32 // n
    xs, ts, gs, ils, tls = x[:b].cuda(), tok[:b].cuda(), gt[:b].cuda(), il[:b], tl[:b]
    step = TrainStep(model, opt, 4337, 5.0, use_graph=True)
    for _ in range(5): step(xs, ils, ts, tls, gs)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): step(xs, ils, ts, tls, gs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print("N = %d: %2d utterances per GPU (%5d frames): %.3f ms/step -> %.2f M frames/s per GPU, x%d = %.2f M frames/s without exchange"
          % (n, b, int(ils.sum()), dt * 1e3, float(ils.sum()) / dt / 1e6, n, n * float(ils.sum()) / dt / 1e6))

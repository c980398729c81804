"""Dev: host-side profile (cProfile) of one EAGER TrainStep at BASELINE config 2."""
import cProfile, pstats, io, os, sys, time, torch
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
xs, ts, gs = x.cuda(), tok.cuda(), gt.cuda()
step = TrainStep(model, opt, 4337, 5.0, use_graph=False)
for _ in range(5): step(xs, il, ts, tl, gs)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): step(xs, il, ts, tl, gs)
torch.cuda.synchronize(); print("eager %.3f ms/step" % ((time.perf_counter() - t) / 20 * 1e3))
t = time.perf_counter()
for _ in range(20): step(xs, il, ts, tl, gs)
print("eager host-only (no final sync) %.3f ms/step" % ((time.perf_counter() - t) / 20 * 1e3)); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step(xs, il, ts, tl, gs)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])

"""Dev: TrainStep(bucket=(1000, 50)) at BASELINE config 2: one capture, batches with changing lengths; time and loss check."""
import os, sys, time, copy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import bench
import transformer.Models as M, transformer.Utils as U
from transformer.Optim import ScheduledOptim
from st_amd import synthetic
from st_amd.trainer import TrainStep
torch.manual_seed(0)
ma = M.Transformer(U.AttrDict(bench.C2)); U.init_parameters(ma); mb = copy.deepcopy(ma)
ma, mb = ma.eval().cuda(), mb.eval().cuda()
sa = TrainStep(ma, ScheduledOptim(ma, 256, U.AttrDict(n_warmup_steps=12000)), 4337, 5.0, use_graph=True, graph_warmup=1, bucket=(1000, 50))
sb = TrainStep(mb, ScheduledOptim(mb, 256, U.AttrDict(n_warmup_steps=12000)), 4337, 5.0, use_graph=False)
batches = []
for seed in range(6):
    x, tok, il, tl, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=seed, t_min=500, l_min=25)
    batches.append((x.cuda(), il, tok.cuda(), tl, gt.cuda()))
for i, b in enumerate(batches):
    la, ga = sa(*b); lb, gb = sb(*b)
    print("batch %d (%5d frames): bucket graph loss %.4f |g| %.4f   eager packed loss %.4f |g| %.4f" % (i, int(b[1].sum()), float(la), float(ga), float(lb), float(gb)))
torch.cuda.synchronize(); t = time.perf_counter()
for k in range(30): sa(*batches[k % 6])
torch.cuda.synchronize(); ta = (time.perf_counter() - t) / 30
t = time.perf_counter()
for k in range(30): sb(*batches[k % 6])
torch.cuda.synchronize(); tb = (time.perf_counter() - t) / 30
print("bucket-graph %.3f ms/step (32,000 padded rows)   eager packed %.3f ms/step (~24,000 rows), batches cycling through 6 length sets" % (ta * 1e3, tb * 1e3))

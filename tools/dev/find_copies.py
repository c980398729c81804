"""Dev tool: which ops of one eager training step issue device-to-device copies / tiny torch kernels."""
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
step = TrainStep(model, opt, 4337, 5.0, use_graph=False)
x, tok, il, tl, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
xg, tg, gg = x.cuda(), tok.cuda(), gt.cuda()
for _ in range(3): step(xg, il, tg, tl, gg)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(xg, il, tg, tl, gg)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if "emcpy" in e.name or "copy" in e.name.lower()]
from collections import Counter
c = Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA: continue
    if e.name in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::to", "aten::_to_copy", "aten::index_select", "aten::zero_", "aten::fill_", "aten::zeros", "aten::add_", "aten::mul_"):
        st = [s for s in (e.stack or []) if "st_amd" in s or "transformer/" in s or "trainer" in s][:2]
        c[(e.name, tuple(st))] += 1
for k, v in c.most_common(40): print(v, k)

"""Dev tool: node census of the captured training-step HIP graph (hipGraphDebugDotPrint via torch's debug_dump)."""
import collections, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import torch
import bench
import transformer.Models as M
import transformer.Utils as U
from st_amd import synthetic
from st_amd.trainer import TrainStep
from transformer.Optim import ScheduledOptim

torch.manual_seed(0)
model = M.Transformer(U.AttrDict(bench.C2)); U.init_parameters(model); model = model.eval().cuda()
optim = ScheduledOptim(model, 256, U.AttrDict(n_warmup_steps=12000))
step = TrainStep(model, optim, 4337, 5.0, use_graph=True)
x, tok, il, tl, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
xg, tg, gg = x.cuda(), tok.cuda(), gt.cuda()
orig = torch.cuda.CUDAGraph
class G(orig):
    def __new__(cls, *a, **k):
        g = orig.__new__(cls)
        return g
made = []
_old_init = orig.__init__
for _ in range(2):
    step(xg, il, tg, tl, gg)
# third call captures: turn debug mode on for the graph object it creates
import st_amd.trainer as T
class Wrap:
    def __call__(self):
        g = orig()
        g.enable_debug_mode()
        made.append(g)
        return g
real = torch.cuda.CUDAGraph
torch.cuda.CUDAGraph = Wrap()
step(xg, il, tg, tl, gg)
torch.cuda.CUDAGraph = real
torch.cuda.synchronize()
made[0].debug_dump("/tmp/step.dot")
txt = open("/tmp/step.dot").read()
labels = re.findall(r'label="([^"]*)"', txt)
c = collections.Counter()
for l in labels:
    name = l.split("\\n")[0][:60]
    name = re.sub(r"\d+", "#", name)
    c[name] += 1
print(len(labels), "nodes")
for n, k in c.most_common(40):
    print("%5d  %s" % (k, n))

"""Dev: the packed-bucket TrainStep of bench.py's loader-proof block alone (six batches with different lengths, one captured
graph over a fixed row capacity), for rocprofv3 --kernel-trace --stats: which launches cost more than on the fixed batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import torch
import bench as B
import transformer.Models as M, transformer.Utils as U
from st_amd import synthetic
from st_amd.trainer import TrainStep
from transformer.Optim import ScheduledOptim
torch.manual_seed(0)
CFG = B.CFG
model = M.Transformer(U.AttrDict(CFG)).cuda().eval()
U.init_parameters(model)
optim = ScheduledOptim(model, 256, U.AttrDict(n_warmup_steps=12000))
bs = []
for sd in range(6):
    bx, bt, bil, btl, bgt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=100 + sd, t_min=500, l_min=25)
    bs.append((bx.cuda(), bil, bt.cuda(), btl, bgt.cuda()))
tc = int(32 * (25 + 50) / 2 * 1.2) // 32 * 32
caps = [(256 * 96, tc), (420 * 64, tc)]
mode = sys.argv[1] if len(sys.argv) > 1 else "packed"
kw = dict(use_graph=True, graph_warmup=1, bucket=(1000, 50), bucket_rows=caps) if mode == "packed" else dict(use_graph=True, graph_warmup=1)
st = TrainStep(model, optim, 4337, max_grad_norm=5.0, **kw)
n = 6 if mode == "packed" else 1
for k in range(12):
    st(*bs[k % n])
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(18):
    st(*bs[k % n])
torch.cuda.synchronize()
print("%s: %.3f ms/step; rows %s" % (mode, (time.perf_counter() - t0) / 18 * 1e3, [int(b[1].sum()) for b in bs[:n]]))

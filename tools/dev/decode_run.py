"""Dev tool: two beam-10 decodes of BASELINE config 5 (for rocprofv3 --kernel-trace)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import torch
import bench
import transformer.Models as M
import transformer.Utils as U
from st_amd import synthetic
from transformer.Decode import Decode
torch.manual_seed(0)
model = M.Transformer(U.AttrDict(bench.C2)); U.init_parameters(model); model = model.eval().cuda()
x, tok, il, tl, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
rec = Decode(U.AttrDict(beam_size=10, n_best=1, max_steps=50, use_graph=(len(sys.argv) < 2)), "cuda", model=model)
x = x.cuda()
rec.decode_batch((x, il)); torch.cuda.synchronize()
t = time.perf_counter(); rec.decode_batch((x, il)); torch.cuda.synchronize(); print("decode s", time.perf_counter() - t)

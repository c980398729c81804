"""Dev: wall time of consecutive Decode.decode_batch calls at the bench's shape (beam 10, B = 32, 50 steps) - what the first
call at a shape pays once (allocator growth, layouts, kernel modules) against the steady state."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import torch
import bench as B
import transformer.Models as M, transformer.Utils as U
from transformer.Decode import Decode
from st_amd import synthetic
torch.manual_seed(0)
model = M.Transformer(U.AttrDict(B.CFG)).cuda().eval()
U.init_parameters(model)
x, tok, in_len, tl, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
xg = x.cuda()
dec = Decode(U.AttrDict(beam_size=10, n_best=1, max_steps=50), "cuda", model=model)
dec.decode_batch((xg[:4], in_len[:4]))
torch.cuda.synchronize()
for i in range(5):
    t0 = time.perf_counter()
    hyps, _ = dec.decode_batch((xg, in_len))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("call %d: %.2f ms = %.0f utterances/s (%d steps)" % (i, dt * 1e3, 32 / dt, len(hyps[0][0])))

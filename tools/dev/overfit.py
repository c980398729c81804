"""Dev: the captured training step on ONE fixed batch for a few hundred steps - the loss must fall (eval and train mode)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import bench
import transformer.Models as M, transformer.Utils as U
from transformer.Optim import ScheduledOptim
from st_amd import synthetic
from st_amd.trainer import TrainStep
x, tok, il, tl, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
xs, ts, gs = x.cuda(), tok.cuda(), gt.cuda()
for training in (False, True):
    torch.manual_seed(0)
    model = M.Transformer(U.AttrDict(bench.C2)); U.init_parameters(model); model = model.cuda().train(training)
    opt = ScheduledOptim(model, 256, U.AttrDict(n_warmup_steps=4000))
    step = TrainStep(model, opt, 4337, 5.0, use_graph=True)
    out = []
    t = time.perf_counter()
    for i in range(1201):
        loss, gn = step(xs, il, ts, tl, gs)
        if i % 150 == 0: out.append("%d: %.3f (|g| %.2f)" % (i, float(loss), float(gn)))
    torch.cuda.synchronize()
    print("train() =", training, " loss by step:", "  ".join(out), " [%.2f ms/step incl. the host reads]" % ((time.perf_counter() - t) / 1201 * 1e3))
    assert torch.isfinite(torch.tensor(float(loss)))

export TMPDIR=/tmp; cd /root/repo
python -m pytest tests/test_kernels_gpu.py -x -q -k "row_chain and 3120" 2>&1 | tail -3
for r in 1 2; do for V in d 1; do
ST_CHAIN_SPLIT=$V python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/speech-tranformer-pytorch_amd')
import bench
import transformer.Models as M, transformer.Utils as U
from transformer.Optim import ScheduledOptim
from st_amd import synthetic
from st_amd.trainer import TrainStep
torch.manual_seed(0)
model = M.Transformer(U.AttrDict(bench.C2)); U.init_parameters(model); model = model.eval().cuda()
opt = ScheduledOptim(model, 256, U.AttrDict(n_warmup_steps=12000))
x, tok, il, tl, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
out = []
for b in (4, 2, 8):
    xs, ts, gs, ils, tls = x[:b].cuda(), tok[:b].cuda(), gt[:b].cuda(), il[:b], tl[:b]
    step = TrainStep(model, opt, 4337, 5.0, use_graph=True)
    for _ in range(6): l = step(xs, ils, ts, tls, gs)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(40): l = step(xs, ils, ts, tls, gs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 40
    out.append("%d utt (%d rows): %.4f ms loss %.4f" % (b, int(ils.sum()), dt * 1e3, float(l[0])))
print("ST_CHAIN_SPLIT=%s  " % os.environ.get("ST_CHAIN_SPLIT"), " | ".join(out))
PY
done; done

import os, socket, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import torch.distributed as dist
from st_amd import dp, synthetic
from st_amd.arena import arena_of
from st_amd.trainer import TrainStep
from transformer.Models import Transformer
from transformer.Optim import ScheduledOptim
from transformer.Utils import AttrDict, init_parameters
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1)
cfg = AttrDict(dict(feature_dim=80, max_inputs_length=200, max_target_length=32, num_enc_layer=2, num_dec_layer=2, n_heads=4, d_k=32, d_v=32, d_model=128, d_inner_hid=256, dropout=0.0, vocab_size=30))
inputs, targets, in_len, tgt_len, truth = synthetic.make_batch(4, 160, 20, 80, 30, seed=1, t_min=60, l_min=6)
res = {}
for tag, with_reducer, graph in (("plain", False, False), ("red-eager", True, False), ("plain-graph", False, True), ("red-graph", True, True)):
    torch.manual_seed(0)
    model = Transformer(cfg).cuda(); init_parameters(model); model.eval()
    opt = ScheduledOptim(model, 128, AttrDict(n_warmup_steps=4000))
    red = dp.GradReducer(arena_of(model), bucket_bytes=64 << 10, force=True) if with_reducer else None
    step = TrainStep(model, opt, 30, 5.0, reducer=red, use_graph=graph, graph_warmup=1)
    x, t, gt = inputs.cuda(), targets.cuda(), truth.cuda()
    out = []
    for i in range(4):
        loss, gnorm = step(x, in_len, t, tgt_len, gt)
        torch.cuda.synchronize()
        out.append((round(float(loss), 5), round(float(gnorm), 4)))
        res[(tag, i)] = (arena_of(model).grad.detach().clone(), arena_of(model).flat.detach().clone())
    print(tag, out)
for i in range(4):
    for tag in ("red-eager", "plain-graph", "red-graph"):
        g0, p0 = res[("plain", i)]; g1, p1 = res[(tag, i)]
        print(i, tag, "grad rel %.3e  param rel %.3e" % (float((g1 - g0).norm() / g0.norm()), float((p1 - p0).norm() / p0.norm())))
dist.destroy_process_group()

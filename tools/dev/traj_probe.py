#!/usr/bin/env python3
"""Dev: where does the 20-step trajectory separate?  Runs the fp64 oracle's trajectory (config 2, 8 utterances, warmup 200),
and at every step loads the ORACLE's weights into the HIP model and compares the gradients at identical weights."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import oracle as orc  # noqa: E402
import transformer.Models as M  # noqa: E402
import transformer.Utils as U  # noqa: E402
from st_amd import functional as F_, synthetic  # noqa: E402
from st_amd.arena import arena_of  # noqa: E402

C2 = dict(feature_dim=80, max_inputs_length=1000, max_target_length=50, num_enc_layer=6, num_dec_layer=6, n_heads=4,
          d_k=64, d_v=64, d_model=256, d_inner_hid=1024, dropout=0.1, vocab_size=4337)
cfg, n_utts, warmup = C2, 8, 200
torch.manual_seed(0)
model = M.Transformer(U.AttrDict(cfg))
U.init_parameters(model)
w0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
model = model.eval().cuda()
x, tokens, in_len, tgt_len, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
x, tokens, in_len, tgt_len, gt = x[:n_utts], tokens[:n_utts], in_len[:n_utts], tgt_len[:n_utts], gt[:n_utts]
xg, tg, gg = x.cuda(), tokens.cuda(), gt.cuda()
L = int(tgt_len.max())
p64 = {k: v.double().cuda() for k, v in w0.items()}
b64 = {"x": xg.double(), "in_len": in_len, "tokens": tg, "tgt_len": tgt_len, "gt": gg}
adam = None
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()
for k in range(1, int(sys.argv[1]) + 1 if len(sys.argv) > 1 else 19):
    truth = orc.train_step(p64, b64, cfg["n_heads"], cfg["d_model"], warmup, k, 5.0, adam_state=adam)
    # HIP gradients at the oracle's weights p64 (before this step's update)
    model.load_state_dict({n: v.float() for n, v in p64.items()})
    arena = arena_of(model)
    arena.refresh() if hasattr(arena, "refresh") else None
    arena.zero_grads()
    lg, t_rows = model.forward_packed(xg, in_len, tg[:, :L], tgt_len)
    truth_gt = gg[:, :L].contiguous().view(-1).index_select(0, t_rows.scatter_index(L))
    loss = torch.nn.CrossEntropyLoss(ignore_index=0)(lg.float(), truth_gt)
    with F_.deferred_wgrads(True):
        loss.backward()
    torch.cuda.synchronize()
    rows, fg, ft = [], [], []
    for n, p in model.named_parameters():
        if "linear_k.bias" in n:
            continue
        g, t = arena.grad_view(p).detach().double(), truth["grads"][n]
        rows.append((rel(g, t), n, t.norm().item(), g.norm().item()))
        fg.append(g.reshape(-1)); ft.append(t.reshape(-1))
    G, T = torch.cat(fg), torch.cat(ft)
    rows.sort(reverse=True)
    print("step %2d loss HIP %.5f oracle %.5f | gnorm HIP %.5f oracle %.5f | global rel %.3e | worst: %s"
          % (k, loss.item(), truth["loss"].item(), G.norm().item(), T.norm().item(), rel(G, T),
             "; ".join("%s %.2e (|t| %.2e |g| %.2e)" % (r[1].replace("layer_stack.", "L"), r[0], r[2], r[3]) for r in rows[:4])), flush=True)
    adam, p64 = truth["adam"], truth["params"]

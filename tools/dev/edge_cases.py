"""Dev check: edge cases of the fused backward plumbing (LnLink / CrossKv / grouped weight gradients)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import torch
from st_amd import functional as F_, synthetic
from st_amd.arena import arena_of
from transformer.Models import Transformer
from transformer.Utils import AttrDict, init_parameters

cfg = AttrDict(dict(feature_dim=80, max_inputs_length=300, max_target_length=30, num_enc_layer=2, num_dec_layer=3,
                    n_heads=4, d_k=64, d_v=64, d_model=256, d_inner_hid=1024, dropout=0.0, vocab_size=500))
torch.manual_seed(0)
model = Transformer(cfg).cuda(); init_parameters(model); model.eval()
crit = torch.nn.CrossEntropyLoss(ignore_index=0)


def batch(B, tmax, lmax, tmin, lmin, seed):
    x, tok, il, tl, gt = synthetic.make_batch(B, tmax, lmax, 80, 500, seed=seed, t_min=tmin, l_min=lmin)
    return x.cuda(), tok.cuda(), il, tl, gt.cuda()


def loss_of(b):
    x, tok, il, tl, gt = b
    logits, _ = model(x, il, tok, tl)
    return crit(logits.reshape(-1, 500), gt.reshape(-1))


def grads():
    return arena_of(model).grad.detach().clone()


# (a) inference under no_grad
with torch.no_grad():
    l = loss_of(batch(3, 200, 20, 50, 5, 1))
print("no_grad forward ok", float(l))
# (b) retain_graph: two backward passes through one graph = twice the gradient
arena_of(model).zero_grads()
l = loss_of(batch(3, 200, 20, 50, 5, 1)); l.backward(retain_graph=True); g1 = grads(); l.backward(); g2 = grads()
print("retain_graph: |g2 - 2 g1| / |g2| =", float((g2 - 2 * g1).norm() / g2.norm()))
# (c) gradient accumulation over two batches = sum of the separate gradients
arena_of(model).zero_grads(); loss_of(batch(3, 200, 20, 50, 5, 1)).backward(); ga = grads()
arena_of(model).zero_grads(); loss_of(batch(2, 250, 25, 60, 6, 2)).backward(); gb = grads()
arena_of(model).zero_grads(); loss_of(batch(3, 200, 20, 50, 5, 1)).backward(); loss_of(batch(2, 250, 25, 60, 6, 2)).backward(); gab = grads()
print("accumulation: |gab - (ga + gb)| / |gab| =", float((gab - ga - gb).norm() / gab.norm()))
# (e) B = 1, very short sequences
for B, tmax, lmax, tmin, lmin in ((1, 40, 3, 40, 3), (1, 7, 1, 7, 1), (2, 65, 2, 1, 1), (5, 300, 30, 290, 29)):
    arena_of(model).zero_grads(); l = loss_of(batch(B, tmax, lmax, tmin, lmin, 3)); l.backward(); torch.cuda.synchronize()
    g = grads()
    print("B=%d T<=%d L<=%d: loss %.4f |g| %.4f finite=%s" % (B, tmax, lmax, float(l), float(g.norm()), bool(torch.isfinite(g).all())))
# (d) frozen encoder
for p in model.encoder.parameters():
    p.requires_grad_(False)
arena_of(model).zero_grads(); l = loss_of(batch(3, 200, 20, 50, 5, 1)); l.backward(); torch.cuda.synchronize()
print("frozen encoder ok", float(l), bool(torch.isfinite(grads()).all()))

import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
from st_amd import functional as F_, rng, synthetic
from st_amd.arena import arena_of
from transformer.Models import Transformer
from transformer.Utils import AttrDict, init_parameters
cfg = AttrDict(dict(feature_dim=80, max_inputs_length=1000, max_target_length=50, num_enc_layer=2, num_dec_layer=3,
                    n_heads=4, d_k=64, d_v=64, d_model=256, d_inner_hid=1024, dropout=0.1, vocab_size=4337))
torch.manual_seed(0)
model = Transformer(cfg).cuda(); init_parameters(model); rng.seed_tensor("cuda")
x, tokens, in_len, tgt_len, gt = synthetic.make_batch(6, 1000, 50, 80, 4337, seed=5, t_min=300, l_min=20)
xs, ts, gs = x.cuda(), tokens.cuda(), gt.cuda()
crit = torch.nn.CrossEntropyLoss(ignore_index=0)
def run(chains):
    model.decoder.use_row_chains = chains
    arena = arena_of(model); arena.zero_grads(); rng.manual_seed(77)
    logits, t_rows = model.forward_packed(xs, in_len, ts, tgt_len)
    truth = gs.contiguous().view(-1).index_select(0, t_rows.scatter_index(gs.shape[1]))
    loss = crit(logits, truth)
    with F_.deferred_wgrads(True):
        loss.backward()
    torch.cuda.synchronize()
    return logits.detach().clone(), loss.item(), arena.grad.detach().clone()
for training in (False, True):
    model.train(training)
    (lg1, l1, g1), (lg0, l0, g0) = run(True), run(False)
    print("training", training, "logits rel", float((lg1 - lg0).norm() / lg0.norm()), "loss", l1, l0, "grad rel", float((g1 - g0).norm() / g0.norm()))
    a = arena_of(model)
    worst = []
    for n, p in model.named_parameters():
        o = a.offset[id(p)]
        d = float((g1[o:o+p.numel()] - g0[o:o+p.numel()]).norm() / (g0[o:o+p.numel()].norm() + 1e-30))
        worst.append((d, n))
    worst.sort(reverse=True)
    print(worst[:6])
print("---- per-site")
model.train(True)
def setp(att, d1, d2):
    for m in model.modules():
        n = type(m).__name__
        if n == "MultiHeadAttention": m.dropout.p = att
        if n == "PositionwiseFeedForward": m.dropout1.p, m.dropout2.p = d1, d2
for name, cfgp in (("attn only", (0.1, 0, 0)), ("d1 only", (0, 0.1, 0)), ("d2 only", (0, 0, 0.1)), ("none(train)", (0, 0, 0))):
    setp(*cfgp)
    (lg1, l1, g1), (lg0, l0, g0) = run(True), run(False)
    print(name, "logits rel", float((lg1 - lg0).norm() / lg0.norm()))
(lg1, l1, g1), (lg0, l0, g0) = run(False), run(False)
print("unfused twice: logits rel", float((lg1 - lg0).norm() / lg0.norm()))
(lg1, l1, g1), (lg0, l0, g0) = run(True), run(True)
print("fused twice: logits rel", float((lg1 - lg0).norm() / lg0.norm()))
model.eval()
(lg1, l1, g1), (lg0, l0, g0) = run(True), run(False)
print("eval again: logits rel", float((lg1 - lg0).norm() / lg0.norm()))

"""Dev: is the product's backward deterministic?  Same model / batch twice in one process: max |g1 - g2| / |g| per
tensor, for the attention-only objective and for the joint CTC objective (PyTorch's CTC backward uses atomics)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import transformer.Models as M, transformer.Utils as U
from transformer.Loss import CTCAttentionLoss
from st_amd import synthetic
cfg = U.AttrDict(dict(feature_dim=80, max_inputs_length=100, max_target_length=20, num_enc_layer=2, num_dec_layer=2, n_heads=4,
                      d_k=32, d_v=32, d_model=128, d_inner_hid=256, dropout=0.0, vocab_size=30))
torch.manual_seed(0)
m = M.Transformer(cfg).cuda().eval()
U.init_parameters(m)
x, tokens, in_len, tgt_len, gt = synthetic.make_batch(3, 70, 9, 80, 30, seed=6, t_min=40, l_min=4)
L = int(tgt_len.max())
tokens, gt = tokens[:, :L].cuda(), gt[:, :L].cuda()
head = CTCAttentionLoss(128, 30, ctc_weight=0.3).cuda()
def grads(joint):
    m.zero_grad(set_to_none=True)
    logits, enc = m.forward_joint(x.cuda(), in_len, tokens, tgt_len)
    if joint:
        loss, _, _ = head(enc, in_len, logits, gt, tgt_len, gt)
    else:
        loss = torch.nn.CrossEntropyLoss(ignore_index=0)(logits.reshape(-1, 30), gt.reshape(-1))
    loss.backward()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
for joint in ((False, True) if len(sys.argv) == 1 else ()):
    worst = 0.0
    for rep in range(5):
        a, b = grads(joint), grads(joint)
        for n in a:
            d = float((a[n] - b[n]).norm() / (a[n].norm() + 1e-30))
            worst = max(worst, d)
    print("joint CTC objective" if joint else "attention objective ", "worst run-to-run relative difference of any gradient tensor: %.3e" % worst)
if len(sys.argv) > 1 and sys.argv[1] == "--cmp":
    pass
if len(sys.argv) > 1:      # cross-process check: python determinism_check.py <out.pt>, twice, then compare with --cmp a b
    if sys.argv[1] == "--cmp":
        a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
        for key in a:
            w = max(float((a[key][n] - b[key][n]).norm() / (a[key][n].norm() + 1e-30)) for n in a[key])
            wn = max(a[key], key=lambda n: float((a[key][n] - b[key][n]).norm() / (a[key][n].norm() + 1e-30)))
            print("%s: worst process-to-process relative difference %.3e (%s)" % (key, w, wn))
    else:
        torch.save({"attention": {n: g.cpu() for n, g in grads(False).items()}, "joint": {n: g.cpu() for n, g in grads(True).items()}}, sys.argv[1])

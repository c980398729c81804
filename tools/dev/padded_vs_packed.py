#!/usr/bin/env python3
"""Dev: logits of the same model and batch through the padded API (Transformer.forward: [B, T, F] inputs, padded layouts) and the
packed one (forward_packed) - the valid rows must agree to the last bits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import oracle as orc  # noqa: E402
import transformer.Models as M  # noqa: E402
import transformer.Utils as U  # noqa: E402

T_cap, L_cap = 400, 30
torch.manual_seed(5)
cfg = U.AttrDict(dict(feature_dim=80, max_inputs_length=T_cap, max_target_length=L_cap, num_enc_layer=2, num_dec_layer=2,
                      n_heads=4, d_k=64, d_v=64, d_model=256, d_inner_hid=512, dropout=0.0, vocab_size=30))
m = M.Transformer(cfg)
U.init_parameters(m)
m = m.eval().cuda()
b = orc.synthetic_batch(4, T_cap, L_cap, 80, 30, seed=22, t_min=260, l_min=4)
print("lens", b["in_len"].tolist(), b["tgt_len"].tolist())
T, L = int(b["in_len"].max()), int(b["tgt_len"].max())
x, tok = b["x"][:, :T].cuda(), b["tokens"][:, :L].cuda()
with torch.no_grad():
    lp, _ = m.forward_packed(x, b["in_len"], tok, b["tgt_len"])
    # padded layouts of the bucket kind: Rows.bucket via prepare_layouts of a TrainStep is internal; the padded API instead
    out = m(x, b["in_len"], tok, b["tgt_len"])
    ld = out[0] if isinstance(out, tuple) else out
valid = (torch.arange(L).view(1, -1) < b["tgt_len"].view(-1, 1)).cuda()
a_, b_ = lp.float(), ld[valid].float()
print("packed vs padded API logits: max |diff| %.3e  rel-L2 %.3e  identical %s" % (float((a_ - b_).abs().max()), float((a_ - b_).norm() / b_.norm()), torch.equal(a_, b_)))

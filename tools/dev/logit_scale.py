"""Dev: is there a systematic scale between the HIP logits and the fp64 oracle's? (least-squares gain per layer output)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import oracle as orc
import transformer.Models as M, transformer.Utils as U
d, dff, ne, nd = 256, 1024, 6, 6
p = orc.xavier_init_(orc.make_params(80, 30, d, dff, ne, nd, 100, 40, dtype=torch.float64), seed=1)
b = orc.synthetic_batch(5, 80, 10, 80, 30, seed=2, t_min=30, l_min=5)
cfg = U.AttrDict(dict(feature_dim=80, max_inputs_length=100, max_target_length=40, num_enc_layer=ne, num_dec_layer=nd, n_heads=4,
                      d_k=64, d_v=64, d_model=d, d_inner_hid=dff, dropout=0.0, vocab_size=30))
for chains in (True, False):
    m = M.Transformer(cfg); m.load_state_dict({k: v.float() for k, v in p.items()}); m = m.eval().cuda()
    m.encoder.use_row_chains = m.decoder.use_row_chains = chains
    L = int(b["tgt_len"].max())
    with torch.no_grad():
        lg, t_rows = m.forward_packed(b["x"].cuda(), b["in_len"], b["tokens"][:, :L].cuda(), b["tgt_len"])
    ref, _ = orc.transformer(p, b["x"].double(), b["in_len"], b["tokens"][:, :L], b["tgt_len"], 4)
    valid = (torch.arange(L).view(1, -1) < b["tgt_len"].view(-1, 1))
    r = ref[valid]
    h = lg.double().cpu()
    gain = float((h * r).sum() / (r * r).sum())
    print("chains", chains, "logits rel-L2 %.3e  least-squares gain %.5f  rel-L2 after removing the gain %.3e  |logit| rms %.3f"
          % (float((h - r).norm() / r.norm()), gain, float((h / gain - r).norm() / r.norm()), float(r.pow(2).mean().sqrt())))

import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
from st_amd import functional as F_, rng, synthetic, native as nv
from st_amd.arena import arena_of
from transformer.Models import Transformer
from transformer.Utils import AttrDict, init_parameters
cfg = AttrDict(dict(feature_dim=80, max_inputs_length=1000, max_target_length=50, num_enc_layer=2, num_dec_layer=3,
                    n_heads=4, d_k=64, d_v=64, d_model=256, d_inner_hid=1024, dropout=0.0, vocab_size=4337))
torch.manual_seed(0)
model = Transformer(cfg).cuda(); init_parameters(model); rng.seed_tensor("cuda")
x, tokens, in_len, tgt_len, gt = synthetic.make_batch(6, 1000, 50, 80, 4337, seed=5, t_min=300, l_min=20)
xs, ts = x.cuda(), tokens.cuda()
def rel(a, b): return float((a.float() - b.float()).norm() / b.float().norm())
for training in (False, True):
    model.train(training)
    rng.manual_seed(77)
    with torch.no_grad():
        arena = arena_of(model)
        in_rows, t_rows = model.prepare_layouts(in_len, tgt_len, ts.shape[1], xs.device)
        with arena.scope():
            enc, _ = model.encoder.forward_rows(xs, in_len, in_rows)
            dec = model.decoder
            y = F_.EmbedFn.apply(dec.tgt_word_emb.weight, dec, ts.contiguous(), t_rows)
            ckv = F_.CrossKv.plan([l.enc_attn for l in dec.layer_stack])
            kv = F_.CrossKvFn.apply(enc, dec.layer_stack[0].enc_attn.linear_k.weight, ckv)
            dc = dec.row_chains(arena)
            yo, pres = dc.forward(dec.layer_stack, y, kv, t_rows, in_rows, True)
            xcur = y
            for l, layer in enumerate(dec.layer_stack):
                a, b, f = pres[l]
                sa, ca, ff = layer.slf_attn._st, layer.enc_attn._st, layer.pos_ffn._st
                scale = 1 / math.sqrt(64)
                # feed the FUSED path's inputs to the unfused kernels stage by stage
                r = F_.MhaFn.compute(xcur if l == 0 else pres[l-1][2].out, None, sa, t_rows, t_rows, True, None, None, True, scale)
                print(training, l, "self: qkv %.2e ctx %.2e out %.2e" % (rel(a.qkv, r[0]), rel(a.ctx, r[2]), rel(a.out, r[5])))
                r2 = F_.MhaFn.compute(a.out, kv, ca, t_rows, in_rows, False, None, F_.CrossKvSlot(ckv, l), True, scale)
                print(training, l, "cross: q %.2e ctx %.2e out %.2e" % (rel(b.qkv, r2[0]), rel(b.ctx, r2[2]), rel(b.out, r2[5])))
                h = torch.empty_like(f.h); o = torch.empty_like(f.out); xh = torch.empty_like(f.out); rs = torch.empty(f.out.shape[0], device="cuda")
                F_.linear_fwd(b.out, ff.w1, h, ff.b1, relu=True)
                nv.gemm_ln(h, ff.w2, ff.b2, b.out, ff.gamma, ff.beta, o, xh, rs, eps=1e-6)
                print(training, l, "ffn: h %.2e out %.2e xhat %.2e" % (rel(f.h, h), rel(f.out, o), rel(f.xhat, xh)))

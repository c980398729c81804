"""Dev: decode throughput with st_beam_advance vs the torch-op formulation (tests/_emul.beam_advance on the GPU)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import torch
import bench
import transformer.Models as M
import transformer.Utils as U
from st_amd import synthetic, native as nv
from transformer.Decode import Decode
torch.manual_seed(0)
model = M.Transformer(U.AttrDict(bench.C2)); U.init_parameters(model); model = model.eval().cuda()
x, tok, il, tl, gt = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
x = x.cuda()
def torch_adv(logits, V, beam, step, eos, scores, tokens, done, lengths, hist_scores, back, toks, order):
    B = scores.shape[0]
    word_lk = torch.log_softmax(logits[:, :V].float(), dim=-1)
    table = (word_lk.view(B, beam, V) + scores.unsqueeze(2)).view(B, beam * V)
    best_scores, best_flat = table.topk(beam, 1, True, True)
    origin = best_flat // V
    token = best_flat - origin * V
    live = ~done
    lv = live.unsqueeze(1)
    hist_scores.index_copy_(0, step, scores.unsqueeze(0))
    scores.copy_(torch.where(lv, best_scores, scores))
    slot_ids = torch.arange(beam, device=scores.device).unsqueeze(0).expand(B, beam)
    origin = torch.where(lv, origin, slot_ids)
    back.index_copy_(0, step, origin.unsqueeze(0))
    toks.index_copy_(0, step, token.unsqueeze(0))
    tokens.copy_(torch.where(lv, token, tokens.view(B, beam)).view(-1))
    lengths.add_(live.to(lengths.dtype))
    done.logical_or_(live & (token[:, 0] == eos))
    order.copy_((origin + (torch.arange(B, device=scores.device) * beam).unsqueeze(1)).view(-1))
kern = nv.beam_advance
for name, fn in (("kernel", kern), ("torch", torch_adv), ("kernel", kern), ("torch", torch_adv)):
    nv.beam_advance = fn
    rec = Decode(U.AttrDict(beam_size=10, n_best=1, max_steps=50), "cuda", model=model)
    rec.decode_batch((x[:4], il[:4])); torch.cuda.synchronize()
    t = time.perf_counter(); rec.decode_batch((x, il)); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(name, "%.1f utt/s  %.3f ms/step" % (32 / dt, dt / 50 * 1e3))

import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import oracle as orc
from oracle import beam_oracle as bo
from tests import test_decode_cpu as dc
from transformer.Decode import Decode
from transformer.Utils import AttrDict
shape = (256, 1024, 6, 6)
p = dc._params(0.0, *shape)
SC = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
p["tgt_word_proj.weight"] = p["tgt_word_proj.weight"] * (SC / 12.0)
batch = orc.synthetic_batch(5, 80, 10, 80, 30, seed=2, t_min=30, l_min=5)
x, in_len = batch["x"], batch["in_len"]
model = dc._model(p, "cuda", *shape)
for ug in (True,):
    dec = Decode(AttrDict(dict(beam_size=10, n_best=2, max_steps=10, use_graph=ug)), "cuda", model=model)
    hyps, scores = dec.decode_batch((x, in_len))
    for b in range(5):
        for n in range(2):
            hyp = hyps[b][n]
            truth = bo.score_hypothesis(p, x[b:b+1].double(), in_len[b:b+1], 4, hyp)
            prefix = torch.tensor([[bo.BOS] + hyp[:-1]]).cuda()
            with torch.no_grad():
                lg, _ = model.forward_packed(x[b:b+1, :int(in_len[b])].cuda(), in_len[b:b+1], prefix, torch.tensor([prefix.shape[1]]))
            lp = torch.log_softmax(lg.float()[:, :30], -1)
            tf = float(sum(lp[t, tok] for t, tok in enumerate(hyp)))
            print(ug, b, n, "decode %.4f  hip-teacher-forced %.4f  fp64 %.4f" % (float(scores[b][n]), tf, truth))

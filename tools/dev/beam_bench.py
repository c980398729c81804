"""Dev: st_beam_advance at the decode bench's shape (B = 32, beam 10, V = 4337): one launch against the row-best + merge pair."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import torch
from st_amd import native as nv
B, beam, V, S = 32, 10, 4337, 64
ld = (V + 127) // 128 * 128
logits = torch.randn(B * beam, ld, device="cuda") * 3
def state():
    sc = torch.zeros(B, beam, device="cuda")
    return dict(scores=sc, tokens=torch.ones(B * beam, dtype=torch.long, device="cuda"), done=torch.zeros(B, dtype=torch.bool, device="cuda"),
                lengths=torch.zeros(B, dtype=torch.long, device="cuda"), hist=torch.zeros(S, B, beam, device="cuda"),
                back=torch.zeros(S, B, beam, dtype=torch.long, device="cuda"), toks=torch.zeros(S, B, beam, dtype=torch.long, device="cuda"),
                order=torch.zeros(B * beam, dtype=torch.long, device="cuda"), step=torch.zeros(1, dtype=torch.long, device="cuda"))
work = torch.zeros(nv.beam_work_words(B, beam), dtype=torch.long, device="cuda")
for name, w in (("one launch", None), ("B x beam workgroups", work), ("one launch", None), ("B x beam workgroups", work)):
    st = state()
    def call():
        st["scores"].zero_()
        nv.beam_advance(logits, V, beam, st["step"], 2, st["scores"], st["tokens"], st["done"], st["lengths"], st["hist"], st["back"], st["toks"], st["order"], work=w)
    for _ in range(5): call()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): call()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print("%-14s %.2f us per step (incl. a ~3 us zero_ of the scores)" % (name, e0.elapsed_time(e1) * 1e3 / 200))

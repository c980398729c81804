import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
from st_amd import native as nv
B, beam, V, S = 32, 10, 4337, 50
dev = "cuda"
logits = torch.randn(B * beam, 4344, device=dev) * 3
sc = torch.zeros(B, beam, device=dev)
st = dict(tokens=torch.ones(B * beam, dtype=torch.long, device=dev), done=torch.zeros(B, dtype=torch.bool, device=dev),
          lengths=torch.zeros(B, dtype=torch.long, device=dev), hist=torch.zeros(S, B, beam, device=dev),
          back=torch.zeros(S, B, beam, dtype=torch.long, device=dev), toks=torch.zeros(S, B, beam, dtype=torch.long, device=dev),
          order=torch.zeros(B * beam, dtype=torch.long, device=dev), step=torch.zeros(1, dtype=torch.long, device=dev))
def run(): nv.beam_advance(logits, V, beam, st["step"], 2, sc, st["tokens"], st["done"], st["lengths"], st["hist"], st["back"], st["toks"], st["order"])
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
print("beam_advance %.1f us" % (s.elapsed_time(e) / 20 * 1e3))

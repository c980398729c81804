import sys, torch
sys.path.insert(0, "/root/repo/speech-tranformer-pytorch_amd")
from st_amd import native as nv
x = torch.zeros(64, dtype=torch.float32, device="cuda"); y = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
N = 400
for _ in range(3):
    for _ in range(N): nv.cast_bf16(x, y)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(N): nv.cast_bf16(x, y)
for _ in range(3): g.replay()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): g.replay()
e.record(); torch.cuda.synchronize()
print("graph: %.2f us per tiny kernel node" % (s.elapsed_time(e) / 10 / N * 1e3))
s.record()
for _ in range(10):
    for _ in range(N): nv.cast_bf16(x, y)
e.record(); torch.cuda.synchronize()
print("eager: %.2f us per tiny kernel launch" % (s.elapsed_time(e) / 10 / N * 1e3))

import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
from st_amd import native as nv
M, N, K = 24060, int(sys.argv[1]) if len(sys.argv) > 1 else 3072, 256
X = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") / 16).to(torch.bfloat16)
b = torch.randn(N, device="cuda"); o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(10): nv.gemm_ws(X, W, o, bias=b)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
lib = nv.load()._cdll
lib.st_ws_prof_read.argtypes = [ctypes.c_void_p]
print("rc", lib.st_ws_prof_read(buf))
for w in range(8):
    pw, pr, pc, pn = buf[4*w:4*w+4]
    if pn: print("wave %d: steps %d  wait+barrier %.0f  role(DMA|store) %.0f  mfma+epilogue %.0f  (ticks per step)" % (w, pn, pw/pn, pr/pn, pc/pn))

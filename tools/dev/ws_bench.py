"""Weight-stationary GEMM (st_gemm_ws) vs the tiled GEMM (st_gemm) on the encoder's K = 256 products: parity + time."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
from st_amd import native as nv
torch.manual_seed(0)
dev = "cuda"

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for M, N, relu in ((24060, 768, False), (24060, 1024, True), (24060, 256, False), (24060, 3072, False), (1206, 768, False), (7000, 1024, True)):
    K = 256
    X = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / 16).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    ref = X.float() @ W.float().t() + b
    if relu: ref = torch.relu(ref)
    o1 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    o2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    epi = nv.EPI_BF16_RELU if relu else nv.EPI_BF16
    nv.gemm(X, W, o1, bias=b, epi=epi)
    nv.gemm_ws(X, W, o2, bias=b, relu=relu)
    torch.cuda.synchronize()
    e1 = ((o1.float() - ref).norm() / ref.norm()).item()
    e2 = ((o2.float() - ref).norm() / ref.norm()).item()
    same = torch.equal(o1, o2)
    t1 = timeit(lambda: nv.gemm(X, W, o1, bias=b, epi=epi))
    t2 = timeit(lambda: nv.gemm_ws(X, W, o2, bias=b, relu=relu))
    fl = 2.0 * M * N * K
    print("M %5d N %4d relu %d | tiled %.3e %6.1f us %6.0f TF | ws %.3e %6.1f us %6.0f TF | bit-equal %s"
          % (M, N, relu, e1, t1, fl / t1 / 1e6, e2, t2, fl / t2 / 1e6, same))

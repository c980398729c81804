#!/usr/bin/env python3
"""Dev tool: build tools/dev/st_gemm_ws.hip (the persistent wave-specialised A/B reference) with -DST_PROF and print per-workgroup phase cycles of one GEMM."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

src = os.path.join(ROOT, "tools", "dev", "st_gemm_ws.hip")
inc = os.path.join(ROOT, "speech-tranformer-pytorch_amd", "csrc")
so = os.path.join(ROOT, "gpurun_out", "libst_prof.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
if not os.path.exists(so):
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-DST_PROF", "-I", inc, "-fPIC",
                    "-shared", src, "-o", so], check=True)
lib = ctypes.CDLL(so)
V = ctypes.c_void_p
lib.st_gemm_ws.argtypes = [V, ctypes.c_int, ctypes.c_int, V, ctypes.c_int, V, ctypes.c_int, V, ctypes.c_int, ctypes.c_int,
                        ctypes.c_int, ctypes.c_int, V, V, ctypes.c_int, ctypes.c_int, ctypes.c_int]
M, N, K = 24060, 1024, 256
X = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda") * 0.5).to(torch.bfloat16)
b = torch.randn(N, device="cuda")
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    rc = lib.st_gemm_ws(st, 0, 0, X.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N, M, N, K, b.data_ptr(), None, 0, 1, 1)
    assert rc == 0
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
lib.st_gemm_ws(st, 0, 0, X.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N, M, N, K, b.data_ptr(), None, 0, 1, 1)
e.record()
torch.cuda.synchronize()
print("kernel us", s.elapsed_time(e) * 1e3)
buf = (ctypes.c_ulonglong * (256 * 8))()
lib.st_prof_read(buf)
import numpy as np
a = np.array(buf[:], dtype=np.float64).reshape(256, 8)
print("cols: barrier-wait, mfma, epilogue, total cycles, tiles, items")
print("mean", a.mean(0)[:6].round(0))
print("min ", a.min(0)[:6].round(0))
print("max ", a.max(0)[:6].round(0))
print("per tile: wait %.0f mfma %.0f epi %.0f total %.0f" % tuple(a[:, i].sum() / a[:, 4].sum() for i in range(4)))

# ---- where do the bytes come from?  alias operand rows (ld = 0) so a panel is a single cache line ----
def run(ldx, ldy, tag):
    for _ in range(3):
        lib.st_gemm_ws(st, 0, 0, X.data_ptr(), ldx, W.data_ptr(), ldy, out.data_ptr(), N, M, N, K, b.data_ptr(), None, 0, 1, 1)
    torch.cuda.synchronize()
    s.record()
    for _ in range(10):
        lib.st_gemm_ws(st, 0, 0, X.data_ptr(), ldx, W.data_ptr(), ldy, out.data_ptr(), N, M, N, K, b.data_ptr(), None, 0, 1, 1)
    e.record()
    torch.cuda.synchronize()
    lib.st_prof_read(buf)
    a = np.array(buf[:], dtype=np.float64).reshape(256, 8)
    print("%-28s %7.2f us | per tile: wait %.0f mfma %.0f epi %.0f total %.0f prefetch-block %.0f" % ((tag, s.elapsed_time(e) * 100) + tuple(
        a[:, i].sum() / a[:, 4].sum() for i in (0, 1, 2, 3, 6))))

run(K, K, "normal")
run(0, 0, "both aliased")
for flag, tag in ((1, "no output stores"), (2, "no operand loads"), (4, "no LDS tile writes"), (6, "no loads, no LDS writes"),
                  (7, "no loads/LDS writes/stores")):
    lib.st_prof_dbg(flag)
    run(K, K, tag)
lib.st_prof_dbg(0)

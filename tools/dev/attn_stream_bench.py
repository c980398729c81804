"""Dev: time the streaming-attention prototype (tools/dev/attn_stream_proto.hip) against st_attn_fwd at the encoder shape."""
import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
from st_amd import native as nv, synthetic
from st_amd import functional as F_
so = "/tmp/attn_stream_proto.so"
lib = ctypes.CDLL(os.environ.get("PROTO_SO", os.path.join(ROOT, "gpurun_in_proto.so")) if False else os.path.join(ROOT, "tools", "dev", "attn_stream_proto.so"))
V = ctypes.c_void_p
lib.proto_attn_stream.argtypes = [V, V, V, ctypes.c_int, V, ctypes.c_int, V, V, ctypes.c_int, V, ctypes.c_int, V]
dev = "cuda"
_, _, in_len, _, _ = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
rows = F_.Rows.packed(in_len, dev)
M, H, d = int(in_len.sum()), 4, 256
qkv = (torch.randn(M, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
ctx = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
lse = torch.empty(H * M, dtype=torch.float32, device=dev)
work = F_.attn_work(rows, rows, False)[0]
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
t_old = timeit(lambda: nv.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], ctx, lse, rows.off, rows.len, rows.off, rows.len, H,
                                   rows.max_len, False, 0.125, work=work, max_k=rows.max_len))
# synthetic fragment streams: per (b, h): ceil(T / 32) tiles x 8 fragments x 64 lanes x 8 bf16, + 32 fragments of padding
lens = in_len.tolist()
ntile = [(t + 31) // 32 for t in lens]
kv_off, tot = [], 0
for n in ntile:
    kv_off.append(tot); tot += H * n * 8 + 4
kv = (torch.randn((tot + 64) * 64 * 8, device=dev) * 0.3).to(torch.bfloat16)
kv_off_t = torch.tensor(kv_off, dtype=torch.int64, device=dev)
items = sorted(((lens[b], (b << 16) | qb) for b in range(32) for qb in range((lens[b] + 255) // 256)), reverse=True)
wk = torch.tensor([i[1] for i in items], dtype=torch.int32, device=dev)
out = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
def new():
    rc = lib.proto_attn_stream(st, kv.data_ptr(), qkv.data_ptr(), 3 * d, out.data_ptr(), d, rows.off.data_ptr(), rows.len.data_ptr(), H,
                               wk.data_ptr(), wk.numel(), kv_off_t.data_ptr())
    assert rc == 0, rc
t_new = timeit(new)
print("encoder self-attention forward: st_attn_fwd %.1f us   streaming prototype %.1f us   (%d work items x %d heads; finite: %s)"
      % (t_old, t_new, wk.numel(), H, bool(torch.isfinite(out.float()).all())))

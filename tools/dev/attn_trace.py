"""Dev: per-workgroup timeline of the long attention forward (needs st_dev_fwd64_trace: a temporary stamp hook in attn_fwd64_kernel (not in the tree; see DESIGN.md section 4, "Round 3, where the time of a launch goes" for what it measured)): item durations, prologue /
loop / epilogue split, concurrency per CU over time."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import torch
from st_amd import native as nv, synthetic
from st_amd.functional import Rows, attn_work
dev = "cuda"
_, _, in_len, _, _ = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
rows = Rows.packed(in_len, dev)
M, H, d = int(in_len.sum()), 4, 256
qkv = (torch.randn(M, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
ctx = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
lse = torch.empty(H * M, dtype=torch.float32, device=dev)
work = attn_work(rows, rows, False, 64, H)[0]
nwg = work.numel() * H
trace = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
lib = nv.load()._cdll
lib.st_dev_fwd64_trace.argtypes = [ctypes.c_void_p]
def run():
    nv.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], ctx, lse, rows.off, rows.len, rows.off, rows.len, H, rows.max_len, False, 0.125,
                work=work, max_k=rows.max_len)
for _ in range(3): run()
torch.cuda.synchronize()
assert lib.st_dev_fwd64_trace(trace.data_ptr()) == 0
run(); torch.cuda.synchronize()
lib.st_dev_fwd64_trace(None)
t = trace.view(nwg, 8).cpu()
live = t[:, 3] > 0
t = t[live]
t0 = t[:, 0].min()
us = lambda x: (x - t0).double() / 100.0          # wall_clock64: 100 MHz
st, l0, l1, en = us(t[:, 0]), us(t[:, 1]), us(t[:, 2]), us(t[:, 3])
print("workgroups %d (of %d launched); kernel span %.1f us" % (t.shape[0], nwg, en.max()))
print("prologue %.2f us avg (max %.2f); loop %.2f avg (min %.2f max %.2f); epilogue %.2f avg (max %.2f); item %.2f avg" % (
    (l0 - st).mean(), (l0 - st).max(), (l1 - l0).mean(), (l1 - l0).min(), (l1 - l0).max(), (en - l1).mean(), (en - l1).max(), (en - st).mean()))
ntile = ((t[:, 6] + 63) // 64).double()
print("loop time per 64-key tile: %.3f us avg" % ((l1 - l0) / ntile).mean())
print("start times: %d workgroups start before 1 us, %d before 5 us; last start %.1f us" % ((st < 1).sum(), (st < 5).sum(), st.max()))
hw = t[:, 4]
cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 16) & 0xf) << 8)     # cu_id, sh/se bits (layout indicative only)
print("distinct hw ids:", len(set(hw.tolist())), " distinct (cu,se) keys:", len(set(cu.tolist())))
for lo in range(0, int(math.ceil(en.max())), 4):
    active = ((st <= lo) & (en > lo)).sum().item()
    print("  t = %2d us: %4d workgroups resident" % (lo, active))

"""Dev: per-workgroup phase timeline of the encoder-sized forward row chain (needs the TR() hook patched into
csrc/st_rowchain.hip: st_dev_chain_trace)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import torch
from st_amd import native as nv, chains
dev = "cuda"
BF16, F32 = torch.bfloat16, torch.float32
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 24060
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF16)
d_, dff = 256, 1024
wo, wqkv, w1, w2 = rnd(d_, d_) * 0.1, rnd(3 * d_, d_) * 0.1, rnd(dff, d_) * 0.1, rnd(d_, dff) * 0.1
vec = lambda n: torch.randn(n, device=dev) * 0.1
bo, bqkv, b1, b2, g0, be0, g1, be1 = vec(d_), vec(3 * d_), vec(dff), vec(d_), vec(d_) + 1, vec(d_), vec(d_) + 1, vec(d_)
cs = chains.ChainSet(dev)
cf = cs.add(chains.blocks_of(wo) + chains.ffn_blocks(w1, w2) + chains.blocks_of(wqkv))
cs.finalize().rebuild()
chf = cs.chain(cf)
E = lambda *s, dtype=BF16: torch.empty(*s, dtype=dtype, device=dev)
ctx, x = rnd(rows, d_), rnd(rows, d_)
c_, xc, rc, h_, y_, xy, ry, p_ = E(rows, d_), E(rows, d_), E(rows, dtype=F32), E(rows, dff), E(rows, d_), E(rows, d_), E(rows, dtype=F32), E(rows, 3 * d_)
run = lambda: nv.row_chain(ctx, chf, pre=(x, bo, g0, be0, c_, xc, rc), ffn=(dff, b1, b2, g1, be1, h_, y_, xy, ry, None, None), post=(3, bqkv, p_))
for _ in range(3): run()
torch.cuda.synchronize()
nwg = (rows + 95) // 96 if rows > 64 * 256 else (rows + 63) // 64 if rows > 32 * 256 else (rows + 31) // 32
trace = torch.zeros(nwg * 32, dtype=torch.int64, device=dev)
lib = nv.load()._cdll if hasattr(nv.load(), "_cdll") else nv.load()
lib.st_dev_chain_trace.argtypes = [ctypes.c_void_p]
assert lib.st_dev_chain_trace(trace.data_ptr()) == 0
run(); torch.cuda.synchronize()
lib.st_dev_chain_trace(None)
full = trace.view(nwg, 32).cpu().double() / 100.0
t = full[:, :20]
t0 = t[:, 0].min()
names = ["start", "prologue+sync", "PRE mma", "PRE epi_ln"] + sum([["c%d mma1" % c, "c%d epi+sync" % c, "c%d mma2+out" % c] for c in range(4)], []) + ["FFN epi_ln", "POST0", "POST1", "POST2"]
print("workgroups %d; span %.1f us; start spread %.2f us; wg duration avg %.1f (min %.1f max %.1f)" % (
    nwg, (t[:, 19].max() - t0), (t[:, 0].max() - t0), (t[:, 19] - t[:, 0]).mean(), (t[:, 19] - t[:, 0]).min(), (t[:, 19] - t[:, 0]).max()))
for i in range(1, 20):
    dt = t[:, i] - t[:, i - 1]
    print("  %-14s %6.2f us avg  (min %5.2f  max %5.2f)   ends at %6.2f avg" % (names[i], dt.mean(), dt.min(), dt.max(), (t[:, i] - t0).mean()))
mma = sum((t[:, i] - t[:, i - 1]).mean() for i in (2, 4, 7, 10, 13)) + sum((t[:, i] - t[:, i - 1]).mean() for i in (6, 9, 12, 15)) + sum((t[:, i] - t[:, i - 1]).mean() for i in (17, 18, 19))
print("phases holding a block_mma: %.1f us of %.1f" % (mma, (t[:, 19] - t[:, 0]).mean()))

for base, nm in ((20, "PRE epi_ln"), (24, "FFN epi_ln")):
    end = t[:, 3] if base == 20 else t[:, 16]
    beg = t[:, 2] if base == 20 else t[:, 15]
    a, b, c = full[:, base], full[:, base + 1], full[:, base + 2]
    print("%s: sum pass + barrier %.2f | squares + barrier %.2f | normalise + LDS + barrier %.2f | copies out %.2f" % (
        nm, (a - beg).mean(), (b - a).mean(), (c - b).mean(), (end - c).mean()))
print("prologue: loads issued at %.2f us, A stored (= A arrived) at %.2f, sync at %.2f" % (
    (full[:, 28] - t[:, 0]).mean(), (full[:, 29] - t[:, 0]).mean(), (t[:, 1] - t[:, 0]).mean()))

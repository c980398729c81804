"""Dev: per-workgroup phase timeline of the pipelined forward row chain (csrc/st_rowchain_pipe.cuh; needs a library built with
ST_DEV_TRACE=1: the TRP() stamps and st_dev_chain_trace).  usage: ST_DEV_TRACE=1 python tools/dev/chain_pipe_trace.py [rows]"""
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
bits = torch.empty(nv.chain_mask_words(rows, dff), dtype=torch.int64, device=dev)
run = lambda: nv.row_chain(ctx, chf, pre=(x, bo, g0, be0, c_, xc, rc), ffn=(dff, b1, b2, g1, be1, h_, y_, xy, ry, None, None, bits), post=(3, bqkv, p_))
for _ in range(3): run()
torch.cuda.synchronize()
nwg = (rows + 95) // 96 if rows > 64 * 256 else (rows + 63) // 64
trace = torch.zeros(nwg * 32, dtype=torch.int64, device=dev)
lib = nv.load()._cdll
lib.st_dev_chain_trace.argtypes = [ctypes.c_void_p]
assert lib.st_dev_chain_trace(trace.data_ptr()) == 0
run(); torch.cuda.synchronize()
lib.st_dev_chain_trace(None)
t = trace.view(nwg, 32).cpu().double()[:, :18] / 100.0
t0 = t[:, 0].min()
names = ["start", "A stored+sync", "PRE block", "R store, LN0", "W1_0 | copies xhat0 out0", "epi0 + sync", "W1_1", "W2_0 | epi1", "W1_2 | copy h0", "W2_1 | epi2",
         "W1_3 | copy h1", "W2_2 | epi3", "W2_3 | copy h2", "LN1", "P0 | copies h3 xhat1", "P1 | epi P0, copy out1", "P2 | epi P1, copy P0", "tail: epi P2, copies P1 P2"]
print("workgroups %d; span %.1f us; start spread %.2f us; wg duration avg %.1f (min %.1f max %.1f)" % (
    nwg, (t[:, 17].max() - t0), (t[:, 0].max() - t0), (t[:, 17] - t[:, 0]).mean(), (t[:, 17] - t[:, 0]).min(), (t[:, 17] - t[:, 0]).max()))
for i in range(1, 18):
    dt = t[:, i] - t[:, i - 1]
    print("  %-30s %6.2f us avg  (min %5.2f  max %5.2f)   ends at %6.2f avg" % (names[i], dt.mean(), dt.min(), dt.max(), (t[:, i] - t0).mean()))

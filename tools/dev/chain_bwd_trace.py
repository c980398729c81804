"""Dev: per-workgroup phase timeline of the pipelined BACKWARD row chain (csrc/st_rowchain_pipe_bwd.cuh; needs a library built with
ST_DEV_TRACE=1).  usage: ST_HIP_LIB=tools/dev/_ab/libst_trace.so python tools/dev/chain_bwd_trace.py [rows]"""
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
g0, g1 = vec(d_) + 1, vec(d_) + 1
cs = chains.ChainSet(dev)
cb = cs.add(chains.t_blocks(chains.blocks_of(wqkv)) + chains.ffn_blocks_bwd(w1, w2) + chains.t_blocks(chains.blocks_of(wo)))
cs.finalize().rebuild()
chb = cs.chain(cb)
E = lambda *s, dtype=BF16: torch.empty(*s, dtype=dtype, device=dev)
xc, xy = rnd(rows, d_), rnd(rows, d_)
dqkv, dss, Hm, O_, Or = rnd(rows, 3 * d_), rnd(rows, d_), torch.relu(rnd(rows, dff)), rnd(rows, d_), rnd(rows, d_) * 0.004
ra, rb = torch.rand(rows, device=dev) + 0.5, torch.rand(rows, device=dev) + 0.5
dsa, dH, dsb, dctx, delta = E(rows, d_), E(rows, dff), E(rows, d_), E(rows, d_), E(4 * rows, dtype=F32)
acc = [torch.zeros(d_, device=dev) for _ in range(6)]
bits = nv.relu_bits_from(Hm)
run = lambda: nv.row_chain_bwd(chb, rows, head=(3, dqkv, dss, xc, ra, g0, None, dsa, acc[0], acc[1], acc[2]),
                               ffn=(dff, bits, 1.0, dH, xy, rb, g1, dsb, acc[3], acc[4], acc[5]), tail=(O_, Or, dctx, delta))
for _ in range(3): run()
torch.cuda.synchronize()
nwg = (rows + 95) // 96 if rows > 64 * 256 else (rows + 63) // 64
trace = torch.zeros(nwg * 32, dtype=torch.int64, device=dev)
lib = nv.load()._cdll
lib.st_dev_chain_trace.argtypes = [ctypes.c_void_p]
assert lib.st_dev_chain_trace(trace.data_ptr()) == 0
run(); torch.cuda.synchronize()
lib.st_dev_chain_trace(None)
t = trace.view(nwg, 32).cpu().double() / 100.0
order = [0, 1, 16, 17, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 18, 19, 12, 13, 14]
names = {0: "start (ring + tile requests issued)", 1: "HEAD: 3 dP blocks (store, sync, block, sync)", 16: "LN a: pass 1 + row-sum exchange", 17: "LN a: dx stored + sync",
         2: "LN a: column pass + atomics + sync", 3: "B1_0 | copy ds_a", 4: "mask epi 0 + sync", 5: "B1_1 | copy dH0", 6: "B2_0 | mask epi 1", 7: "B1_2 | copy dH1",
         8: "B2_1 | mask epi 2", 9: "B1_3 | copy dH2", 10: "B2_2 | mask epi 3", 11: "B2_3 | copy dH3, xhat_b in + sync", 18: "LN b: pass 1 + exchange",
         19: "LN b: dx stored + sync", 12: "LN b: column pass + atomics + sync", 13: "TAIL block | copy ds_b; O/Ores in + sync", 14: "dctx epilogue, copy, delta"}
t0 = t[:, 0].min()
print("workgroups %d; span %.1f us; start spread %.2f us; wg duration avg %.1f (min %.1f max %.1f)" % (
    nwg, t[:, 14].max() - t0, t[:, 0].max() - t0, (t[:, 14] - t[:, 0]).mean(), (t[:, 14] - t[:, 0]).min(), (t[:, 14] - t[:, 0]).max()))
for a, b in zip(order[:-1], order[1:]):
    dt = t[:, b] - t[:, a]
    print("  %-48s %6.2f us avg  (min %5.2f  max %5.2f)   ends at %6.2f avg" % (names[b], dt.mean(), dt.min(), dt.max(), (t[:, b] - t0).mean()))
dur = (t[:, 14] - t[:, 0])
end = t[:, 14] - t0
print("by XCD (blockIdx % 8): mean duration / latest end:", ["%.1f / %.1f" % (float(dur[x::8].mean()), float(end[x::8].max())) for x in range(8)])
order_ = torch.argsort(end, descending=True)[:10].tolist()
print("latest ten workgroups (index, start, end):", [(i, round(float(t[i, 0] - t0), 1), round(float(end[i]), 1)) for i in order_])
order_ = torch.argsort(end)[:10].tolist()
print("earliest ten workgroups (index, start, end):", [(i, round(float(t[i, 0] - t0), 1), round(float(end[i]), 1)) for i in order_])
q = torch.tensor([float(end[i * 25:(i + 1) * 25].mean()) for i in range(10)])
print("mean end by blockIdx decile:", [round(float(v), 1) for v in q])

"""Dev: per-stage accuracy of the forward row chain against fp64 (each stage's reference is computed from the KERNEL's own upstream
bf16 tensors, so one stage's error does not leak into the next).  Run with and without ST_CHAIN_PIPE=0."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
import torch
from st_amd import native as nv, chains
dev, BF16, F32, F64 = "cuda", torch.bfloat16, torch.float32, torch.float64
torch.manual_seed(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 24060
d_, dff = 256, 1024
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(BF16)
wo, wqkv, w1, w2 = rnd(d_, d_, sc=d_ ** -0.5), rnd(3 * d_, d_, sc=d_ ** -0.5), rnd(dff, d_, sc=d_ ** -0.5), rnd(d_, dff, sc=dff ** -0.5)
vec = lambda n, sc=0.1: torch.randn(n, device=dev) * sc
bo, bqkv, b1, b2, g0, be0, g1, be1 = vec(d_), vec(3 * d_), vec(dff), vec(d_), vec(d_) + 1, vec(d_), vec(d_) + 1, vec(d_)
cs = chains.ChainSet(dev)
cf = cs.add(chains.blocks_of(wo) + chains.ffn_blocks(w1, w2) + chains.blocks_of(wqkv))
cs.finalize().rebuild()
chf = cs.chain(cf)
E_ = lambda *s, dtype=BF16: torch.empty(*s, dtype=dtype, device=dev)
ctx, x = rnd(rows, d_), rnd(rows, d_) + 0.3
out0, xh0, r0, H, out1, xh1, r1, P = E_(rows, d_), E_(rows, d_), E_(rows, dtype=F32), E_(rows, dff), E_(rows, d_), E_(rows, d_), E_(rows, dtype=F32), E_(rows, 3 * d_)
ks = 0.125 * 1.4426950408889634
bits = torch.zeros(nv.chain_mask_words(rows, dff), dtype=torch.int64, device=dev)
nv.row_chain(ctx, chf, pre=(x, bo, g0, be0, out0, xh0, r0), ffn=(dff, b1, b2, g1, be1, H, out1, xh1, r1, None, None, bits), post=(3, bqkv, P), post_kscale=ks if "KS" in os.environ else 0.0)
torch.cuda.synchronize()
D = lambda t: t.to(F64)
def ln(v, g, b):
    m = v.mean(1, keepdim=True); var = ((v - m) ** 2).mean(1, keepdim=True)
    n = (v - m) / torch.sqrt(var + 1e-6)
    return n, n * D(g) + D(b), 1 / torch.sqrt(var + 1e-6)
rel = lambda a, b: float((D(a) - b).norm() / b.norm())
n0, o0, rs0 = ln(D(ctx) @ D(wo).t() + D(bo) + D(x), g0, be0)
h = torch.relu(D(out0) @ D(w1).t() + D(b1))
n1, o1, rs1 = ln(D(H) @ D(w2).t() + D(b2) + D(out0), g1, be1)
p = D(out1) @ D(wqkv).t() + D(bqkv)
if "KS" in os.environ: p[:, d_:2 * d_] *= ks
bf = lambda t: float((t.to(BF16).to(F64) - t).norm() / t.norm())      # what rounding the exact result to bf16 costs
print("rows %d  ST_CHAIN_PIPE=%s" % (rows, os.environ.get("ST_CHAIN_PIPE", "(default)")))
for name, got, ref in (("xhat0", xh0, n0), ("out0", out0, o0), ("rstd0", r0, rs0.squeeze(1)), ("H", H, h), ("xhat1", xh1, n1), ("out1", out1, o1), ("rstd1", r1, rs1.squeeze(1)), ("P", P, p)):
    print("  %-6s rel-L2 vs fp64 %.4e   (bf16 rounding of the exact value: %.4e)" % (name, rel(got, ref), bf(ref) if got.dtype == BF16 else 0.0))

import hashlib
for name, t in (("out0", out0), ("xhat0", xh0), ("rstd0", r0), ("H", H), ("out1", out1), ("xhat1", xh1), ("rstd1", r1), ("P", P), ("bits", bits)):
    print("  sha %-6s %s" % (name, hashlib.sha1(t.cpu().view(torch.uint8).numpy().tobytes()).hexdigest()[:12]))
ref_bits = nv.relu_bits_from(H)
print("  relu bits equal to bits recomputed from H:", bool(torch.equal(bits, ref_bits)), "differing words:", int((bits != ref_bits).sum()))
x_ = (bits ^ ref_bits)
idx = torch.nonzero(x_).flatten()
pc = sum(bin(int(v) & 0xffffffffffffffff).count("1") for v in x_[idx].cpu().tolist())
print("  differing bits %d in %d words; word index range %s .. %s of %d; first few:" % (pc, idx.numel(), int(idx.min()) if idx.numel() else None, int(idx.max()) if idx.numel() else None, bits.numel()),
      [(int(i), hex(int(x_[i]) & 0xffffffffffffffff)) for i in idx[:3].cpu().tolist()], [(int(i), hex(int(x_[i]) & 0xffffffffffffffff)) for i in idx[-3:].cpu().tolist()])

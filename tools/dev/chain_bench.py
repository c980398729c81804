"""Dev: native.row_chain vs the separate kernels at decoder-sized M: correctness and time."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
from st_amd import native as nv, chains, rng
BF16, F32 = torch.bfloat16, torch.float32
dev = "cuda"
torch.manual_seed(0)
def rnd(*s, sc=0.5): return (torch.randn(*s, device=dev) * sc).to(BF16)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
_flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
def cold(fn, n=20):
    tot = 0.0
    for _ in range(n):
        _flush.fill_(1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / n * 1e3
def md(a, b): return (a.float() - b.float()).abs().max().item()
d, dff = 256, 1024
wo, wq, wqkv = rnd(d, d, sc=0.06), rnd(d, d, sc=0.06), rnd(3 * d, d, sc=0.06)
w1, w2 = rnd(dff, d, sc=0.06), rnd(d, dff, sc=0.03)
vec = lambda n, sc=0.1: torch.randn(n, device=dev) * sc
bo, bq, bqkv, b1, b2 = vec(d), vec(d), vec(3 * d), vec(dff), vec(d)
g0, be0, g1, be1 = torch.rand(d, device=dev) + 0.5, vec(d), torch.rand(d, device=dev) + 0.5, vec(d)
cs = chains.ChainSet(dev)
c_f1 = cs.add(chains.blocks_of(wo) + chains.blocks_of(wq))
c_f2 = cs.add(chains.blocks_of(wo) + chains.ffn_blocks(w1, w2) + chains.blocks_of(wqkv))
c_ffn = cs.add(chains.ffn_blocks(w1, w2))
cs.finalize(); cs.rebuild()
print("rebuild %.1f us for %d blocks" % (timeit(cs.rebuild), cs.table.shape[0]))
E = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=dev)
for M in [int(a) for a in sys.argv[1:]] or (320, 1206):
    ctx, x = rnd(M, d), rnd(M, d)
    # ---- F1: wo + LN, q
    o0, xh0, r0, q0 = E(M, d), E(M, d), E(M, dt=F32), E(M, d)
    o1, xh1, r1, q1 = torch.zeros_like(o0), torch.zeros_like(xh0), torch.zeros_like(r0), torch.zeros_like(q0)
    def old1():
        nv.gemm_ln(ctx, wo, bo, x, g0, be0, o0, xh0, r0, eps=1e-6)
        nv.gemm(o0, wq, q0, bias=bq)
    ch1 = cs.chain(c_f1)
    def new1(): nv.row_chain(ctx, ch1, pre=(x, bo, g0, be0, o1, xh1, r1), post=(1, bq, q1))
    old1(); new1(); torch.cuda.synchronize()
    print("F1 M %5d  d(out) %.3g d(xhat) %.3g d(rstd) %.2g d(q) %.3g   old %.1f us new %.1f us" % (
        M, md(o0, o1), md(xh0, xh1), ((r0 - r1).abs() / r0).max().item(), md(q0, q1), timeit(old1), timeit(new1)), " cold: old %.1f new %.1f" % (cold(old1), cold(new1)))
    # ---- F2: wo + LN, FFN, qkv (with dropout)
    for drop in (False, True):
        dr1, dr2 = (rng.site(dev, 0.1), rng.site(dev, 0.1)) if drop else (None, None)
        c0, xc0, rc0, h0, y0, xy0, ry0, p0 = E(M, d), E(M, d), E(M, dt=F32), E(M, dff), E(M, d), E(M, d), E(M, dt=F32), E(M, 3 * d)
        c1, xc1, rc1, h1, y1, xy1, ry1, p1 = [torch.zeros_like(t) for t in (c0, xc0, rc0, h0, y0, xy0, ry0, p0)]
        def old2():
            nv.gemm_ln(ctx, wo, bo, x, g0, be0, c0, xc0, rc0, eps=1e-6)
            nv.gemm(c0, w1, h0, bias=b1, epi=nv.EPI_BF16_RELU, drop=dr1)
            nv.gemm_ln(h0, w2, b2, c0, g1, be1, y0, xy0, ry0, eps=1e-6, drop=dr2, drop_where=2 if dr2 else 0)
            nv.gemm(y0, wqkv, p0, bias=bqkv)
        ch2 = cs.chain(c_f2)
        def new2():
            nv.row_chain(ctx, ch2, pre=(x, bo, g0, be0, c1, xc1, rc1), ffn=(dff, b1, b2, g1, be1, h1, y1, xy1, ry1, dr1, dr2),
                         post=(3, bqkv, p1))
        old2(); new2(); torch.cuda.synchronize()
        print("F2 M %5d drop %d  d(c) %.3g d(h) %.3g d(y) %.3g d(xhat) %.3g d(qkv) %.3g  zero-pattern h %s y %s   old %.1f us new %.1f us" % (
            M, drop, md(c0, c1), md(h0, h1), md(y0, y1), md(xy0, xy1), md(p0, p1), bool(((h0 == 0) == (h1 == 0)).all()),
            bool(((y0 == 0) == (y1 == 0)).all()), timeit(old2), timeit(new2)), " cold: old %.1f new %.1f" % (cold(old2), cold(new2)))
    # ---- FFN alone
    h1, y1, xy1, ry1 = E(M, dff), E(M, d), E(M, d), E(M, dt=F32)
    ch3 = cs.chain(c_ffn)
    print("FFN M %5d  new %.1f us" % (M, timeit(lambda: nv.row_chain(x, ch3, ffn=(dff, b1, b2, g1, be1, h1, y1, xy1, ry1, None, None)))))

"""Dev: native.row_chain_bwd vs the separate kernels: time."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
from st_amd import native as nv, chains
BF16, F32 = torch.bfloat16, torch.float32
dev = "cuda"
torch.manual_seed(0)
def rnd(*s, sc=0.5): return (torch.randn(*s, device=dev) * sc).to(BF16)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
d, dff = 256, 1024
wqkv, w1, w2, wo = rnd(3 * d, d, sc=0.06), rnd(dff, d, sc=0.06), rnd(d, dff, sc=0.03), rnd(d, d, sc=0.06)
cs = chains.ChainSet(dev)
cid = cs.add(chains.t_blocks(chains.blocks_of(wqkv)) + chains.ffn_blocks_bwd(w1, w2) + chains.t_blocks(chains.blocks_of(wo)))
c1 = cs.add(chains.t_blocks(chains.blocks_of(wo)) + chains.t_blocks(chains.blocks_of(wo)))
c2 = cs.add(chains.ffn_blocks_bwd(w1, w2))
cs.finalize(); cs.rebuild()
ch, ch1 = cs.chain(cid), cs.chain(c1)
for M in [int(a) for a in sys.argv[1:]] or (1206, 24060):
    dqkv, ds_s, xa, xb, H, O, Ores = rnd(M, 3 * d), rnd(M, d), rnd(M, d), rnd(M, d), torch.relu(rnd(M, dff)), rnd(M, d), rnd(M, d, sc=0.002)
    ra, rb = torch.rand(M, device=dev) + 0.5, torch.rand(M, device=dev) + 0.5
    HB = nv.relu_bits_from(H)
    ga, gb = torch.rand(d, device=dev) + 0.5, torch.rand(d, device=dev) + 0.5
    Z = lambda *s, dt=BF16: torch.zeros(*s, dtype=dt, device=dev)
    ds_a, dH, ds_b, dctx, delta = Z(M, d), Z(M, dff), Z(M, d), Z(M, d), Z(4 * M, dt=F32)
    acc = [Z(d, dt=F32) for _ in range(6)]
    def new():
        nv.row_chain_bwd(ch, M, head=(3, dqkv, ds_s, xa, ra, ga, None, ds_a, acc[0], acc[1], acc[2]),
                         ffn=(dff, HB, 1.0, dH, xb, rb, gb, ds_b, acc[3], acc[4], acc[5]), tail=(O, Ores, dctx, delta))
    def old():
        nv.gemm_lnbwd(dqkv, wqkv, ds_s, xa, ra, ga, ds_a, acc[0], acc[1], acc[2])
        nv.gemm(ds_a, w2, dH, epi=nv.EPI_BF16_MASK, aux=H, y_cmajor=True)
        nv.gemm_lnbwd(dH, w1, ds_a, xb, rb, gb, ds_b, acc[3], acc[4], acc[5])
        nv.gemm(ds_b, wo, dctx, epi=nv.EPI_BF16_DELTA, aux=O, y_cmajor=True, delta=delta, head_dim=64, aux2=Ores)
    def new1():
        nv.row_chain_bwd(ch1, M, head=(1, ds_s, ds_b, xa, ra, ga, None, ds_a, acc[0], acc[1], acc[2]), tail=(O, Ores, dctx, delta))
    def old1():
        nv.gemm_lnbwd(ds_s, wo, ds_b, xa, ra, ga, ds_a, acc[0], acc[1], acc[2])
        nv.gemm(ds_a, wo, dctx, epi=nv.EPI_BF16_DELTA, aux=O, y_cmajor=True, delta=delta, head_dim=64, aux2=Ores)
    def new_noatomic():
        nv.row_chain_bwd(ch, M, head=(3, dqkv, ds_s, xa, ra, ga, None, ds_a, None, None, None),
                         ffn=(dff, HB, 1.0, dH, xb, rb, gb, ds_b, None, None, None), tail=(O, Ores, dctx, delta))
    def new_ffn_only():
        nv.row_chain_bwd(cs.chain(c2), M, ds_in=ds_s, ffn=(dff, HB, 1.0, dH, xb, rb, gb, ds_b, None, None, None))
    print("   no atomics %.1f us   ffn only (no atomics) %.1f us" % (timeit(new_noatomic), timeit(new_ffn_only)))
    print("M %5d  B2 (head3+ffn+tail): old %.1f us new %.1f us    B1 (head1+tail): old %.1f new %.1f" % (M, timeit(old), timeit(new), timeit(old1), timeit(new1)))

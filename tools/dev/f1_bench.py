import os, sys, math, torch
ROOT='/root/repo'
for p in (ROOT, os.path.join(ROOT, "speech-tranformer-pytorch_amd")): sys.path.insert(0, p)
from st_amd import native as nv, synthetic, chains
from st_amd.functional import Rows, attn_work
BF16, F32 = torch.bfloat16, torch.float32
dev='cuda'
def rnd(*s, dtype=BF16): return (torch.randn(*s, device=dev)*0.5).to(dtype)
_, _, in_len, tgt_len, _ = synthetic.make_batch(32, 1000, 50, 80, 4337, seed=0, t_min=500, l_min=25)
d, H = 256, 4
for name, ql, kl in (("train", tgt_len, in_len), ("decode", torch.full((32,), 10, dtype=torch.int64), in_len)):
    qr, kr = Rows.packed(ql, dev), Rows.packed(kl, dev)
    M, Mk = int(ql.sum()), int(kl.sum())
    wo, wq = rnd(d, d), rnd(d, d)
    cs = chains.ChainSet(dev); cid = cs.add(chains.blocks_of(wo) + chains.blocks_of(wq)); cs.finalize().rebuild(); ch = cs.chain(cid)
    A, R, kv = rnd(M, d), rnd(M, d), rnd(Mk, 2*d)
    bo, bq, g0, be0 = rnd(d, dtype=F32), rnd(d, dtype=F32), rnd(d, dtype=F32)+1, rnd(d, dtype=F32)
    out, xh, rs, q, O, ores, lse = rnd(M, d), rnd(M, d), torch.zeros(M, device=dev), rnd(M, d), rnd(M, d), rnd(M, d), torch.zeros(H*M, device=dev)
    work = attn_work(qr, kr, False, 64, H)[0]
    for _ in range(40):
        nv.attn_f1_fwd(A, ch, (R, bo, g0, be0, out, xh, rs), (1, bq, q), kv[:, :d], kv[:, d:], O, lse, qr.off, qr.len, kr.off, kr.len, H, int(ql.max()), 0.125, work=work, max_k=int(kl.max()), ores=ores if name == "train" else None)
    torch.cuda.synchronize()

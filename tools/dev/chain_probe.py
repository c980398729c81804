"""Dev: per-phase wall times inside st_row_chain (forward, wo+LN | FFN | qkv) from s_memrealtime stamps of wave 0 of
workgroup 0.  Builds a probed copy of the library under /tmp (the product source is not touched)."""
import ctypes, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
from st_amd import build
src = os.path.join(ROOT, "speech-tranformer-pytorch_amd", "csrc")
dst = "/tmp/csrc_probe"
shutil.rmtree(dst, ignore_errors=True); shutil.copytree(src, dst)
p = os.path.join(dst, "st_rowchain.hip")
s = open(p).read()
def rep(old, new, count=1):
    global s
    assert s.count(old) >= 1, old
    s = s.replace(old, new, count)
rep('namespace {\n', 'namespace {\n__device__ unsigned long long g_probe[64];\n#define PROBE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_probe[i] = __builtin_amdgcn_s_memrealtime(); } while (0)\n')
rep('  Ctx<MT> c;\n  c.tid = threadIdx.x;', '  PROBE(0);\n  Ctx<MT> c;\n  c.tid = threadIdx.x;')
rep('  const Drop d1 = make_drop(a.drop1), d2 = make_drop(a.drop2), off = make_drop(DropArgs{nullptr, 0u, 0, 1.f});\n  __syncthreads();\n',
    '  const Drop d1 = make_drop(a.drop1), d2 = make_drop(a.drop2), off = make_drop(DropArgs{nullptr, 0u, 0, 1.f});\n  __syncthreads();\n  PROBE(1);\n')
rep('    block_mma(c, cur, acc);\n    // xhat is staged in the A tile', '    block_mma(c, cur, acc);\n    PROBE(2);\n    // xhat is staged in the A tile')
rep('    bf16* t = cur; cur = f1; f1 = t;\n  }\n  if (FFN) {', '    bf16* t = cur; cur = f1; f1 = t;\n    PROBE(3);\n  }\n  if (FFN) {')
rep('      block_mma(c, cur, acc1);\n      epi_store<true, DROP>(c, acc1, a.b1 + ch * 256, hc, d1, ch * 256, dff,\n',
    '      block_mma(c, cur, acc1);\n      PROBE(4 + 4 * ch);\n      epi_store<true, DROP>(c, acc1, a.b1 + ch * 256, hc, d1, ch * 256, dff,\n')
rep('      __syncthreads();\n      block_mma(c, hc, acc2);\n      tile_out(c, hc, a.H + ch * 256, dff);\n',
    '      __syncthreads();\n      PROBE(5 + 4 * ch);\n      block_mma(c, hc, acc2);\n      tile_out(c, hc, a.H + ch * 256, dff);\n')
rep('      block_mma(c, hc, acc2);\n      tile_out(c, hc, a.H + ch * 256, dff);\n',
    '      block_mma(c, hc, acc2);\n      PROBE(6 + 4 * ch);\n      tile_out(c, hc, a.H + ch * 256, dff);\n      PROBE(7 + 4 * ch);\n')
rep('    if (tx == f0) { f0 = f1; f1 = tx; }', '    PROBE(20);\n    if (tx == f0) { f0 = f1; f1 = tx; }')
rep('      block_mma(c, cur, acc);\n      epi_store<false, false>(c, acc, a.bp + u * 256, st, off, 0, 0);\n      __syncthreads();\n      tile_out(c, st, a.P + u * 256, a.ldp);\n',
    '      block_mma(c, cur, acc);\n      PROBE(21 + 3 * u);\n      epi_store<false, false>(c, acc, a.bp + u * 256, st, off, 0, 0);\n      __syncthreads();\n      PROBE(22 + 3 * u);\n      tile_out(c, st, a.P + u * 256, a.ldp);\n      PROBE(23 + 3 * u);\n')
# backward kernel: stamps 32.. (HEAD: 32 entry, 33 tiles in, 34+u after block u, 38 after LN backward; FFN: 40+4ch.. ; 56 LN backward; TAIL 57..59)
rep('  Ctx<MT> c;\n  c.tid = threadIdx.x; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6); c.l = c.tid & 63; c.hi = c.l >> 5; c.r = c.l & 31;\n  c.row0 = blockIdx.x * RB; c.nvalid = min(RB, a.M - c.row0);\n  c.ws = a.wfrag + (size_t)c.wave * a.wave_frags * 64;\n#pragma unroll',
    '  PROBE(32);\n  Ctx<MT> c;\n  c.tid = threadIdx.x; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6); c.l = c.tid & 63; c.hi = c.l >> 5; c.r = c.l & 31;\n  c.row0 = blockIdx.x * RB; c.nvalid = min(RB, a.M - c.row0);\n  c.ws = a.wfrag + (size_t)c.wave * a.wave_frags * 64;\n#pragma unroll')
rep('    if (a.nb > 0) tile_load(c, a.dP, a.ldp, nxt);\n', '    if (a.nb > 0) tile_load(c, a.dP, a.ldp, nxt);\n    PROBE(33);\n')
rep('      block_mma(c, t0, acc);\n      __syncthreads();                           // every wave is past its MFMAs on this block of dP: t0 may be rewritten\n',
    '      block_mma(c, t0, acc);\n      __syncthreads();\n      PROBE(34 + u);\n')
rep('    cur = t0; fa = t1; fb = t2;\n    __syncthreads();                             // the column pass has read t1 / t2: free from here\n',
    '    cur = t0; fa = t1; fb = t2;\n    __syncthreads();\n    PROBE(38);\n')
rep('      block_mma(c, cur, acc1);                   // ds x W2[:, chunk]: the hidden gradient before the mask\n',
    '      PROBE(40 + 4 * ch);\n      block_mma(c, cur, acc1);\n      PROBE(41 + 4 * ch);\n')
rep('      __syncthreads();\n      block_mma(c, hc, acc2);                    // dH chunk x W1[chunk, :]\n      tile_out(c, hc, a.dH + ch * 256, dff);\n',
    '      __syncthreads();\n      PROBE(42 + 4 * ch);\n      block_mma(c, hc, acc2);\n      PROBE(43 + 4 * ch);\n      tile_out(c, hc, a.dH + ch * 256, dff);\n')
rep('    fa = cur; fb = tx; cur = td;\n    __syncthreads();                             // the column pass has read fa / fb\n',
    '    fa = cur; fb = tx; cur = td;\n    __syncthreads();\n    PROBE(56);\n')
rep('    block_mma(c, cur, acc);\n    tile_store(c, ro, fa);', '    PROBE(57);\n    block_mma(c, cur, acc);\n    PROBE(58);\n    tile_store(c, ro, fa);')
rep('    for (int i = c.tid; i < 4 * RB; i += 512) {', '    PROBE(59);\n    for (int i = c.tid; i < 4 * RB; i += 512) {')
rep('extern "C" int st_wfrag_depth(void) { return DEPTH; }'
, 'extern "C" int st_wfrag_depth(void) { return DEPTH; }\nextern "C" int st_chain_probe(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_probe), sizeof(g_probe)); }')
open(p, "w").write(s)
lib = "/tmp/libst_probe.so"
subprocess.run(["/opt/rocm/bin/hipcc"] + build.FLAGS + [os.path.join(dst, f) for f in build.SOURCES] + ["-o", lib], check=True, cwd=dst,
               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
build.LIB = lib
import torch
from st_amd import native as nv, chains
BF16, F32 = torch.bfloat16, torch.float32
dev = "cuda"
rnd = lambda *sh, sc=0.1: (torch.randn(*sh, device=dev) * sc).to(BF16)
vec = lambda n: torch.randn(n, device=dev) * 0.1
d, dff = 256, 1024
wo, wqkv, w1, w2 = rnd(d, d), rnd(3 * d, d), rnd(dff, d), rnd(d, dff)
bo, bqkv, b1, b2, g0, be0, g1, be1 = vec(d), vec(3 * d), vec(dff), vec(d), vec(d) + 1, vec(d), vec(d) + 1, vec(d)
cs = chains.ChainSet(dev)
cf = cs.add(chains.blocks_of(wo) + chains.ffn_blocks(w1, w2) + chains.blocks_of(wqkv))
cs.finalize().rebuild()
ch = cs.chain(cf)
E = lambda *sh, dt=BF16: torch.empty(*sh, dtype=dt, device=dev)
names = {1: "prologue (ring issue, touch, A/R tiles, barrier)", 2: "PRE mma", 3: "PRE LN epilogue + copies out", 20: "FFN LN epilogue + copies out"}
for c4 in range(4):
    names.update({4 + 4 * c4: "chunk %d W1 mma" % c4, 5 + 4 * c4: "chunk %d relu epilogue + barrier" % c4, 6 + 4 * c4: "chunk %d W2 mma" % c4, 7 + 4 * c4: "chunk %d H copy out" % c4})
for u in range(3):
    names.update({21 + 3 * u: "POST %d mma" % u, 22 + 3 * u: "POST %d epilogue + barrier" % u, 23 + 3 * u: "POST %d copy out" % u})
order = [0, 1, 2, 3] + list(range(4, 20)) + [20] + list(range(21, 30))
for M in [int(a) for a in sys.argv[1:]] or (1206, 24060):
    ctx, x = rnd(M, d, sc=0.5), rnd(M, d, sc=0.5)
    outs = (E(M, d), E(M, d), E(M, dt=F32), E(M, dff), E(M, d), E(M, d), E(M, dt=F32), E(M, 3 * d))
    c_, xc, rc, h_, y_, xy, ry, p_ = outs
    run = lambda: nv.row_chain(ctx, ch, pre=(x, bo, g0, be0, c_, xc, rc), ffn=(dff, b1, b2, g1, be1, h_, y_, xy, ry, None, None), post=(3, bqkv, p_))
    acc = None
    for it in range(12):
        run(); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 64)()
        nv.load()._cdll.st_chain_probe(buf)
        t = [buf[i] for i in range(64)]
        if it >= 2:
            dl = [(t[order[k]] - t[order[k - 1]]) * 10 for k in range(1, len(order))]
            acc = dl if acc is None else [a + b for a, b in zip(acc, dl)]
    print("M = %d: phases of workgroup 0 / wave 0, ns (mean of 10)" % M)
    tot = 0
    for k in range(1, len(order)):
        v = acc[k - 1] / 10; tot += v
        print("  %-50s %8.0f" % (names[order[k]], v))
    print("  %-50s %8.0f" % ("total (first stamp to last)", tot))

# ---- backward chain: qkv^T + LN backward | FFN^T + LN backward | wo^T + delta
cb = chains.ChainSet(dev)
ib = cb.add(chains.t_blocks(chains.blocks_of(wqkv)) + chains.ffn_blocks_bwd(w1, w2) + chains.t_blocks(chains.blocks_of(wo)))
cb.finalize().rebuild()
chb = cb.chain(ib)
bnames = {33: "HEAD: G / xhat tiles in, first dP block requested", 34: "HEAD block 0 (store, barrier, mma, barrier)", 35: "HEAD block 1", 36: "HEAD block 2",
          38: "HEAD LayerNorm backward + column sums", 56: "FFN xhat in + LayerNorm backward + column sums", 57: "TAIL O / Ores requested", 58: "TAIL mma", 59: "TAIL delta epilogue + copy out"}
for c4 in range(4):
    bnames.update({40 + 4 * c4: "chunk %d H mask values requested" % c4 if c4 == 0 else "chunk %d dH copy out + H mask request" % c4, 41 + 4 * c4: "chunk %d W2^T mma" % c4,
                   42 + 4 * c4: "chunk %d mask epilogue + barrier" % c4, 43 + 4 * c4: "chunk %d W1^T mma" % c4})
border = [32, 33, 34, 35, 36, 38] + list(range(40, 56)) + [56, 57, 58, 59]
for M in [int(a) for a in sys.argv[1:]] or (1206, 24060):
    dqkv, dss, Hm, O_, Or = rnd(M, 3 * d), rnd(M, d), torch.relu(rnd(M, dff)), rnd(M, d), rnd(M, d, sc=0.004)
    xc, xy = rnd(M, d, sc=1.0), rnd(M, d, sc=1.0)
    ra, rb = torch.rand(M, device=dev) + 0.5, torch.rand(M, device=dev) + 0.5
    dsa, dH, dsb, dctx, delta = E(M, d), E(M, dff), E(M, d), E(M, d), E(4 * M, dt=F32)
    accs = [torch.zeros(d, device=dev) for _ in range(6)]
    run = lambda: nv.row_chain_bwd(chb, M, head=(3, dqkv, dss, xc, ra, g0, None, dsa, accs[0], accs[1], accs[2]),
                                   ffn=(dff, nv.relu_bits_from(Hm), 1.0, dH, xy, rb, g1, dsb, accs[3], accs[4], accs[5]), tail=(O_, Or, dctx, delta))
    acc = None
    for it in range(12):
        run(); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 64)()
        nv.load()._cdll.st_chain_probe(buf)
        t = [buf[i] for i in range(64)]
        if it >= 2:
            dl = [(t[border[k]] - t[border[k - 1]]) * 10 for k in range(1, len(border))]
            acc = dl if acc is None else [a + b for a, b in zip(acc, dl)]
    print("BACKWARD M = %d: phases of workgroup 0 / wave 0, ns (mean of 10)" % M)
    tot = 0
    for k in range(1, len(border)):
        v = acc[k - 1] / 10; tot += v
        print("  %-55s %8.0f" % (bnames[border[k]], v))
    print("  %-55s %8.0f" % ("total (first stamp to last)", tot))

"""Dev: wall time per streamed tile inside the attention kernels (s_memrealtime stamps of thread 0 of workgroup 0 = the
heaviest work item) - forward, dQ body, dK/dV body at the encoder shape.  Builds a probed copy of the library under /tmp."""
import ctypes, math, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "speech-tranformer-pytorch_amd"))
from st_amd import build
src = os.path.join(ROOT, "speech-tranformer-pytorch_amd", "csrc")
dst = "/tmp/csrc_probe_attn"
shutil.rmtree(dst, ignore_errors=True); shutil.copytree(src, dst)
p = os.path.join(dst, "st_attn.hip")
s = open(p).read()
def rep(old, new):
    global s
    assert s.count(old) >= 1, old
    s = s.replace(old, new, 1)
rep('namespace {\n', 'namespace {\n__device__ unsigned long long g_probe[128];\n#define PROBE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (i) < 128) g_probe[i] = __builtin_amdgcn_s_memrealtime(); } while (0)\n')
rep('''  if (ntiles <= 0) return;   // (workgroup-uniform) nothing visible: the accumulators stay zero
  load(0, 0);
  if (ntiles > 1) load(1, 1);
  store(0);
  __syncthreads();
  int it = 0;
  for (; it + 3 < ntiles; it += 2) {
    load(0, it + 2);
    compute(0, it);
    store(1);
    __syncthreads();
    load(1, it + 3);
    compute(1, it + 1);
    store(0);
    __syncthreads();
  }''', '''  if (ntiles <= 0) return;
  PROBE(0);
  load(0, 0);
  if (ntiles > 1) load(1, 1);
  store(0);
  __syncthreads();
  PROBE(1);
  int it = 0;
  for (; it + 3 < ntiles; it += 2) {
    load(0, it + 2);
    compute(0, it);
    PROBE(2 + 2 * it);
    store(1);
    __syncthreads();
    PROBE(3 + 2 * it);
    load(1, it + 3);
    compute(1, it + 1);
    PROBE(4 + 2 * it);
    store(0);
    __syncthreads();
    PROBE(5 + 2 * it);
  }
  PROBE(120);''')
rep('  __syncthreads();   // the epilogue reuses the tile buffers\n}', '  __syncthreads();   // the epilogue reuses the tile buffers\n  PROBE(121);\n}')
s += '\nextern "C" int st_attn_probe(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_probe), sizeof(g_probe)); }\n'
# kernel entry / exit stamps
rep('  int b, h, tile;\n  decode_item(a, blockIdx.x, b, h, tile);\n  const int lq = a.q_len[b], lk = a.k_len[b];\n  const int q0 = tile * QROWS;\n  if (q0 >= lq) return;\n  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hi = l >> 5;\n  const int qw = KS > 1 ? (wave & 1) : wave, kp = KS > 1 ? (wave >> 1) : 0;   // query block, key half',
    '  PROBE(126);\n  int b, h, tile;\n  decode_item(a, blockIdx.x, b, h, tile);\n  const int lq = a.q_len[b], lk = a.k_len[b];\n  const int q0 = tile * QROWS;\n  if (q0 >= lq) return;\n  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hi = l >> 5;\n  const int qw = KS > 1 ? (wave & 1) : wave, kp = KS > 1 ? (wave >> 1) : 0;   // query block, key half')
open(p, "w").write(s)
lib = "/tmp/libst_probe_attn.so"
subprocess.run(["/opt/rocm/bin/hipcc"] + build.FLAGS + [os.path.join(dst, f) for f in build.SOURCES] + ["-o", lib], check=True, cwd=dst,
               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
build.LIB = lib
import torch
from st_amd import native as nv, synthetic
from st_amd.functional import Rows, attn_work
import bench
BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32
dev = "cuda"
_, _, in_len, tgt_len, _ = synthetic.make_batch(bench.BATCH, bench.T_MAX, bench.L_MAX, bench.C2["feature_dim"], bench.C2["vocab_size"])
M, d, H = int(in_len.sum()), 256, 4
rows = Rows.packed(in_len, dev)
wq, wk = attn_work(rows, rows, False)
qkv = (torch.randn(M, 3 * d, device=dev) * 0.5).to(BF16)
Q, K, V = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
O, dO = torch.empty(M, d, dtype=BF16, device=dev), (torch.randn(M, d, device=dev) * 0.5).to(BF16)
lse, delta = torch.empty(H * M, dtype=F32, device=dev), torch.empty(H * M, dtype=F32, device=dev)
dQ, dK, dV = (torch.empty(M, d, dtype=BF16, device=dev) for _ in range(3))
scale, mx = 1 / math.sqrt(64), int(in_len.max())
runs = {"forward": lambda: nv.attn_fwd(Q, K, V, O, lse, rows.off, rows.len, rows.off, rows.len, H, mx, False, scale, work=wq, max_k=mx),
        "dQ body": lambda: nv.attn_bwd(Q, K, V, O, dO, lse, delta, dQ, dK, dV, rows.off, rows.len, rows.off, rows.len, H, mx, mx, False, scale, parts=1, work_q=wq, work_k=wk),
        "dK/dV body": lambda: nv.attn_bwd(Q, K, V, O, dO, lse, delta, dQ, dK, dV, rows.off, rows.len, rows.off, rows.len, H, mx, mx, False, scale, parts=2, work_q=wq, work_k=wk)}
runs["forward"](); runs["dQ body"]()
for name, fn in runs.items():
    acc = None
    for it in range(8):
        fn(); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 128)()
        nv.load()._cdll.st_attn_probe(buf)
        t = [buf[i] * 10 for i in range(128)]
        if it >= 2:
            n_it = 14          # T = 1000: 16 tiles, 7 double iterations
            row = [t[0] - t[126], t[1] - t[0]] + [t[2 + k] - (t[1 + k] if k else t[1]) for k in range(2 * n_it)] + [t[121] - t[120]]
            acc = row if acc is None else [a + b for a, b in zip(acc, row)]
    acc = [a / 6 for a in acc]
    comp, sync = acc[2::2][:n_it], acc[3::2][:n_it]
    print("%-11s entry->stream %5.0f ns | first tiles in LDS %5.0f | per tile: compute %s ... mean %5.0f | store+barrier %s ... mean %5.0f | tail %5.0f" % (
        name, acc[0], acc[1], " ".join("%4.0f" % c for c in comp[:4]), sum(comp) / len(comp), " ".join("%4.0f" % c for c in sync[:4]), sum(sync) / len(sync), acc[-1]))

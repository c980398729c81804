"""Dev: patch csrc/st_rowchain.hip with the TR() phase stamps tools/dev/chain_trace.py reads (wall_clock64 per phase, thread
0 of every workgroup).  Apply, rebuild, run chain_trace.py on the GPU, then `git checkout` the file: the stamps do not ship."""
import os, sys
p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "speech-tranformer-pytorch_amd", "csrc", "st_rowchain.hip")
s = open(p).read()
def rep(old, new, count=1):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, count)
rep('''template <bool PRE, bool FFN, bool POST, bool DROP, int MT>
__global__ __launch_bounds__(512, 1) void row_chain_kernel(ChainArgs a) {''', '''template <bool PRE, bool FFN, bool POST, bool DROP, int MT>
__global__ __launch_bounds__(512, 1) void row_chain_kernel(ChainArgs a) {
  TR(0);''')
rep('''// v = acc + bias + res; LayerNorm over the 256 columns''', '''__device__ long long* g_trace = nullptr;
__device__ int g_ln = 0;
#define TR(i) do { if (g_trace && threadIdx.x == 0) g_trace[blockIdx.x * 32 + (i)] = wall_clock64(); } while (0)
// v = acc + bias + res; LayerNorm over the 256 columns''')
# inside epi_ln: stamps 20.. (first call) / 24.. (second call)
rep('''    if (c.hi == 0) red[0][(c.wave * MT + mt) * 32 + c.r] = sum[mt];
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[0][(w * MT + mt) * 32 + c.r];
    mean[mt] = s * (1.f / DM);''', '''    if (c.hi == 0) red[0][(c.wave * MT + mt) * 32 + c.r] = sum[mt];
  }
  __syncthreads();
  TR(20 + lnbase);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[0][(w * MT + mt) * 32 + c.r];
    mean[mt] = s * (1.f / DM);''')
rep('''    if (c.hi == 0) red[1][(c.wave * MT + mt) * 32 + c.r] = sq[mt];
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[1][(w * MT + mt) * 32 + c.r];
    rstd[mt] = rsqrtf''', '''    if (c.hi == 0) red[1][(c.wave * MT + mt) * 32 + c.r] = sq[mt];
  }
  __syncthreads();
  TR(21 + lnbase);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[1][(w * MT + mt) * 32 + c.r];
    rstd[mt] = rsqrtf''')
rep('''  __syncthreads();
  if (g_xhat) tile_out(c, t_xhat, g_xhat, DM);
  tile_out(c, t_out, g_out, DM);
}''', '''  __syncthreads();
  TR(22 + lnbase);
  if (g_xhat) tile_out(c, t_xhat, g_xhat, DM);
  tile_out(c, t_out, g_out, DM);
}''')
rep('''                                       float (*red)[NW * 32 * MT], bf16* g_out, bf16* g_xhat, float* g_rstd) {
  const int j0 = c.wave * 32;
  float sum[MT], sq[MT], mean[MT], rstd[MT];''', '''                                       float (*red)[NW * 32 * MT], bf16* g_out, bf16* g_xhat, float* g_rstd, int lnbase = 0) {
  const int j0 = c.wave * 32;
  float sum[MT], sq[MT], mean[MT], rstd[MT];''')
rep("epi_ln<DROP>(c, acc2, a.b2, cur, a.g1, a.be1, a.eps, d2, tx, cur, red, a.out1, a.xhat1, a.rstd1);",
    "epi_ln<DROP>(c, acc2, a.b2, cur, a.g1, a.be1, a.eps, d2, tx, cur, red, a.out1, a.xhat1, a.rstd1, 4);")
rep('''  {   // A and the residual are requested together (one global round trip), then stored
    TileRegs<MT> ra, rr;
    tile_load(c, a.A, a.lda, ra);
    if (PRE) tile_load(c, a.R, a.ldr, rr);
    tile_store(c, ra, cur);''', '''  {   // A and the residual are requested together (one global round trip), then stored
    TileRegs<MT> ra, rr;
    tile_load(c, a.A, a.lda, ra);
    if (PRE) tile_load(c, a.R, a.ldr, rr);
    TR(28);
    tile_store(c, ra, cur);
    TR(29);''')
rep('''  const Drop d1 = make_drop(a.drop1), d2 = make_drop(a.drop2), off = make_drop(DropArgs{nullptr, 0u, 0, 1.f});
  __syncthreads();

  if (PRE) {
    f32x16 acc[MT];
    zero_acc(acc);
    block_mma(c, cur, acc);
''', '''  const Drop d1 = make_drop(a.drop1), d2 = make_drop(a.drop2), off = make_drop(DropArgs{nullptr, 0u, 0, 1.f});
  __syncthreads();
  TR(1);

  if (PRE) {
    f32x16 acc[MT];
    zero_acc(acc);
    block_mma(c, cur, acc);
    TR(2);
''')
rep('''    bf16* t = cur; cur = f1; f1 = t;
  }
  if (FFN) {
    const int dff = a.nc * 256;''', '''    bf16* t = cur; cur = f1; f1 = t;
    TR(3);
  }
  if (FFN) {
    const int dff = a.nc * 256;''')
rep('''      block_mma(c, cur, acc1);
      epi_store<true, DROP>(c, acc1, a.b1 + ch * 256, hc, d1, ch * 256, dff,
                            a.relu_bits ? a.relu_bits + ((size_t)(blockIdx.x * a.nc + ch) * NW + c.wave) * 64 : nullptr);
      __syncthreads();
      block_mma(c, hc, acc2);
      tile_out(c, hc, a.H + ch * 256, dff);
    }''', '''      block_mma(c, cur, acc1);
      TR(4 + 3 * ch);
      epi_store<true, DROP>(c, acc1, a.b1 + ch * 256, hc, d1, ch * 256, dff,
                            a.relu_bits ? a.relu_bits + ((size_t)(blockIdx.x * a.nc + ch) * NW + c.wave) * 64 : nullptr);
      __syncthreads();
      TR(5 + 3 * ch);
      block_mma(c, hc, acc2);
      tile_out(c, hc, a.H + ch * 256, dff);
      TR(6 + 3 * ch);
    }''')
rep('''    if (tx == f0) { f0 = f1; f1 = tx; }      // f0 = the tile free right now, f1 = xhat (still being copied out)
  }
  if (POST) {''', '''    if (tx == f0) { f0 = f1; f1 = tx; }      // f0 = the tile free right now, f1 = xhat (still being copied out)
    TR(16);
  }
  if (POST) {''')
rep('''      tile_out(c, st, a.P + u * 256, a.ldp);
    }
  }
  {
    int tsum = 0;''', '''      tile_out(c, st, a.P + u * 256, a.ldp);
      TR(17 + u);
    }
  }
  {
    int tsum = 0;''')
rep('''extern "C" int st_wfrag_depth(void) { return DEPTH; }''', '''extern "C" int st_wfrag_depth(void) { return DEPTH; }
extern "C" int st_dev_chain_trace(long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &p, sizeof(p)); }''')
open(p, "w").write(s)
print("patched", p)

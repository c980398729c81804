// st_gemm_lnbwd: a data-gradient GEMM whose output is immediately the input of a LayerNorm backward - both in
// one launch (the backward analogue of st_gemm_ln):
//
//   dy = bf16( dY W  (+ aux) )                       dY [M, Kc] natural, W [Kc, N] contraction-major, N = d_model
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma          (LayerNorm backward, per row)
//   dgamma += sum_rows dy * xhat,   dbeta += sum_rows dy,   dbias += sum_rows dx
//
// i.e. st_gemm(ST_EPI_BF16 / ST_EPI_BF16_ADD) followed by st_ln_bwd, without `dy` ever going to HBM (a 2 x M x N
// byte round trip) and without the second launch.  Where it applies: the gradient a sublayer hands to the
// sublayer before it - dx = dh W1 + ds out of the feed-forward (transformer/SubLayers.py:25-27 backward) feeding
// the LayerNorm of the attention sublayer (transformer/Attention.py:94 backward), and so on down the stack.
// `dy` is rounded to bf16 once, as acc + aux (the two-kernel path rounds the GEMM result and then the sum: its dy
// differs from this one by that double rounding, ~0.3 bf16 ulp rms - 3e-3 of the step's gradient norm, measured).
//
// Structure = st_gemm_ln.hip (256 threads, 32 * (4 / (N / 128)) x N tile, wave block 32 rows x 128 columns with the
// accumulator transposed: row statistics are lane-local) with the weight tile contraction-major (read with
// ds_read_b64_tr_b16 from the XOR-swizzled image of st_gemm_sym.hip).  Epilogue, per wave, through two private
// [32][128] LDS patches: xhat and aux are staged coalesced; pass 1 (row per lane) forms dy and the two row sums;
// the column sums of dy / dy*xhat are taken in the coalesced layout (8 columns per thread); pass 2 forms dx, which
// leaves as 256-byte row segments while its column sum is taken.
#include "st_common.cuh"
#include <cstdlib>

#ifndef ST_GEMM_SC1
#define ST_GEMM_SC1 0      // development: bf16 output rows write-through (store16_wt)
#endif
namespace {

constexpr int BK = 32;
constexpr int NS = BK + 8;   // X tile row stride (80 B)

__device__ __attribute__((aligned(16))) float g_zero_lb[4];

struct LnBwdArgs {
  const bf16* X; int ldx;       // dY  [M, Kc] natural
  const bf16* W; int ldw;       // W   [Kc, N] contraction-major (row c holds the N outputs)
  int M, Kc;
  const bf16* aux; int ldaux;   // optional addend [M, N]
  const bf16* xhat;             // [M, N], ld = N
  const float* rstd;            // [M]
  const float* gamma;           // [N]
  bf16* out; int ldo;           // dx [M, N]
  float* dgamma; float* dbeta; float* dbias;   // [N] each, atomically accumulated (nullable)
  DropArgs drop;                // the forward dropped the LayerNorm OUTPUT (SubLayers.py:27): dy <- dy * keep / (1 - p)
};

// TALL (N = 256, encoder-sized M, long contraction): 8 waves, 4 x 2, BM = 128 - as in st_gemm_ln.hip.
// MB = 2 (N = 512, encoder-sized M, long contraction): two 32-row blocks per wave = 128-row tiles, as in st_gemm_ln.hip (the
// 32 x 512 weight k-tile then feeds 16 MFMAs per wave instead of 8).
template <int N, bool TALL = false, int MB = 1> struct Geo {
  static constexpr int WN = N / 128, WM = (TALL ? 8 : 4) / WN, BM = 32 * WM * MB, NT = 64 * WM * WN;
  static constexpr int XE = BM * NS, YE = BK * N, BUF = XE + YE;
  static constexpr int SMEM_E = 2 * BUF > WM * WN * 2 * 4096 ? 2 * BUF : WM * WN * 2 * 4096;   // operand buffers | 2 patches per wave
};

__device__ __forceinline__ int cm_col(int crow, int col) { return col ^ ((crow & 3) << 5); }

// X tile: BM x 32, natural
template <int ROWS, int NT = 256>
struct StageX {
  static constexpr int CHUNKS = ROWS * 4, CH = (CHUNKS + NT - 1) / NT;
  bf16x8 v[CH];
  static __device__ __forceinline__ int chunk_id(int p) { return (threadIdx.x + p * NT) % CHUNKS; }
  static __device__ __forceinline__ void offsets(uint32_t (&off)[CH], int ld, int row0, int nrows) {
#pragma unroll
    for (int p = 0; p < CH; ++p) {
      const int id = chunk_id(p);
      off[p] = ((uint32_t)min(row0 + (id >> 2), nrows - 1) * (uint32_t)ld + (id & 3) * 8) * 2u;
    }
  }
  __device__ __forceinline__ void load(const uint32_t (&off)[CH], const bf16* __restrict__ base, int c0, int c_end) {
    const char* kb = reinterpret_cast<const char*>(base) + (size_t)c0 * 2;
    if (c0 + BK <= c_end) {
#pragma unroll
      for (int p = 0; p < CH; ++p) v[p] = *reinterpret_cast<const bf16x8*>(kb + off[p]);
    } else {
#pragma unroll
      for (int p = 0; p < CH; ++p) {
        const bool ok = c0 + (chunk_id(p) & 3) * 8 < c_end;
        v[p] = *reinterpret_cast<const bf16x8*>(ok ? kb + off[p] : reinterpret_cast<const char*>(g_zero_lb));
      }
    }
  }
  __device__ __forceinline__ void store(bf16* tile) const {
#pragma unroll
    for (int p = 0; p < CH; ++p) {
      const int id = chunk_id(p);
      *reinterpret_cast<bf16x8*>(tile + (id >> 2) * NS + (id & 3) * 8) = v[p];
    }
  }
};

// W tile: 32 c-rows x N, contraction-major, swizzled
template <int N, int NT = 256>
struct StageW {
  static constexpr int CPR = N / 8, CH = BK * CPR / NT;
  bf16x8 v[CH];
  static __device__ __forceinline__ void offsets(uint32_t (&off)[CH], int ld) {
#pragma unroll
    for (int p = 0; p < CH; ++p) {
      const int id = threadIdx.x + p * NT;
      off[p] = ((uint32_t)(id / CPR) * (uint32_t)ld + (id % CPR) * 8) * 2u;
    }
  }
  __device__ __forceinline__ void load(const uint32_t (&off)[CH], const bf16* __restrict__ base, int ld, int c0,
                                       int c_end) {
    const char* kb = reinterpret_cast<const char*>(base) + (size_t)c0 * ld * 2;
    if (c0 + BK <= c_end) {
#pragma unroll
      for (int p = 0; p < CH; ++p) v[p] = *reinterpret_cast<const bf16x8*>(kb + off[p]);
    } else {
#pragma unroll
      for (int p = 0; p < CH; ++p) {
        const bool ok = c0 + (int)((threadIdx.x + p * NT) / CPR) < c_end;
        v[p] = *reinterpret_cast<const bf16x8*>(ok ? kb + off[p] : reinterpret_cast<const char*>(g_zero_lb));
      }
    }
  }
  __device__ __forceinline__ void store(bf16* tile) const {
#pragma unroll
    for (int p = 0; p < CH; ++p) {
      const int id = threadIdx.x + p * NT, crow = id / CPR;
      *reinterpret_cast<bf16x8*>(tile + crow * N + cm_col(crow, (id % CPR) * 8)) = v[p];
    }
  }
};

// transposing fragment of the swizzled W tile: for output column col0 + (lane & 31), c-rows kk*16 + hi*8 + {0..3, 4..7}
template <int N>
__device__ __forceinline__ bf16x8 frag_w(const bf16* tile, int col0, int kk) {
  const int l = threadIdx.x & 63, hi = l >> 5, t = l & 15;
  const int ca = kk * 16 + hi * 8 + (t >> 2);
  const int col = col0 + ((l >> 4) & 1) * 16 + 4 * (t & 3);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ST_LDS bf16x4*)(tile + ca * N + cm_col(ca, col)));
  const bf16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ST_LDS bf16x4*)(tile + (ca + 4) * N + cm_col(ca + 4, col)));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = up[0]; f[5] = up[1]; f[6] = up[2]; f[7] = up[3];
  return f;
}

// wave-private [32][128] patch, 16-byte chunks XOR-swizzled by the row: element (row, col)
__device__ __forceinline__ int patch_at(int row, int col) { return row * 128 + ((((col >> 3) ^ (row & 15)) << 3) | (col & 7)); }

template <int N, bool DROP, bool TALL = false, int MB = 1>
__global__ __launch_bounds__((TALL ? 512 : 256), (TALL ? 1 : 2)) void gemm_lnbwd_kernel(LnBwdArgs a) {
  static_assert(MB == 1 || TALL, "two row blocks per wave: 8-wave workgroups only");
  using G = Geo<N, TALL, MB>;
  __shared__ __attribute__((aligned(16))) bf16 smem[G::SMEM_E];
  __shared__ float red_all[MB * 2 * G::WM * G::WN * 32];     // one exchange area per row block (one barrier each)
  __shared__ float csum[3][G::WM][N];
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
  const int wm = wave / G::WN, wn = wave % G::WN;
  const int i0 = blockIdx.x * G::BM, i_base0 = i0 + wm * 32 * MB;      // the wave's rows: i_base0 + mb * 32 + (lane & 31)
  auto xs = [&](int buf) { return smem + buf * G::BUF; };
  auto ys = [&](int buf) { return smem + buf * G::BUF + G::XE; };

  f32x16 accs[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int b = 0; b < 4; ++b) accs[mb][b] = zero16();

  using SX = StageX<G::BM, G::NT>;
  using SW = StageW<N, G::NT>;
  uint32_t offx[SX::CH], offw[SW::CH];
  SX::offsets(offx, a.ldx, i0, a.M);
  SW::offsets(offw, a.ldw);
  SX ax, bx;
  SW aw, bw;
  auto loadA = [&](int kt) { ax.load(offx, a.X, kt * BK, a.Kc); aw.load(offw, a.W, a.ldw, kt * BK, a.Kc); };
  auto loadB = [&](int kt) { bx.load(offx, a.X, kt * BK, a.Kc); bw.load(offw, a.W, a.ldw, kt * BK, a.Kc); };
  auto storeA = [&]() { ax.store(xs(0)); aw.store(ys(0)); };
  auto storeB = [&]() { bx.store(xs(1)); bw.store(ys(1)); };
  auto compute = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 xf[MB], wf[4];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) xf[mb] = frag_nat(xs(buf), NS, (wm * MB + mb) * 32 + r, kk * 16 + hi * 8);
#pragma unroll
      for (int b = 0; b < 4; ++b) wf[b] = frag_w<N>(ys(buf), wn * 128 + b * 32, kk);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int b = 0; b < 4; ++b) accs[mb][b] = mfma32(wf[b], xf[mb], accs[mb][b]);
    }
  };
  const int nk = (a.Kc + BK - 1) / BK;
  loadA(0);
  if (nk > 1) loadB(1);
  storeA();
  __syncthreads();
  int kt = 0;
  for (; kt + 3 < nk; kt += 2) {
    loadA(kt + 2);
    compute(0);
    storeB();
    __syncthreads();
    loadB(kt + 3);
    compute(1);
    storeA();
    __syncthreads();
  }
  if (kt + 2 < nk) loadA(kt + 2);
  // coalesced chunks of this wave's [32][128] block (chunk id = p*64 + lane -> row id >> 4, 16-byte chunk id & 15):
  // the addend now (its latency hides under the tail), xhat right after the last MFMAs
  bf16x8 auxv[8];
  auto load_aux = [&](int mb) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int id = p * 64 + l, rr = min(i_base0 + mb * 32 + (id >> 4), a.M - 1), c = id & 15;
      auxv[p] = *reinterpret_cast<const bf16x8*>(a.aux + (size_t)rr * a.ldaux + wn * 128 + c * 8);
    }
  };
  if (a.aux) load_aux(0);
  compute(0);
  if (kt + 1 < nk) {
    storeB();
    __syncthreads();
    compute(1);
    if (kt + 2 < nk) {
      storeA();
      __syncthreads();
      compute(0);
    }
  }
  bf16x8 xhv[8];
  auto load_xhat = [&](int mb) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int id = p * 64 + l, rr = min(i_base0 + mb * 32 + (id >> 4), a.M - 1), c = id & 15;
      xhv[p] = *reinterpret_cast<const bf16x8*>(a.xhat + (size_t)rr * N + wn * 128 + c * 8);
    }
  };
  load_xhat(0);
  __syncthreads();   // the patches reuse the operand buffers

  bf16* p1 = smem + wave * 8192;        // xhat
  bf16* p2 = p1 + 4096;                 // aux -> dy -> dx
  const float* gm = a.gamma + wn * 128;
  const Drop dr = make_drop(a.drop);
  const int cc = l & 15;
  float ag[8], ab[8], axs[8];           // column sums over all the wave's row blocks
#pragma unroll
  for (int e = 0; e < 8; ++e) ag[e] = ab[e] = axs[e] = 0.f;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {     // one 32-row block at a time through the wave's two patches
  f32x16 (&acc)[4] = accs[mb];
  float* red = red_all + mb * 2 * G::WM * G::WN * 32;
  const int i_base = i_base0 + mb * 32;
  if (mb > 0) {
    if (a.aux) load_aux(mb);
    load_xhat(mb);
  }
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int id = p * 64 + l, rr = id >> 4, c = id & 15;
    *reinterpret_cast<bf16x8*>(p1 + rr * 128 + ((c ^ (rr & 15)) << 3)) = xhv[p];
    if (a.aux) *reinterpret_cast<bf16x8*>(p2 + rr * 128 + ((c ^ (rr & 15)) << 3)) = auxv[p];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  // ---- pass 1 (row per lane): dy = bf16(acc + aux); row sums of g = dy*gamma and g*xhat; dy back into the patch
  const int i = i_base + r;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int jl = b * 32 + 8 * g + 4 * hi, at = patch_at(r, jl);
      const bf16x4 xh4 = *reinterpret_cast<const bf16x4*>(p1 + at);
      bf16x4 ad4 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
      if (a.aux) ad4 = *reinterpret_cast<const bf16x4*>(p2 + at);
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(gm + jl);
      bf16x4 dy4;
      uint32_t bits = 0;
      if (DROP) bits = dr.bits(drop_counter_rc(i, wn * 128 + jl, N));   // the mask st_gemm_ln drew (drop_where = 2)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dy4[e] = (bf16)(acc[b][4 * g + e] + (float)ad4[e]);       // one rounding (see the header)
        float d = (float)dy4[e];
        if (DROP) d = dr.keep(bits, e) ? d * dr.scale : 0.f;
        const float gg = d * g4[e];
        acc[b][4 * g + e] = gg;                                    // keep g = dy * gamma
        s1 += gg;
        s2 += gg * (float)xh4[e];
      }
      *reinterpret_cast<bf16x4*>(p2 + at) = dy4;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  s1 += wave_xor32(s1);
  s2 += wave_xor32(s2);
  if (G::WN > 1) {
    if (hi == 0) {
      red[(wm * G::WN + wn) * 32 + r] = s1;
      red[G::WM * G::WN * 32 + (wm * G::WN + wn) * 32 + r] = s2;
    }
    __syncthreads();
    s1 = s2 = 0.f;
#pragma unroll
    for (int w = 0; w < G::WN; ++w) {
      s1 += red[(wm * G::WN + w) * 32 + r];
      s2 += red[G::WM * G::WN * 32 + (wm * G::WN + w) * 32 + r];
    }
  }
  const float m1 = s1 * (1.f / N), m2 = s2 * (1.f / N);
  const float rs = a.rstd[min(i, a.M - 1)];

  // ---- column sums of dy and dy*xhat in the coalesced layout: thread = 8 columns (chunk lane & 15), 8 of the 32 rows
  const int nvalid = min(32, a.M - i_base);
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int rr = (p * 64 + l) >> 4;
    const bf16x8 dyv = *reinterpret_cast<const bf16x8*>(p2 + rr * 128 + ((cc ^ (rr & 15)) << 3));
    const bf16x8 xv = *reinterpret_cast<const bf16x8*>(p1 + rr * 128 + ((cc ^ (rr & 15)) << 3));
    if (rr < nvalid) {
      uint32_t bits[2] = {0, 0};
      if (DROP) {
        bits[0] = dr.bits(drop_counter_rc(i_base + rr, wn * 128 + cc * 8, N));
        bits[1] = dr.bits(drop_counter_rc(i_base + rr, wn * 128 + cc * 8 + 4, N));
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float d = (float)dyv[e];
        if (DROP) d = dr.keep(bits[e >> 2], e & 3) ? d * dr.scale : 0.f;
        ab[e] += d;
        ag[e] += d * (float)xv[e];
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  // ---- pass 2 (row per lane): dx = rstd * (g - m1 - xhat * m2) into the patch
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int jl = b * 32 + 8 * g + 4 * hi, at = patch_at(r, jl);
      const bf16x4 xh4 = *reinterpret_cast<const bf16x4*>(p1 + at);
      bf16x4 dx4;
#pragma unroll
      for (int e = 0; e < 4; ++e) dx4[e] = (bf16)(rs * (acc[b][4 * g + e] - m1 - (float)xh4[e] * m2));
      *reinterpret_cast<bf16x4*>(p2 + at) = dx4;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  // dx leaves as 256-byte row segments; its column sum (the bias gradient of the Linear in front of the LN)
  bf16* gout = a.out + (size_t)i_base * a.ldo + wn * 128;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int rr = (p * 64 + l) >> 4;
    const bf16x8 dxv = *reinterpret_cast<const bf16x8*>(p2 + rr * 128 + ((cc ^ (rr & 15)) << 3));
    if (rr < nvalid) {
      store16<ST_GEMM_SC1>(gout + (size_t)rr * a.ldo + cc * 8, dxv);
#pragma unroll
      for (int e = 0; e < 8; ++e) axs[e] += (float)dxv[e];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the next row block rewrites the patches)
  __builtin_amdgcn_sched_barrier(0);
  }   // mb
  // ---- reduce the column sums: lanes l, l^16, l^32 share a chunk; then the WM waves of a column block through LDS
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ag[e] += __shfl_xor(ag[e], 16, 64); ag[e] += __shfl_xor(ag[e], 32, 64);
    ab[e] += __shfl_xor(ab[e], 16, 64); ab[e] += __shfl_xor(ab[e], 32, 64);
    axs[e] += __shfl_xor(axs[e], 16, 64); axs[e] += __shfl_xor(axs[e], 32, 64);
  }
  if (l < 16) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      csum[0][wm][wn * 128 + cc * 8 + e] = ag[e];
      csum[1][wm][wn * 128 + cc * 8 + e] = ab[e];
      csum[2][wm][wn * 128 + cc * 8 + e] = axs[e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < N; c += G::NT) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < G::WM; ++w) { t0 += csum[0][w][c]; t1 += csum[1][w][c]; t2 += csum[2][w][c]; }
    if (a.dgamma) atomicAdd(a.dgamma + c, t0);
    if (a.dbeta) atomicAdd(a.dbeta + c, t1);
    if (a.dbias) atomicAdd(a.dbias + c, t2);
  }
}

}  // namespace

extern "C" int st_gemm_lnbwd(hipStream_t stream, const void* dY, int lddy, const void* W, int ldw, int M, int N, int Kc,
                             const void* aux, int ldaux, const void* xhat, const float* rstd, const float* gamma,
                             void* dx, int lddx, float* dgamma, float* dbeta, float* dbias, const unsigned* drop_seed,
                             unsigned drop_salt, int drop_thresh, float drop_scale) {
  if (M <= 0) return 0;
  if ((lddy & 7) || (ldw & 7) || (lddx & 7) || (Kc & 7) || (aux && (ldaux & 7)) || !xhat || !rstd || !gamma || !dx) return -1;
  if (ldw < N) return -2;
  LnBwdArgs a;
  a.X = (const bf16*)dY; a.ldx = lddy; a.W = (const bf16*)W; a.ldw = ldw; a.M = M; a.Kc = Kc;
  a.aux = (const bf16*)aux; a.ldaux = ldaux; a.xhat = (const bf16*)xhat; a.rstd = rstd; a.gamma = gamma;
  a.out = (bf16*)dx; a.ldo = lddx; a.dgamma = dgamma; a.dbeta = dbeta; a.dbias = dbias;
  const bool drop = drop_seed != nullptr && drop_thresh > 0;
  a.drop.seed = drop ? drop_seed : nullptr; a.drop.salt = drop_salt; a.drop.thresh = drop ? drop_thresh : 0;
  a.drop.scale = drop ? drop_scale : 1.f;
  // more 64-row tiles than CUs but fewer than two rounds of them, long contraction: 128-row tiles on 8 waves
  // (measured like st_gemm_ln: helps K >= 512 only)
  if (N == 256 && Kc >= 512 && M > 64 * 256 && M <= 128 * 256) {
    const dim3 grid((M + 127) / 128);
    if (drop) hipLaunchKernelGGL((gemm_lnbwd_kernel<256, true, true>), grid, dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((gemm_lnbwd_kernel<256, false, true>), grid, dim3(512), 0, stream, a);
    ST_CHECK_LAUNCH();
    return 0;
  }
  if (N == 512 && M > 64 * 128) {      // (round 3) d_model 512, encoder-sized M: 8 waves, as in st_gemm_ln.hip
    static const bool mb1 = getenv("ST_GEMM_LN_MB1") != nullptr;      // development: the 64-row tiles everywhere
    if (mb1 || Kc < 1024) {            // 64-row tiles
      const dim3 grid((M + 63) / 64);
      if (drop) hipLaunchKernelGGL((gemm_lnbwd_kernel<512, true, true>), grid, dim3(512), 0, stream, a);
      else hipLaunchKernelGGL((gemm_lnbwd_kernel<512, false, true>), grid, dim3(512), 0, stream, a);
    } else {                           // 128-row tiles: two row blocks per wave
      const dim3 grid((M + 127) / 128);
      if (drop) hipLaunchKernelGGL((gemm_lnbwd_kernel<512, true, true, 2>), grid, dim3(512), 0, stream, a);
      else hipLaunchKernelGGL((gemm_lnbwd_kernel<512, false, true, 2>), grid, dim3(512), 0, stream, a);
    }
    ST_CHECK_LAUNCH();
    return 0;
  }
#define ST_LB(NN)                                                                                                  \
  do {                                                                                                             \
    const dim3 grid((M + Geo<NN>::BM - 1) / Geo<NN>::BM);                                                          \
    if (drop) hipLaunchKernelGGL((gemm_lnbwd_kernel<NN, true>), grid, dim3(256), 0, stream, a);                    \
    else hipLaunchKernelGGL((gemm_lnbwd_kernel<NN, false>), grid, dim3(256), 0, stream, a);                        \
  } while (0)
  if (N == 128) ST_LB(128);
  else if (N == 256) ST_LB(256);
  else if (N == 512) ST_LB(512);
  else return -3;
#undef ST_LB
  ST_CHECK_LAUNCH();
  return 0;
}

// st_gemm_ln: GEMM + bias (+ReLU) (+residual) + LayerNorm (+positional-encoding add) in one launch.
//
//   out = LN(act(x W^T + b) + res) * gamma + beta (+ pe[pos[i]])        N = d_model: full rows per workgroup
//
// Reference lines replaced: output_linear + residual + layernorm (transformer/Attention.py:92-94), fc2 +
// residual + layernorm (transformer/SubLayers.py:26-27), the encoder front-end Linear + ReLU + LayerNorm and
// the positional-encoding add (transformer/Models.py:28-33,42-44).
//
// Same structure as st_gemm_sym.hip (every wave loads, multiplies and stores; 2-3 workgroups per CU cover
// one another's latencies): 256 threads, BM x N tile with BM = 32 * (4 / (N / 128)) rows, k-tiles of 32 held
// two-deep in registers and double-buffered in LDS, fixed per-thread byte offsets (no address arithmetic or
// guards in the loop).  Each wave owns a 32-row x 128-column block with the accumulator TRANSPOSED (x row on
// the lane): LayerNorm's row statistics are an in-lane sum, one exchange with lane ^ 32 and (N > 128) one
// LDS round between the waves that share a row.  Values leave through an LDS patch as 256-byte row segments.
#include "st_common.cuh"
#include <cstdlib>

#ifndef ST_GEMM_SC1
#define ST_GEMM_SC1 0      // development: bf16 output rows write-through (store16_wt)
#endif
namespace {

constexpr int BK = 32;
constexpr int NS = BK + 8;   // LDS row stride (80 B): conflict-free ds_read_b128 over 16 rows

__device__ __attribute__((aligned(16))) float g_zero_ln[4];

struct GemmLnArgs {
  const bf16* X; int ldx;     // [M,K] natural
  const bf16* W;              // [N,K] natural, ld = K
  int M, K;
  const float* bias;          // [N]
  const bf16* res; int ldres; // residual [M,N] or null
  const float* gamma; const float* beta;
  float eps;
  int relu;                   // ReLU before LN (front-end, Models.py:28-33)
  const float* pe; const int* pos;  // optional PE table [max_len,N] and per-row position (Models.py:43-44)
  bf16* out; int ldo;         // LN output (+PE)
  bf16* xhat;                 // normalised value (saved for backward), ld = N
  float* rstd;                // [M]
  bf16* pre;                  // optional: pre-LN value (front-end ReLU mask), ld = N
  DropArgs drop;              // training-mode dropout (null seed = off)
  int drop_where;             // 1: on act(xW^T+b) before the LayerNorm (Models.py:31); 2: on the LayerNorm output (SubLayers.py:27)
};

// Geometry: WM x WN waves, each 32 rows x 128 columns (4 MFMA tiles).  N = 128: 4 x 1 (BM = 128),
// N = 256: 2 x 2 (BM = 64), N = 512: 1 x 4 (BM = 32).
// TALL (N = 256, encoder-sized M): 8 waves, 4 x 2, BM = 128 - twice the rows per weight tile that goes through LDS and
// two waves per SIMD inside ONE workgroup (M / 64 = 376 tiles leave 136 of the 256 CUs with a single 4-wave workgroup).
// MB = 2 (N = 512, encoder-sized M): every wave owns TWO 32-row blocks of its 128 columns - 128-row tiles, so that the
// 512 x 32 weight k-tile (32 KB through one CU's 64 B/clk vector-memory path) feeds 16 MFMAs per wave instead of 8: the
// 64-row tile asked for 70 B/clk at full MFMA rate.
template <int N, bool TALL = false, int MB = 1> struct Geo {
  static constexpr int WN = N / 128, WM = (TALL ? 8 : 4) / WN, BM = 32 * WM * MB, NT = 64 * WM * WN;
  static constexpr int XE = BM * NS, YE = N * NS, BUF = XE + YE;
  static constexpr int SMEM_E = 2 * BUF > WM * WN * 4096 ? 2 * BUF : WM * WN * 4096;   // operand double buffer | one patch per wave
};

// This thread's 16-byte chunks of a ROWS x 32 operand tile: fixed byte offsets from the k-tile base.
template <int ROWS, int NT = 256>   // NT: threads that cooperate on a tile (a 4-wave group, or all 8 waves of a TALL workgroup)
struct Stage {
  static constexpr int CHUNKS = ROWS * 4;                 // [ROWS][4 chunks]
  static constexpr int CH = (CHUNKS + NT - 1) / NT;       // per thread (ROWS = 32: threads 128.. duplicate 0..127)
  bf16x8 v[CH];
  static __device__ __forceinline__ int chunk_id(int p) { return ((threadIdx.x & (NT - 1)) + p * NT) % CHUNKS; }
  static __device__ __forceinline__ void offsets(uint32_t (&off)[CH], int ld, int row0, int nrows) {
#pragma unroll
    for (int p = 0; p < CH; ++p) {
      const int id = chunk_id(p);
      const int row = min(row0 + (id >> 2), nrows - 1);   // clamped rows only feed outputs that are never stored
      off[p] = ((uint32_t)row * (uint32_t)ld + (id & 3) * 8) * 2u;
    }
  }
  __device__ __forceinline__ void load(const uint32_t (&off)[CH], const bf16* __restrict__ base, int c0, int c_end) {
    const char* kb = reinterpret_cast<const char*>(base) + (size_t)c0 * 2;
    if (c0 + BK <= c_end) {
#pragma unroll
      for (int p = 0; p < CH; ++p) v[p] = *reinterpret_cast<const bf16x8*>(kb + off[p]);
    } else {   // the k-tile that crosses K: out-of-range chunks come from a zero buffer (branch-free)
#pragma unroll
      for (int p = 0; p < CH; ++p) {
        const bool ok = c0 + (chunk_id(p) & 3) * 8 < c_end;
        const char* ptr = ok ? kb + off[p] : reinterpret_cast<const char*>(g_zero_ln);
        v[p] = *reinterpret_cast<const bf16x8*>(ptr);
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

// Store a wave's [32 rows][128 cols] block of values (row-per-lane registers, column of (b, g, e) =
// b*32 + 8g + 4hi + e; value(b, g) yields e = 0..3) through its private LDS patch as 256-byte row segments.
template <typename F>
__device__ __forceinline__ void ln_store_block(bf16* patch, bf16* gbase, int ld, int nvalid_rows, F value) {
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int jl = b * 32 + 8 * g + 4 * hi;
      const f32x4 v4 = value(b, g);   // columns jl .. jl+3 of this lane's row
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)v4[e];
      *reinterpret_cast<bf16x4*>(patch + r * 128 + (((jl >> 3) ^ (r & 15)) << 3) + (jl & 7)) = o;
      if (g == 3) __builtin_amdgcn_sched_barrier(0);   // keep the per-column vector loads of one 32-column block together
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int id = p * 64 + l, rr = id >> 4, c = id & 15;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(patch + rr * 128 + ((c ^ (rr & 15)) << 3));
    if (rr < nvalid_rows) store16<ST_GEMM_SC1>(gbase + (size_t)rr * ld + c * 8, v);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// The row-wise epilogue.  `acc` holds x W^T for rows i_base + (lane & 31), columns wn*128 + ...; the
// residual block (if any) already sits in the wave's patch (ln_stage_res).
template <int N, int DROPW, bool TALL>
__device__ __forceinline__ void ln_epilogue(const GemmLnArgs& a, f32x16 (&acc)[4], bool have_res, int i_base, int wm,
                                            int wn, bf16* patch, float* red) {
  constexpr int WN = Geo<N, TALL>::WN, WM = Geo<N, TALL>::WM;
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
  const int i = i_base + r;
  const bool row_ok = i < a.M;
  const int nvalid = min(32, a.M - i_base);
  const Drop dr = make_drop(a.drop);
  constexpr bool drop_pre = DROPW == 1, drop_out = DROPW == 2;   // compile-time: the eval-mode epilogue carries no dropout code
  float sum = 0.f;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int jl = b * 32 + 8 * g + 4 * hi, j = wn * 128 + jl;
      const f32x4 bb = *reinterpret_cast<const f32x4*>(a.bias + j);
      bf16x4 rr = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
      if (have_res) rr = *reinterpret_cast<const bf16x4*>(patch + r * 128 + (((jl >> 3) ^ (r & 15)) << 3) + (jl & 7));
      uint32_t bits = 0;
      if (drop_pre) bits = dr.bits(drop_counter_rc(i, j, N));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[b][4 * g + e] + bb[e];
        if (a.relu) v = fmaxf(v, 0.f);
        if (drop_pre) v = dr.keep(bits, e) ? v * dr.scale : 0.f;
        v += (float)rr[e];
        acc[b][4 * g + e] = v;
        sum += v;
      }
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  sum += wave_xor32(sum);
  if (WN > 1) {
    if (hi == 0) red[(wm * WN + wn) * 32 + r] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int w = 0; w < WN; ++w) sum += red[(wm * WN + w) * 32 + r];
  }
  const float mean = sum * (1.f / N);
  float sq = 0.f;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float d = acc[b][q] - mean;
      sq += d * d;
    }
  sq += wave_xor32(sq);
  if (WN > 1) {
    if (hi == 0) red[WM * WN * 32 + (wm * WN + wn) * 32 + r] = sq;
    __syncthreads();
    sq = 0.f;
#pragma unroll
    for (int w = 0; w < WN; ++w) sq += red[WM * WN * 32 + (wm * WN + w) * 32 + r];
  }
  const float rstd = rsqrtf(sq * (1.f / N) + a.eps);
  if (a.rstd && row_ok && wn == 0 && hi == 0) a.rstd[i] = rstd;
  const float* perow = (a.pe != nullptr && row_ok) ? a.pe + (size_t)a.pos[i] * N + wn * 128 : nullptr;
  const float* gm = a.gamma + wn * 128;
  const float* bt = a.beta + wn * 128;
  if (a.pre)
    ln_store_block(patch, a.pre + (size_t)i_base * N + wn * 128, N, nvalid, [&](int b, int g) {
      return f32x4{acc[b][4 * g], acc[b][4 * g + 1], acc[b][4 * g + 2], acc[b][4 * g + 3]};
    });
  if (a.xhat)
    ln_store_block(patch, a.xhat + (size_t)i_base * N + wn * 128, N, nvalid, [&](int b, int g) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (acc[b][4 * g + e] - mean) * rstd;
      return v;
    });
  ln_store_block(patch, a.out + (size_t)i_base * a.ldo + wn * 128, a.ldo, nvalid, [&](int b, int g) {
    const int j4 = b * 32 + 8 * g + 4 * hi;
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(gm + j4);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bt + j4);
    f32x4 p4 = {0.f, 0.f, 0.f, 0.f};
    if (perow) p4 = *reinterpret_cast<const f32x4*>(perow + j4);
    uint32_t bits = 0;
    if (drop_out) bits = dr.bits(drop_counter_rc(i, wn * 128 + j4, N));
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = (acc[b][4 * g + e] - mean) * rstd * g4[e] + b4[e] + p4[e];
      if (drop_out) v[e] = dr.keep(bits, e) ? v[e] * dr.scale : 0.f;
    }
    return v;
  });
}

// KG = 2 (decoder-sized M, long K: ~20 workgroups whose k-loop is a serial latency chain): 8 waves, the second
// 4-wave group takes the second half of K with its own LDS buffers and hands its accumulators to the first through
// LDS; the first group alone runs the LayerNorm epilogue.  One such workgroup per CU.
template <int N, int DROPW, int KG, bool TALL = false, int MB = 1>
__global__ __launch_bounds__((TALL ? 512 : 256 * KG), ((KG > 1 || TALL) ? 1 : 2)) void gemm_ln_kernel(GemmLnArgs a) {
  static_assert(!(TALL && KG > 1), "one or the other");
  static_assert(MB == 1 || TALL, "two row blocks per wave: 8-wave workgroups only");
  using G = Geo<N, TALL, MB>;
  __shared__ __attribute__((aligned(16))) bf16 smem_all[KG > 1 ? KG * 2 * G::BUF : G::SMEM_E];
  __shared__ float red[2 * G::WM * G::WN * 32];
  const int grp = KG > 1 ? (int)(threadIdx.x >> 8) : 0;
  bf16* smem = smem_all + grp * 2 * G::BUF;
  const int wave = (threadIdx.x >> 6) & (G::WM * G::WN - 1), l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
  const int wm = wave / G::WN, wn = wave % G::WN;
  const int i0 = blockIdx.x * G::BM, i_base = i0 + wm * 32 * MB;      // the wave's rows: i_base + mb * 32 + (lane & 31)
  auto xs = [&](int buf) { return smem + buf * G::BUF; };
  auto ys = [&](int buf) { return smem + buf * G::BUF + G::XE; };

  f32x16 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[mb][b] = zero16();

  using SX = Stage<G::BM, G::NT>;
  using SY = Stage<N, G::NT>;
  uint32_t offx[SX::CH], offy[SY::CH];
  SX::offsets(offx, a.ldx, i0, a.M);
  SY::offsets(offy, a.K, 0, N);
  // two k-tiles in flight in registers (A: even tiles, B: odd tiles) + one in LDS being consumed
  SX ax, bx;
  SY ay, by;
  // this group's k-range [k_begin, k_end): both groups run the same number of k-tiles (surplus tiles read zeros)
  const int nk = ((a.K + BK - 1) / BK + KG - 1) / KG;
  const int k_begin = grp * nk * BK, k_end = min(a.K, k_begin + nk * BK);
  auto loadA = [&](int kt) { ax.load(offx, a.X, k_begin + kt * BK, k_end); ay.load(offy, a.W, k_begin + kt * BK, k_end); };
  auto loadB = [&](int kt) { bx.load(offx, a.X, k_begin + kt * BK, k_end); by.load(offy, a.W, k_begin + kt * BK, k_end); };
  auto storeA = [&]() { ax.store(xs(0)); ay.store(ys(0)); };
  auto storeB = [&]() { bx.store(xs(1)); by.store(ys(1)); };
  auto compute = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 xf[MB], yf[4];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) xf[mb] = frag_nat(xs(buf), NS, (wm * MB + mb) * 32 + r, kk * 16 + hi * 8);
#pragma unroll
      for (int b = 0; b < 4; ++b) yf[b] = frag_nat(ys(buf), NS, wn * 128 + b * 32 + r, kk * 16 + hi * 8);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[mb][b] = mfma32(yf[b], xf[mb], acc[mb][b]);
    }
  };
  loadA(0);
  if (nk > 1) loadB(1);
  storeA();
  __syncthreads();
  int kt = 0;
  for (; kt + 3 < nk; kt += 2) {   // steady state: no conditionals -> counted s_waitcnt vmcnt()
    loadA(kt + 2);
    compute(0);
    storeB();
    __syncthreads();
    loadB(kt + 3);
    compute(1);
    storeA();
    __syncthreads();
  }
  // tail: 1..3 tiles left; tile kt is in LDS buffer 0, tile kt+1 (if any) in the B registers
  if (kt + 2 < nk) loadA(kt + 2);
  // this lane's 8 residual chunks of the wave's [32][128] block (chunk id = p*64 + lane -> row id >> 4,
  // 16-byte column chunk id & 15): requested now, parked in the wave's patch after the k-loop
  bf16x8 resv[8];
  auto load_res = [&](int mb) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int id = p * 64 + l, rr = min(i_base + mb * 32 + (id >> 4), a.M - 1), c = id & 15;
      resv[p] = *reinterpret_cast<const bf16x8*>(a.res + (size_t)rr * a.ldres + wn * 128 + c * 8);
    }
  };
  if (a.res && grp == 0) load_res(0);
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
  __syncthreads();   // the patches reuse the operand buffers
  if (KG > 1) {   // group 1 -> LDS ([register][lane]: conflict-free both ways) -> group 0
    float* xch = reinterpret_cast<float*>(smem_all) + wave * (64 * 64) + l;
    if (grp == 1) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int t = 0; t < 16; ++t) xch[(b * 16 + t) * 64] = acc[0][b][t];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[0][b][t] += xch[(b * 16 + t) * 64];
    __syncthreads();   // (the four remaining waves) every hand-over block is read before a patch overwrites it
  }
  bf16* patch = smem + wave * 4096;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {      // one 32-row block at a time through the wave's patch
    if (a.res) {
      if (mb > 0) load_res(mb);
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int id = p * 64 + l, rr = id >> 4, c = id & 15;
        *reinterpret_cast<bf16x8*>(patch + rr * 128 + ((c ^ (rr & 15)) << 3)) = resv[p];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    ln_epilogue<N, DROPW, TALL>(a, acc[mb], a.res != nullptr, i_base + mb * 32, wm, wn, patch, red);
  }
}

}  // namespace

extern "C" int st_gemm_ln(hipStream_t stream, const void* X, int ldx, const void* W, int M, int N, int K,
                          const float* bias, const void* res, int ldres, const float* gamma, const float* beta,
                          float eps, int relu, const float* pe, const int* pos, void* out, int ldo, void* xhat,
                          float* rstd, void* pre, const unsigned* drop_seed, unsigned drop_salt, int drop_thresh,
                          float drop_scale, int drop_where) {
  if (M <= 0) return 0;
  if ((ldx & 7) || (K & 7) || (ldo & 7) || (res && (ldres & 7)) || !bias || !gamma || !beta || !out) return -1;
  if (pe && !pos) return -2;
  GemmLnArgs a;
  a.X = (const bf16*)X; a.ldx = ldx; a.W = (const bf16*)W; a.M = M; a.K = K; a.bias = bias;
  a.res = (const bf16*)res; a.ldres = ldres; a.gamma = gamma; a.beta = beta; a.eps = eps; a.relu = relu;
  a.pe = pe; a.pos = pos; a.out = (bf16*)out; a.ldo = ldo; a.xhat = (bf16*)xhat; a.rstd = rstd; a.pre = (bf16*)pre;
  const bool drop = drop_seed != nullptr && drop_thresh > 0 && (drop_where == 1 || drop_where == 2);
  a.drop.seed = drop ? drop_seed : nullptr; a.drop.salt = drop_salt; a.drop.thresh = drop ? drop_thresh : 0;
  a.drop.scale = drop ? drop_scale : 1.f; a.drop_where = drop ? drop_where : 0;
#define ST_LN(NN, KG)                                                                                              \
  do {                                                                                                             \
    const dim3 grid((M + Geo<NN>::BM - 1) / Geo<NN>::BM), blk(256 * KG);                                           \
    if (a.drop_where == 1) hipLaunchKernelGGL((gemm_ln_kernel<NN, 1, KG>), grid, blk, 0, stream, a);               \
    else if (a.drop_where == 2) hipLaunchKernelGGL((gemm_ln_kernel<NN, 2, KG>), grid, blk, 0, stream, a);          \
    else hipLaunchKernelGGL((gemm_ln_kernel<NN, 0, KG>), grid, blk, 0, stream, a);                                 \
  } while (0)
  // few workgroups and a long contraction: split K inside an 8-wave workgroup (N = 512 would not fit its LDS)
  const bool split = N <= 256 && K >= 512 && (M + Geo<256>::BM - 1) / Geo<256>::BM <= 128;
  // more 64-row tiles than CUs, but fewer than two rounds of them: 128-row tiles on 8 waves (see Geo)
  // (measured, M = 24060: K = 1024 32.7 -> 30.1 us; K = 256 16.8 -> 17.1 us, so only for the long contraction)
  const bool tall = N == 256 && K >= 512 && M > 64 * 256 && M <= 128 * 256;
  if (N == 128) { if (split) ST_LN(128, 2); else ST_LN(128, 1); }
  else if (N == 256 && tall) {
    const dim3 grid((M + 127) / 128), blk(512);
    if (a.drop_where == 1) hipLaunchKernelGGL((gemm_ln_kernel<256, 1, 1, true>), grid, blk, 0, stream, a);
    else if (a.drop_where == 2) hipLaunchKernelGGL((gemm_ln_kernel<256, 2, 1, true>), grid, blk, 0, stream, a);
    else hipLaunchKernelGGL((gemm_ln_kernel<256, 0, 1, true>), grid, blk, 0, stream, a);
  }
  else if (N == 256) { if (split) ST_LN(256, 2); else ST_LN(256, 1); }
  else if (N == 512 && M > 64 * 128) {
    // (round 3) 128-row tiles on 8 waves for encoder-sized M and long contractions (two row blocks per wave, see Geo): BM = 32
    // re-staged the whole weight matrix for every 32 rows (752 workgroups x 0.5-1 MB at config 3), BM = 64 still asked more of
    // the vector-memory path than it delivers (M = 24060, K = 2048: 117 -> 96 us)
    static const bool mb1 = getenv("ST_GEMM_LN_MB1") != nullptr;      // development: the 64-row tiles
    if (mb1 || K < 1024) {      // short contractions are epilogue-bound: 47.4 us (64 rows) vs 48.8 (128) at K = 512
      const dim3 grid((M + 63) / 64), blk(512);
      if (a.drop_where == 1) hipLaunchKernelGGL((gemm_ln_kernel<512, 1, 1, true>), grid, blk, 0, stream, a);
      else if (a.drop_where == 2) hipLaunchKernelGGL((gemm_ln_kernel<512, 2, 1, true>), grid, blk, 0, stream, a);
      else hipLaunchKernelGGL((gemm_ln_kernel<512, 0, 1, true>), grid, blk, 0, stream, a);
    } else {
      const dim3 grid((M + 127) / 128), blk(512);
      if (a.drop_where == 1) hipLaunchKernelGGL((gemm_ln_kernel<512, 1, 1, true, 2>), grid, blk, 0, stream, a);
      else if (a.drop_where == 2) hipLaunchKernelGGL((gemm_ln_kernel<512, 2, 1, true, 2>), grid, blk, 0, stream, a);
      else hipLaunchKernelGGL((gemm_ln_kernel<512, 0, 1, true, 2>), grid, blk, 0, stream, a);
    }
  }
  else if (N == 512) ST_LN(512, 1);
  else return -3;
#undef ST_LN
  ST_CHECK_LAUNCH();
  return 0;
}

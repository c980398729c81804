// st_wgrad_wide: weight gradients dW[n][k] += sum_m dY[m][n] X[m][k] for ENCODER-sized row counts (m ~ 24 k rows against
// outputs of 256 x 256 .. 256 x 1024), all problems of a backward pass in one launch.  Reference lines replaced: the
// autograd of nn.Linear in transformer/Attention.py:74-76,92 and transformer/SubLayers.py:25-26.
//
// Why not the 128 x 128 tiles of st_gemm_sym.hip (gemm_wgrad_group_kernel): both operands are contraction-major, so every
// fragment is a transposing LDS read, and a 64 x 64 wave tile needs one fragment per MFMA - PMC: 2.3 LDS instructions per
// MFMA, matrix pipe 27 % busy, and 8 token splits x 16 tiles re-read each operand 2-8 times through L2.  Here:
//   * workgroup tile 256 x 256 (8 waves, wave tile 64 x 128 = 2 x 4 MFMA blocks, 128 accumulator registers): 0.75 fragment
//     reads and 0.25 LDS stores per MFMA, each operand column block crosses L2 -> CU once per 256 output columns;
//   * one workgroup per CU and ~one workgroup per CU in the grid: the token axis is cut into as many splits as fill the
//     256 CUs (3 for config 2's 85 tiles), so a tile receives 3 rounds of fp32 atomics instead of 8;
//   * k-tile = 32 tokens (32 KiB of LDS), FOUR LDS buffers and two register stages: a tile is requested four k-steps
//     before it is multiplied and stored to LDS two steps before; fragments are double-buffered in registers across
//     k-steps, so the one barrier of a k-step sits behind MFMAs that cover the reads issued in front of it;
//   * XCD-aware walk: workgroup b runs on XCD b % 8; consecutive items (the tiles of one problem and one split, which
//     share an operand) go to the same XCD, so the shared operand crosses the fabric once (PMC: 1.22 GB fetched per
//     launch for 1.18 GB of operands).
//   * (round 5) the two waves of a SIMD run half a k-step apart - one set of four waves owns the LDS pipe while the other owns
//     the matrix pipes (ST_KSTEP): config 3's launches 644 -> 569 us on average (its 768 + 208 items are not HBM-bound); config 2's
//     one launch is (1.34 GB at 4.7 TB/s with all 255 workgroups streaming) and stays at 286 us.
// Measured (MI355X, config 2, 24 encoder problems = 227 GFLOP): 413 us -> 284 us without bias gradients, 297 us with
// them; the launch then reads HBM at ~4.9 TB/s - the bound.  One workgroup alone on a CU runs at 57 % of the MFMA peak.
#include "st_common.cuh"
#include <cstdlib>

namespace {

constexpr int BK = 32;                 // tokens per k-tile
constexpr int HALF_E = BK * 128;       // one [BK][128] image: unpadded token rows, 64-byte groups XOR-swizzled (cm_col)
constexpr int NBUF = 4;
// LDS: [4 operand halves: X columns 0-127, 128-255, dY columns 0-127, 128-255][NBUF buffers][HALF_E] - a wave reads one X
// half and one dY half, so buffer and k-offset fit the 16-bit immediate of its DS instructions (one address register
// per fragment for all buffers)
__device__ __forceinline__ bf16* half_at(bf16* smem, int half, int buf) { return smem + (half * NBUF + buf) * HALF_E; }

__device__ __attribute__((aligned(16))) float g_zero_f32[4];

// same swizzle as st_gemm_sym.hip: bits 5-6 of the column XOR (token row & 3) - conflict-free ds_read_b64_tr_b16
__device__ __forceinline__ int cm_col(int crow, int col) { return col ^ ((crow & 3) << 5); }

struct WideProblem {
  const bf16* X; const bf16* Y; float* D; float* bias;
  int ldx, ldy, ldd, M, N, Kc, c_per_split, tiles_i, tiles_j, splits;
};
constexpr int WIDE_MAX = 48;   // descriptors travel in the kernel-argument segment (3.7 KB of 4 KB)
struct WideArgs {
  int n, per_xcd;
  int first[WIDE_MAX + 1];     // first work item of problem p; first[n] = number of items
  WideProblem p[WIDE_MAX];
};

// This thread's two 16-byte chunks of a [32 tokens][256 columns] operand tile: chunk id -> token id >> 5, columns
// (id & 31) * 8.  Column blocks past the matrix edge are clamped onto the tile's first block (they only feed outputs
// that are never stored); tokens past c_end read zeros.
struct Addr {
  uint32_t off[2];
  __device__ __forceinline__ void init(int tid, int ld, int col0, int ncols) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int id = tid + p * 512;
      int col = col0 + (id & 31) * 8;
      if (col >= ncols) col = col0;
      off[p] = ((uint32_t)(id >> 5) * (uint32_t)ld + col) * 2u;
    }
  }
};

struct Stage {
  bf16x8 v[2];
  template <bool FULL>   // FULL: the caller knows that the k-tile lies inside [.., c_end) - no guard, no branch
  __device__ __forceinline__ void load(int tid, const Addr& ad, const bf16* __restrict__ base, int ld, int c0, int c_end) {
    const char* kb = reinterpret_cast<const char*>(base) + (size_t)c0 * ld * 2;
    if (FULL || c0 + BK <= c_end) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        uint32_t o = ad.off[p];
        asm volatile("" : "+v"(o));   // keep the offset 32-bit: the load takes the k-tile's SGPR base + this VGPR
        v[p] = *reinterpret_cast<const bf16x8*>(kb + o);
      }
    } else {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const bool ok = c0 + ((tid + p * 512) >> 5) < c_end;
        v[p] = *reinterpret_cast<const bf16x8*>(ok ? kb + ad.off[p] : reinterpret_cast<const char*>(g_zero_f32));
      }
    }
  }
  __device__ __forceinline__ void store(int tid, bf16* smem, int op, int buf) const {   // op: 0 = X, 1 = dY
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int id = tid + p * 512, row = id >> 5, ch = id & 31;
      *reinterpret_cast<bf16x8*>(half_at(smem, op * 2 + (ch >> 4), buf) + row * 128 + cm_col(row, (ch & 15) * 8)) = v[p];
    }
  }
};

// This lane's address in buffer 0 / tokens 0-15 of the fragment of 32 operand columns blk0 .. blk0+31 (lane-local column
// l & 31) of a half image; other buffers and the second 16 tokens are constant offsets from it (16 tokens keep row & 3).
__device__ __forceinline__ const bf16* frag_ptr(const bf16* half, int blk0) {
  const int l = threadIdx.x & 63, t = l & 15, ca = (l >> 5) * 8 + (t >> 2);
  return half + ca * 128 + cm_col(ca, blk0 + ((l >> 4) & 1) * 16 + 4 * (t & 3));
}
// transposing read (st_common.cuh: frag_tr): token rows ca + (t >> 2) and ca + 4 + (t >> 2)
__device__ __forceinline__ bf16x8 read_frag(const bf16* p, int buf, int kk) {
  const bf16* pa = p + buf * HALF_E + kk * 16 * 128;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ST_LDS bf16x4*)(pa));
  const bf16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ST_LDS bf16x4*)(pa + 4 * 128));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = up[0]; f[5] = up[1]; f[6] = up[2]; f[7] = up[3];
  return f;
}

// One work item: output tile (ti, tj) of problem `a` over token split ts.  CS: also the bias gradient.
template <bool CS, bool STAG>
__device__ __forceinline__ void wide_body(const WideProblem& a, int local, bf16* smem) {
  const int tiles = a.tiles_i * a.tiles_j;
  const int ts = local / tiles, ti = (local % tiles) % a.tiles_i, tj = (local % tiles) / a.tiles_i;
  const int i0 = ti * 256, j0 = tj * 256;
  const int c_begin = ts * a.c_per_split, c_end = min(a.Kc, c_begin + a.c_per_split);
  const int nk = (c_end - c_begin + BK - 1) / BK, nk_full = (c_end - c_begin) / BK;

  const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63, hi = l >> 5, r = l & 31;
  const int wm = __builtin_amdgcn_readfirstlane(wave >> 1), wn = __builtin_amdgcn_readfirstlane(wave & 1);   // wave tile: X columns i0 + 64 wm .. +63, dY columns j0 + 128 wn .. +127

  f32x16 acc[2][4];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = zero16();
  // bias gradient sum_m dY[m][j] in the ti == 0 workgroups: wave (wm, wn) takes the 32-column block wm of its 128 dY
  // columns (its fragment 0: a wave's dY fragments y = 0..3 are the blocks (y + wm) & 3) and feeds that fragment to ONE v_mfma_f32_16x16x32_bf16 per 16 tokens (4 accumulator registers).  Read as a
  // 16 x 32 A operand, lane l of the 32-row fragment is row l & 15 with k-group l >> 4: columns j and j + 16 share a row
  // (k-groups of different parity).  The B operand is 1 where (n & 1) == (k-group & 1): output column 0 sums the even
  // k-groups = all 16 tokens of columns 0-15, output column 1 the odd ones = columns 16-31.
  const bool do_cs = CS && ti == 0;
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};
  bf16x8 pat;
#pragma unroll
  for (int e = 0; e < 8; ++e) pat[e] = (bf16)((((l & 15) ^ (l >> 4)) & 1) ? 0.f : 1.f);

  const bf16* const X = a.X;
  const bf16* const Y = a.Y;
  const int ldx = a.ldx, ldy = a.ldy;
  Addr adx, ady;
  adx.init(tid, ldx, i0, a.M);
  ady.init(tid, ldy, j0, a.N);
  Stage sx0, sy0, sx1, sy1;
  auto load = [&](Stage& sx, Stage& sy, int kt) {
    sx.load<false>(tid, adx, X, ldx, c_begin + kt * BK, c_end);
    sy.load<false>(tid, ady, Y, ldy, c_begin + kt * BK, c_end);
  };
  auto load_full = [&](Stage& sx, Stage& sy, int kt) {
    sx.load<true>(tid, adx, X, ldx, c_begin + kt * BK, c_end);
    sy.load<true>(tid, ady, Y, ldy, c_begin + kt * BK, c_end);
  };
  auto store = [&](const Stage& sx, const Stage& sy, int buf) {
    sx.store(tid, smem, 0, buf);
    sy.store(tid, smem, 1, buf);
  };
  // Fragments are double-buffered in registers and software-pipelined ACROSS k-steps: while the MFMAs of one 16-token
  // half run, the transposing reads of the next half (of this tile, then of the next tile) are in flight - the
  // workgroup's waves hit the barrier together, so without this every wave reads while no wave multiplies.
  struct Frags { bf16x8 x[2], y[4]; };
  Frags f0, f1;
  const bf16* px[2];
  const bf16* py[4];
#pragma unroll
  for (int t = 0; t < 2; ++t) px[t] = frag_ptr(half_at(smem, wm >> 1, 0), ((wm & 1) * 2 + t) * 32);
#pragma unroll
  for (int y = 0; y < 4; ++y) py[y] = frag_ptr(half_at(smem, 2 + wn, 0), ((y + wm) & 3) * 32);   // block wm first: see cs
  auto read = [&](Frags& f, int buf, int kk) {
#pragma unroll
    for (int t = 0; t < 2; ++t) f.x[t] = read_frag(px[t], buf, kk);
#pragma unroll
    for (int y = 0; y < 4; ++y) f.y[y] = read_frag(py[y], buf, kk);
  };
  auto mma = [&](const Frags& f) {
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) acc[x][y] = mfma32(f.y[y], f.x[x], acc[x][y]);
    if (CS) {
      if (do_cs) cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.y[0], pat, cs, 0, 0, 0);
    }
  };
  // STAG: the two waves of a SIMD (wave w and w + 4) run HALF A K-STEP APART.  The LDS pipe (96 KB of transposing fragment reads at
  // ~120 B/clk + 32 KB of 16-byte writes at ~70 B/clk per k-step) is busy about as long as the matrix pipe (16 MFMAs per wave), and
  // with one barrier per k-step all eight waves read together and multiply together.  Here a k-step is two barrier intervals: in
  // one a wave stores / requests / reads ALL fragments of its tile (48 registers, as f0 + f1 held before), in the other it only
  // multiplies; waves 4-7 enter one interval late (an extra barrier in front, waves 0-3 take theirs behind), so in every interval
  // four waves own the LDS pipe and the other four the matrix pipes.  Tile k is read in intervals 2k (waves 0-3) and 2k + 1 (4-7);
  // its buffer is rewritten with tile k + 4's predecessor k + 2 ... by stores in intervals 2k + 4 and 2k + 5: no new hazard.
#define ST_KSTEP(SX, SY, CUR, LOAD, KT)                                  \
  if (STAG) {                                                            \
    store(SX, SY, ((CUR) + 2) & 3);                                      \
    LOAD(SX, SY, (KT) + 4);                                              \
    read(f0, CUR, 0);                                                    \
    read(f1, CUR, 1);                                                    \
    __builtin_amdgcn_sched_barrier(0);                                   \
    __syncthreads();                                                     \
    mma(f0);                                                             \
    mma(f1);                                                             \
    __builtin_amdgcn_sched_barrier(0);                                   \
    __syncthreads();                                                     \
  } else {                                                               \
    /* tile kt in buffer `cur`, its first half already in f0: tile kt+2 leaves its register stage for buffer (kt+2) & 3, */ \
    /* tile kt+4 is requested into the stage; second half -> f1 | MFMAs f0 | first half of tile kt+1 -> f0 (its buffer   */ \
    /* was stored one step ago and the last barrier made it visible) | MFMAs f1 | barrier                                */ \
    store(SX, SY, ((CUR) + 2) & 3);                                      \
    LOAD(SX, SY, (KT) + 4);                                              \
    read(f1, CUR, 1);                                                    \
    __builtin_amdgcn_sched_barrier(0);                                   \
    mma(f0);                                                             \
    __builtin_amdgcn_sched_barrier(0);                                   \
    read(f0, ((CUR) + 1) & 3, 0);                                        \
    __builtin_amdgcn_sched_barrier(0);                                   \
    mma(f1);                                                             \
    __builtin_amdgcn_sched_barrier(0);   /* the MFMAs stay in front of the barrier: they cover the reads of f0 */ \
    __syncthreads();                                                     \
  }

  // buffers 0, 1 <- tiles 0, 1; register stages 0, 1 <- tiles 2, 3 (tiles past nk read zeros)
  load(sx0, sy0, 0);
  load(sx1, sy1, 1);
  store(sx0, sy0, 0);
  store(sx1, sy1, 1);
  load(sx0, sy0, 2);
  load(sx1, sy1, 3);
  __syncthreads();
  const bool late = STAG && __builtin_amdgcn_readfirstlane(wave) >= 4;
  if (STAG) { if (late) __syncthreads(); }
  else read(f0, 0, 0);
  int kt = 0;
  // steady state without conditionals, so the compiler's s_waitcnt vmcnt() stays counted
  for (; kt + 8 <= nk_full; kt += 4) {
    ST_KSTEP(sx0, sy0, 0, load_full, kt)
    ST_KSTEP(sx1, sy1, 1, load_full, kt + 1)
    ST_KSTEP(sx0, sy0, 2, load_full, kt + 2)
    ST_KSTEP(sx1, sy1, 3, load_full, kt + 3)
  }
  for (; kt < nk; kt += 4) {   // the last 1..7 tiles (workgroup-uniform exits)
    ST_KSTEP(sx0, sy0, 0, load, kt)
    if (kt + 1 >= nk) break;
    ST_KSTEP(sx1, sy1, 1, load, kt + 1)
    if (kt + 2 >= nk) break;
    ST_KSTEP(sx0, sy0, 2, load, kt + 2)
    if (kt + 3 >= nk) break;
    ST_KSTEP(sx1, sy1, 3, load, kt + 3)
  }
  if (STAG && !late) __syncthreads();      // the early waves' share of the late waves' last interval
#undef ST_KSTEP

  // D^T[j][i] += acc: the lane index i is the contiguous axis of dW -> coalesced fp32 atomics
  const int ib = i0 + wm * 64, jb = j0 + wn * 128;
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int i = ib + x * 32 + r;
    if (i >= a.M) continue;
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int j = jb + ((y + wm) & 3) * 32 + acc_row(t, hi);
        if (j < a.N) atomicAdd(a.D + (size_t)j * a.ldd + i, acc[x][y][t]);
      }
  }
  if (do_cs && (l & 15) < 2) {   // C layout of the 16 x 16 MFMA: column l & 15, rows 4 (l >> 4) + t
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = jb + wm * 32 + (l & 1) * 16 + 4 * (l >> 4) + t;
      if (j < a.N) atomicAdd(a.bias + j, cs[t]);
    }
  }
}

template <bool STAG>
__global__ __launch_bounds__(512, 1) void wgrad_wide_kernel(WideArgs g) {
  __shared__ __attribute__((aligned(16))) bf16 smem[4 * NBUF * HALF_E];
  const int item = (int)(blockIdx.x & 7) * g.per_xcd + (int)(blockIdx.x >> 3);
  if (item >= g.first[g.n]) return;
  int pi = 0;
  while (pi + 1 < g.n && item >= g.first[pi + 1]) ++pi;   // workgroup-uniform
  const WideProblem& a = g.p[pi];
  if (a.bias != nullptr) wide_body<true, STAG>(a, item - g.first[pi], smem);
  else wide_body<false, STAG>(a, item - g.first[pi], smem);
}

}  // namespace

// Same arguments as st_wgrad_group (include/st_hip.h); `splits` cuts the token axis, and the caller picks it so that the
// launch has about one workgroup per CU: sum over problems of ceil(K_in / 256) * ceil(N_out / 256) * splits ~ 256.
extern "C" int st_wgrad_wide(hipStream_t stream, int n, const void* const* X, const int* ldx, const void* const* dY,
                             const int* lddy, float* const* dW, const int* lddw, float* const* db, const int* tokens,
                             const int* K_in, const int* N_out, const int* splits) {
  for (int base = 0; base < n; base += WIDE_MAX) {
    WideArgs g;
    g.n = 0;
    g.first[0] = 0;
    for (int q = base; q < n && q < base + WIDE_MAX; ++q) {
      if (tokens[q] <= 0 || K_in[q] <= 0 || N_out[q] <= 0) continue;
      if ((ldx[q] & 7) || (lddy[q] & 7)) return -1;
      if (ldx[q] < ((K_in[q] + 7) & ~7) || lddy[q] < ((N_out[q] + 7) & ~7)) return -3;
      WideProblem& p = g.p[g.n];
      p.X = (const bf16*)X[q]; p.ldx = ldx[q]; p.Y = (const bf16*)dY[q]; p.ldy = lddy[q]; p.D = dW[q]; p.ldd = lddw[q];
      p.bias = db ? db[q] : nullptr; p.M = K_in[q]; p.N = N_out[q]; p.Kc = tokens[q];
      int s = splits[q] < 1 ? 1 : splits[q];
      int per = (p.Kc + s - 1) / s;
      per = (per + BK - 1) / BK * BK;
      p.c_per_split = per;
      p.splits = (p.Kc + per - 1) / per;
      p.tiles_i = (p.M + 255) / 256; p.tiles_j = (p.N + 255) / 256;
      g.first[g.n + 1] = g.first[g.n] + p.splits * p.tiles_i * p.tiles_j;
      ++g.n;
    }
    if (g.n == 0) continue;
    g.per_xcd = (g.first[g.n] + 7) / 8;
    static const bool lockstep = [] { const char* e = getenv("ST_WGRAD_STAG"); return e != nullptr && e[0] == '0'; }();      // development: all waves in step
    if (lockstep) hipLaunchKernelGGL(wgrad_wide_kernel<false>, dim3(8 * g.per_xcd), dim3(512), 0, stream, g);
    else hipLaunchKernelGGL(wgrad_wide_kernel<true>, dim3(8 * g.per_xcd), dim3(512), 0, stream, g);
    ST_CHECK_LAUNCH();
  }
  return 0;
}

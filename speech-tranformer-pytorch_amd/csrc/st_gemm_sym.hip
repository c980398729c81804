// st_gemm: symmetric (every wave loads, multiplies and stores) bf16 MFMA GEMM for gfx950.
//
//   D[i][j] = sum_c X(i,c) * Y(j,c)            fp32 accumulate
//   forward  y  = x W^T      : X = x  [M,K] natural,      Y = W  [N,K] natural
//   dgrad    dx = dy W       : X = dy [M,N] natural,      Y = W  [N,K] contraction-major (c = n)
//   wgrad    dW = dy^T x     : X = x  [M,K] contr.-major, Y = dy [M,N] contraction-major (c = m)
// (contraction-major operands are read from LDS with ds_read_b64_tr_b16 - no transposed copy of a
// weight or an activation ever goes to HBM).  Reference lines replaced: transformer/Attention.py:74-76,92,
// transformer/SubLayers.py:25-26, transformer/Models.py:145,151 and their autograd.
//
// Why this shape (measured on MI355X with in-kernel cycle probes, tools/prof_gemm.py):
//  * a persistent, wave-specialised variant (st_gemm.hip: 4 loader + 4 consumer waves, one workgroup per
//    CU) spent ~12k cycles per 128x128x256 tile against 2k cycles of MFMA work - with ONE compute wave per
//    SIMD every dependent VALU instruction of the epilogue / address code pays full pipeline latency and
//    nothing overlaps an LDS round trip; LDS-DMA (global_load_lds) additionally tops out at ~10 B/clk/CU;
//  * so: 256-thread workgroups, 40 KiB of LDS and <= 128 VGPRs -> 3-4 workgroups (waves per SIMD) per CU,
//    each running the plain loop  [global -> registers two k-tiles ahead] -> [LDS double buffer] -> MFMA.
//    Other resident workgroups cover a workgroup's prologue, LDS latencies and epilogue.
// Tile 128 x 128 x 32; wave tile 64 x 64 = 2 x 2 v_mfma_f32_32x32x16_bf16; accumulator TRANSPOSED (X row on
// the lane, Y rows over registers).  bf16 outputs leave through an LDS patch as 256-byte row segments.
#include "st_common.cuh"

#ifndef ST_GEMM_SC1
#define ST_GEMM_SC1 0      // development: bf16 output rows write-through (store16_wt)
#endif
namespace {

#ifndef ST_GEMM_BK
#define ST_GEMM_BK 64
#endif
#ifndef ST_GEMM_OCC
#define ST_GEMM_OCC 2
#endif
constexpr int BK = ST_GEMM_BK;
constexpr int NS = BK + 8;    // natural tile row stride (144 B): conflict-free ds_read_b128 over 16 rows
constexpr int CS_CM = 128;    // contraction-major tile: unpadded c-rows, 64-byte groups XOR-swizzled by (c-row & 3) - see cm_col()
constexpr int TILE_E = (128 * NS > BK * CS_CM) ? 128 * NS : BK * CS_CM;   // elements of either tile image
constexpr int CH = BK / 16;   // 16-byte chunks per thread per operand tile
constexpr int CPR = BK / 8;   // chunks per natural row

__device__ __attribute__((aligned(16))) float g_zero_f32[4];

enum Epi { EPI_BF16 = 0, EPI_BF16_RELU = 1, EPI_F32 = 2, EPI_BF16_MASK = 3, EPI_BF16_ADD = 4, EPI_F32_ATOMIC = 5,
           EPI_F32_ATOMIC_T = 6, EPI_BF16_DELTA = 7 };

struct GemmArgs {
  const bf16* X; int ldx;
  const bf16* Y; int ldy;
  void* D; int ldd;
  int M, N, Kc;
  const float* bias;
  const bf16* aux; int ldaux;
  const bf16* aux2;   // EPI_BF16_DELTA, optional: second addend of the delta factor (aux + aux2), ld = ldaux
  int c_per_split, tiles_i, tiles_j, splits;
  int head_dim;    // EPI_BF16_DELTA: columns per attention head (32, 64 or 128)
  // Y (and the bias) may be a stack of equally spaced blocks - the same weight of consecutive identical layers as it
  // lies in the parameter arena: row r of the stack is row (r & (2^yseg_shift - 1)) of block r >> yseg_shift, blocks
  // yseg_extra + 2^yseg_shift * ldy elements apart (bias blocks: bias_extra + 2^yseg_shift).  yseg_shift = 31: one block.
  int yseg_shift; long yseg_extra, bias_extra;
  DropArgs drop;   // EPI_BF16_RELU: dropout after the ReLU (SubLayers.py:25); EPI_BF16_MASK: .scale on the survivors
  // EPI_BF16 with splits > 1 (st_gemm_splitk): fp32 partial tiles of the contraction splits and one ticket per output tile
  float* split_ws; unsigned* split_tickets;
  // EPI_BF16 (st_gemm_kscale): output columns [cs_lo, cs_hi) - multiples of 32 - leave as (acc + bias) * cs, scaled in fp32 before
  // their one rounding: the pre-scaled KEYS of a q | k | v projection (st_attn_common.cuh: k_prescaled)
  int cs_lo, cs_hi; float cs;
};

// Column of a contraction-major tile element after the swizzle: bits 5-6 of the column are XORed with (c-row & 3).
// A ds_read_b64_tr_b16 wave-instruction touches, per 32 lanes, 4 consecutive c-rows x 2 column blocks of 32 bytes;
// with 256-byte c-rows they would all sit on the same 16 banks - the XOR spreads them over all 64 (PMC showed 31 %
// of the LDS cycles of the weight-gradient kernel as bank conflicts with a padded stride instead).
__device__ __forceinline__ int cm_col(int crow, int col) { return col ^ ((crow & 3) << 5); }

// This thread's share (4 x 16 bytes) of a 128 x 64 operand tile.  The byte offsets from the k-tile's
// (wave-uniform) base are fixed for the whole kernel, so the inner loop carries no address arithmetic and no
// guards: rows past the edge are clamped onto a valid row (they only feed outputs that are never stored);
// only a k-tile that crosses c_end takes the guarded path (out-of-range chunks come from a zero buffer).
template <bool CM>
struct Addr {
  uint32_t off[CH];
  __device__ __forceinline__ void init(int tid, int ld, int row0, int nrows) {
#pragma unroll
    for (int p = 0; p < CH; ++p) {
      const int id = tid + p * 256;
      if (!CM) {                                                // [128 rows][CPR chunks]
        const int row = min(row0 + id / CPR, nrows - 1);
        off[p] = ((uint32_t)row * (uint32_t)ld + (id % CPR) * 8) * 2u;
      } else {                                                  // [64 c-rows][16 chunks]
        int row = row0 + (id & 15) * 8;
        if (row >= nrows) row = row0;
        off[p] = ((uint32_t)(id >> 4) * (uint32_t)ld + row) * 2u;
      }
    }
  }
};

template <bool CM>
struct Stage {
  bf16x8 v[CH];
  __device__ __forceinline__ void load(int tid, const Addr<CM>& ad, const bf16* __restrict__ base, int ld, int c0,
                                       int c_end) {
    const char* kb = reinterpret_cast<const char*>(base) + (CM ? (size_t)c0 * ld * 2 : (size_t)c0 * 2);
    if (c0 + BK <= c_end) {
#pragma unroll
      for (int p = 0; p < CH; ++p) v[p] = *reinterpret_cast<const bf16x8*>(kb + ad.off[p]);
    } else {
#pragma unroll
      for (int p = 0; p < CH; ++p) {
        const int id = tid + p * 256;
        const bool ok = CM ? (c0 + (id >> 4) < c_end) : (c0 + (id % CPR) * 8 < c_end);
        const char* ptr = ok ? kb + ad.off[p] : reinterpret_cast<const char*>(g_zero_f32);
        v[p] = *reinterpret_cast<const bf16x8*>(ptr);
      }
    }
  }
  __device__ __forceinline__ void store(int tid, bf16* tile) const {
#pragma unroll
    for (int p = 0; p < CH; ++p) {
      const int id = tid + p * 256;
      if (!CM) *reinterpret_cast<bf16x8*>(tile + (id / CPR) * NS + (id % CPR) * 8) = v[p];
      else *reinterpret_cast<bf16x8*>(tile + (id >> 4) * CS_CM + cm_col(id >> 4, (id & 15) * 8)) = v[p];
    }
  }
};

template <bool CM>
__device__ __forceinline__ bf16x8 read_frag(const bf16* tile, int blk_row0, int kk) {
  const int l = threadIdx.x & 63, hi = l >> 5;
  if (!CM) return frag_nat(tile, NS, blk_row0 + (l & 31), kk * 16 + hi * 8);
  // transposing read (st_common.cuh: frag_tr) on the swizzled image: c-rows ca + (t >> 2) and ca + 4 + (t >> 2)
  const int t = l & 15, ca = kk * 16 + hi * 8 + (t >> 2);
  const int col = blk_row0 + ((l >> 4) & 1) * 16 + 4 * (t & 3);
  const bf16* pa = tile + ca * CS_CM + cm_col(ca, col);
  const bf16* pb = tile + (ca + 4) * CS_CM + cm_col(ca + 4, col);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ST_LDS bf16x4*)(pa));
  const bf16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ST_LDS bf16x4*)(pb));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = up[0]; f[5] = up[1]; f[6] = up[2]; f[7] = up[3];
  return f;
}

// KG = 2 (weight gradients): two 4-wave groups share one output tile, each taking half of the workgroup's
// contraction range with its own LDS buffers; group 1 hands its accumulators to group 0 through LDS, so a
// CU still runs 8 waves but issues HALF the fp32 atomics of two independent split-K workgroups (the atomic
// epilogue is what bounds split-K here: 64 splits take 62 us where 16 take 34 us on dW[1024,256], m = 24060).
// LDS of one workgroup: per group 2 buffers x (X + Y tile) = 72 KiB; KG = 2 also needs 4 x 96 x 64 floats for the
// accumulator hand-over.
template <int KG> struct SmemSize {
  static constexpr int XCH_E = KG > 1 ? 4 * 96 * 64 * 2 : 0;
  static constexpr int E = KG * 4 * TILE_E > XCH_E ? KG * 4 * TILE_E : XCH_E;
};

// XCD-local tile walk of a single-problem launch: workgroup b runs on XCD b % 8 (own L2).  forward / dgrad: an XCD owns
// X row-tiles i = x, x+8, ... and its consecutive workgroups take the tiles_j tiles that share one row-tile (the panel
// crosses the fabric once); weight gradients: an XCD owns splits s = x, x+8, ... and walks their (i, j) tiles.
// -> false: this workgroup lies in the padding of the grid.
__device__ __forceinline__ bool xcd_walk(const GemmArgs& a, int bid, int& ts, int& ti, int& tj) {
  const int xcd = bid & 7, q = bid >> 3;
  const int minor = a.splits > 1 ? a.tiles_i * a.tiles_j : a.tiles_j;
  const int major = xcd + 8 * (q / minor), mi = q % minor;
  if (major >= (a.splits > 1 ? a.splits : a.tiles_i)) return false;
  ts = 0;
  if (a.splits > 1) { ts = major; ti = mi % a.tiles_i; tj = mi / a.tiles_i; }
  else { ti = major; tj = mi; }
  return true;
}

// Output tile (ti, tj) over token split ts of one problem.
template <bool XT, bool YT, int EPI, int KG, bool DROP = false>
__device__ __forceinline__ void gemm_body(const GemmArgs& a, int ts, int ti, int tj, bf16* smem_all) {
  const int grp = KG > 1 ? (int)(threadIdx.x >> 8) : 0, tid = threadIdx.x & 255;
  bf16* smem = smem_all + grp * 4 * TILE_E;
  const int i0 = ti * 128, j0 = tj * 128;
  int c_begin = ts * a.c_per_split, c_end = min(a.Kc, c_begin + a.c_per_split);
  int nk = (c_end - c_begin + BK - 1) / BK;
  if (KG > 1) {   // both groups run the same number of k-tiles (workgroup-wide barriers); surplus tiles read zeros
    nk = (nk + KG - 1) / KG;
    c_begin += grp * nk * BK;
    c_end = min(c_end, c_begin + nk * BK);
  }

  const int wave = tid >> 6, l = tid & 63, hi = l >> 5, r = l & 31;
  const int wm = wave >> 1, wn = wave & 1;
  auto xs = [&](int buf) { return smem + buf * 2 * TILE_E; };
  auto ys = [&](int buf) { return smem + buf * 2 * TILE_E + TILE_E; };

  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = zero16();

  // weight-gradient launches can also produce the bias gradient: sum_c Y(j,c) = one more MFMA per Y
  // fragment against a fragment of ones, on the (otherwise idle) matrix pipe of the ti == 0 workgroups
  constexpr bool CSUM = (EPI == EPI_F32_ATOMIC_T);
  const bool do_cs = CSUM && a.bias != nullptr && ti == 0 && wm == 0;
  f32x16 cs[2] = {zero16(), zero16()};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  Addr<XT> adx;
  Addr<YT> ady;
  adx.init(tid, a.ldx, i0, a.M);
  ady.init(tid, a.ldy, j0, a.N);
  // block of a stacked Y this tile / k-tile reads (tiles never straddle blocks: block rows are a multiple of 128)
  const bf16* const Yn = a.Y + (YT ? 0 : (long)(j0 >> a.yseg_shift) * a.yseg_extra);
  auto ybase = [&](int c0) { return YT ? a.Y + (long)(c0 >> a.yseg_shift) * a.yseg_extra : Yn; };
  // two k-tiles in flight in registers (A: even tiles, B: odd tiles) + one in LDS being consumed
  Stage<XT> ax, bx;
  Stage<YT> ay, by;
  auto loadA = [&](int kt) {
    ax.load(tid, adx, a.X, a.ldx, c_begin + kt * BK, c_end);
    ay.load(tid, ady, ybase(c_begin + kt * BK), a.ldy, c_begin + kt * BK, c_end);
  };
  auto loadB = [&](int kt) {
    bx.load(tid, adx, a.X, a.ldx, c_begin + kt * BK, c_end);
    by.load(tid, ady, ybase(c_begin + kt * BK), a.ldy, c_begin + kt * BK, c_end);
  };
  auto storeA = [&]() { ax.store(tid, xs(0)); ay.store(tid, ys(0)); };
  auto storeB = [&]() { bx.store(tid, xs(1)); by.store(tid, ys(1)); };
  auto compute = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 xf[2], yf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        xf[t] = read_frag<XT>(xs(buf), (wm * 2 + t) * 32, kk);
        yf[t] = read_frag<YT>(ys(buf), (wn * 2 + t) * 32, kk);
      }
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = mfma32(yf[y], xf[x], acc[x][y]);
      if (CSUM) {
        if (do_cs) {
#pragma unroll
          for (int y = 0; y < 2; ++y) cs[y] = mfma32(yf[y], ones, cs[y]);
        }
      }
    }
  };
  loadA(0);
  if (nk > 1) loadB(1);
  storeA();
  __syncthreads();
  int kt = 0;
  // steady state: no conditionals, so the compiler's s_waitcnt vmcnt() stays counted (the loads of tile
  // kt+2 remain in flight across the LDS store of tile kt+1)
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
  // tail: 1..3 tiles left; tile kt is in LDS buffer 0, tile kt+1 (if any) in the B registers
  if (kt + 2 < nk) loadA(kt + 2);
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
  __syncthreads();   // the epilogue reuses the operand buffers

  // ---- epilogues ------------------------------------------------------------------------------
  const int ib = i0 + wm * 64, jb = j0 + wn * 64;
  if (KG > 1) {   // group 1 -> LDS -> group 0 ([register][lane] layout: conflict-free both ways)
    float* xch = reinterpret_cast<float*>(smem_all) + wave * (96 * 64) + l;
    if (grp == 1) {
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int t = 0; t < 16; ++t) xch[((x * 2 + y) * 16 + t) * 64] = acc[x][y][t];
      if (do_cs) {
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int t = 0; t < 16; ++t) xch[(64 + y * 16 + t) * 64] = cs[y][t];
      }
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[x][y][t] += xch[((x * 2 + y) * 16 + t) * 64];
    if (do_cs) {
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int t = 0; t < 16; ++t) cs[y][t] += xch[(64 + y * 16 + t) * 64];
    }
    // the bf16 epilogues stage the tile in the same LDS: every wave must have read its hand-over block first
    if (EPI != EPI_F32_ATOMIC_T) __syncthreads();
  }
  if constexpr (EPI == EPI_BF16 && !DROP) {
    if (a.splits > 1) {
      // split-K with a bf16 result (st_gemm_splitk: few output tiles, a long contraction): every split leaves its fp32
      // partial tile in the scratch with write-through stores and draws the tile's ticket; the last one adds the partials
      // in split order (the result does not depend on who is last) and runs the epilogue.  No fence, no atomic adds: see
      // row_chain_split_kernel (st_rowchain.hip) for the protocol, DESIGN.md for what fp32 atomics cost here.
      __shared__ bool last;
      float* slot = a.split_ws + (size_t)(tj * a.tiles_i + ti) * a.splits * (256 * 64);
      float* mine = slot + (size_t)ts * (256 * 64);
      // (two floats per 8-byte access: half the memory instructions of the exchange)
      auto pack2 = [](float lo, float hi_) { return ((unsigned long long)__float_as_uint(hi_) << 32) | __float_as_uint(lo); };
      unsigned long long* mine8 = reinterpret_cast<unsigned long long*>(mine);
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int t = 0; t < 16; t += 2)
            __hip_atomic_store(mine8 + ((x * 2 + y) * 8 + t / 2) * 256 + tid, pack2(acc[x][y][t], acc[x][y][t + 1]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
      ST_PUBLISH_FENCE();
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if (tid == 0) {
        unsigned* tk = a.split_tickets + (tj * a.tiles_i + ti);
        last = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(a.splits - 1);
        if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if (!last) return;
      ST_MERGER_FENCE();
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = zero16();
      for (int p = 0; p < a.splits; ++p) {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(slot + (size_t)p * (256 * 64));
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
              const unsigned long long v = __hip_atomic_load(src + ((x * 2 + y) * 8 + t / 2) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              acc[x][y][t] += __uint_as_float((unsigned)v);
              acc[x][y][t + 1] += __uint_as_float((unsigned)(v >> 32));
            }
      }
      __syncthreads();      // (the bf16 epilogue stages the tile in LDS)
    }
  }
  if (EPI == EPI_F32_ATOMIC_T) {
    // D^T[j][i] += acc: the lane index i is the contiguous axis of dW -> coalesced fp32 atomics
    float* D = reinterpret_cast<float*>(a.D);
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int i = ib + x * 32 + r;
      if (i >= a.M) continue;
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int j = jb + y * 32 + acc_row(t, hi);
          if (j < a.N) atomicAdd(D + (size_t)j * a.ldd + i, acc[x][y][t]);
        }
    }
    if (do_cs && r == 0) {   // every column of cs holds the same sums: lanes 0 and 32 carry all 32 rows
      float* db = const_cast<float*>(a.bias);
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int j = jb + y * 32 + acc_row(t, hi);
          if (j < a.N) atomicAdd(db + j, cs[y][t]);
        }
    }
    return;
  }
  // bias vectors of this wave's 64 columns: branch-free, all in flight together
  f32x4 bv[2][4];
#pragma unroll
  for (int y = 0; y < 2; ++y)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int j = jb + y * 32 + 8 * g + 4 * hi;
      bv[y][g] = *reinterpret_cast<const f32x4*>((EPI != EPI_BF16_DELTA && a.bias != nullptr && j < a.N)
                                                     ? a.bias + j + (YT ? 0 : (long)(j >> a.yseg_shift) * a.bias_extra)
                                                     : g_zero_f32);
    }
  if (EPI == EPI_F32 || EPI == EPI_F32_ATOMIC) {   // logits (row-per-lane fp32 vectors)
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int i = ib + x * 32 + r;
      if (i >= a.M) continue;
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int j = jb + y * 32 + 8 * g + 4 * hi;
          if (j >= a.N) continue;
          f32x4 v = {acc[x][y][4 * g], acc[x][y][4 * g + 1], acc[x][y][4 * g + 2], acc[x][y][4 * g + 3]};
          v += bv[y][g];
          float* d = reinterpret_cast<float*>(a.D) + (size_t)i * a.ldd + j;
          if (EPI == EPI_F32) *reinterpret_cast<f32x4*>(d) = v;
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(d + e, v[e]);
          }
        }
    }
    return;
  }
  // bf16 outputs: the whole 128 x 128 tile goes through LDS (row stride 136) and leaves as 256-byte row
  // segments (a row-per-lane accumulator stored directly = 64 scattered 8-byte writes per instruction).
  constexpr int PS = 136;
  bf16* ct = smem;   // the operand buffers are free (barrier above)
  const Drop dr = make_drop(a.drop);
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int il = (wm * 2 + x) * 32 + r;
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int jl = (wn * 2 + y) * 32 + 8 * g + 4 * hi;
        bf16x4 o;
        uint32_t bits = 0;
        if (EPI == EPI_BF16_RELU && DROP) bits = dr.bits(drop_counter_rc(i0 + il, j0 + jl, a.N));   // training only
        const float cs = (EPI == EPI_BF16 && j0 + jl >= a.cs_lo && j0 + jl < a.cs_hi) ? a.cs : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[x][y][4 * g + e] + bv[y][g][e];
          if (EPI == EPI_BF16) v *= cs;
          if (EPI == EPI_BF16_RELU) {
            v = fmaxf(v, 0.f);
            if (DROP) v = dr.keep(bits, e) ? v * dr.scale : 0.f;
          }
          if (EPI == EPI_BF16_MASK) v *= a.drop.scale;   // 1 unless the forward dropped h (SubLayers.py:25)
          o[e] = (bf16)v;
        }
        *reinterpret_cast<bf16x4*>(ct + il * PS + jl) = o;
      }
  }
  // the mask / addend chunks this thread will need: requested before the barrier, all in flight together
  bf16x8 auxv[8], auxw[8];
  if (EPI == EPI_BF16_MASK || EPI == EPI_BF16_ADD || EPI == EPI_BF16_DELTA) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int id = p * 256 + tid, i = i0 + (id >> 4), j = j0 + (id & 15) * 8;
      auxv[p] = gload8(a.aux + (size_t)i * a.ldaux + j, i < a.M && j < a.N);
      if (EPI == EPI_BF16_DELTA) auxw[p] = gload8(a.aux2 + (size_t)i * a.ldaux + j, a.aux2 != nullptr && i < a.M && j < a.N);
    }
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int id = p * 256 + tid, il = id >> 4, jl = (id & 15) * 8;
    const int i = i0 + il, j = j0 + jl;
    bf16x8 v = *reinterpret_cast<const bf16x8*>(ct + il * PS + jl);
    if (EPI == EPI_BF16_MASK) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ((float)auxv[p][e] > 0.f) ? v[e] : (bf16)0.f;
    } else if (EPI == EPI_BF16_ADD) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] + (float)auxv[p][e]);
    } else if (EPI == EPI_BF16_DELTA) {
      // delta[h][i] = sum over head h's columns of D(i, .) * aux(i, .): the attention backward's rowsum(dO * O),
      // produced where dO is produced.  A head's columns sit in head_dim / 8 neighbouring lanes of one row.
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) part += (float)v[e] * ((float)auxv[p][e] + (float)auxw[p][e]);
      part += __shfl_xor(part, 1, 64);
      part += __shfl_xor(part, 2, 64);
      if (a.head_dim >= 64) part += __shfl_xor(part, 4, 64);
      if (a.head_dim == 128) part += __shfl_xor(part, 8, 64);
      const int cph = a.head_dim >> 3;   // 16-byte chunks per head
      if (i < a.M && j < a.N && ((id & 15) % cph) == 0)
        const_cast<float*>(a.bias)[(size_t)(j / a.head_dim) * a.M + i] = part;
    }
    if (i < a.M && j < a.N) store16<ST_GEMM_SC1>(reinterpret_cast<bf16*>(a.D) + (size_t)i * a.ldd + j, v);
  }
}

template <bool XT, bool YT, int EPI, int KG, bool DROP = false>
__global__ __launch_bounds__(256 * KG, KG > 1 ? 1 : ST_GEMM_OCC) void gemm_sym_kernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) bf16 smem_all[SmemSize<KG>::E];
  int ts, ti, tj;
  if (!xcd_walk(a, blockIdx.x, ts, ti, tj)) return;
  gemm_body<XT, YT, EPI, KG, DROP>(a, ts, ti, tj, smem_all);
}

// Several weight-gradient problems in ONE launch (the decoder's are ~20 workgroups each and pure latency when
// launched one by one).  The descriptors travel in the kernel-argument segment - no device-side table.
constexpr int GROUP_MAX = 48;   // 48 descriptors = 3.6 KB of the 4 KB kernel-argument segment; the decoder's group is 43 problems
struct GroupProblem {
  const bf16* X; const bf16* Y; float* D; float* bias;
  int ldx, ldy, ldd, M, N, Kc, c_per_split, tiles_i, tiles_j, splits;
};
struct GroupArgs {
  int n;
  int first[GROUP_MAX + 1];   // first workgroup of problem p; first[n] = grid size
  GroupProblem p[GROUP_MAX];
};

// Two 4-wave workgroups per CU, one k-range each (no intra-workgroup k split as in the single-problem launch): with
// thousands of workgroups in a group, one workgroup's prologue and atomic epilogue hide behind its neighbour's k-loop.
__global__ __launch_bounds__(256, 2) void gemm_wgrad_group_kernel(GroupArgs g) {
  __shared__ __attribute__((aligned(16))) bf16 smem_all[SmemSize<1>::E];
  int pi = 0;
  while (pi + 1 < g.n && (int)blockIdx.x >= g.first[pi + 1]) ++pi;   // workgroup-uniform
  const GroupProblem& p = g.p[pi];
  GemmArgs a;
  a.X = p.X; a.ldx = p.ldx; a.Y = p.Y; a.ldy = p.ldy; a.D = p.D; a.ldd = p.ldd;
  a.M = p.M; a.N = p.N; a.Kc = p.Kc; a.bias = p.bias; a.aux = nullptr; a.ldaux = 0; a.aux2 = nullptr;
  a.c_per_split = p.c_per_split; a.tiles_i = p.tiles_i; a.tiles_j = p.tiles_j; a.splits = p.splits;
  a.head_dim = 0; a.yseg_shift = 31; a.yseg_extra = 0; a.bias_extra = 0; a.cs_lo = a.cs_hi = 0; a.cs = 1.f;
  a.drop.seed = nullptr; a.drop.salt = 0; a.drop.thresh = 0; a.drop.scale = 1.f;
  // compact walk (no XCD padding): these are decoder-sized problems of 4 .. 70 tiles whose operands fit every L2 - the
  // XCD-local walk of the single-problem launch would put a problem with two row tiles and one split on XCDs 0 and 1 only
  const int local = (int)blockIdx.x - g.first[pi], tiles = p.tiles_i * p.tiles_j;
  gemm_body<true, true, EPI_F32_ATOMIC_T, 1>(a, local / tiles, (local % tiles) % p.tiles_i, (local % tiles) / p.tiles_i, smem_all);
}

template <bool XT, bool YT>
int launch(hipStream_t stream, const GemmArgs& a, int epi, dim3 grid) {
#define ST_CASE(E, KG) \
  case E: hipLaunchKernelGGL((gemm_sym_kernel<XT, YT, E, KG>), grid, dim3(256 * KG), 0, stream, a); break;
  if (epi == EPI_BF16_RELU && a.drop.seed != nullptr) {   // training-mode dropout: its own instantiation
    hipLaunchKernelGGL((gemm_sym_kernel<XT, YT, EPI_BF16_RELU, 1, true>), grid, dim3(256), 0, stream, a);
    return 0;
  }
  // few output tiles and a long contraction (the vocabulary projection's input gradient: 20 tiles, K = 4344): the two
  // wave groups of an 8-wave workgroup take half of K each
  if (epi == EPI_BF16 && a.tiles_i * a.tiles_j <= 128 && a.Kc >= 2048) {
    hipLaunchKernelGGL((gemm_sym_kernel<XT, YT, EPI_BF16, 2>), grid, dim3(512), 0, stream, a);
    return 0;
  }
  switch (epi) {
    ST_CASE(EPI_BF16, 1) ST_CASE(EPI_BF16_RELU, 1) ST_CASE(EPI_F32, 1) ST_CASE(EPI_BF16_MASK, 1)
    ST_CASE(EPI_BF16_ADD, 1) ST_CASE(EPI_F32_ATOMIC, 1) ST_CASE(EPI_F32_ATOMIC_T, 2) ST_CASE(EPI_BF16_DELTA, 1)
    default: return -1;
  }
#undef ST_CASE
  return 0;
}

}  // namespace

namespace {
// split-K plan shared by st_gemm and st_wgrad_group: returns the grid size
int plan_splits(GemmArgs& a, int splits) {
  int per = (a.Kc + splits - 1) / splits;
  per = (per + BK - 1) / BK * BK;
  splits = (a.Kc + per - 1) / per;
  a.c_per_split = per;
  a.tiles_i = (a.M + 127) / 128; a.tiles_j = (a.N + 127) / 128; a.splits = splits;
  const int minor = splits > 1 ? a.tiles_i * a.tiles_j : a.tiles_j;
  const int major_n = splits > 1 ? splits : a.tiles_i;
  return 8 * ((major_n + 7) / 8) * minor;
}
}  // namespace

extern "C" int st_wgrad_group(hipStream_t stream, int n, const void* const* X, const int* ldx, const void* const* dY,
                              const int* lddy, float* const* dW, const int* lddw, float* const* db, const int* tokens,
                              const int* K_in, const int* N_out, const int* splits) {
  for (int base = 0; base < n; base += GROUP_MAX) {
    GroupArgs g;
    g.n = 0;
    g.first[0] = 0;
    for (int q = base; q < n && q < base + GROUP_MAX; ++q) {
      if (tokens[q] <= 0 || K_in[q] <= 0 || N_out[q] <= 0) continue;
      if ((ldx[q] & 7) || (lddy[q] & 7) || (N_out[q] & 3)) return -1;
      if (ldx[q] < ((K_in[q] + 7) & ~7) || lddy[q] < ((N_out[q] + 7) & ~7)) return -3;
      GemmArgs a;
      a.M = K_in[q]; a.N = N_out[q]; a.Kc = tokens[q];
      plan_splits(a, splits[q] < 1 ? 1 : splits[q]);
      const int grid = a.splits * a.tiles_i * a.tiles_j;
      GroupProblem& p = g.p[g.n];
      p.X = (const bf16*)X[q]; p.ldx = ldx[q]; p.Y = (const bf16*)dY[q]; p.ldy = lddy[q]; p.D = dW[q]; p.ldd = lddw[q];
      p.bias = db ? db[q] : nullptr; p.M = a.M; p.N = a.N; p.Kc = a.Kc; p.c_per_split = a.c_per_split;
      p.tiles_i = a.tiles_i; p.tiles_j = a.tiles_j; p.splits = a.splits;
      g.first[g.n + 1] = g.first[g.n] + grid;
      ++g.n;
    }
    if (g.n == 0) continue;
    hipLaunchKernelGGL(gemm_wgrad_group_kernel, dim3(g.first[g.n]), dim3(256), 0, stream, g);
    ST_CHECK_LAUNCH();
  }
  return 0;
}

namespace {
int gemm_impl(hipStream_t stream, int x_cmajor, int y_cmajor, const void* X, int ldx, const void* Y,
              int ldy, void* D, int ldd, int M, int N, int Kc, float* bias, const void* aux,
              int ldaux, int epi, int splits, const unsigned* drop_seed, unsigned drop_salt,
              int drop_thresh, float drop_scale, int y_block_rows, long y_block_stride,
              long bias_block_stride, const void* aux2, void* split_work = nullptr, long long split_bytes = 0, int cs_lo = 0,
              int cs_hi = 0, float cs = 1.f) {
  if (M <= 0 || N <= 0 || Kc <= 0) return 0;
  int yseg_shift = 31;
  if (y_block_rows > 0) {   // power-of-two multiple of 128 rows per block; forward and dgrad operands only
    if (y_block_rows < 128 || (y_block_rows & (y_block_rows - 1)) || x_cmajor || (y_block_stride & 7) ||
        (bias_block_stride & 3)) return -7;
    if (epi == EPI_F32_ATOMIC || epi == EPI_F32_ATOMIC_T || epi == EPI_BF16_DELTA) return -7;
    yseg_shift = __builtin_ctz(y_block_rows);
  }
  if ((ldx & 7) || (ldy & 7) || (N & 3) || epi < 0 || epi > 7) return -1;
  if (x_cmajor && !y_cmajor) return -2;  // not needed by any caller
  // contraction-major operands are read in 8-row chunks: the caller guarantees the buffer is
  // padded (ld >= round_up(rows, 8)); rows beyond M / N only feed outputs that are never stored.
  if ((x_cmajor && ldx < ((M + 7) & ~7)) || (y_cmajor && ldy < ((N + 7) & ~7))) return -3;
  if (epi != EPI_F32 && epi != EPI_F32_ATOMIC && epi != EPI_F32_ATOMIC_T && ((ldd & 7) || (N & 7))) return -4;
  if ((epi == EPI_BF16_MASK || epi == EPI_BF16_ADD || epi == EPI_BF16_DELTA) && (ldaux & 7)) return -5;
  if (epi == EPI_BF16_DELTA && (!bias || !aux || (splits != 32 && splits != 64 && splits != 128) || (N % splits))) return -6;
  GemmArgs a;
  a.head_dim = epi == EPI_BF16_DELTA ? splits : 0;   // this epilogue takes the head width in the `splits` slot
  if (splits < 1) splits = 1;
  if (epi != EPI_F32_ATOMIC && epi != EPI_F32_ATOMIC_T && !(epi == EPI_BF16 && split_work)) splits = 1;
  a.split_ws = nullptr; a.split_tickets = nullptr;
  a.cs_lo = cs_lo; a.cs_hi = cs_hi; a.cs = cs;
  a.X = (const bf16*)X; a.ldx = ldx; a.Y = (const bf16*)Y; a.ldy = ldy; a.D = D; a.ldd = ldd;
  a.M = M; a.N = N; a.Kc = Kc; a.bias = bias; a.aux = (const bf16*)aux; a.ldaux = ldaux;
  a.aux2 = epi == EPI_BF16_DELTA ? (const bf16*)aux2 : nullptr;
  a.yseg_shift = yseg_shift;
  // per block: what the plain row stride does not already cover (rows run along the contraction axis for dgrad)
  a.yseg_extra = y_block_rows > 0 ? y_block_stride - (long)y_block_rows * ldy : 0;
  a.bias_extra = y_block_rows > 0 ? bias_block_stride - (long)y_block_rows : 0;
  const bool drop = epi == EPI_BF16_RELU && drop_seed != nullptr && drop_thresh > 0;
  a.drop.seed = drop ? drop_seed : nullptr; a.drop.salt = drop_salt; a.drop.thresh = drop ? drop_thresh : 0;
  a.drop.scale = (drop || epi == EPI_BF16_MASK) && drop_scale > 0.f ? drop_scale : 1.f;
  dim3 grid(plan_splits(a, splits));
  if (epi == EPI_BF16 && a.splits > 1) {      // scratch: 1024 tickets (fixed place), then splits x 64 KB per output tile
    const long long tiles = (long long)a.tiles_i * a.tiles_j;
    if (tiles > 1024 || split_bytes < 4096 + tiles * a.splits * (256 * 64 * 4)) return -8;
    a.split_tickets = (unsigned*)split_work;
    a.split_ws = (float*)split_work + 1024;
  }
  int rc;
  if (!x_cmajor && !y_cmajor) rc = launch<false, false>(stream, a, epi, grid);
  else if (!x_cmajor && y_cmajor) rc = launch<false, true>(stream, a, epi, grid);
  else rc = launch<true, true>(stream, a, epi, grid);
  if (rc) return rc;
  ST_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// D (bf16) = X Y^T with the contraction cut `splits` ways over workgroups (few output tiles, long Kc): see gemm_body
extern "C" int st_gemm_splitk(hipStream_t stream, int y_cmajor, const void* X, int ldx, const void* Y, int ldy, void* D, int ldd,
                              int M, int N, int Kc, int splits, void* work, long long work_bytes) {
  if (!work || splits < 1) return -1;
  return gemm_impl(stream, 0, y_cmajor, X, ldx, Y, ldy, D, ldd, M, N, Kc, nullptr, nullptr, 0, EPI_BF16, splits, nullptr, 0u, 0, 1.f,
                   0, 0, 0, nullptr, work, work_bytes);
}

// D (bf16 [M, N]) = (X Y^T + bias), columns [col_lo, col_hi) multiplied by `scale` in fp32 before the rounding: the q | k | v
// projection (Attention.py:74-76) with its KEY block pre-scaled for the attention kernels' k_prescaled mode
extern "C" int st_gemm_kscale(hipStream_t stream, const void* X, int ldx, const void* Y, int ldy, void* D, int ldd, int M, int N,
                              int Kc, float* bias, int col_lo, int col_hi, float scale) {
  if ((col_lo & 31) || (col_hi & 31) || col_lo < 0 || col_hi < col_lo || col_hi > N) return -9;
  return gemm_impl(stream, 0, 0, X, ldx, Y, ldy, D, ldd, M, N, Kc, bias, nullptr, 0, EPI_BF16, 1, nullptr, 0u, 0, 1.f, 0, 0, 0, nullptr,
                   nullptr, 0, col_lo, col_hi, scale);
}

extern "C" int st_gemm_stacked(hipStream_t stream, int x_cmajor, int y_cmajor, const void* X, int ldx, const void* Y,
                               int ldy, void* D, int ldd, int M, int N, int Kc, float* bias, const void* aux,
                               int ldaux, int epi, int splits, const unsigned* drop_seed, unsigned drop_salt,
                               int drop_thresh, float drop_scale, int y_block_rows, long y_block_stride,
                               long bias_block_stride) {
  return gemm_impl(stream, x_cmajor, y_cmajor, X, ldx, Y, ldy, D, ldd, M, N, Kc, bias, aux, ldaux, epi, splits, drop_seed,
                   drop_salt, drop_thresh, drop_scale, y_block_rows, y_block_stride, bias_block_stride, nullptr);
}

extern "C" int st_gemm(hipStream_t stream, int x_cmajor, int y_cmajor, const void* X, int ldx, const void* Y, int ldy,
                       void* D, int ldd, int M, int N, int Kc, float* bias, const void* aux, int ldaux, int epi,
                       int splits, const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale,
                       const void* aux2) {
  return gemm_impl(stream, x_cmajor, y_cmajor, X, ldx, Y, ldy, D, ldd, M, N, Kc, bias, aux, ldaux, epi, splits,
                   drop_seed, drop_salt, drop_thresh, drop_scale, 0, 0, 0, aux2);
}

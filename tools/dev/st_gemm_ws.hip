// bf16 MFMA GEMMs for the speech-transformer training step (gfx950).
//
//   D[i][j] = sum_c X(i,c) * Y(j,c)            fp32 accumulate
//
// Each operand is either "natural" ([rows][c], c contiguous) or "contraction-major"
// ([c][rows], rows contiguous; read from LDS with ds_read_b64_tr_b16).  That one
// kernel family covers the three GEMMs of every nn.Linear in the reference
// (transformer/Attention.py:74-76,92, transformer/SubLayers.py:25-26,
// transformer/Models.py:28-33,145):
//   forward  y  = x W^T      : X = x  [M,K] natural,      Y = W  [N,K] natural
//   dgrad    dx = dy W       : X = dy [M,N] natural,      Y = W  [N,K] contraction-major (c = n)
//   wgrad    dW = dy^T x     : X = x  [M,K] contr.-major, Y = dy [M,N] contraction-major (c = m)
// so no transposed copy of a weight or an activation is ever written to HBM.
//
// Structure (why): every GEMM of this model has a short contraction (256..1024, or a split of
// the token axis), so a 128x128 output tile needs only ~1 us of MFMA time while an HBM/L2 tile
// fetch takes ~2 us under load - a tile-per-workgroup kernel with one tile of prefetch measured
// 3x slower than its own MFMA + HBM bounds.  The kernels are therefore PERSISTENT (one
// 512-thread workgroup per CU walking a list of output tiles) with the waves SPECIALISED:
//   waves 4-7  loaders  : keep the next 4 (tile, k-step) items of 128x64 operand tiles in flight in
//                         their REGISTERS (global_load_dwordx4, across output-tile boundaries) and
//                         copy one item per step into a 2-slot LDS double buffer (ds_write_b128).
//                         Registers, not LDS-DMA: measured in-kernel, global_load_lds tops out at
//                         ~10 B/clk/CU whatever the source (even with every operand row aliased to
//                         one cache line), a quarter of what the MFMA side of these GEMMs needs,
//                         while L2 -> VGPR streams at up to 64 B/clk/CU and the VGPR file of the four
//                         loader waves holds 128 KiB in flight (LDS could hold 96);
//   waves 0-3  consumers: 2x2 waves x (2x2 v_mfma_f32_32x32x16_bf16), epilogue from registers
//                         through a wave-private LDS patch to coalesced 128-byte row segments.
// One s_barrier per item orders ring slots between the two roles.  The accumulator is kept
// TRANSPOSED (X row i on the lane, Y row j over registers): per-row LayerNorm statistics and
// row-contiguous stores need no cross-lane traffic beyond one exchange with lane ^ 32.
#include "st_common.cuh"

namespace {

constexpr int BK = 64;    // contraction elements per LDS tile = one 128-byte row of a natural tile
constexpr int RING = 4;   // ring slots of st_gemm (3 items in flight)

__device__ __attribute__((aligned(16))) bf16 g_zero_chunk[8];  // zero-initialised source for padding lanes
__device__ __attribute__((aligned(16))) float g_zero_f32[4];   // stands in for an absent bias vector

enum Epi { EPI_BF16 = 0, EPI_BF16_RELU = 1, EPI_F32 = 2, EPI_BF16_MASK = 3, EPI_BF16_ADD = 4, EPI_F32_ATOMIC = 5,
           EPI_F32_ATOMIC_T = 6 };

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Loader wave: wait until at most `younger` items (IPW wave-instructions each) are still in flight.
template <int IPW>
__device__ __forceinline__ void wait_items(int younger) {
  if (younger >= 2) wait_vmcnt<2 * IPW>();
  else if (younger == 1) wait_vmcnt<IPW>();
  else wait_vmcnt<0>();
}

// ---- operand tiles: HBM -> LDS by LDS-DMA, no VGPR staging ------------------------------------
// One wave-instruction moves 64 x 16 bytes to wave_base + 16 * lane, so the LDS image is lane-linear
// and the bank-conflict swizzle lives on the per-lane SOURCE address; fragment reads undo it with the
// same XOR (an involution):
//   natural  [ROWS][64] : 16-byte chunk p of row r   holds logical chunk p ^ ((r >> 1) & 7)
//                         -> any 16 consecutive rows read conflict-free with ds_read_b128
//   c-major  [64][ROWS] : chunk p of contraction row c holds logical chunk p ^ ((c & 3) << 2)
//                         -> the 4 c-rows of a ds_read_b64_tr_b16 land in 4 different 64-byte bank groups
// Lanes outside the matrix (row >= nrows or c >= c_end) read a 16-byte zero chunk instead.
// `lw` = index of the issuing wave among the 4 waves that share the tile (wave-uniform).
template <int ROWS>
__host__ __device__ constexpr int tile_elems() { return ROWS * BK; }

template <int ROWS, bool CM>
__device__ __forceinline__ void issue_tile(bf16* tile, const bf16* __restrict__ base, int ld, int row0, int nrows,
                                           int c0, int c_end, int lw) {
  static_assert(ROWS % 32 == 0 && (!CM || ROWS == 128), "tile shape");
  const int i = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t < ROWS / 32; ++t) {
    const int I = t * 4 + lw;  // wave-instruction index: 1 KiB of the tile each
    const bf16* src;
    if (!CM) {
      const int r = I * 8 + (i >> 3), p = i & 7;
      const int row = row0 + r, c = c0 + 8 * (p ^ ((r >> 1) & 7));
      src = (row < nrows && c < c_end) ? base + (size_t)row * ld + c : g_zero_chunk;
    } else {
      const int cr = I * 4 + (i >> 4), p = i & 15;
      const int c = c0 + cr, row = row0 + 8 * (p ^ ((cr & 3) << 2));
      src = (c < c_end && row < nrows) ? base + (size_t)c * ld + row : g_zero_chunk;
    }
    lds_dma16(src, lds_addr(tile + I * 512));
  }
}

// Register-staged variant of issue_tile: the same LDS image, filled in two steps - load() puts this
// wave's share of a tile (ROWS/32 x 16 bytes per lane) in flight into VGPRs, store() writes it to LDS
// later.  Ordinary loads: the compiler's counted s_waitcnt vmcnt keeps younger items in flight.
template <int ROWS, bool CM>
struct TileRegs {
  static_assert(ROWS % 32 == 0 && (!CM || ROWS == 128), "tile shape");
  bf16x8 v[ROWS / 32];
  __device__ __forceinline__ void load(const bf16* __restrict__ base, int ld, int row0, int nrows, int c0, int c_end,
                                       int lw) {
    const int i = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < ROWS / 32; ++t) {
      const int I = t * 4 + lw;
      if (!CM) {
        const int r = I * 8 + (i >> 3), p = i & 7;
        const int row = row0 + r, c = c0 + 8 * (p ^ ((r >> 1) & 7));
        v[t] = gload8(base + (size_t)row * ld + c, row < nrows && c < c_end);
      } else {
        const int cr = I * 4 + (i >> 4), p = i & 15;
        const int c = c0 + cr, row = row0 + 8 * (p ^ ((cr & 3) << 2));
        v[t] = gload8(base + (size_t)c * ld + row, c < c_end && row < nrows);
      }
    }
  }
  __device__ __forceinline__ void store(bf16* tile, int lw) const {
    const int i = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < ROWS / 32; ++t) *reinterpret_cast<bf16x8*>(tile + (t * 4 + lw) * 512 + i * 8) = v[t];
  }
};

// Fragment (8 contraction elements kk*16 + hi*8 .. +7 of operand row blk_row0 + (lane & 31)).
template <int ROWS, bool CM>
__device__ __forceinline__ bf16x8 read_frag(const bf16* tile, int blk_row0, int kk) {
  const int l = threadIdx.x & 63, hi = l >> 5;
  if (!CM) {
    const int r = blk_row0 + (l & 31), q = kk * 2 + hi;
    return *reinterpret_cast<const bf16x8*>(tile + r * BK + ((q ^ ((r >> 1) & 7)) << 3));
  }
  const int t = l & 15, col = blk_row0 + ((l >> 4) & 1) * 16 + 4 * (t & 3);
  const int ca = kk * 16 + hi * 8 + (t >> 2);            // ca & 3 == t >> 2
  const bf16* pa = tile + ca * ROWS + (((col >> 3) ^ ((t >> 2) << 2)) << 3) + (col & 7);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ST_LDS bf16x4*)(pa));
  const bf16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ST_LDS bf16x4*)(pa + 4 * ROWS));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = up[0]; f[5] = up[1]; f[6] = up[2]; f[7] = up[3];
  return f;
}

#ifdef ST_PROF
__device__ unsigned long long g_prof[256 * 8];   // per workgroup: barrier-wait, mfma, epilogue, total cycles, tiles
#define PROF_NOW() __builtin_readcyclecounter()
__device__ int g_dbg;   // bit 0: skip output stores, bit 1: skip operand loads, bit 2: skip LDS tile writes
#define DBG(bit) (g_dbg & (bit))
#else
#define DBG(bit) 0
#endif

struct GemmArgs {
  const bf16* X; int ldx;
  const bf16* Y; int ldy;
  void* D; int ldd;
  int M, N, Kc;
  const float* bias;      // [N] or null
  const bf16* aux; int ldaux;  // mask source (EPI_BF16_MASK) or addend (EPI_BF16_ADD)
  int epi;
  int c_per_split;        // contraction elements per split
  int tiles_i, tiles_j, splits;
};

// ---- the (output tile, k-step) item stream of one persistent workgroup --------------------------
// XCD-aware tile order.  Workgroup w runs on XCD w % 8 (its own 4 MiB L2; L2s are not shared), so
// tiles that read the same operand panel should meet on ONE XCD at about the same time:
//   forward / dgrad : XCD x owns the X row-tiles i = x, x+8, ...; its workgroups walk (i, j) with j
//                     fastest, so the tiles_j tiles that share X row-tile i run side by side on that XCD -
//                     the panel crosses the fabric once and is re-read from L2 (the weights are small
//                     and end up resident in every L2);
//   weight gradients: XCD x owns the splits s = x, x+8, ... and walks the (i, j) tiles of one split.
// (i-fastest order across all workgroups - every re-read of an X panel from a different XCD - measured
// no faster than a tile-per-workgroup launch: the fabric, not HBM or MFMA, was the limit.)
struct Work {
  int q, kt, nk, i0, j0, c_begin, c_end;   // q = index into this XCD's tile list
  bool valid;
};

__device__ __forceinline__ void work_load(Work& w, const GemmArgs& a, int xcd) {
  const int minor = a.splits > 1 ? a.tiles_i * a.tiles_j : a.tiles_j;
  const int major_n = a.splits > 1 ? a.splits : a.tiles_i;
  const int major = xcd + 8 * (w.q / minor), mi = w.q % minor;
  w.valid = major < major_n;
  if (!w.valid) return;
  int ts = 0, ti, tj;
  if (a.splits > 1) { ts = major; ti = mi % a.tiles_i; tj = mi / a.tiles_i; }
  else { ti = major; tj = mi; }
  w.i0 = ti * 128;
  w.j0 = tj * 128;
  w.c_begin = ts * a.c_per_split;
  w.c_end = min(a.Kc, w.c_begin + a.c_per_split);
  w.nk = (w.c_end - w.c_begin + BK - 1) / BK;
  w.kt = 0;
}

__device__ __forceinline__ void work_init(Work& w, const GemmArgs& a) {
  w.q = blockIdx.x >> 3;                 // slot of this workgroup on its XCD
  work_load(w, a, blockIdx.x & 7);
}

__device__ __forceinline__ void work_next(Work& w, const GemmArgs& a) {
  if (++w.kt >= w.nk) {
    w.q += gridDim.x >> 3;               // gridDim.x is a multiple of 8
    work_load(w, a, blockIdx.x & 7);
  }
}

template <bool XT, bool YT>
__device__ __forceinline__ void gemm_loader(const GemmArgs& a, bf16* ring, int lw) {
  constexpr int XE = tile_elems<128>(), SLOT = 2 * XE, D = 4;   // D items in flight in registers
  TileRegs<128, XT> rx[D];
  TileRegs<128, YT> ry[D];
  Work cur, ahead;
  work_init(cur, a);
  ahead = cur;
  auto fetch = [&](TileRegs<128, XT>& x, TileRegs<128, YT>& y) {
    const int c0 = ahead.c_begin + ahead.kt * BK;
    x.load(a.X, a.ldx, ahead.i0, a.M, c0, ahead.c_end, lw);
    y.load(a.Y, a.ldy, ahead.j0, a.N, c0, ahead.c_end, lw);
    work_next(ahead, a);
  };
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (ahead.valid) fetch(rx[d], ry[d]);
  int k = 0;
  for (;;) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (!cur.valid) return;
      // item k -> slot k & 1: the consumers left that slot (item k-2) before they reached barrier k-1
      bf16* slot = ring + (k & 1) * SLOT;
      if (!DBG(4)) {
        rx[d].store(slot, lw);
        ry[d].store(slot + XE, lw);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();        // barrier k: item k is readable
      if (ahead.valid) {
        if (!DBG(2)) fetch(rx[d], ry[d]);
        else work_next(ahead, a);
      }
      ++k;
      work_next(cur, a);
    }
  }
}

// Epilogue of one consumer wave: its 64 x 64 sub-tile at (i0 + wm*64, j0 + wn*64).
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x16 (&acc)[2][2], const bf16x8 (&auxv)[8],
                                              const f32x4 (&bv)[2][4], int i0, int j0, bf16* patch, int wm, int wn) {
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
  const int ib = i0 + wm * 64, jb = j0 + wn * 64;
  if (a.epi == EPI_F32_ATOMIC_T) {
    // D^T[j][i] += acc: the lane index i is the CONTIGUOUS axis of the output, so every atomic
    // instruction covers 2 x 128 contiguous bytes (coalesced L2 atomics).
    float* D = reinterpret_cast<float*>(a.D);
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int i = ib + x * 32 + r;
      if (i >= a.M) continue;
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int j = jb + y * 32 + acc_row(q, hi);
          if (j < a.N) atomicAdd(D + (size_t)j * a.ldd + i, acc[x][y][q]);
        }
    }
    return;
  }
  if (a.epi == EPI_F32 || a.epi == EPI_F32_ATOMIC) {   // logits / plain split-K: row-per-lane fp32
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
          if (a.epi == EPI_F32) *reinterpret_cast<f32x4*>(d) = v;
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(d + e, v[e]);
          }
        }
    }
    return;
  }
  // bf16 outputs: registers -> wave-private [64][64] LDS patch (16-byte chunks XOR-swizzled by row)
  // -> 128-byte row segments.  A row-per-lane accumulator stored directly is 64 scattered 8-byte
  // writes per instruction and store-issue bound.
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int il = x * 32 + r;
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int jl = y * 32 + 8 * g + 4 * hi;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[x][y][4 * g + e];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bv[y][g][e];
        if (a.epi == EPI_BF16_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(patch + il * 64 + (((jl >> 3) ^ (il & 7)) << 3) + (jl & 7)) = o;
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int id = p * 64 + l, rr = id >> 3, c = id & 7;
    const int i = ib + rr, j = jb + c * 8;
    bf16x8 v = *reinterpret_cast<const bf16x8*>(patch + rr * 64 + ((c ^ (rr & 7)) << 3));
    if (a.epi == EPI_BF16_MASK) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ((float)auxv[p][e] > 0.f) ? v[e] : (bf16)0.f;
    } else if (a.epi == EPI_BF16_ADD) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] + (float)auxv[p][e]);
    }
    if (i < a.M && j < a.N && !DBG(1)) *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(a.D) + (size_t)i * a.ldd + j) = v;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // patch reads retired before the next tile rewrites it
}

template <bool XT, bool YT>
__device__ __forceinline__ void gemm_consumer(const GemmArgs& a, const bf16* ring, bf16* patch, int wave) {
  constexpr int XE = tile_elems<128>(), SLOT = 2 * XE;
  const int wm = wave >> 1, wn = wave & 1;
  Work cur;
  work_init(cur, a);
  f32x16 acc[2][2];
  bf16x8 auxv[8];
  f32x4 bv[2][4];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = zero16();
#pragma unroll
  for (int p = 0; p < 8; ++p) auxv[p] = zero_bf8();
  const bool use_aux = a.epi == EPI_BF16_MASK || a.epi == EPI_BF16_ADD;
  int f = 0;
#ifdef ST_PROF
  unsigned long long p_wait = 0, p_mma = 0, p_epi = 0, p_tiles = 0, p_pre = 0, p_start = PROF_NOW();
#endif
  while (cur.valid) {
#ifdef ST_PROF
    const unsigned long long t0 = PROF_NOW();
#endif
    if (cur.kt == 0) {
      // bias vectors of this wave's 64 columns: requested at the first k-step, branch-free and all in
      // flight together (in the epilogue they cost a full loaded-memory round trip, ~4k cycles per tile)
      const int hi = (threadIdx.x & 63) >> 5;
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int j = cur.j0 + wn * 64 + y * 32 + 8 * g + 4 * hi;
          bv[y][g] = *reinterpret_cast<const f32x4*>((a.bias != nullptr && j < a.N) ? a.bias + j : g_zero_f32);
        }
    }
    if (use_aux && cur.kt == 0) {
      // the mask / addend block of this wave's sub-tile: 8 coalesced 16-byte chunks per lane, requested
      // at the first k-step so that their latency hides under the tile's MFMAs
      const int l = threadIdx.x & 63;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int id = p * 64 + l, i = cur.i0 + wm * 64 + (id >> 3), j = cur.j0 + wn * 64 + (id & 7) * 8;
        auxv[p] = gload8(a.aux + (size_t)i * a.ldaux + j, i < a.M && j < a.N);
      }
    }
#ifdef ST_PROF
    const unsigned long long t0b = PROF_NOW();
#endif
    __builtin_amdgcn_s_barrier();   // item f is in its slot
#ifdef ST_PROF
    const unsigned long long t1 = PROF_NOW();
    p_pre += t0b - t0;
#endif
    const bf16* xs = ring + (f & 1) * SLOT;
    const bf16* ys = xs + XE;
    // All 16 fragment reads of the item are issued up front into their own registers; the MFMAs then
    // start as the first fragments arrive.  (With one consumer wave per SIMD nothing else hides an
    // LDS round trip: the compiler's read -> lgkmcnt(0) -> 2 MFMAs -> read ... schedule on 3 recycled
    // fragment registers ran the matrix pipe at ~30 %.)
    bf16x8 xf[BK / 16][2], yf[BK / 16][2];
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        xf[kk][t] = read_frag<128, XT>(xs, (wm * 2 + t) * 32, kk);
        yf[kk][t] = read_frag<128, YT>(ys, (wn * 2 + t) * 32, kk);
      }
    __builtin_amdgcn_sched_barrier(0);   // keep the reads above the MFMAs (the scheduler would sink them back)
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = mfma32(yf[kk][y], xf[kk][x], acc[x][y]);
#ifdef ST_PROF
    asm volatile("s_nop 0" : "+v"(acc[1][1]));
    const unsigned long long t2 = PROF_NOW();
    p_wait += t1 - t0b; p_mma += t2 - t1;
#endif
    if (cur.kt == cur.nk - 1) {
      gemm_epilogue(a, acc, auxv, bv, cur.i0, cur.j0, patch, wm, wn);
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = zero16();
#ifdef ST_PROF
      p_epi += PROF_NOW() - t2; ++p_tiles;
#endif
    }
    ++f;
    work_next(cur, a);
  }
#ifdef ST_PROF
  if (wave == 0 && (threadIdx.x & 63) == 0) {
    unsigned long long* o = g_prof + blockIdx.x * 8;
    o[0] = p_wait; o[1] = p_mma; o[2] = p_epi; o[3] = PROF_NOW() - p_start; o[4] = p_tiles; o[5] = f; o[6] = p_pre;
  }
#endif
}

template <bool XT, bool YT>
__global__ __launch_bounds__(512) void gemm_kernel(GemmArgs a) {
  constexpr int SLOT = 2 * tile_elems<128>();
  __shared__ __attribute__((aligned(1024))) bf16 smem[2 * SLOT + 4 * 4096];   // 2 x 32 KiB tiles + 4 x 8 KiB patches
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= 4) gemm_loader<XT, YT>(a, smem, wave - 4);
  else gemm_consumer<XT, YT>(a, smem, smem + 2 * SLOT + wave * 4096, wave);
}

// ---- GEMM + bias (+ReLU) (+residual) + LayerNorm (+positional-encoding add) -------------------
// out = LN(act(x W^T + b) + res) * gamma + beta (+ pe[pos[i]])      N = d_model, full rows per workgroup
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
  int tiles;                  // row tiles
};

// Geometry: consumers are WM x WN waves, each 32 rows x 128 columns (4 MFMA tiles); BM = 32 * WM rows
// per workgroup tile, all N columns.  N = 128: 4 x 1, N = 256: 2 x 2, N = 512: 1 x 4.
template <int N> struct LnGeo {
  static constexpr int WN = N / 128, WM = 4 / WN, BM = 32 * WM;
  static constexpr int XE = BM * BK, YE = N * BK, SLOT = XE + YE;
  static constexpr int IPW = BM / 32 + N / 32;   // LDS-DMA instructions per loader wave per item
};

// Store a wave's [32 rows][128 cols] block of values (row-per-lane registers, column of (b, g, e) =
// b*32 + 8g + 4hi + e) through its private LDS patch as 256-byte row segments.
template <typename F>
__device__ __forceinline__ void ln_store_block(bf16* patch, bf16* gbase, int ld, int nvalid_rows, F value) {
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int jl = b * 32 + 8 * g + 4 * hi;
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)value(b, 4 * g + e);
      *reinterpret_cast<bf16x4*>(patch + r * 128 + (((jl >> 3) ^ (r & 15)) << 3) + (jl & 7)) = o;
      if (g == 3) __builtin_amdgcn_sched_barrier(0);   // keep the per-column vector loads of one 32-column block together
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int id = p * 64 + l, rr = id >> 4, c = id & 15;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(patch + rr * 128 + ((c ^ (rr & 15)) << 3));
    if (rr < nvalid_rows) *reinterpret_cast<bf16x8*>(gbase + (size_t)rr * ld + c * 8) = v;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// The row-wise epilogue shared by the persistent and the simple kernel.  `acc` holds x W^T for rows
// i_base + (lane & 31), columns wn*128 + ...; `resv` the 8 residual chunks this lane prefetched
// (chunk id = p*64 + lane -> row id >> 4, 16-byte column chunk id & 15 of the wave's block).
template <int N, typename SyncFn>
__device__ __forceinline__ void ln_epilogue(const GemmLnArgs& a, f32x16 (&acc)[4], bool have_res, int i_base, int wm,
                                            int wn, bf16* patch, float* red, SyncFn sync) {
  constexpr int WN = LnGeo<N>::WN, WM = LnGeo<N>::WM;
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
  const int i = i_base + r;
  const bool row_ok = i < a.M;
  const int nvalid = min(32, a.M - i_base);
  float sum = 0.f;   // (the residual block already sits in the patch: ln_stage_res)
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int jl = b * 32 + 8 * g + 4 * hi, j = wn * 128 + jl;
      const f32x4 bb = *reinterpret_cast<const f32x4*>(a.bias + j);
      bf16x4 rr = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
      if (have_res) rr = *reinterpret_cast<const bf16x4*>(patch + r * 128 + (((jl >> 3) ^ (r & 15)) << 3) + (jl & 7));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[b][4 * g + e] + bb[e];
        if (a.relu) v = fmaxf(v, 0.f);
        v += (float)rr[e];
        acc[b][4 * g + e] = v;
        sum += v;
      }
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  sum += wave_xor32(sum);
  if (WN > 1) {
    if (hi == 0) red[(wm * WN + wn) * 32 + r] = sum;
    sync();
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
    sync();
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
    ln_store_block(patch, a.pre + (size_t)i_base * N + wn * 128, N, nvalid, [&](int b, int q) { return acc[b][q]; });
  if (a.xhat)
    ln_store_block(patch, a.xhat + (size_t)i_base * N + wn * 128, N, nvalid,
                   [&](int b, int q) { return (acc[b][q] - mean) * rstd; });
  ln_store_block(patch, a.out + (size_t)i_base * a.ldo + wn * 128, a.ldo, nvalid, [&](int b, int q) {
    const int j4 = b * 32 + 8 * (q >> 2) + 4 * hi;   // the four e = q & 3 of one (b, g) share these vector loads
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(gm + j4);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bt + j4);
    f32x4 p4 = {0.f, 0.f, 0.f, 0.f};
    if (perow) p4 = *reinterpret_cast<const f32x4*>(perow + j4);
    return (acc[b][q] - mean) * rstd * g4[q & 3] + b4[q & 3] + p4[q & 3];
  });
}

// Park the prefetched residual chunks in the wave's patch (row-per-lane reads in ln_epilogue).
__device__ __forceinline__ void ln_stage_res(bf16* patch, const bf16x8 (&resv)[8]) {
  const int l = threadIdx.x & 63;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int id = p * 64 + l, rr = id >> 4, c = id & 15;
    *reinterpret_cast<bf16x8*>(patch + rr * 128 + ((c ^ (rr & 15)) << 3)) = resv[p];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// Prefetch this lane's 8 residual chunks of the wave's [32][128] block (issued at the first k-step
// of a tile so their latency hides under the tile's MFMAs).
template <int N>
__device__ __forceinline__ void ln_prefetch_res(const GemmLnArgs& a, bf16x8 (&resv)[8], int i_base, int wn) {
  const int l = threadIdx.x & 63;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int id = p * 64 + l, rr = id >> 4, c = id & 15;
    resv[p] = gload8(a.res + (size_t)(i_base + rr) * a.ldres + wn * 128 + c * 8, i_base + rr < a.M);
  }
}

// Persistent, wave-specialised version (N = 128, 256): 3-slot ring.
template <int N>
__global__ __launch_bounds__(512) void gemm_ln_kernel(GemmLnArgs a) {
  using G = LnGeo<N>;
  constexpr int RL = 3;
  __shared__ __attribute__((aligned(1024))) bf16 smem[RL * G::SLOT + 4 * 4096 + 1024];
  bf16* patches = smem + RL * G::SLOT;
  float* red = reinterpret_cast<float*>(patches + 4 * 4096);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nk = (a.K + BK - 1) / BK;
  const int my_tiles = (a.tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int items = my_tiles * nk;

  if (wave >= 4) {   // ---- loader ----
    const int lw = wave - 4;
    int issued = 0;
    auto issue = [&]() {
      const int tile = blockIdx.x + (issued / nk) * gridDim.x, kt = issued % nk;
      bf16* slot = smem + (issued % RL) * G::SLOT;
      issue_tile<G::BM, false>(slot, a.X, a.ldx, tile * G::BM, a.M, kt * BK, a.K, lw);
      issue_tile<N, false>(slot + G::XE, a.W, a.K, 0, N, kt * BK, a.K, lw);
      ++issued;
    };
#pragma unroll
    for (int p = 0; p < RL - 1; ++p)
      if (issued < items) issue();
    for (int done = 0; done < items; ++done) {
      wait_items<G::IPW>(min(issued - done - 1, 1));
      __builtin_amdgcn_s_barrier();
      if (issued < items) issue();
      if (G::WN > 1 && (done % nk) == nk - 1) {   // mirror the consumers' two statistics barriers
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
      }
    }
    return;
  }
  // ---- consumer ----
  const int wm = wave / G::WN, wn = wave % G::WN;
  bf16* patch = patches + wave * 4096;
  f32x16 acc[4];
  bf16x8 resv[8];
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = zero16();
  for (int f = 0; f < items; ++f) {
    const int tile = blockIdx.x + (f / nk) * gridDim.x, kt = f % nk;
    const int i_base = tile * G::BM + wm * 32;
    if (kt == 0 && a.res) ln_prefetch_res<N>(a, resv, i_base, wn);
    if (kt == nk - 1 && a.res) ln_stage_res(patch, resv);   // registers free again before the epilogue
    __builtin_amdgcn_s_barrier();
    const bf16* xs = smem + (f % RL) * G::SLOT;
    const bf16* ys = xs + G::XE;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {   // two half-items: 10 fragment reads in flight, then 8 MFMAs
      bf16x8 xf[2], yf[2][4];
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        xf[k2] = read_frag<G::BM, false>(xs, wm * 32, h2 * 2 + k2);
#pragma unroll
        for (int b = 0; b < 4; ++b) yf[k2][b] = read_frag<N, false>(ys, wn * 128 + b * 32, h2 * 2 + k2);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = mfma32(yf[k2][b], xf[k2], acc[b]);
    }
    if (kt == nk - 1) {
      ln_epilogue<N>(a, acc, a.res != nullptr, i_base, wm, wn, patch, red, [] { __builtin_amdgcn_s_barrier(); });
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = zero16();
    }
  }
}

// Simple version (N = 512: the ring of the persistent kernel does not fit next to a 64 KiB weight tile):
// one 32-row tile per 256-thread workgroup, double-buffered LDS-DMA.
template <int N>
__global__ __launch_bounds__(256) void gemm_ln_simple_kernel(GemmLnArgs a) {
  using G = LnGeo<N>;
  __shared__ __attribute__((aligned(1024))) bf16 smem[2 * G::SLOT + 1024];
  bf16* patches = smem;   // the output patches alias the ring (used after the k-loop only)
  float* red = reinterpret_cast<float*>(smem + 2 * G::SLOT);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / G::WN, wn = wave % G::WN;
  const int i0 = blockIdx.x * G::BM, i_base = i0 + wm * 32;
  f32x16 acc[4];
  bf16x8 resv[8];
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = zero16();
  if (a.res) {
    ln_prefetch_res<N>(a, resv, i_base, wn);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int p = 0; p < 8; ++p) touch(resv[p]);
  }
  const int nk = (a.K + BK - 1) / BK;
  issue_tile<G::BM, false>(smem, a.X, a.ldx, i0, a.M, 0, a.K, wave);
  issue_tile<N, false>(smem + G::XE, a.W, a.K, 0, N, 0, a.K, wave);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    wait_vmcnt<0>();
    __syncthreads();
    if (kt + 1 < nk) {
      bf16* nxt = smem + (cur ^ 1) * G::SLOT;
      issue_tile<G::BM, false>(nxt, a.X, a.ldx, i0, a.M, (kt + 1) * BK, a.K, wave);
      issue_tile<N, false>(nxt + G::XE, a.W, a.K, 0, N, (kt + 1) * BK, a.K, wave);
    }
    const bf16* xs = smem + cur * G::SLOT;
    const bf16* ys = xs + G::XE;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {   // two half-items: 10 fragment reads in flight, then 8 MFMAs
      bf16x8 xf[2], yf[2][4];
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        xf[k2] = read_frag<G::BM, false>(xs, wm * 32, h2 * 2 + k2);
#pragma unroll
        for (int b = 0; b < 4; ++b) yf[k2][b] = read_frag<N, false>(ys, wn * 128 + b * 32, h2 * 2 + k2);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = mfma32(yf[k2][b], xf[k2], acc[b]);
    }
  }
  __syncthreads();   // every wave is done with the ring before the patches overwrite it
  if (a.res) ln_stage_res(patches + wave * 4096, resv);
  ln_epilogue<N>(a, acc, a.res != nullptr, i_base, wm, wn, patches + wave * 4096, red, [] { __syncthreads(); });
}

}  // namespace

#ifdef ST_PROF
extern "C" int st_prof_dbg(int v) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &v, sizeof(int)); }
extern "C" int st_prof_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 256 * 8);
}
#endif

// Development reference only (see st_gemm_sym.hip for the production st_gemm): the persistent,
// wave-specialised variant, kept for A/B measurements with tools/prof_gemm.py.
extern "C" int st_gemm_ws(hipStream_t stream, int x_cmajor, int y_cmajor, const void* X, int ldx, const void* Y, int ldy,
                       void* D, int ldd, int M, int N, int Kc, const float* bias, const void* aux, int ldaux, int epi,
                       int splits) {
  if (M <= 0 || N <= 0 || Kc <= 0) return 0;
  if ((ldx & 7) || (ldy & 7) || (N & 3) || epi < 0 || epi > 6) return -1;
  if (x_cmajor && !y_cmajor) return -2;  // not needed by any caller
  // contraction-major operands are read in 8-row chunks: the caller guarantees the buffer is
  // padded (ld >= round_up(rows, 8)); rows beyond M / N only feed outputs that are never stored.
  if ((x_cmajor && ldx < ((M + 7) & ~7)) || (y_cmajor && ldy < ((N + 7) & ~7))) return -3;
  if (epi != EPI_F32 && epi != EPI_F32_ATOMIC && epi != EPI_F32_ATOMIC_T && ((ldd & 7) || (N & 7))) return -4;
  if ((epi == EPI_BF16_MASK || epi == EPI_BF16_ADD) && (ldaux & 7)) return -5;
  if (splits < 1) splits = 1;
  if (epi != EPI_F32_ATOMIC && epi != EPI_F32_ATOMIC_T) splits = 1;
  GemmArgs a;
  a.X = (const bf16*)X; a.ldx = ldx; a.Y = (const bf16*)Y; a.ldy = ldy; a.D = D; a.ldd = ldd;
  a.M = M; a.N = N; a.Kc = Kc; a.bias = bias; a.aux = (const bf16*)aux; a.ldaux = ldaux; a.epi = epi;
  int per = (Kc + splits - 1) / splits;
  per = (per + BK - 1) / BK * BK;
  splits = (Kc + per - 1) / per;
  a.c_per_split = per;
  a.tiles_i = (M + 127) / 128; a.tiles_j = (N + 127) / 128; a.splits = splits;
  const int total = a.tiles_i * a.tiles_j * splits;
  dim3 grid(total < 256 ? ((total + 7) & ~7) : 256), block(512);   // one persistent workgroup per CU, 8 | grid
  if (!x_cmajor && !y_cmajor) hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, 0, stream, a);
  else if (!x_cmajor && y_cmajor) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, 0, stream, a);
  ST_CHECK_LAUNCH();
  return 0;
}

extern "C" int st_gemm_ln(hipStream_t stream, const void* X, int ldx, const void* W, int M, int N, int K,
                          const float* bias, const void* res, int ldres, const float* gamma, const float* beta,
                          float eps, int relu, const float* pe, const int* pos, void* out, int ldo, void* xhat,
                          float* rstd, void* pre) {
  if (M <= 0) return 0;
  if ((ldx & 7) || (K & 7) || (ldo & 7) || (res && (ldres & 7)) || !bias || !gamma || !beta || !out) return -1;
  if (pe && !pos) return -2;
  GemmLnArgs a;
  a.X = (const bf16*)X; a.ldx = ldx; a.W = (const bf16*)W; a.M = M; a.K = K; a.bias = bias;
  a.res = (const bf16*)res; a.ldres = ldres; a.gamma = gamma; a.beta = beta; a.eps = eps; a.relu = relu;
  a.pe = pe; a.pos = pos; a.out = (bf16*)out; a.ldo = ldo; a.xhat = (bf16*)xhat; a.rstd = rstd; a.pre = (bf16*)pre;
  if (N == 128) {
    a.tiles = (M + 127) / 128;
    hipLaunchKernelGGL((gemm_ln_kernel<128>), dim3(a.tiles < 256 ? a.tiles : 256), dim3(512), 0, stream, a);
  } else if (N == 256) {
    a.tiles = (M + 63) / 64;
    hipLaunchKernelGGL((gemm_ln_kernel<256>), dim3(a.tiles < 256 ? a.tiles : 256), dim3(512), 0, stream, a);
  } else if (N == 512) {
    a.tiles = (M + 31) / 32;
    hipLaunchKernelGGL((gemm_ln_simple_kernel<512>), dim3(a.tiles), dim3(256), 0, stream, a);
  } else {
    return -3;
  }
  ST_CHECK_LAUNCH();
  return 0;
}

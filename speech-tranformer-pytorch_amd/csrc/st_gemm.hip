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
//   wgrad    dW = dy^T x     : X = dy [M,N] contr.-major, Y = x  [M,K] contraction-major (c = m)
// so no transposed copy of a weight or an activation is ever written to HBM.
//
// Tile: 128 x 128 x 32, 256 threads = 2 x 2 waves, each wave 64 x 64 = 2 x 2
// v_mfma_f32_32x32x16_bf16 tiles.  The accumulator is kept TRANSPOSED (the
// X-row index i is the lane, the Y-row index j runs over registers) so the
// epilogue owns whole output rows per lane: 8-byte packed bf16 row stores and
// lane-local LayerNorm statistics (st_gemm_ln).
#include "st_common.cuh"

namespace {

constexpr int BK = 64;  // contraction elements per LDS tile = one 128-byte row of a natural tile

#define ST_AS1 __attribute__((address_space(1)))

__device__ __attribute__((aligned(16))) bf16 g_zero_chunk[8];  // zero-initialised source for padding lanes

enum Epi { EPI_BF16 = 0, EPI_BF16_RELU = 1, EPI_F32 = 2, EPI_BF16_MASK = 3, EPI_BF16_ADD = 4, EPI_F32_ATOMIC = 5,
           EPI_F32_ATOMIC_T = 6 };

// ---- operand tiles: HBM -> LDS by LDS-DMA (global_load_lds_dwordx4), no VGPR staging --------------
// One wave-instruction moves 64 x 16 bytes to wave_base + 16 * lane, so the LDS image is lane-linear
// and the bank-conflict swizzle lives on the per-lane SOURCE address; fragment reads undo it with the
// same XOR (an involution):
//   natural  [ROWS][64] : 16-byte chunk p of row r   holds logical chunk p ^ ((r >> 1) & 7)
//                         -> any 16 consecutive rows read conflict-free with ds_read_b128
//   c-major  [64][ROWS] : chunk p of contraction row c holds logical chunk p ^ ((c & 3) << 2)
//                         -> the 4 c-rows of a ds_read_b64_tr_b16 land in 4 different 64-byte bank groups
// Lanes outside the matrix (row >= nrows or c >= c_end) read a 16-byte zero chunk instead.
template <int ROWS>
__host__ __device__ constexpr int tile_elems() { return ROWS * BK; }

template <int ROWS, bool CM>
__device__ __forceinline__ void issue_tile(bf16* tile, const bf16* __restrict__ base, int ld, int row0, int nrows,
                                           int c0, int c_end) {
  static_assert(ROWS % 32 == 0 && (!CM || ROWS == 128), "tile shape");
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t < ROWS / 32; ++t) {
    const int I = t * 4 + w;  // wave-instruction index: 1 KiB of the tile each
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
    __builtin_amdgcn_global_load_lds((const ST_AS1 void*)src, (ST_LDS void*)(tile + I * 512), 16, 0, 0);
  }
}

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

template <bool XT, bool YT>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
  constexpr int BM = 128, BN = 128;
  constexpr int XE = tile_elems<BM>(), YE = tile_elems<BN>();
  __shared__ __attribute__((aligned(1024))) bf16 smem[2 * (XE + YE)];
  auto xs = [&](int buf) { return smem + buf * (XE + YE); };
  auto ys = [&](int buf) { return smem + buf * (XE + YE) + XE; };

  // 1-D grid; workgroup b runs on XCD b % 8 (each XCD has its own L2).  Forward / dgrad: the i-tile
  // is the fastest index, so one XCD sees a fixed subset of X row-tiles for every j (X streams from
  // HBM once and is re-read from that XCD's L2; the small weight operand is shared by all XCDs).
  // Weight gradients: the split index is fastest, so the (i, j) tiles of one contraction range meet
  // in the same L2.
  int bid = blockIdx.x, ti, tj, ts;
  if (a.splits > 1) { ts = bid % a.splits; bid /= a.splits; ti = bid % a.tiles_i; tj = bid / a.tiles_i; }
  else { ts = 0; ti = bid % a.tiles_i; tj = bid / a.tiles_i; }
  const int j0 = tj * BN, i0 = ti * BM;
  const int c_begin = ts * a.c_per_split;
  const int c_end = min(a.Kc, c_begin + a.c_per_split);
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, hi = l >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = zero16();

  const int nk = (c_end - c_begin + BK - 1) / BK;
  if (nk > 0) {
    issue_tile<BM, XT>(xs(0), a.X, a.ldx, i0, a.M, c_begin, c_end);
    issue_tile<BN, YT>(ys(0), a.Y, a.ldy, j0, a.N, c_begin, c_end);
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    // The compiler drains this wave's LDS-DMA (vmcnt(0)) in front of the barrier: past it tile kt is
    // visible to every wave and buffer cur^1 (read during iteration kt-1) is free again.
    __syncthreads();
    if (kt + 1 < nk) {
      issue_tile<BM, XT>(xs(cur ^ 1), a.X, a.ldx, i0, a.M, c_begin + (kt + 1) * BK, c_end);
      issue_tile<BN, YT>(ys(cur ^ 1), a.Y, a.ldy, j0, a.N, c_begin + (kt + 1) * BK, c_end);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 xf[2], yf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        xf[t] = read_frag<BM, XT>(xs(cur), (wm * 2 + t) * 32, kk);
        yf[t] = read_frag<BN, YT>(ys(cur), (wn * 2 + t) * 32, kk);
      }
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = mfma32(yf[y], xf[x], acc[x][y]);
    }
  }

  // ---- weight-gradient epilogue: D^T[j][i] += acc.  The lane index i is the CONTIGUOUS axis of
  // the output, so every atomic instruction covers 2 x 128 contiguous bytes (coalesced L2 atomics).
  if (a.epi == EPI_F32_ATOMIC_T) {
    float* D = reinterpret_cast<float*>(a.D);
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int i = i0 + (wm * 2 + x) * 32 + (l & 31);
      if (i >= a.M) continue;
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = j0 + (wn * 2 + y) * 32 + acc_row(r, hi);
          if (j < a.N) atomicAdd(D + (size_t)j * a.ldd + i, acc[x][y][r]);
        }
    }
    return;
  }

  // ---- fp32 epilogues (logits, plain split-K): lane owns row i, registers run over j --------------
  if (a.epi == EPI_F32 || a.epi == EPI_F32_ATOMIC) {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int i = i0 + (wm * 2 + x) * 32 + (l & 31);
      if (i >= a.M) continue;
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int j = j0 + (wn * 2 + y) * 32 + 8 * g + 4 * hi;
          if (j >= a.N) continue;
          f32x4 v = {acc[x][y][4 * g], acc[x][y][4 * g + 1], acc[x][y][4 * g + 2], acc[x][y][4 * g + 3]};
          if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + j);
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

  // ---- bf16 epilogues: the tile goes through LDS so that HBM sees whole 256-byte row segments ----
  // (a row-per-lane accumulator stored directly is 64 scattered 8-byte writes per instruction and
  // store-issue bound).  Row stride 136 elements: the 8-byte accumulator writes are <= 2-way
  // conflicted, the 16-byte read-back is conflict-free.
  constexpr int CS = BN + 8;
  bf16* ct = smem;
  __syncthreads();   // every wave is done with the operand tiles
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int il = (wm * 2 + x) * 32 + (l & 31);
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int jl = (wn * 2 + y) * 32 + 8 * g + 4 * hi;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[x][y][4 * g + e];
        if (a.bias && j0 + jl < a.N) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(a.bias + j0 + jl);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bb[e];
        }
        if (a.epi == EPI_BF16_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(ct + il * CS + jl) = o;
      }
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < BM * BN / 8 / 256; ++p) {
    const int id = p * 256 + threadIdx.x;
    const int il = id / (BN / 8), jl = (id % (BN / 8)) * 8;
    const int i = i0 + il, j = j0 + jl;
    if (i >= a.M || j >= a.N) continue;
    bf16x8 v = *reinterpret_cast<const bf16x8*>(ct + il * CS + jl);
    if (a.epi == EPI_BF16_MASK) {
      const bf16x8 m = *reinterpret_cast<const bf16x8*>(a.aux + (size_t)i * a.ldaux + j);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ((float)m[e] > 0.f) ? v[e] : (bf16)0.f;
    } else if (a.epi == EPI_BF16_ADD) {
      const bf16x8 m = *reinterpret_cast<const bf16x8*>(a.aux + (size_t)i * a.ldaux + j);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] + (float)m[e]);
    }
    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(a.D) + (size_t)i * a.ldd + j) = v;
  }
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
};

template <int N>
__global__ __launch_bounds__(256) void gemm_ln_kernel(GemmLnArgs a) {
  constexpr int WN = N / 128, WM = 4 / WN, BM = 32 * WM;
  static_assert(N == 128 || N == 256 || N == 512, "d_model must be 128, 256 or 512");
  constexpr int XE = tile_elems<BM>(), YE = tile_elems<N>();
  __shared__ __attribute__((aligned(1024))) bf16 smem[2 * (XE + YE)];
  // cross-wave row statistics alias the tile buffers (used only after the k-loop; keeps N = 256 at
  // 80 KiB of LDS = two workgroups per CU)
  float (*red)[WM][WN][32] = reinterpret_cast<float (*)[WM][WN][32]>(smem);
  auto xs = [&](int buf) { return smem + buf * (XE + YE); };
  auto ys = [&](int buf) { return smem + buf * (XE + YE) + XE; };

  const int i0 = blockIdx.x * BM;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, hi = l >> 5;
  const int wm = wave / WN, wn = wave % WN;

  f32x16 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = zero16();

  const int nk = (a.K + BK - 1) / BK;
  issue_tile<BM, false>(xs(0), a.X, a.ldx, i0, a.M, 0, a.K);
  issue_tile<N, false>(ys(0), a.W, a.K, 0, N, 0, a.K);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    __syncthreads();   // LDS-DMA of tile kt drained + visible; buffer cur^1 free (see gemm_kernel)
    if (kt + 1 < nk) {
      issue_tile<BM, false>(xs(cur ^ 1), a.X, a.ldx, i0, a.M, (kt + 1) * BK, a.K);
      issue_tile<N, false>(ys(cur ^ 1), a.W, a.K, 0, N, (kt + 1) * BK, a.K);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const bf16x8 xf = read_frag<BM, false>(xs(cur), wm * 32, kk);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const bf16x8 yf = read_frag<N, false>(ys(cur), wn * 128 + b * 32, kk);
        acc[b] = mfma32(yf, xf, acc[b]);
      }
    }
  }

  if (WN > 1) __syncthreads();   // every wave is done reading the tiles before `red` overwrites them
  const int i = i0 + wm * 32 + (l & 31);
  const bool row_ok = i < a.M;
  // v = act(acc + bias) + residual ; column of (b, r): wn*128 + b*32 + acc_row(r, hi)
  float sum = 0.f;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int j = wn * 128 + b * 32 + 8 * g + 4 * hi;
      const f32x4 bb = *reinterpret_cast<const f32x4*>(a.bias + j);
      bf16x4 rr = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
      if (a.res && row_ok) rr = *reinterpret_cast<const bf16x4*>(a.res + (size_t)i * a.ldres + j);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[b][4 * g + e] + bb[e];
        if (a.relu) v = fmaxf(v, 0.f);
        v += (float)rr[e];
        acc[b][4 * g + e] = v;
        sum += v;
      }
    }
  }
  sum += wave_xor32(sum);
  if (WN > 1) {
    if (hi == 0) red[0][wm][wn][l & 31] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int w = 0; w < WN; ++w) sum += red[0][wm][w][l & 31];
  }
  const float mean = sum * (1.f / N);
  float sq = 0.f;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = acc[b][r] - mean;
      sq += d * d;
    }
  sq += wave_xor32(sq);
  if (WN > 1) {
    if (hi == 0) red[1][wm][wn][l & 31] = sq;
    __syncthreads();
    sq = 0.f;
#pragma unroll
    for (int w = 0; w < WN; ++w) sq += red[1][wm][w][l & 31];
  }
  const float rstd = rsqrtf(sq * (1.f / N) + a.eps);
  if (!row_ok) return;
  if (a.rstd && wn == 0 && hi == 0) a.rstd[i] = rstd;
  const float* perow = (a.pe != nullptr) ? a.pe + (size_t)a.pos[i] * N : nullptr;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int j = wn * 128 + b * 32 + 8 * g + 4 * hi;
      const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + j);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(a.beta + j);
      f32x4 pe = {0.f, 0.f, 0.f, 0.f};
      if (perow) pe = *reinterpret_cast<const f32x4*>(perow + j);
      bf16x4 xh, yo, pr;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = acc[b][4 * g + e];
        const float h = (v - mean) * rstd;
        xh[e] = (bf16)h;
        yo[e] = (bf16)(h * gm[e] + bt[e] + pe[e]);
        pr[e] = (bf16)v;
      }
      *reinterpret_cast<bf16x4*>(a.out + (size_t)i * a.ldo + j) = yo;
      if (a.xhat) *reinterpret_cast<bf16x4*>(a.xhat + (size_t)i * N + j) = xh;
      if (a.pre) *reinterpret_cast<bf16x4*>(a.pre + (size_t)i * N + j) = pr;
    }
  }
}

}  // namespace

extern "C" int st_gemm(hipStream_t stream, int x_cmajor, int y_cmajor, const void* X, int ldx, const void* Y, int ldy,
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
  dim3 grid(a.tiles_i * a.tiles_j * splits), block(256);
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
  if ((ldx & 7) || (K & 7) || !bias || !gamma || !beta || !out) return -1;
  if (pe && !pos) return -2;
  GemmLnArgs a;
  a.X = (const bf16*)X; a.ldx = ldx; a.W = (const bf16*)W; a.M = M; a.K = K; a.bias = bias;
  a.res = (const bf16*)res; a.ldres = ldres; a.gamma = gamma; a.beta = beta; a.eps = eps; a.relu = relu;
  a.pe = pe; a.pos = pos; a.out = (bf16*)out; a.ldo = ldo; a.xhat = (bf16*)xhat; a.rstd = rstd; a.pre = (bf16*)pre;
  dim3 block(256);
  if (N == 128) hipLaunchKernelGGL((gemm_ln_kernel<128>), dim3((M + 127) / 128), block, 0, stream, a);
  else if (N == 256) hipLaunchKernelGGL((gemm_ln_kernel<256>), dim3((M + 63) / 64), block, 0, stream, a);
  else if (N == 512) hipLaunchKernelGGL((gemm_ln_kernel<512>), dim3((M + 31) / 32), block, 0, stream, a);
  else return -3;
  ST_CHECK_LAUNCH();
  return 0;
}

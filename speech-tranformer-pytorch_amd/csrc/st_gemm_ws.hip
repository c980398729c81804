// st_gemm_ws: weight-stationary streaming GEMM for the short-contraction products of the encoder (gfx950).
//
//   D[M, N] = epi( X[M, K] W[N, K]^T + bias ),     K = 256 (d_model),  N a multiple of 256,  M large (all frames)
//
// Reference lines replaced: the q/k/v projections (transformer/Attention.py:74-76), fc1 + ReLU
// (transformer/SubLayers.py:25), the stacked key/value projections of the decoder-encoder attentions - every
// nn.Linear whose input width is d_model.
//
// Why another GEMM: these products are tall and skinny (24,060 x {768, 1024, 3072} x 256).  A 128 x 128 output tile
// (st_gemm_sym.hip) re-loads a 64 KB slab of each operand for 0.86 us of matrix-core work and pays a prologue and an
// epilogue per tile.  Here the WEIGHT never moves: a workgroup (4 waves) owns 256 output columns, each wave keeps the
// 64 x 256 weight block of its 64 columns in 128 VGPRs (2 x 16 MFMA A-fragments) for the whole kernel, and the
// activations stream past in 32-row chunks through a 3-slot LDS ring filled by LDS-DMA (global_load_lds_dwordx4,
// counted vmcnt): one barrier per chunk, no prologue / epilogue bubbles between output tiles, X is the only streamed
// operand and every X fragment read from LDS feeds two MFMAs.
//
// What measurement on the MI355X said about the first versions (8 waves x 32 columns in lock-step; s_memtime per
// phase, tools/dev/ws_bench.py): the phases of a step - barrier, LDS-DMA issue, fragment reads, 16 MFMAs, epilogue,
// stores - simply ADD UP (0.44 + 0.57 + 0.2 + ... us) because the two waves of a SIMD leave the same barrier together
// and run the same in-order instruction stream: while both issue loads or convert outputs the matrix pipe idles.
// Hence this shape: TWO independent 4-wave workgroups per CU (one wave each per SIMD) whose barriers are unrelated,
// so one workgroup's MFMA segment (32 MFMAs per wave and chunk = 1024 cycles of the SIMD's matrix pipe) overlaps the
// other's barrier / load / epilogue segment; and every wave loads, multiplies and stores (no loader / storer roles).
//
//   step i:  s_barrier | LDS-DMA(i + 2) | 16 ds_read_b128 + 32 MFMA (two accumulator chains) | vmcnt | epilogue + stores(i)
//
// vmcnt: stores share the counter with loads and may retire out of order with them, so the wait for chunk i+1 counts
// only the YOUNGER loads (chunk i+2's four) - i.e. it also waits for the stores of step i-1, which have had a whole
// step to be acknowledged by then.
//
// LDS images: chunk [row / 16][32 k-slices][row % 16] x 16 B - the 16 fragment reads of a lane are one base register
// plus immediates (k-step kk at + 512 kk) and each 16-lane group reads one full 256-byte bank row; LDS-DMA writes
// lane-linear, so the image is produced by the per-lane SOURCE addresses (cdna_hip_programming.md 5.4 rule 21).
// Epilogue: wave-private [32][64] patch (16-byte slots XOR-swizzled by row / 2) -> 128-byte row segments to HBM.
#include "st_common.cuh"

#ifndef ST_GEMM_SC1
#define ST_GEMM_SC1 0
#endif
namespace {

constexpr int WS_K = 256;                    // contraction length (held in registers per wave)
constexpr int WS_CH = 32;                    // rows per streamed chunk
constexpr int WS_NB = 3;                     // ring slots
constexpr int WS_SLOT = WS_CH * WS_K * 2;    // bytes per slot (16 KB)
constexpr int WS_PATCH = 32 * 64 * 2;        // wave-private epilogue patch (4 KB)
constexpr int WS_SMEM = WS_NB * WS_SLOT + 4 * WS_PATCH;   // 64 KB: two workgroups per CU

enum WsEpi { WS_BF16 = 0, WS_RELU = 1 };

struct WsArgs {
  const bf16* X; int ldx;
  const bf16* W; int ldw;          // [N, K] natural
  bf16* D; int ldd;
  int M, N;
  const float* bias;               // [N] or null
  int nslices, ranks, nchunks;     // N / 256, workgroups per slice, ceil(M / 32)
  // W (and the bias) may be a stack of equally spaced blocks (st_gemm_stacked): slice s lies in block
  // s >> wseg_shift (blocks of 2^wseg_shift slices), wseg_extra / bias_extra elements beyond the plain row stride
  int wseg_shift; long wseg_extra, bias_extra;
  DropArgs drop;                   // WS_RELU: dropout after the ReLU (SubLayers.py:25)
};

__device__ __attribute__((aligned(16))) float g_zero_ws[4];

template <int EPI, bool DROP>
__global__ __launch_bounds__(256, 2) void gemm_ws_kernel(WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
  // consecutive workgroups = the slices of one rank: they stream the same X rows at about the same time (the chunk
  // misses HBM once and is then served from the L2s / Infinity Cache)
  const int slice = blockIdx.x % a.nslices, rank = blockIdx.x / a.nslices;
  const int nch = (a.nchunks - rank + a.ranks - 1) / a.ranks;     // chunks rank, rank + ranks, ...
  if (nch <= 0) return;

  // ---- LDS-DMA: wave w fills LDS bytes [4096 w, 4096 (w + 1)) of every chunk image (four 1 KB instructions).
  // Image piece (row, c) - the 16 bytes of k-slice c of chunk row `row` - lives at ((row >> 4) * 32 + c) * 256 +
  // (row & 15) * 16; instruction q = 4 w + j covers pieces c = 4 (q & 7) + (lane >> 4), row = 16 (q >> 3) + (lane & 15).
  const unsigned lds0 = lds_addr(smem);
  unsigned voff[4];                                                // byte offset of this lane's source piece from the chunk's row 0
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = 4 * wave + j;
    voff[j] = (unsigned)((16 * (q >> 3) + (l & 15)) * a.ldx + (4 * (q & 7) + (l >> 4)) * 8) * 2u;
  }
  auto issue = [&](int ci) {
    const int g = rank + min(ci, nch - 1) * a.ranks;               // beyond the end: re-load the last chunk (keeps vmcnt counted)
    const unsigned dst = lds0 + (unsigned)(ci % WS_NB) * WS_SLOT + (unsigned)wave * 4096u;
    const char* base = reinterpret_cast<const char*>(a.X) + (size_t)g * WS_CH * a.ldx * 2;
    if ((g + 1) * WS_CH <= a.M) {          // (uniform) whole chunk inside the matrix
#pragma unroll
      for (int j = 0; j < 4; ++j) lds_dma16(base + voff[j], dst + 1024u * j);
    } else {                               // rows past the end are clamped onto the last row (never stored)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = 4 * wave + j, row = min(g * WS_CH + 16 * (q >> 3) + (l & 15), a.M - 1);
        lds_dma16(a.X + (size_t)row * a.ldx + (4 * (q & 7) + (l >> 4)) * 8, dst + 1024u * j);
      }
    }
  };
  issue(0);                            // slot 0 = [0, 16 KB); the weight staging below uses [16 KB, 48 KB)

  // ---- this wave's weight block (rows n0 .. n0+63 of W, all of K) -> 2 x 16 A-fragments in registers ------------
  // coalesced 16-byte loads (256 B per row and instruction) -> wave-private swizzled LDS image -> row fragments
  // (fragment-shaped loads straight from global touch 32 rows per instruction: 3.3 us of prologue, measured)
  const int n0 = slice * 256 + wave * 64;
  const long wblk = (long)(slice >> a.wseg_shift);
  const bf16* wbase = a.W + wblk * a.wseg_extra + (size_t)n0 * a.ldw;
  bf16x8 wfA[WS_K / 16], wfB[WS_K / 16];
  {
    // all 32 coalesced loads first (one HBM / L2 latency), then each group of 8 registers goes through the wave's
    // LDS image and comes back as fragments IN PLACE: group (blk, h) = rows 32 blk .., k half h
    char* stage = smem + WS_SLOT + wave * 8192;                    // [32 rows][256 B]
    const int srow = l >> 4, ss = l & 15;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int blk = pass >> 1, h = pass & 1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(wbase + (size_t)(32 * blk + 4 * j + srow) * a.ldw + h * 128 + ss * 8);
        if (blk == 0) wfA[h * 8 + j] = t; else wfB[h * 8 + j] = t;
      }
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int blk = pass >> 1, h = pass & 1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = 4 * j + srow;
        *reinterpret_cast<bf16x8*>(stage + row * 256 + ((ss ^ (row & 15)) << 4)) = blk == 0 ? wfA[h * 8 + j] : wfB[h * 8 + j];
      }
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const bf16x8 f = *reinterpret_cast<const bf16x8*>(stage + r * 256 + (((2 * kk + hi) ^ (r & 15)) << 4));
        if (blk == 0) wfA[h * 8 + kk] = f; else wfB[h * 8 + kk] = f;
      }
    }
  }
  // the accumulators' C inputs: bias of column n0 + 32 blk + acc_row(t, hi) in register t
  f32x16 biasA, biasB;
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const float* bp = a.bias ? a.bias + wblk * a.bias_extra + n0 + 8 * gq + 4 * hi : g_zero_ws;
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp);
    const f32x4 c4 = *reinterpret_cast<const f32x4*>(a.bias ? bp + 32 : g_zero_ws);
#pragma unroll
    for (int e = 0; e < 4; ++e) { biasA[4 * gq + e] = b4[e]; biasB[4 * gq + e] = c4[e]; }
  }
  // the compiler's own loads are retired HERE, so that it places no s_waitcnt vmcnt() of its own inside the loop (it
  // cannot see the LDS-DMAs and would drain them)
#pragma unroll
  for (int kk = 0; kk < WS_K / 16; ++kk) { touch(wfA[kk]); touch(wfB[kk]); }
  touch(biasA); touch(biasB);
  __syncthreads();                     // every wave has read its weight image: the staging area becomes ring slots 1, 2
  issue(1);

  // fragment reads of this lane (chunk row r, k-slice 2 kk + hi): rdbase + 512 kk
  const int rdbase = (r >> 4) * 8192 + (r & 15) * 16 + hi * 256;
  char* const patch = smem + WS_NB * WS_SLOT + wave * WS_PATCH;
  // epilogue offsets: this lane's 8-byte pieces of patch row r (16-byte slot blk*4 + gq, half hi) ...
  int ep[4];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) ep[gq] = r * 128 + ((gq ^ ((r >> 1) & 7)) << 4) + hi * 8;    // block B: slot ^ 4 -> + or - 64
  const int epB = (((r >> 1) & 4) ? -64 : 64);
  // ... and the 16-byte pieces (row 8 j + lane / 8, slot lane % 8) it stores
  const int srow0 = l >> 3, sc = l & 7;
  int sp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 8 * j + srow0;
    sp[j] = row * 128 + ((sc ^ ((row >> 1) & 7)) << 4);
  }
  const Drop dr = make_drop(a.drop);

  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // chunk 0 has landed (chunk 1's four loads may be in flight)
  for (int ci = 0; ci < nch; ++ci) {
    __builtin_amdgcn_s_barrier();      // every wave's part of chunk ci is in LDS; everybody is done with chunk ci-1's slot
    asm volatile("" ::: "memory");
    issue(ci + 2);                     // into the slot chunk ci-1 occupied
    const char* xs = smem + (ci % WS_NB) * WS_SLOT + rdbase;
    f32x16 cA, cB;
    {
      bf16x8 xf = *reinterpret_cast<const bf16x8*>(xs);
      cA = mfma32(wfA[0], xf, biasA);
      cB = mfma32(wfB[0], xf, biasB);
#pragma unroll
      for (int kk = 1; kk < WS_K / 16; ++kk) {
        xf = *reinterpret_cast<const bf16x8*>(xs + kk * 512);
        cA = mfma32(wfA[kk], xf, cA);
        cB = mfma32(wfB[kk], xf, cB);
      }
    }
    // chunk ci+1's loads of this wave have landed once at most chunk ci+2's four are outstanding
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    // ---- epilogue: row = lane & 31 of the chunk, columns n0 + 32 blk + acc_row(t, hi) ------------------------------
    const int grow = (rank + ci * a.ranks) * WS_CH;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        uint32_t bits = 0;
        if (EPI == WS_RELU && DROP) bits = dr.bits(drop_counter_rc(grow + r, n0 + 32 * blk + 8 * gq + 4 * hi, a.N));
        f32x4 v4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = blk ? cB[4 * gq + e] : cA[4 * gq + e];
          if (EPI == WS_RELU) {
            v = fmaxf(v, 0.f);
            if (DROP) v = dr.keep(bits, e) ? v * dr.scale : 0.f;
          }
          v4[e] = v;
        }
        *reinterpret_cast<bf16x4*>(patch + ep[gq] + (blk ? epB : 0)) = __builtin_convertvector(v4, bf16x4);
      }
    bf16x8 ov[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ov[j] = *reinterpret_cast<const bf16x8*>(patch + sp[j]);
    bf16* drow = a.D + (size_t)(grow + srow0) * a.ldd + n0 + sc * 8;
    if (grow + WS_CH <= a.M) {         // (uniform) no per-row guards
#pragma unroll
      for (int j = 0; j < 4; ++j) store16<ST_GEMM_SC1>(drow + (size_t)(8 * j) * a.ldd, ov[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (grow + 8 * j + srow0 < a.M) store16<ST_GEMM_SC1>(drow + (size_t)(8 * j) * a.ldd, ov[j]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (dummy) DMAs must not outlive the workgroup's LDS
}

}  // namespace

extern "C" int st_gemm_ws(hipStream_t stream, const void* X, int ldx, const void* W, int ldw, void* D, int ldd, int M,
                          int N, int K, const float* bias, int epi, const unsigned* drop_seed, unsigned drop_salt,
                          int drop_thresh, float drop_scale, int w_block_rows, long w_block_stride,
                          long bias_block_stride) {
  if (M <= 0 || N <= 0) return 0;
  if (K != WS_K || (N & 255) || (ldx & 7) || (ldw & 7) || (ldd & 7) || epi < 0 || epi > 1) return -1;
  WsArgs a;
  a.X = (const bf16*)X; a.ldx = ldx; a.W = (const bf16*)W; a.ldw = ldw; a.D = (bf16*)D; a.ldd = ldd; a.M = M; a.N = N;
  a.bias = bias;
  a.nslices = N / 256;
  a.nchunks = (M + WS_CH - 1) / WS_CH;
  a.wseg_shift = 31; a.wseg_extra = 0; a.bias_extra = 0;
  if (w_block_rows > 0) {          // stacked weights: blocks of w_block_rows rows (a power-of-two multiple of 256)
    if ((w_block_rows & 255) || (w_block_rows & (w_block_rows - 1)) || (w_block_stride & 7)) return -2;
    a.wseg_shift = __builtin_ctz(w_block_rows / 256);
    a.wseg_extra = w_block_stride - (long)w_block_rows * ldw;
    a.bias_extra = bias_block_stride - (long)w_block_rows;
  }
  // two workgroups per CU: the 512 slots are shared by the slices
  int ranks = 512 / a.nslices;
  if (ranks < 1) ranks = 1;
  if (ranks > a.nchunks) ranks = a.nchunks;
  a.ranks = ranks;
  const bool drop = epi == WS_RELU && drop_seed != nullptr && drop_thresh > 0;
  a.drop.seed = drop ? drop_seed : nullptr; a.drop.salt = drop_salt; a.drop.thresh = drop ? drop_thresh : 0;
  a.drop.scale = drop ? drop_scale : 1.f;
  const dim3 grid(ranks * a.nslices), block(256);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ws_kernel<WS_BF16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ws_kernel<WS_RELU, false>), hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ws_kernel<WS_RELU, true>), hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
    attr_set = true;
  }
  if (epi == WS_BF16) hipLaunchKernelGGL((gemm_ws_kernel<WS_BF16, false>), grid, block, WS_SMEM, stream, a);
  else if (!drop) hipLaunchKernelGGL((gemm_ws_kernel<WS_RELU, false>), grid, block, WS_SMEM, stream, a);
  else hipLaunchKernelGGL((gemm_ws_kernel<WS_RELU, true>), grid, block, WS_SMEM, stream, a);
  ST_CHECK_LAUNCH();
  return 0;
}

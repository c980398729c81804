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
// epilogue per tile; measured 300-480 TFLOP/s, bound by the per-tile latency chain, not by the matrix cores.  Here the
// WEIGHT never moves: a workgroup owns 256 output columns, each of its 8 waves keeps the 32 x 256 weight block of its
// 32 columns in 64 VGPRs (16 MFMA A-fragments) for the whole kernel, and the activations stream past in 32-row chunks
// through an LDS ring filled by LDS-DMA (global_load_lds_dwordx4) with a counted vmcnt: ONE barrier per chunk, no
// prologue / epilogue bubbles between output tiles, X is the only streamed operand.
//
// Pipeline (iteration i, all 8 waves, one s_barrier):   DMA(i + 5) | MFMA(i) || epilogue(i - 1) | store(i - 2)
//   * MFMA(i): 16 ds_read_b128 (X fragments, B operand: lane = x row) + 16 v_mfma_f32_32x32x16_bf16 into a transposed
//     accumulator (lane = x row, registers = 16 of the wave's 32 columns);
//   * epilogue(i - 1) works on the PREVIOUS chunk's accumulator in the same instruction stream, so its VALU work
//     (bias, activation, bf16 packing) fills the issue slots between the MFMAs instead of following them (measured:
//     with the epilogue after the chain, both waves of a SIMD - released by the same barrier - first queue for the
//     matrix pipe and then for the VALU: 1750 cycles per chunk against 1024 of MFMA);
//   * the packed rows go to a workgroup-wide [32][256] LDS patch (double-buffered); one barrier later store(i - 2)
//     writes them to HBM as whole 512-byte rows;
//   * roles: waves 0-3 issue every LDS-DMA and never store, waves 4-7 store and never wait on vmcnt - stores share the
//     vmcnt counter with loads and may retire out of order with them, so a wave that did both would have to wait for
//     its stores' acknowledgements before trusting a counted wait for its DMAs.
//
// The kernel is instruction-ISSUE bound, not MFMA- or memory-bound (s_memtime per phase: a SIMD issues one instruction
// per ~4 cycles whatever its kind, so a 32-cycle MFMA pays for at most ~7 other instructions of the SIMD's two waves):
// everything around the 16 MFMAs of a chunk is kept to a minimum -
//   * chunk image [row / 16][32 k-slices][row % 16] x 16 B: the 16 fragment reads of a lane are ONE base register plus
//     immediates (k-step kk at + 512 kk) and each 16-lane group reads one full 256-byte bank row (conflict-free);
//     LDS-DMA writes lane-linear, so the image is produced by the per-lane SOURCE address (64-byte pieces of 16 rows
//     per instruction);
//   * the bias enters as the C operand of the first MFMA (no accumulator initialisation, no add);
//   * LDS-DMA source offsets, patch and store offsets are per-lane constants computed once.
// Patch and weight-staging images: 16-byte slots XOR-swizzled by (row & 15) (conflict-free 8 / 16-byte accesses).
#include "st_common.cuh"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int WS_K = 256;                    // contraction length (held in registers per wave)
constexpr int WS_CH = 32;                    // rows per streamed chunk
constexpr int WS_NB = 6;                     // ring slots
constexpr int WS_SLOT = WS_CH * WS_K * 2;    // bytes per slot (16 KB)
constexpr int WS_PATCH = WS_CH * 256 * 2;    // workgroup-wide epilogue patch (16 KB), two of them
constexpr int WS_SMEM = WS_NB * WS_SLOT + 2 * WS_PATCH;   // 128 KB

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
  int dbg;
};

__device__ __attribute__((aligned(16))) float g_zero_ws[4];

template <int EPI, bool DROP>
__global__ __launch_bounds__(512, 2) void gemm_ws_kernel(WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
  // consecutive workgroups = the slices of one rank: they stream the same X rows at the same time (the chunk misses
  // HBM once and is then served from the L2s / Infinity Cache)
  const int slice = blockIdx.x % a.nslices, rank = blockIdx.x / a.nslices;
  const int nch = (a.nchunks - rank + a.ranks - 1) / a.ranks;     // chunks rank, rank + ranks, ...
  if (nch <= 0) return;
  const bool loader = wave < 4;      // (wave-uniform) LDS-DMA issue | global stores

  // ---- LDS-DMA: loader wave w fills LDS bytes [4096 w, 4096 (w + 1)) of every chunk image (four 1 KB instructions).
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
  if (loader) {
#pragma unroll
    for (int p = 0; p < 4; ++p) issue(p);      // slots 0-3 = [0, 64 KB); the weight staging below uses [64, 128 KB)
  }

  // ---- this wave's weight block (rows n0 .. n0+31 of W, all of K) -> 16 A-fragments in registers -----------------
  // coalesced 16-byte loads (256 B per row and instruction) -> wave-private swizzled LDS image -> row fragments;
  // fragment-shaped loads straight from global would touch 32 rows per instruction (3.3 us of the prologue, measured)
  const int n0 = slice * 256 + wave * 32;
  const long wblk = (long)(slice >> a.wseg_shift);
  const bf16* wbase = a.W + wblk * a.wseg_extra + (size_t)n0 * a.ldw;
  bf16x8 wf[WS_K / 16];
  {
    char* stage = smem + 64 * 1024 + wave * 8192;                  // [32 rows][256 B]
    const int srow = l >> 4, ss = l & 15;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x8 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = *reinterpret_cast<const bf16x8*>(wbase + (size_t)(4 * j + srow) * a.ldw + h * 128 + ss * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = 4 * j + srow;
        *reinterpret_cast<bf16x8*>(stage + row * 256 + ((ss ^ (row & 15)) << 4)) = t[j];
      }
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
        wf[h * 8 + kk] = *reinterpret_cast<const bf16x8*>(stage + r * 256 + (((2 * kk + hi) ^ (r & 15)) << 4));
    }
  }
  f32x16 bias16;                       // the accumulator's C input: bias of column n0 + acc_row(t, hi) in register t
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias ? a.bias + wblk * a.bias_extra + n0 + 8 * gq + 4 * hi : g_zero_ws);
#pragma unroll
    for (int e = 0; e < 4; ++e) bias16[4 * gq + e] = b4[e];
  }
  // the compiler's own loads are retired HERE, so that it places no s_waitcnt vmcnt() of its own inside the loop (it
  // cannot see the LDS-DMAs and would drain them)
#pragma unroll
  for (int kk = 0; kk < WS_K / 16; ++kk) touch(wf[kk]);
  touch(bias16);
  __syncthreads();                   // every wave has read its weight image: the staging area becomes ring slots 4, 5
  if (loader) issue(4);

  // fragment reads of this lane (chunk row r, k-slice 2 kk + hi): rdbase + 512 kk
  const int rdbase = (r >> 4) * 8192 + (r & 15) * 16 + hi * 256;
  char* const patch0 = smem + WS_NB * WS_SLOT;
  const Drop dr = make_drop(a.drop);
  // store(c): thread t of waves 4-7 moves 16-byte pieces id = j * 256 + t (row id >> 5, slot id & 31) of the patch
  const int st = (int)threadIdx.x - 256;
  int sp[4], so[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int id = j * 256 + st, row = id >> 5, c = id & 31;
    sp[j] = row * 512 + (((c & 16) | ((c ^ row) & 15)) << 4);
    so[j] = row * a.ldd + c * 8;
  }
  // epilogue: this lane's four 8-byte pieces of patch row r
  int ep[4];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int c = wave * 4 + gq;
    ep[gq] = r * 512 + (((c & 16) | ((c ^ r) & 15)) << 4) + hi * 8;
  }

  // One pipeline step: [barrier] DMA(ci + 5) | store(ci - 2) | MFMA(ci) -> cur || epilogue(ci - 1) <- prev.
  // MMA / EPI are compile-time so that the steady state is ONE basic block: the scheduler can then place the
  // epilogue's VALU work and LDS writes in the issue slots between the 16 dependent MFMAs.
  auto step = [&](auto mma_c, auto epi_c, int ci, f32x16& cur, const f32x16& prev) {
    constexpr bool MMA = decltype(mma_c)::value, EPI_ON = decltype(epi_c)::value;
    if (loader && !(a.dbg & 2)) {
      // chunk ci's loads have landed once at most the 4 * 4 younger ones (chunks ci+1 .. ci+4) are outstanding
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WS_NB - 2) * 4) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's patch writes have reached the LDS
    __builtin_amdgcn_s_barrier();    // chunk ci is in LDS; patch (ci-1) & 1 is complete; everybody is done with chunk ci-1
    asm volatile("" ::: "memory");
    if (loader) {
      if (!(a.dbg & 2)) issue(ci + WS_NB - 1);         // into the slot chunk ci-1 occupied
    } else if (ci >= 2 && !(a.dbg & 1)) {            // store(ci - 2): 256 threads x 4 x 16 B = the 32 x 512 B patch, whole rows
      const char* pt = patch0 + (ci & 1) * WS_PATCH;
      const int grow = (rank + (ci - 2) * a.ranks) * WS_CH;
      bf16* drow = a.D + (size_t)grow * a.ldd + slice * 256;
      bf16x8 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const bf16x8*>(pt + sp[j]);
      if (grow + WS_CH <= a.M) {     // (workgroup-uniform) the whole chunk is inside the matrix: no per-row guards
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<bf16x8*>(drow + so[j]) = v[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (grow + ((j * 256 + st) >> 5) < a.M) *reinterpret_cast<bf16x8*>(drow + so[j]) = v[j];
      }
    }
    bf16x8 xf[WS_K / 16];
    if (MMA) {
      const char* xs = smem + (ci % WS_NB) * WS_SLOT;
      if (a.dbg & 4) {
#pragma unroll
        for (int kk = 0; kk < WS_K / 16; ++kk) xf[kk] = wf[(kk + 1) & 15];
      } else {
#pragma unroll
        for (int kk = 0; kk < WS_K / 16; ++kk) xf[kk] = *reinterpret_cast<const bf16x8*>(xs + rdbase + kk * 512);
      }
      if (a.dbg & 8) {
#pragma unroll
        for (int kk = 0; kk < WS_K / 16; ++kk) asm volatile("" :: "v"(xf[kk]));
        cur = bias16;
      } else {
      // two independent accumulator chains (even / odd k-steps): a dependent MFMA cannot issue before its
      // predecessor's result is back, which is longer than the 32-cycle issue interval
      f32x16 c0 = mfma32(wf[0], xf[0], bias16), c1 = mfma32(wf[1], xf[1], zero16());
#pragma unroll
      for (int kk = 2; kk < WS_K / 16; kk += 2) {
        c0 = mfma32(wf[kk], xf[kk], c0);
        c1 = mfma32(wf[kk + 1], xf[kk + 1], c1);
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) cur[t] = c0[t] + c1[t];
      }
    }
    if (EPI_ON && !(a.dbg & 16)) {                    // epilogue(ci - 1): row = lane & 31 of that chunk, columns n0 + acc_row(t, hi)
      char* pt = patch0 + ((ci - 1) & 1) * WS_PATCH;
      const int grow = (rank + (ci - 1) * a.ranks) * WS_CH;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        uint32_t bits = 0;
        if (EPI == WS_RELU && DROP) bits = dr.bits(drop_counter_rc(grow + r, n0 + 8 * gq + 4 * hi, a.N));
        f32x4 v4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = prev[4 * gq + e];
          if (EPI == WS_RELU) {
            v = fmaxf(v, 0.f);
            if (DROP) v = dr.keep(bits, e) ? v * dr.scale : 0.f;
          }
          v4[e] = v;
        }
        *reinterpret_cast<bf16x4*>(pt + ep[gq]) = __builtin_convertvector(v4, bf16x4);
      }
    }
    if (MMA && EPI_ON) {             // 16 x { 1 MFMA, a few epilogue VALU }, an LDS write every fourth slot
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, EPI == WS_RELU ? 2 : 1, 0);
        if ((i & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    }
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;

  f32x16 acc_a = zero16(), acc_b = zero16();      // chunk ci accumulates into acc_a (ci even) / acc_b (ci odd)
  step(T_{}, F_{}, 0, acc_a, acc_b);
  int ci = 1;
  for (; ci + 1 < nch; ci += 2) {
    step(T_{}, T_{}, ci, acc_b, acc_a);
    step(T_{}, T_{}, ci + 1, acc_a, acc_b);
  }
  if (ci < nch) {                                  // one steady step left (nch even)
    step(T_{}, T_{}, ci, acc_b, acc_a);
    step(F_{}, T_{}, nch, acc_a, acc_b);           // epilogue(nch - 1): nch - 1 = ci is odd -> acc_b
  } else {
    step(F_{}, T_{}, nch, acc_b, acc_a);           // nch odd: the last chunk (even) sits in acc_a
  }
  step(F_{}, F_{}, nch + 1, acc_a, acc_b);         // store(nch - 1)
  if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (dummy) DMAs must not outlive the workgroup's LDS
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
  { const char* e = getenv("ST_WS_DBG"); a.dbg = e ? atoi(e) : 0; }
  a.nslices = N / 256;
  a.nchunks = (M + WS_CH - 1) / WS_CH;
  a.wseg_shift = 31; a.wseg_extra = 0; a.bias_extra = 0;
  if (w_block_rows > 0) {          // stacked weights: blocks of w_block_rows rows (a power-of-two multiple of 256)
    if ((w_block_rows & 255) || (w_block_rows & (w_block_rows - 1)) || (w_block_stride & 7)) return -2;
    a.wseg_shift = __builtin_ctz(w_block_rows / 256);
    a.wseg_extra = w_block_stride - (long)w_block_rows * ldw;
    a.bias_extra = bias_block_stride - (long)w_block_rows;
  }
  // one workgroup per CU: the 256 CUs are shared by the slices
  int ranks = 256 / a.nslices;
  if (ranks < 1) ranks = 1;
  if (ranks > a.nchunks) ranks = a.nchunks;
  a.ranks = ranks;
  const bool drop = epi == WS_RELU && drop_seed != nullptr && drop_thresh > 0;
  a.drop.seed = drop ? drop_seed : nullptr; a.drop.salt = drop_salt; a.drop.thresh = drop ? drop_thresh : 0;
  a.drop.scale = drop ? drop_scale : 1.f;
  const dim3 grid(ranks * a.nslices), block(512);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ws_kernel<WS_BF16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ws_kernel<WS_RELU, false>), hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ws_kernel<WS_RELU, true>), hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
    attr_set = true;
  }
  if (epi == WS_BF16) hipLaunchKernelGGL((gemm_ws_kernel<WS_BF16, false>), grid, block, WS_SMEM, stream, a);
  else if (!drop) hipLaunchKernelGGL((gemm_ws_kernel<WS_RELU, false>), grid, block, WS_SMEM, stream, a);
  else hipLaunchKernelGGL((gemm_ws_kernel<WS_RELU, true>), grid, block, WS_SMEM, stream, a);
  ST_CHECK_LAUNCH();
  return 0;
}

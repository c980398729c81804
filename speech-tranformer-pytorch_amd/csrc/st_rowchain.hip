// st_rowchain: chains of row-wise layers for decoder-sized row counts as ONE launch.
//
// With ~1,200 target rows (or 320 beam hypotheses in decode) every GEMM of the decoder is a launch of ~20-40 workgroups
// whose duration is launch ramp + prologue + epilogue, not arithmetic: a GEMM + LayerNorm costs
// ~7.7 us before its first k-tile.  Between two attention kernels everything the decoder does is row-wise, so one launch
// can run it all for a block of rows:
//
//   PRE   cur = LN(A Wo^T + bo + R) * g0 + be0                 output_linear + residual + layernorm (Attention.py:92-94)
//   FFN   h   = dropout1(relu(cur W1^T + b1))                  (SubLayers.py:25)
//         cur = dropout2(LN(h W2^T + b2 + cur) * g1 + be1)     (SubLayers.py:26-27)
//   POST  P   = cur Wp^T + bp,  Wp [256 nb, 256]               the next attention's q (nb = 1) or q|k|v (nb = 3) projection
//                                                              (Attention.py:74-76)
//
// each part optional: PRE + POST is what follows the decoder's self-attention, PRE + FFN + POST what follows its
// encoder-decoder attention (POST being the next layer's q|k|v).  A workgroup owns 32 rows for the whole chain; the
// activations live in LDS; every intermediate the backward pass reads is also written to HBM exactly as the separate
// kernels write it.  The weights are STREAMED: all GEMMs are cut into 256 x 256 blocks; each of the 8 waves owns 32
// output columns of every block and reads exactly the weight fragments it multiplies, in the order it multiplies them,
// from a per-wave "fragment stream" (st_wfrag_build lays the weight blocks out as 1 KB MFMA A-operand fragments in
// consumption order): one fully coalesced 1 KB load per MFMA, 16 of them in flight per wave, no LDS staging of
// weights, no barrier inside a block.  d_ff is walked in chunks of 256 hidden columns (W1 block -> LDS -> W2 block
// accumulates), so the hidden tile never exceeds 16 KB of LDS.  Measured (M = 1206, d_ff 1024): the feed-forward
// sublayer alone 25.5 us as two launches -> 15.9 us, bound by one CU's 64 B/clk vector-memory path (1 MB of weights
// per workgroup).
#include "st_rowchain_common.cuh"
#include "st_rowchain_pipe.cuh"
#include "st_rowchain_pipe512.cuh"
#include <cstdlib>
#include <type_traits>

namespace {

// The forward chains' epilogue vectors as an LDS copy (32- and 64-row workgroups; LVec in st_rowchain_common.cuh): units of 256 floats -
// 0 bo, 1 g0, 2 be0, 3 b2, 4 g1, 5 be1, 6 .. 6 + nc - 1 the chunks of b1, then the nb blocks of bp.  The launch passes
// (6 + nc + nb) KB of dynamic LDS.  Wave w fetches units w, w + 8, w + 16, w + 24 (one 16-byte piece per lane) together with the
// activation tiles and stores them behind the tiles' stores; chains with more than 32 units take further rounds.
extern __shared__ __attribute__((aligned(16))) float chain_vecs[];
__device__ __forceinline__ const float* chain_vec_src(const ChainArgs& a, int u) {
  if (u < 6) return u == 0 ? a.bo : u == 1 ? a.g0 : u == 2 ? a.be0 : u == 3 ? a.b2 : u == 4 ? a.g1 : a.be1;
  if (u < 6 + a.nc) return a.b1 ? a.b1 + (u - 6) * 256 : nullptr;
  return a.bp ? a.bp + (u - 6 - a.nc) * 256 : nullptr;
}
struct VecStage {
  f32x4 v[4];
  __device__ __forceinline__ void load(const ChainArgs& a, int wave, int l, int u0) {
    const int n = 6 + a.nc + a.nb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = u0 + wave + 8 * i;
      const float* src = u < n ? chain_vec_src(a, u) : nullptr;
      v[i] = src ? *reinterpret_cast<const f32x4*>(src + 4 * l) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __device__ __forceinline__ void store(const ChainArgs& a, int wave, int l, int u0) const {
    const int n = 6 + a.nc + a.nb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = u0 + wave + 8 * i;
      if (u < n) *reinterpret_cast<f32x4*>(chain_vecs + u * 256 + 4 * l) = v[i];
    }
  }
};
template <bool LDSV> struct ChainVecs;
template <> struct ChainVecs<true> {
  using V = LVec;
  const ChainArgs& a;
  __device__ __forceinline__ LVec at(int u) const { return LVec{(const ST_LDS float*)chain_vecs + u * 256}; }
  __device__ __forceinline__ V bo() const { return at(0); }
  __device__ __forceinline__ V g0() const { return at(1); }
  __device__ __forceinline__ V be0() const { return at(2); }
  __device__ __forceinline__ V b2() const { return at(3); }
  __device__ __forceinline__ V g1() const { return at(4); }
  __device__ __forceinline__ V be1() const { return at(5); }
  __device__ __forceinline__ V b1(int ch) const { return at(6 + ch); }
  __device__ __forceinline__ V bp(int u) const { return at(6 + a.nc + u); }
};
template <> struct ChainVecs<false> {
  using V = const float*;
  const ChainArgs& a;
  __device__ __forceinline__ V bo() const { return a.bo; }
  __device__ __forceinline__ V g0() const { return a.g0; }
  __device__ __forceinline__ V be0() const { return a.be0; }
  __device__ __forceinline__ V b2() const { return a.b2; }
  __device__ __forceinline__ V g1() const { return a.g1; }
  __device__ __forceinline__ V be1() const { return a.be1; }
  __device__ __forceinline__ V b1(int ch) const { return a.b1 + ch * 256; }
  __device__ __forceinline__ V bp(int u) const { return a.bp + u * 256; }
};

// MT = row tiles of 32 per workgroup.  1: decoder-sized row counts (as many workgroups as possible).  2: in between (a
// strong-scaling shard of the batch).  3: encoder-sized
// ones (24,060 rows = 251 workgroups = one round of the 256 CUs; every weight fragment feeds three MFMAs, so a
// workgroup's MFMA time matches its weight stream; three 50 KB activation tiles fill the LDS).
template <bool PRE, bool FFN, bool POST, bool DROP, int MT>
__global__ __launch_bounds__(512, 1) void row_chain_kernel(ChainArgs a) {
  constexpr int RB = 32 * MT, TE = RB * AS;
  // three activation tiles: `cur` (the running activation) and two free ones that serve, in turn, as residual, hidden
  // chunks, xhat and output staging.  A tile is rewritten only after a workgroup barrier that every wave reaches after its
  // last read of it (the comments at each site name that barrier).
  __shared__ __attribute__((aligned(16))) bf16 tiles[3 * TE];
  __shared__ float red[2][NW * 32 * MT];
  Ctx<MT> c;
  c.tid = threadIdx.x; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6); c.l = c.tid & 63; c.hi = c.l >> 5; c.r = c.l & 31;
  c.row0 = blockIdx.x * RB; c.nvalid = min(RB, a.M - c.row0);
  c.ws = a.wfrag + (size_t)c.wave * a.wave_frags * 64;   // wave-uniform: the loads take an SGPR base + lane offset
#pragma unroll
  for (int i = 0; i < Ring<MT>::D; ++i) c.ring[i] = c.ws[i * 64 + c.l];   // the first fragments go out before anything else
  c.ws += Ring<MT>::D * 64;
  // Warm the L2 of this workgroup's XCD with the WHOLE chain's streams - and those of the chain that runs next (they lie
  // right behind in st_amd.chains' buffer; an attention kernel runs in between): the streams are read once per step, so
  // a wave's 16 KB in flight would otherwise meet the HBM latency block after block (cold caches, M = 1206: 40.4 us for
  // the 12-block chain against 22.6 us with the streams cached; 32.8 us with its own lines touched up front).  The
  // workgroups that share an XCD (dispatch is round-robin over the 8 XCDs) deal the 128-byte lines among their threads;
  // the values are only consumed at the very end.  MT = 3 (hundreds of workgroups, ~31 per XCD, all starting at the same
  // fragment): one line per thread covers the chain's own streams - without it every workgroup of an XCD waits for the
  // same HBM fetches block after block (the stream runs at the latency-bound rate of a single requester).
  constexpr int NTOUCH = MT == 1 ? TOUCH : 1;
  int touched[NTOUCH];
  {
    const int nlines = NW * (a.wave_frags + (MT == 1 ? a.next_frags : 0)) * 8;
    const int xcd = blockIdx.x & 7, nr = ((int)gridDim.x - xcd + 7) >> 3;
    const char* sb = reinterpret_cast<const char*>(a.wfrag);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NTOUCH; ++t) {
      const int ln = min(((int)blockIdx.x >> 3) * 512 + c.tid + t * nr * 512, nlines - 1);
      touched[t] = *reinterpret_cast<const int*>(sb + (size_t)ln * 128);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  bf16* cur = tiles;             // A
  bf16* f0 = tiles + TE;         // residual, then the first free tile
  bf16* f1 = tiles + 2 * TE;
  constexpr bool LV = MT <= 2;      // 32- and 64-row workgroups: the epilogue vectors from LDS (96-row ones have none to spare)
  const ChainVecs<LV> vec{a};
  {   // A and the residual are requested together (one global round trip), then stored
    TileRegs<MT> ra, rr;
    VecStage vs;
    tile_load(c, a.A, a.lda, ra);
    if (PRE) tile_load(c, a.R, a.ldr, rr);
    if (LV) vs.load(a, c.wave, c.l, 0);
    tile_store(c, ra, cur);
    if (PRE) tile_store(c, rr, f0);
    if (LV) {
      vs.store(a, c.wave, c.l, 0);
      for (int u0 = 32; u0 < 6 + a.nc + a.nb; u0 += 32) { vs.load(a, c.wave, c.l, u0); vs.store(a, c.wave, c.l, u0); }
    }
  }
  const Drop d1 = make_drop(a.drop1), d2 = make_drop(a.drop2), off = make_drop(DropArgs{nullptr, 0u, 0, 1.f});
  __syncthreads();

  if (PRE) {
    f32x16 acc[MT];
    zero_acc(acc);
    block_mma(c, cur, acc);
    // xhat is staged in the A tile (every wave is past its MFMAs on it at epi_ln's first barrier), the output in f1
    epi_ln<false>(c, acc, vec.bo(), f0, vec.g0(), vec.be0(), a.eps, off, cur, f1, red, a.out0, a.xhat0, a.rstd0);
    // now: cur = f1; free: f0 (the residual: last read before epi_ln's barriers) and, one barrier later, the A tile (xhat0 is
    // still being copied out of it)
    bf16* t = cur; cur = f1; f1 = t;
  }
  if (FFN) {
    const int dff = a.nc * 256;
    f32x16 acc2[MT];
    zero_acc(acc2);
    for (int ch = 0; ch < a.nc; ++ch) {
      // hidden chunks alternate f0, f1; chunk ch's tile is rewritten by chunk ch + 2 with the barrier of chunk ch + 1 in
      // between; f1's first use (chunk 1) lies behind chunk 0's barrier, which every wave reaches after PRE's copies out
      bf16* hc = (ch & 1) ? f1 : f0;
      f32x16 acc1[MT];
      zero_acc(acc1);
      block_mma(c, cur, acc1);
      epi_store<true, DROP>(c, acc1, vec.b1(ch), hc, d1, ch * 256, dff,
                            a.relu_bits ? a.relu_bits + ((size_t)(blockIdx.x * a.nc + ch) * NW + c.wave) * 64 : nullptr);
      __syncthreads();
      block_mma(c, hc, acc2);
      if (a.H) tile_out(c, hc, a.H + ch * 256, dff);      // (no H: inference - nobody reads the hidden activation back)
    }
    // xhat goes to the tile the LAST chunk did not use (last read one chunk earlier), the output replaces cur in place
    // (its residual reads precede epi_ln's barriers, its MFMA reads too)
    bf16* tx = (a.nc & 1) ? f1 : f0;
    epi_ln<DROP>(c, acc2, vec.b2(), cur, vec.g1(), vec.be1(), a.eps, d2, tx, cur, red, a.out1, a.xhat1, a.rstd1);
    if (tx == f0) { f0 = f1; f1 = tx; }      // f0 = the tile free right now, f1 = xhat (still being copied out)
  }
  if (POST) {
    for (int u = 0; u < a.nb; ++u) {
      // staging alternates f0, f1: f0 is free (see above), f1 one barrier later; a tile is rewritten two blocks later
      bf16* st = (u & 1) ? f1 : f0;
      f32x16 acc[MT];
      zero_acc(acc);
      block_mma(c, cur, acc);
      epi_store<false, false>(c, acc, vec.bp(u), st, off, 0, 0, nullptr, u == 1 ? a.post_kscale : 1.f);
      __syncthreads();
      tile_out(c, st, a.P + u * 256, a.ldp);
    }
  }
  {
    int tsum = 0;
#pragma unroll
    for (int t = 0; t < NTOUCH; ++t) tsum ^= touched[t];
    if (tsum == 0x5a5a5a5a && a.M < 0) red[0][0] = 1.f;      // (never true: keeps the warm-up loads alive)
  }
}

// The same chain with the feed-forward sublayer's hidden dimension cut over nc WORKGROUPS per 32-row block (decoder-sized
// row counts only: M / 32 x nc workgroups must fit one round of the chip).  A 12-block chain is 12 weight blocks through
// ONE compute unit at the latency-bound rate of a single requester (~2.2 us per 128 KB block: ten workgroups of a decode
// step, 38 of a training step, on a 256-CU chip).  Only the hidden dimension can be cut without an exchange per block: the
// nc chunks of d_ff are independent until their contributions to the second GEMM are added.  So workgroup (block, part)
// runs PRE (replicated - one block), chunk `part` of the feed-forward (two blocks) and leaves its 32 x 256 fp32 partial in
// `split_ws` with write-through stores; the LAST of the block's nc workgroups to draw the block's ticket adds the partials
// in index order (the result does not depend on who is last), and runs the LayerNorm epilogue and POST: 6 blocks + one
// merge on the critical path instead of 12.  No fence (an agent-scope release writes the whole L2 back: +60-80 us,
// tools/dev/merge_probe.hip): system-scope relaxed stores, s_waitcnt, a device-scope ticket, device-scope loads.
// Part 0 writes PRE's saved tensors; every part writes its chunk of H and its ReLU bits.
template <bool PRE, bool POST, bool DROP>
__global__ __launch_bounds__(512, 1) void row_chain_split_kernel(ChainArgs a) {
  constexpr int MT = 1, RB = 32, TE = RB * AS;
  __shared__ __attribute__((aligned(16))) bf16 tiles[3 * TE];
  __shared__ float red[2][NW * 32 * MT];
  __shared__ bool last;
  Ctx<MT> c;
  c.tid = threadIdx.x; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6); c.l = c.tid & 63; c.hi = c.l >> 5; c.r = c.l & 31;
  const int parts = a.split_parts, cpp = a.nc / parts, block = blockIdx.x / parts, part = blockIdx.x - block * parts;
  c.row0 = block * RB; c.nvalid = min(RB, a.M - c.row0);
  const bf16x8* base = a.wfrag + (size_t)c.wave * a.wave_frags * 64;
  // stream order: [PRE] [W1_0 W2_0] .. [W1_(nc-1) W2_(nc-1)] [POST blocks]; this workgroup multiplies PRE, the two blocks of
  // each of its cpp chunks (consecutive in the stream) and - if it turns out to be the last - POST.  The ring always holds the
  // block being multiplied and is refilled from c.ws, so c.ws is pointed at the block that comes NEXT for this workgroup
  // before every block_mma that is not followed by its stream neighbour.
  auto at_block = [&](int b) { return base + (size_t)b * 16 * 64; };
  const int b_chunk = (PRE ? 1 : 0) + 2 * part * cpp, b_post = (PRE ? 1 : 0) + 2 * a.nc;
  c.ws = at_block(PRE ? 0 : b_chunk);
#pragma unroll
  for (int i = 0; i < Ring<MT>::D; ++i) c.ring[i] = c.ws[i * 64 + c.l];
  c.ws += Ring<MT>::D * 64;
  int touched[TOUCH];
  {
    const int nlines = NW * (a.wave_frags + a.next_frags) * 8;
    const int xcd = blockIdx.x & 7, nr = ((int)gridDim.x - xcd + 7) >> 3;
    const char* sb = reinterpret_cast<const char*>(a.wfrag);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < TOUCH; ++t) {
      const int ln = min(((int)blockIdx.x >> 3) * 512 + c.tid + t * nr * 512, nlines - 1);
      touched[t] = *reinterpret_cast<const int*>(sb + (size_t)ln * 128);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  bf16* cur = tiles;
  bf16* f0 = tiles + TE;
  bf16* f1 = tiles + 2 * TE;
  const ChainVecs<true> vec{a};
  {
    TileRegs<MT> ra, rr;
    VecStage vs;
    tile_load(c, a.A, a.lda, ra);
    if (PRE) tile_load(c, a.R, a.ldr, rr);
    vs.load(a, c.wave, c.l, 0);
    tile_store(c, ra, cur);
    if (PRE) tile_store(c, rr, f0);
    vs.store(a, c.wave, c.l, 0);
    for (int u0 = 32; u0 < 6 + a.nc + a.nb; u0 += 32) { vs.load(a, c.wave, c.l, u0); vs.store(a, c.wave, c.l, u0); }
  }
  const Drop d1 = make_drop(a.drop1), d2 = make_drop(a.drop2), off = make_drop(DropArgs{nullptr, 0u, 0, 1.f});
  __syncthreads();
  if (PRE) {
    f32x16 acc[MT];
    zero_acc(acc);
    c.ws = at_block(b_chunk);
    block_mma(c, cur, acc);
    const bool w = part == 0;
    epi_ln<false>(c, acc, vec.bo(), f0, vec.g0(), vec.be0(), a.eps, off, cur, f1, red, w ? a.out0 : nullptr, w ? a.xhat0 : nullptr,
                  w ? a.rstd0 : nullptr);
    bf16* t = cur; cur = f1; f1 = t;
  }
  f32x16 acc2[MT];
  zero_acc(acc2);
  {
    const int dff = a.nc * 256;
    for (int j = 0; j < cpp; ++j) {
      // hidden tiles alternate f0, f1 as in row_chain_kernel: f0 is free (the residual's last read precedes epi_ln's barriers;
      // without PRE never written), f1 (PRE's xhat staging, copied out by part 0) is first written behind chunk 0's barrier
      const int ch = part * cpp + j;
      bf16* hc = (j & 1) ? f1 : f0;
      f32x16 acc1[MT];
      zero_acc(acc1);
      block_mma(c, cur, acc1);                  // W1 chunk; refills: the W2 chunk right behind it
      epi_store<true, DROP>(c, acc1, vec.b1(ch), hc, d1, ch * 256, dff,
                            a.relu_bits ? a.relu_bits + ((size_t)(block * a.nc + ch) * NW + c.wave) * 64 : nullptr);
      __syncthreads();
      if (j + 1 == cpp) c.ws = at_block(b_post);      // (POST's first block, on the chance that this workgroup is the last)
      block_mma(c, hc, acc2);
      if (a.H) tile_out(c, hc, a.H + ch * 256, dff);
    }
  }
  // ---- the partial leaves through the L2 (write-through), the ticket says who merges
  float* slot = a.split_ws + (size_t)block * parts * (512 * 16);
  {
    float* mine = slot + (size_t)part * (512 * 16);
    unsigned long long* mine8 = reinterpret_cast<unsigned long long*>(mine);      // (two floats per 8-byte access)
#pragma unroll
    for (int i = 0; i < 16; i += 2)
      __hip_atomic_store(mine8 + (i / 2) * 512 + c.tid,
                         ((unsigned long long)__float_as_uint(acc2[0][i + 1]) << 32) | __float_as_uint(acc2[0][i]), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    ST_PUBLISH_FENCE();
    __builtin_amdgcn_s_waitcnt(0);
  }
  __syncthreads();
  if (c.tid == 0)
    last = __hip_atomic_fetch_add(a.split_tickets + block, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(parts - 1);
  __syncthreads();
  if (last) {
    ST_MERGER_FENCE();
    if (c.tid == 0) __hip_atomic_store(a.split_tickets + block, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    zero_acc(acc2);
    for (int p = 0; p < parts; ++p) {
      const unsigned long long* src8 = reinterpret_cast<const unsigned long long*>(slot + (size_t)p * (512 * 16));
      unsigned long long v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __hip_atomic_load(src8 + i * 512 + c.tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc2[0][2 * i] += __uint_as_float((unsigned)v[i]);
        acc2[0][2 * i + 1] += __uint_as_float((unsigned)(v[i] >> 32));
      }
    }
    // xhat goes to f1 (the A tile / PRE's xhat staging: last read two barriers ago), the output replaces cur in place
    epi_ln<DROP>(c, acc2, vec.b2(), cur, vec.g1(), vec.be1(), a.eps, d2, f1, cur, red, a.out1, a.xhat1, a.rstd1);
    if (POST) {
      for (int u = 0; u < a.nb; ++u) {
        bf16* st = (u & 1) ? f1 : f0;
        f32x16 acc[MT];
        zero_acc(acc);
        block_mma(c, cur, acc);
        epi_store<false, false>(c, acc, vec.bp(u), st, off, 0, 0, nullptr, u == 1 ? a.post_kscale : 1.f);
        __syncthreads();
        tile_out(c, st, a.P + u * 256, a.ldp);
      }
    }
  }
  {
    int tsum = 0;
#pragma unroll
    for (int t = 0; t < TOUCH; ++t) tsum ^= touched[t];
    if (tsum == 0x5a5a5a5a && a.M < 0) red[0][0] = 1.f;      // (never true: keeps the warm-up loads alive)
  }
}

// =====================================================================================================================
// Backward chains: the same row blocks, weight blocks read transposed (st_wfrag_build), in the reverse order of the
// forward chain:
//
//   HEAD  dy = sum_u dP[:, 256u..] Wp_u + G                      the data gradient of the NEXT attention's projection(s)
//         ds_a = LayerNorm-backward(dropout-backward(dy); xhat_a, rstd_a, gamma_a)      (== st_gemm_lnbwd)
//         dgamma_a += sum_rows dy xhat,  dbeta_a += sum_rows dy,  dbias_a += sum_rows ds_a
//   FFN   dH = (ds W2) masked by H > 0, x mask_scale              (== st_gemm EPI_BF16_MASK; SubLayers.py:25-26 backward)
//         dy = dH W1 + ds;  ds_b = LayerNorm-backward(dy; xhat_b, rstd_b, gamma_b) and its three column sums
//   TAIL  dctx = ds Wo;  delta[h][i] = sum over head h of dctx (O + Ores)        (== st_gemm EPI_BF16_DELTA)
//
// (ds = the running gradient: HEAD's ds_a, else the input DS).  HEAD + TAIL follows the decoder-encoder attention's
// backward kernel, HEAD + FFN + TAIL the self-attention's (decoder: of the next layer; encoder likewise).
struct ChainBwdArgs {
  int M;
  const bf16x8* wfrag; int wave_frags; int next_frags;
  // HEAD
  int nb; const bf16* dP; int ldp; const bf16* G; int ldg;
  const bf16* xhat_a; const float* rstd_a; const float* gamma_a; DropArgs drop_a;
  bf16* ds_a; float* dgamma_a; float* dbeta_a; float* dbias_a;
  const bf16* DS;                    // no HEAD: the chain input [M, 256], ld 256
  // FFN
  int nc; const unsigned long long* relu_bits; float mask_scale; bf16* dH;
  const bf16* xhat_b; const float* rstd_b; const float* gamma_b;
  bf16* ds_b; float* dgamma_b; float* dbeta_b; float* dbias_b;
  // TAIL
  const bf16* O; const bf16* Ores; int ldo; bf16* dctx; int lddc; float* delta;
  // split feed-forward (row_chain_bwd_split_kernel): as ChainArgs
  float* split_ws; unsigned* split_tickets; int split_parts;
  // pipelined kernel (st_rowchain_pipe_bwd.cuh): [workgroups][a: dgamma, dbeta, dbias | b: ..][256] column sums instead of atomics
  float* colsum_ws;
};

// LayerNorm backward over the 256 columns held by the 8 waves.  acc = the GEMM result; t_aux holds the addend and receives
// dy (bf16, rounded once as acc + addend, before the dropout mask) in place; t_xhat the saved normalised values; dx goes
// to t_dx and to HBM; then the three column sums (256 columns x the block's rows, two threads per column) are added
// atomically.  Barriers: two for the row sums, one before the column pass / copy out; the caller needs one more before
// t_aux / t_xhat are rewritten.
template <bool DROP, int MT>
__device__ __forceinline__ void epi_lnbwd(const Ctx<MT>& c, f32x16 (&acc)[MT], bf16* t_aux, const bf16* t_xhat, bf16* t_dx,
                                          const float* g_rstd, const float* gamma, const Drop& d, float (*red)[NW * 32 * MT],
                                          bf16* g_dx, float* dgamma, float* dbeta, float* dbias) {
  const int j0 = c.wave * 32;
  float s1[MT], s2[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) s1[mt] = s2[mt] = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int jl = j0 + 8 * g + 4 * c.hi;
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + jl);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = mt * 32 + c.r, at = row * AS + jl;
      const bf16x4 xh4 = *reinterpret_cast<const bf16x4*>(t_xhat + at);
      const bf16x4 ad4 = *reinterpret_cast<const bf16x4*>(t_aux + at);
      uint32_t bits = 0;
      if (DROP) bits = d.bits(drop_counter_rc(c.row0 + row, jl, DM));   // the mask the forward drew on this LayerNorm's output
      bf16x4 dy4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dy4[e] = (bf16)(acc[mt][4 * g + e] + (float)ad4[e]);       // one rounding, as st_gemm_lnbwd
        float v = (float)dy4[e];
        if (DROP && d.on()) v = d.keep(bits, e) ? v * d.scale : 0.f;
        const float gg = v * g4[e];
        acc[mt][4 * g + e] = gg;                                    // keep g = dy * gamma
        s1[mt] += gg;
        s2[mt] += gg * (float)xh4[e];
      }
      *reinterpret_cast<bf16x4*>(t_aux + at) = dy4;
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    s1[mt] += wave_xor32(s1[mt]);
    s2[mt] += wave_xor32(s2[mt]);
    if (c.hi == 0) {
      red[0][(c.wave * MT + mt) * 32 + c.r] = s1[mt];
      red[1][(c.wave * MT + mt) * 32 + c.r] = s2[mt];
    }
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      a1 += red[0][(w * MT + mt) * 32 + c.r];
      a2 += red[1][(w * MT + mt) * 32 + c.r];
    }
    const int row = mt * 32 + c.r;
    const float m1 = a1 * (1.f / DM), m2 = a2 * (1.f / DM), rs = row < c.nvalid ? g_rstd[c.row0 + row] : 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int jl = j0 + 8 * g + 4 * c.hi, at = row * AS + jl;
      const bf16x4 xh4 = *reinterpret_cast<const bf16x4*>(t_xhat + at);
      bf16x4 dx4;
#pragma unroll
      for (int e = 0; e < 4; ++e) dx4[e] = (bf16)(rs * (acc[mt][4 * g + e] - m1 - (float)xh4[e] * m2));
      *reinterpret_cast<bf16x4*>(t_dx + at) = dx4;
    }
  }
  __syncthreads();
  if (g_dx) tile_out(c, t_dx, g_dx, DM);
  // column sums: thread = (column, half of the rows); the upper half hands its sums over through LDS (red is free: the row
  // sums were consumed before the barrier above), so a workgroup issues ONE atomic per column and quantity - the adds of
  // all workgroups to one address serialise in the L2 (measured at 251 workgroups: 17 us of a 94 us launch with two)
  {
    const int col = c.tid & 255, half = c.tid >> 8, rows = 16 * MT;
    float cg = 0.f, cb = 0.f, cx = 0.f;
    for (int i = 0; i < rows; ++i) {
      const int row = half * rows + i;
      float v = (float)t_aux[row * AS + col];
      if (DROP && d.on()) {
        const uint32_t bits = d.bits(drop_counter_rc(c.row0 + row, col & ~3, DM));
        v = d.keep(bits, col & 3) ? v * d.scale : 0.f;
      }
      cb += v;
      cg += v * (float)t_xhat[row * AS + col];
      cx += (float)t_dx[row * AS + col];
    }
    float* xch = &red[0][0];          // 2 * NW * 32 * MT >= 768 floats for MT >= 2; MT = 1: 512 -> two rounds
    if (MT >= 2) {
      if (half) { xch[col] = cg; xch[256 + col] = cb; xch[512 + col] = cx; }
      __syncthreads();
      if (!half) {
        if (dgamma) atomicAdd(dgamma + col, cg + xch[col]);
        if (dbeta) atomicAdd(dbeta + col, cb + xch[256 + col]);
        if (dbias) atomicAdd(dbias + col, cx + xch[512 + col]);
      }
    } else {
      if (half) { xch[col] = cg; xch[256 + col] = cb; }
      __syncthreads();
      if (!half) {
        if (dgamma) atomicAdd(dgamma + col, cg + xch[col]);
        if (dbeta) atomicAdd(dbeta + col, cb + xch[256 + col]);
      }
      __syncthreads();
      if (half) xch[col] = cx;
      __syncthreads();
      if (!half && dbias) atomicAdd(dbias + col, cx + xch[col]);
    }
  }
}

template <bool HEAD, bool FFN, bool TAIL, bool DROP, int MT>
__global__ __launch_bounds__(512, 1) void row_chain_bwd_kernel(ChainBwdArgs a) {
  constexpr int RB = 32 * MT, TE = RB * AS;
  __shared__ __attribute__((aligned(16))) bf16 tiles[3 * TE];
  __shared__ float red[2][NW * 32 * MT];
  Ctx<MT> c;
  c.tid = threadIdx.x; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6); c.l = c.tid & 63; c.hi = c.l >> 5; c.r = c.l & 31;
  c.row0 = blockIdx.x * RB; c.nvalid = min(RB, a.M - c.row0);
  c.ws = a.wfrag + (size_t)c.wave * a.wave_frags * 64;
#pragma unroll
  for (int i = 0; i < Ring<MT>::D; ++i) c.ring[i] = c.ws[i * 64 + c.l];
  c.ws += Ring<MT>::D * 64;
  constexpr int NTOUCH = MT == 1 ? TOUCH : 1;
  int touched[NTOUCH];
  {       // warm-up of this (and, at decoder size, the next) chain's streams: see row_chain_kernel
    const int nlines = NW * (a.wave_frags + (MT == 1 ? a.next_frags : 0)) * 8;
    const int xcd = blockIdx.x & 7, nr = ((int)gridDim.x - xcd + 7) >> 3;
    const char* sb = reinterpret_cast<const char*>(a.wfrag);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NTOUCH; ++t) {
      const int ln = min(((int)blockIdx.x >> 3) * 512 + c.tid + t * nr * 512, nlines - 1);
      touched[t] = *reinterpret_cast<const int*>(sb + (size_t)ln * 128);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  bf16* t0 = tiles; bf16* t1 = tiles + TE; bf16* t2 = tiles + 2 * TE;
  const Drop da = make_drop(a.drop_a), off = make_drop(DropArgs{nullptr, 0u, 0, 1.f});
  bf16* cur;          // the running gradient ds
  bf16 *fa, *fb;      // the two other tiles

  if (HEAD) {
    // t0: the dP blocks one after the other, then ds_a; t1: G -> dy; t2: xhat_a
    // G, xhat_a and the first dP block are requested together (one global round trip, not three), then stored
    // (block u + 1 of dP is requested before the MFMAs on block u and stored after them: its latency is hidden)
    TileRegs<MT> nxt;
    {
      TileRegs<MT> rg, rx;
      if (a.G) tile_load(c, a.G, a.ldg, rg);
      tile_load(c, a.xhat_a, DM, rx);
      if (a.nb > 0) tile_load(c, a.dP, a.ldp, nxt);
      if (a.G) tile_store(c, rg, t1);
      else {
#pragma unroll
        for (int p = 0; p < 2 * MT; ++p) *reinterpret_cast<bf16x8*>(t1 + ((c.tid + p * 512) >> 5) * AS + ((c.tid + p * 512) & 31) * 8) = zero_bf8();
      }
      tile_store(c, rx, t2);
    }
    f32x16 acc[MT];
    zero_acc(acc);
    for (int u = 0; u < a.nb; ++u) {
      tile_store(c, nxt, t0);
      __syncthreads();
      if (u + 1 < a.nb) tile_load(c, a.dP + (u + 1) * 256, a.ldp, nxt);
      block_mma(c, t0, acc);
      __syncthreads();                           // every wave is past its MFMAs on this block of dP: t0 may be rewritten
    }
    if (a.nb == 0) __syncthreads();              // (bare LayerNorm backward: the G / xhat tiles must be visible)
    epi_lnbwd<DROP>(c, acc, t1, t2, t0, a.rstd_a, a.gamma_a, da, red, a.ds_a, a.dgamma_a, a.dbeta_a, a.dbias_a);
    cur = t0; fa = t1; fb = t2;
    __syncthreads();                             // the column pass has read t1 / t2: free from here
  } else {
    tile_in(c, a.DS, DM, t0);
    cur = t0; fa = t1; fb = t2;
    __syncthreads();
  }

  if (FFN) {
    const int dff = a.nc * 256;
    f32x16 acc2[MT];
    zero_acc(acc2);
    TileRegs<MT> xr;                             // xhat_b on its way to LDS (live in the last chunk only)
    // one hidden chunk; LAST: xhat_b is requested under the chunk's second block of MFMAs (the last chunk is peeled off the
    // loop so that its 8 MT registers are not live through the whole loop)
    auto chunk = [&](int ch, auto last) {
      bf16* hc = (ch & 1) ? fb : fa;             // rewritten two chunks later, the next chunk's barrier in between
      // this lane's ReLU / dropout mask bits of the chunk (st_row_chain wrote them), requested before the MFMAs
      const unsigned long long relu = a.relu_bits[((size_t)(blockIdx.x * a.nc + ch) * NW + c.wave) * 64 + c.l];
      f32x16 acc1[MT];
      zero_acc(acc1);
      block_mma(c, cur, acc1);                   // ds x W2[:, chunk]: the hidden gradient before the mask
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bf16 v = (bf16)(acc1[mt][4 * g + e] * a.mask_scale);
            const int b = mt * 16 + 4 * g + e;
            const bool on = b < 32 ? (((uint32_t)relu >> b) & 1u) : (((uint32_t)(relu >> 32) >> (b - 32)) & 1u);
            o[e] = on ? v : (bf16)0.f;
          }
          *reinterpret_cast<bf16x4*>(hc + (mt * 32 + c.r) * AS + c.wave * 32 + 8 * g + 4 * c.hi) = o;
        }
      __syncthreads();
      if (decltype(last)::value) tile_load(c, a.xhat_b, DM, xr);
      block_mma(c, hc, acc2);                    // dH chunk x W1[chunk, :]
      tile_out(c, hc, a.dH + ch * 256, dff);
    };
    for (int ch = 0; ch + 1 < a.nc; ++ch) chunk(ch, std::false_type{});
    chunk(a.nc - 1, std::true_type{});
    // dy = acc2 + ds (in place over the ds tile), xhat_b into the tile the last chunk did not use, ds_b into the other
    bf16* tx = (a.nc & 1) ? fb : fa;
    bf16* td = (a.nc & 1) ? fa : fb;
    tile_store(c, xr, tx);                       // (tx: last read one chunk earlier, behind the last chunk's barrier)
    __syncthreads();                             // xhat_b visible; every wave is past its MFMAs on the last chunk (td)
    epi_lnbwd<false>(c, acc2, cur, tx, td, a.rstd_b, a.gamma_b, off, red, a.ds_b, a.dgamma_b, a.dbeta_b, a.dbias_b);
    fa = cur; fb = tx; cur = td;
    __syncthreads();                             // the column pass has read fa / fb
  }

  if (TAIL) {
    // O -> fa, Ores -> fb (requested before the MFMAs, stored after them); dctx is staged in the ds tile once every wave is
    // past its MFMAs on it
    TileRegs<MT> ro, rr_;
    tile_load(c, a.O, a.ldo, ro);
    if (a.Ores) tile_load(c, a.Ores, a.ldo, rr_);
    f32x16 acc[MT];
    zero_acc(acc);
    block_mma(c, cur, acc);
    tile_store(c, ro, fa);
    if (a.Ores) tile_store(c, rr_, fb);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = mt * 32 + c.r;
      float part = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int at = row * AS + c.wave * 32 + 8 * g + 4 * c.hi;
        const bf16x4 o4 = *reinterpret_cast<const bf16x4*>(fa + at);
        bf16x4 r4 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        if (a.Ores) r4 = *reinterpret_cast<const bf16x4*>(fb + at);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (bf16)acc[mt][4 * g + e];
          part += (float)o[e] * ((float)o4[e] + (float)r4[e]);
        }
        *reinterpret_cast<bf16x4*>(cur + at) = o;
      }
      part += wave_xor32(part);
      if (c.hi == 0) red[0][(c.wave * MT + mt) * 32 + c.r] = part;     // this wave's 32 columns of the row: half a head
    }
    __syncthreads();
    tile_out(c, cur, a.dctx, a.lddc);
    // delta[h][row]: heads are 64 columns = two waves
    for (int i = c.tid; i < 4 * RB; i += 512) {
      const int h = i / RB, row = i % RB, mt = row >> 5, r = row & 31;
      if (row < c.nvalid)
        a.delta[(size_t)h * a.M + c.row0 + row] = red[0][((2 * h) * MT + mt) * 32 + r] + red[0][((2 * h + 1) * MT + mt) * 32 + r];
    }
  }
  {
    int tsum = 0;
#pragma unroll
    for (int t = 0; t < NTOUCH; ++t) tsum ^= touched[t];
    if (tsum == 0x5a5a5a5a && a.M < 0) red[0][0] = 1.f;
  }
}

}  // namespace
#include "st_rowchain_pipe_bwd.cuh"
#include "st_rowchain_pipe512_bwd.cuh"
namespace {
// Block descriptor table of st_wfrag_build: 4 x int64 per 256 x 256 weight block
//   [0] address of the block's first element (row n0, column k0 of a row-major bf16 matrix)
//   [1] leading dimension of that matrix (elements) | transposed << 32: the block is read as its TRANSPOSE (the data
//       gradient's operand: output index = the weight's column, contraction over its rows)
//   [2] destination: fragment index of the block inside a wave's stream (block position * 16) | (wave stride in
//       fragments) << 32
//   [3] destination: ADDRESS of the chain's wave-0 stream (so one table - one launch - can fill several buffers)
// piece (block, wave, ks, lane) = 8 consecutive k of weight row n0 + wave*32 + (lane & 31): the MFMA A operand of that lane.
__global__ __launch_bounds__(256) void wfrag_build_kernel(const long long* __restrict__ table) {
  const long long* d = table + (size_t)blockIdx.x * 4;
  const bf16* src = reinterpret_cast<const bf16*>(d[0]);
  const long long ld = d[1] & 0xffffffffll, frag0 = d[2] & 0xffffffffll, wstride = d[2] >> 32;
  const bool transposed = (d[1] >> 32) & 1;
  const int wave = blockIdx.y;
  bf16x8* dst = reinterpret_cast<bf16x8*>(d[3]) + ((size_t)wave * wstride + frag0) * 64;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int id = p * 256 + threadIdx.x, ks = id >> 6, lane = id & 63;
    const int o = wave * 32 + (lane & 31), c0 = ks * 16 + (lane >> 5) * 8;      // output index, first contraction index
    if (!transposed) dst[id] = *reinterpret_cast<const bf16x8*>(src + (size_t)o * ld + c0);
    else {
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(c0 + j) * ld + o];      // (coalesced across the lanes' consecutive o)
      dst[id] = v;
    }
  }
}

}  // namespace

extern "C" int st_wfrag_depth(void) { return DEPTH; }
#ifdef ST_DEV_TRACE
extern "C" int st_dev_chain_trace(long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_chain_trace), &p, sizeof(p)); }
#endif

extern "C" int st_wfrag_build(hipStream_t stream, const long long* table, int n_blocks) {
  if (n_blocks <= 0) return 0;
  if (!table) return -1;
  hipLaunchKernelGGL(wfrag_build_kernel, dim3(n_blocks, NW), dim3(256), 0, stream, table);
  ST_CHECK_LAUNCH();
  return 0;
}

namespace {
// The backward chain with the feed-forward's hidden dimension over nc workgroups per 32-row block - the mirror image of
// row_chain_split_kernel: workgroup (block, part) runs HEAD (replicated; part 0 writes ds_a and adds the column sums),
// chunk `part` (dH chunk = (ds W2[:, chunk]) masked; its contribution dH W1[chunk, :] to dy), leaves the 32 x 256 fp32
// partial in the scratch and draws the block's ticket; the last one adds the partials in chunk order and runs the second
// LayerNorm backward (+ its column sums) and TAIL.
template <bool HEAD, bool TAIL, bool DROP>
__global__ __launch_bounds__(512, 1) void row_chain_bwd_split_kernel(ChainBwdArgs a) {
  constexpr int MT = 1, RB = 32, TE = RB * AS;
  __shared__ __attribute__((aligned(16))) bf16 tiles[3 * TE];
  __shared__ float red[2][NW * 32 * MT];
  __shared__ bool last;
  Ctx<MT> c;
  c.tid = threadIdx.x; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6); c.l = c.tid & 63; c.hi = c.l >> 5; c.r = c.l & 31;
  const int parts = a.split_parts, cpp = a.nc / parts, block = blockIdx.x / parts, part = blockIdx.x - block * parts;
  c.row0 = block * RB; c.nvalid = min(RB, a.M - c.row0);
  const bf16x8* base = a.wfrag + (size_t)c.wave * a.wave_frags * 64;
  auto at_block = [&](int b) { return base + (size_t)b * 16 * 64; };
  const int nbh = HEAD ? a.nb : 0, b_chunk = nbh + 2 * part * cpp, b_tail = nbh + 2 * a.nc;
  c.ws = at_block(nbh > 0 ? 0 : b_chunk);
#pragma unroll
  for (int i = 0; i < Ring<MT>::D; ++i) c.ring[i] = c.ws[i * 64 + c.l];
  c.ws += Ring<MT>::D * 64;
  int touched[TOUCH];
  {
    const int nlines = NW * (a.wave_frags + a.next_frags) * 8;
    const int xcd = blockIdx.x & 7, nr = ((int)gridDim.x - xcd + 7) >> 3;
    const char* sb = reinterpret_cast<const char*>(a.wfrag);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < TOUCH; ++t) {
      const int ln = min(((int)blockIdx.x >> 3) * 512 + c.tid + t * nr * 512, nlines - 1);
      touched[t] = *reinterpret_cast<const int*>(sb + (size_t)ln * 128);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  bf16* t0 = tiles; bf16* t1 = tiles + TE; bf16* t2 = tiles + 2 * TE;
  const Drop da = make_drop(a.drop_a), off = make_drop(DropArgs{nullptr, 0u, 0, 1.f});
  bf16* cur;
  bf16 *fa, *fb;
  if (HEAD) {
    TileRegs<MT> nxt;
    {
      TileRegs<MT> rg, rx;
      if (a.G) tile_load(c, a.G, a.ldg, rg);
      tile_load(c, a.xhat_a, DM, rx);
      if (a.nb > 0) tile_load(c, a.dP, a.ldp, nxt);
      if (a.G) tile_store(c, rg, t1);
      else {
#pragma unroll
        for (int p = 0; p < 2 * MT; ++p) *reinterpret_cast<bf16x8*>(t1 + ((c.tid + p * 512) >> 5) * AS + ((c.tid + p * 512) & 31) * 8) = zero_bf8();
      }
      tile_store(c, rx, t2);
    }
    f32x16 acc[MT];
    zero_acc(acc);
    for (int u = 0; u < a.nb; ++u) {
      tile_store(c, nxt, t0);
      __syncthreads();
      if (u + 1 < a.nb) tile_load(c, a.dP + (u + 1) * 256, a.ldp, nxt);
      if (u + 1 == a.nb) c.ws = at_block(b_chunk);          // (the block after HEAD's last: this workgroup's chunk)
      block_mma(c, t0, acc);
      __syncthreads();
    }
    if (a.nb == 0) __syncthreads();
    const bool w = part == 0;
    epi_lnbwd<DROP>(c, acc, t1, t2, t0, a.rstd_a, a.gamma_a, da, red, w ? a.ds_a : nullptr, w ? a.dgamma_a : nullptr,
                    w ? a.dbeta_a : nullptr, w ? a.dbias_a : nullptr);
    cur = t0; fa = t1; fb = t2;
    __syncthreads();
  } else {
    tile_in(c, a.DS, DM, t0);
    cur = t0; fa = t1; fb = t2;
    __syncthreads();
  }
  f32x16 acc2[MT];
  zero_acc(acc2);
  TileRegs<MT> xr;
  {
    const int dff = a.nc * 256;
    for (int j = 0; j < cpp; ++j) {
      const int ch = part * cpp + j;
      bf16* hc = (j & 1) ? fb : fa;            // (as row_chain_bwd_kernel: rewritten two chunks later, the next chunk's barrier in between)
      const unsigned long long relu = a.relu_bits[((size_t)(block * a.nc + ch) * NW + c.wave) * 64 + c.l];
      f32x16 acc1[MT];
      zero_acc(acc1);
      block_mma(c, cur, acc1);                   // ds x W2[:, chunk]; refills: W1[chunk, :] right behind
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bf16 v = (bf16)(acc1[0][4 * g + e] * a.mask_scale);
          const int b = 4 * g + e;
          o[e] = (((uint32_t)relu >> b) & 1u) ? v : (bf16)0.f;
        }
        *reinterpret_cast<bf16x4*>(hc + c.r * AS + c.wave * 32 + 8 * g + 4 * c.hi) = o;
      }
      __syncthreads();
      if (j + 1 == cpp) {
        tile_load(c, a.xhat_b, DM, xr);          // (used by the last arriver only; asked for by all: nobody knows yet)
        c.ws = at_block(b_tail);
      }
      block_mma(c, hc, acc2);                    // dH chunk x W1[chunk, :]
      tile_out(c, hc, a.dH + ch * 256, dff);
    }
  }
  float* slot = a.split_ws + (size_t)block * parts * (512 * 16);
  {
    float* mine = slot + (size_t)part * (512 * 16);
    unsigned long long* mine8 = reinterpret_cast<unsigned long long*>(mine);      // (two floats per 8-byte access)
#pragma unroll
    for (int i = 0; i < 16; i += 2)
      __hip_atomic_store(mine8 + (i / 2) * 512 + c.tid,
                         ((unsigned long long)__float_as_uint(acc2[0][i + 1]) << 32) | __float_as_uint(acc2[0][i]), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    ST_PUBLISH_FENCE();
    __builtin_amdgcn_s_waitcnt(0);
  }
  __syncthreads();
  if (c.tid == 0)
    last = __hip_atomic_fetch_add(a.split_tickets + block, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(parts - 1);
  __syncthreads();
  if (last) {
    ST_MERGER_FENCE();
    if (c.tid == 0) __hip_atomic_store(a.split_tickets + block, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    zero_acc(acc2);
    for (int p = 0; p < parts; ++p) {
      const unsigned long long* src8 = reinterpret_cast<const unsigned long long*>(slot + (size_t)p * (512 * 16));
      unsigned long long v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __hip_atomic_load(src8 + i * 512 + c.tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc2[0][2 * i] += __uint_as_float((unsigned)v[i]);
        acc2[0][2 * i + 1] += __uint_as_float((unsigned)(v[i] >> 32));
      }
    }
    // dy = acc2 + ds (in place over the ds tile), xhat_b into the tile the last chunk did not use, ds_b into the other (both
    // behind the merge's barriers: every wave is past its MFMAs and copies on them)
    bf16* tx = (cpp & 1) ? fb : fa;
    bf16* td = (cpp & 1) ? fa : fb;
    tile_store(c, xr, tx);
    __syncthreads();
    epi_lnbwd<false>(c, acc2, cur, tx, td, a.rstd_b, a.gamma_b, off, red, a.ds_b, a.dgamma_b, a.dbeta_b, a.dbias_b);
    fa = cur; fb = tx; cur = td;
    __syncthreads();
    if (TAIL) {
      TileRegs<MT> ro, rr_;
      tile_load(c, a.O, a.ldo, ro);
      if (a.Ores) tile_load(c, a.Ores, a.ldo, rr_);
      f32x16 acc[MT];
      zero_acc(acc);
      block_mma(c, cur, acc);
      tile_store(c, ro, fa);
      if (a.Ores) tile_store(c, rr_, fb);
      __syncthreads();
      {
        const int row = c.r;
        float part_ = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int at = row * AS + c.wave * 32 + 8 * g + 4 * c.hi;
          const bf16x4 o4 = *reinterpret_cast<const bf16x4*>(fa + at);
          bf16x4 r4 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
          if (a.Ores) r4 = *reinterpret_cast<const bf16x4*>(fb + at);
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = (bf16)acc[0][4 * g + e];
            part_ += (float)o[e] * ((float)o4[e] + (float)r4[e]);
          }
          *reinterpret_cast<bf16x4*>(cur + at) = o;
        }
        part_ += wave_xor32(part_);
        if (c.hi == 0) red[0][c.wave * 32 + c.r] = part_;
      }
      __syncthreads();
      tile_out(c, cur, a.dctx, a.lddc);
      for (int i = c.tid; i < 4 * RB; i += 512) {
        const int h = i / RB, row = i % RB;
        if (row < c.nvalid) a.delta[(size_t)h * a.M + c.row0 + row] = red[0][(2 * h) * 32 + row] + red[0][(2 * h + 1) * 32 + row];
      }
    }
  }
  {
    int tsum = 0;
#pragma unroll
    for (int t = 0; t < TOUCH; ++t) tsum ^= touched[t];
    if (tsum == 0x5a5a5a5a && a.M < 0) red[0][0] = 1.f;
  }
}
}  // namespace

namespace {
// row tiles per workgroup: the smallest of 1, 2, 3 that gives at most one round of workgroups (256 CUs)
// row tiles of 32 per workgroup (1, 2 or 3): the choice that needs the least time for M rows at one workgroup per CU -
// rounds of 256 workgroups x the time a workgroup of that height takes (forward chain, measured: 27 / 41 / 59 us; the
// backward's ratios are the same).  One round of the tallest tile that fits wins (24,060 rows: 251 x 96), but just past a
// full round the shorter tile's two rounds beat the taller one's (26,880 rows: 94.5 us with 96-row tiles, 75.1 with 64-row
// ones; 32,000: 97.2 vs 84.2).
int row_tiles(int M) {
  static const int force = [] { const char* e = getenv("ST_CHAIN_MT"); return e ? atoi(e) : 0; }();      // development switch
  if (force >= 1 && force <= 3) return force;
  const int cost[4] = {0, 27, 41, 59};
  int best = 1, best_cost = 1 << 30;
  for (int mt = 1; mt <= 3; ++mt) {
    const int tiles = (M + 32 * mt - 1) / (32 * mt), c = ((tiles + 255) / 256) * cost[mt];
    if (c < best_cost) { best = mt; best_cost = c; }
  }
  return best;
}
}  // namespace

namespace {
// Workgroups per 32-row block that share the feed-forward's hidden dimension (row_chain_split_kernel / _bwd_split_kernel): the
// largest divisor of nc (the 256-column chunks of d_ff, at most 8) that keeps blocks x parts within one round of the chip
// (256 workgroups, and the ticket table's 256 entries); 0 = no split.  Decoder-sized M: parts = nc (1,206 rows: 38 x 4); a
// 4-utterance shard's encoder (3,120 rows = 98 blocks): 2 - two chunks per part.
int split_parts_for(int blocks, int nc) {
  if (nc < 2 || nc > 8 || blocks > 128) return 0;
  for (int p = nc; p >= 2; --p)
    if (nc % p == 0 && blocks * p <= 256) return p;
  return 0;
}
}  // namespace

// 64-bit words of the relu_bits buffer st_row_chain writes and st_row_chain_bwd reads for M rows and this d_ff
extern "C" int st_row_chain_mask_words(int M, int d_ff) {
  if (M <= 0 || d_ff <= 0) return 0;
  const int mt = row_tiles(M);
  return ((M + 32 * mt - 1) / (32 * mt)) * (d_ff / 256) * NW * 64;
}

extern "C" int st_row_chain(hipStream_t stream, int M, const void* wfrag, int n_blocks, int next_blocks, float eps, const void* A, int lda,
                            const void* R, int ldr, const float* bo, const float* g0, const float* be0, void* out0, void* xhat0,
                            float* rstd0, int d_ff, const float* b1, const float* b2, const float* g1, const float* be1, void* H,
                            unsigned long long* relu_bits, void* out1, void* xhat1, float* rstd1, const unsigned* drop_seed, unsigned drop1_salt,
                            int drop1_thresh, float drop1_scale, unsigned drop2_salt, int drop2_thresh, float drop2_scale,
                            int post_blocks, const float* bp, void* P, int ldp, void* split_work, long long split_bytes,
                            float post_kscale) {
  if (M <= 0) return 0;
  const bool pre = R != nullptr, ffn = d_ff > 0, post = post_blocks > 0;
  if (!A || !wfrag || (lda & 7) || (!pre && !ffn && !post)) return -1;
  if (pre && ((ldr & 7) || !bo || !g0 || !be0)) return -2;      // (out0 may be NULL: nobody outside the chain reads it - inference)
  if (ffn && ((d_ff & 255) || !b1 || !b2 || !g1 || !be1 || !out1)) return -3;
  if (post && (!bp || !P || (ldp & 7) || ldp < 256 * post_blocks)) return -4;
  if (n_blocks != (pre ? 1 : 0) + (ffn ? 2 * (d_ff / 256) : 0) + post_blocks) return -5;
  ChainArgs a;
  a.M = M; a.wfrag = (const bf16x8*)wfrag; a.wave_frags = n_blocks * 16 + DEPTH; a.eps = eps;
  a.next_frags = next_blocks > 0 ? next_blocks * 16 + DEPTH : 0;
  a.A = (const bf16*)A; a.lda = lda; a.R = (const bf16*)R; a.ldr = ldr; a.bo = bo; a.g0 = g0; a.be0 = be0;
  a.out0 = (bf16*)out0; a.xhat0 = (bf16*)xhat0; a.rstd0 = rstd0;
  a.nc = d_ff / 256; a.b1 = b1; a.b2 = b2; a.g1 = g1; a.be1 = be1; a.H = (bf16*)H; a.relu_bits = relu_bits; a.out1 = (bf16*)out1; a.xhat1 = (bf16*)xhat1;
  a.rstd1 = rstd1;
  const bool on1 = ffn && drop_seed && drop1_thresh > 0, on2 = ffn && drop_seed && drop2_thresh > 0;
  a.drop1.seed = on1 ? drop_seed : nullptr; a.drop1.salt = drop1_salt; a.drop1.thresh = on1 ? drop1_thresh : 0;
  a.drop1.scale = on1 ? drop1_scale : 1.f;
  a.drop2.seed = on2 ? drop_seed : nullptr; a.drop2.salt = drop2_salt; a.drop2.thresh = on2 ? drop2_thresh : 0;
  a.drop2.scale = on2 ? drop2_scale : 1.f;
  a.nb = post_blocks; a.bp = bp; a.P = (bf16*)P; a.ldp = ldp;
  a.post_kscale = (post_blocks == 3 && post_kscale > 0.f) ? post_kscale : 1.f;      // (0 = not given)
  const int mt = row_tiles(M);
  const dim3 grid((M + 32 * mt - 1) / (32 * mt)), blk(512);
  const bool drop = on1 || on2;
  // the feed-forward's hidden dimension over nc workgroups per row block (row_chain_split_kernel): decoder-sized M only
  a.split_ws = nullptr; a.split_tickets = nullptr; a.split_parts = 0;
  const int sp = split_parts_for((int)grid.x, a.nc);
  if (split_work && ffn && mt == 1 && sp >= 2) {
    // layout: 256 tickets (one per row block; at a FIXED place - launches of different M share the scratch, and a ticket
    // must never lie where another launch leaves partial sums), then the partials
    const size_t words = (size_t)grid.x * sp * 512 * 16;
    if (split_bytes < (long long)((256 + words) * 4)) return -6;
    a.split_tickets = (unsigned*)split_work;
    a.split_ws = (float*)split_work + 256;
    a.split_parts = sp;
    const dim3 sgrid(grid.x * sp);
    const size_t vec_bytes = (size_t)(6 + a.nc + a.nb) * 1024;      // the epilogue vectors' LDS copy (ChainVecs)
    if (vec_bytes > 48 * 1024) return -7;
#define ST_SPLIT(PRE_, POST_)                                                                                        \
  do {                                                                                                               \
    if (drop) hipLaunchKernelGGL((row_chain_split_kernel<PRE_, POST_, true>), sgrid, blk, vec_bytes, stream, a);     \
    else hipLaunchKernelGGL((row_chain_split_kernel<PRE_, POST_, false>), sgrid, blk, vec_bytes, stream, a);         \
  } while (0)
    if (pre && post) ST_SPLIT(true, true);
    else if (pre) ST_SPLIT(true, false);
    else if (post) ST_SPLIT(false, true);
    else ST_SPLIT(false, false);
#undef ST_SPLIT
    ST_CHECK_LAUNCH();
    return 0;
  }
  // 64- / 96-row workgroups, the encoder's shapes (PRE + FFN [+ POST], POST alone): the pipelined kernel (st_rowchain_pipe.cuh)
  static const bool pipe_on = [] { const char* e = getenv("ST_CHAIN_PIPE"); return !(e && (e[0] == '0' || e[0] == 'b')); }();      // development switch
  if (pipe_on && mt >= 2 && ((pre && ffn && (post_blocks == 0 || post_blocks == 1 || post_blocks == 3)) || (!pre && !ffn && post_blocks == 3))) {
#ifdef ST_DEV_CHAIN_NULL
  {       // development: which saved tensors does the launch's time hang on? (bit 0 H, 1 xhat, 2 out0, 3 P, 4 out1, 5 mask bits)
    static const int nul = [] { const char* e = getenv("ST_CHAIN_NULL"); return e ? atoi(e) : 0; }();
    if (nul & 1) a.H = nullptr;
    if (nul & 2) a.xhat0 = a.xhat1 = nullptr;
    if (nul & 4) a.out0 = nullptr;
    if (nul & 8) a.P = nullptr;
    if (nul & 16) a.out1 = nullptr;
    if (nul & 32) a.relu_bits = nullptr;
  }
#endif
#define ST_PIPE(PRE_, FFN_, NB_)                                                                                      \
  do {                                                                                                                \
    if (mt == 3) {                                                                                                    \
      if (drop) hipLaunchKernelGGL((row_chain_pipe_kernel<PRE_, FFN_, NB_, true, 3>), grid, blk, 0, stream, a);       \
      else hipLaunchKernelGGL((row_chain_pipe_kernel<PRE_, FFN_, NB_, false, 3>), grid, blk, 0, stream, a);           \
    } else {                                                                                                          \
      if (drop) hipLaunchKernelGGL((row_chain_pipe_kernel<PRE_, FFN_, NB_, true, 2>), grid, blk, 0, stream, a);       \
      else hipLaunchKernelGGL((row_chain_pipe_kernel<PRE_, FFN_, NB_, false, 2>), grid, blk, 0, stream, a);           \
    }                                                                                                                 \
  } while (0)
    if (pre && post_blocks == 3) ST_PIPE(true, true, 3);
    else if (pre && post_blocks == 1) ST_PIPE(true, true, 1);
    else if (pre) ST_PIPE(true, true, 0);
    else if (mt == 3) hipLaunchKernelGGL((row_chain_pipe_kernel<false, false, 3, false, 3>), grid, blk, 0, stream, a);
    else hipLaunchKernelGGL((row_chain_pipe_kernel<false, false, 3, false, 2>), grid, blk, 0, stream, a);
#undef ST_PIPE
    ST_CHECK_LAUNCH();
    return 0;
  }
  const size_t vec1 = (size_t)(6 + a.nc + a.nb) * 1024;      // 32- / 64-row workgroups: the epilogue vectors' LDS copy (ChainVecs)
  if (mt <= 2 && vec1 > 48 * 1024) return -7;      // (d_ff up to 9,984)
#define ST_CHAIN(PRE_, FFN_, POST_)                                                                               \
  do {                                                                                                            \
    if (mt == 3) {                                                                                                \
      if (drop) hipLaunchKernelGGL((row_chain_kernel<PRE_, FFN_, POST_, true, 3>), grid, blk, 0, stream, a);      \
      else hipLaunchKernelGGL((row_chain_kernel<PRE_, FFN_, POST_, false, 3>), grid, blk, 0, stream, a);          \
    } else if (mt == 2) {                                                                                         \
      if (drop) hipLaunchKernelGGL((row_chain_kernel<PRE_, FFN_, POST_, true, 2>), grid, blk, vec1, stream, a);   \
      else hipLaunchKernelGGL((row_chain_kernel<PRE_, FFN_, POST_, false, 2>), grid, blk, vec1, stream, a);       \
    } else {                                                                                                      \
      if (drop) hipLaunchKernelGGL((row_chain_kernel<PRE_, FFN_, POST_, true, 1>), grid, blk, vec1, stream, a);   \
      else hipLaunchKernelGGL((row_chain_kernel<PRE_, FFN_, POST_, false, 1>), grid, blk, vec1, stream, a);       \
    }                                                                                                             \
  } while (0)
  if (pre && ffn && post) ST_CHAIN(true, true, true);
  else if (pre && ffn) ST_CHAIN(true, true, false);
  else if (pre && post) ST_CHAIN(true, false, true);
  else if (ffn && post) ST_CHAIN(false, true, true);
  else if (ffn) ST_CHAIN(false, true, false);
  else if (pre) ST_CHAIN(true, false, false);
  else ST_CHAIN(false, false, true);
#undef ST_CHAIN
  ST_CHECK_LAUNCH();
  return 0;
}

// ---- d_model = 512 (st_rowchain_pipe512.cuh): PRE + FFN [+ the six 256-column blocks of a q | k | v projection], 64-row workgroups
extern "C" int st_row_chain512_mask_words(int M, int d_ff) {
  if (M <= 0 || d_ff <= 0) return 0;
  return ((M + 63) / 64) * (d_ff / 256) * NW * 64;
}

extern "C" int st_row_chain512(hipStream_t stream, int M, const void* wfrag, int n_blocks, int next_blocks, float eps, const void* A, int lda,
                               const void* R, int ldr, const float* bo, const float* g0, const float* be0, void* out0, void* xhat0,
                               float* rstd0, int d_ff, const float* b1, const float* b2, const float* g1, const float* be1, void* H,
                               unsigned long long* relu_bits, void* out1, void* xhat1, float* rstd1, const unsigned* drop_seed,
                               unsigned drop1_salt, int drop1_thresh, float drop1_scale, unsigned drop2_salt, int drop2_thresh,
                               float drop2_scale, int post_blocks, const float* bp, void* P, int ldp, float post_kscale) {
  if (M <= 0) return 0;
  if (!A || !wfrag || (lda & 7) || !R || (ldr & 7) || !bo || !g0 || !be0) return -1;
  if (d_ff <= 0 || (d_ff & 255) || !b1 || !b2 || !g1 || !be1 || !out1) return -3;
  if (post_blocks != 0 && post_blocks != 6) return -4;
  if (post_blocks && (!bp || !P || (ldp & 7) || ldp < 256 * post_blocks)) return -4;
  if (n_blocks != 4 + 4 * (d_ff / 256) + 2 * post_blocks) return -5;
  ChainArgs a;
  a.M = M; a.wfrag = (const bf16x8*)wfrag; a.wave_frags = n_blocks * 16 + DEPTH; a.eps = eps;
  a.next_frags = next_blocks > 0 ? next_blocks * 16 + DEPTH : 0;
  a.A = (const bf16*)A; a.lda = lda; a.R = (const bf16*)R; a.ldr = ldr; a.bo = bo; a.g0 = g0; a.be0 = be0;
  a.out0 = (bf16*)out0; a.xhat0 = (bf16*)xhat0; a.rstd0 = rstd0;
  a.nc = d_ff / 256; a.b1 = b1; a.b2 = b2; a.g1 = g1; a.be1 = be1; a.H = (bf16*)H; a.relu_bits = relu_bits; a.out1 = (bf16*)out1;
  a.xhat1 = (bf16*)xhat1; a.rstd1 = rstd1;
  const bool on1 = drop_seed && drop1_thresh > 0, on2 = drop_seed && drop2_thresh > 0;
  a.drop1.seed = on1 ? drop_seed : nullptr; a.drop1.salt = drop1_salt; a.drop1.thresh = on1 ? drop1_thresh : 0;
  a.drop1.scale = on1 ? drop1_scale : 1.f;
  a.drop2.seed = on2 ? drop_seed : nullptr; a.drop2.salt = drop2_salt; a.drop2.thresh = on2 ? drop2_thresh : 0;
  a.drop2.scale = on2 ? drop2_scale : 1.f;
  a.nb = post_blocks; a.bp = bp; a.P = (bf16*)P; a.ldp = ldp;
  a.post_kscale = (post_blocks == 6 && post_kscale > 0.f) ? post_kscale : 1.f;
  a.split_ws = nullptr; a.split_tickets = nullptr; a.split_parts = 0;
  const dim3 grid((M + 63) / 64), blk(512);
  const bool drop = on1 || on2;
  if (post_blocks) {
    if (drop) hipLaunchKernelGGL((row_chain512_kernel<true, true>), grid, blk, 0, stream, a);
    else hipLaunchKernelGGL((row_chain512_kernel<true, false>), grid, blk, 0, stream, a);
  } else {
    if (drop) hipLaunchKernelGGL((row_chain512_kernel<false, true>), grid, blk, 0, stream, a);
    else hipLaunchKernelGGL((row_chain512_kernel<false, false>), grid, blk, 0, stream, a);
  }
  ST_CHECK_LAUNCH();
  return 0;
}

extern "C" int st_row_chain_bwd(hipStream_t stream, int M, const void* wfrag, int n_blocks, int next_blocks,
                                int head_blocks, const void* dP, int ldp, const void* G, int ldg, const void* xhat_a,
                                const float* rstd_a, const float* gamma_a, const unsigned* drop_seed, unsigned drop_salt,
                                int drop_thresh, float drop_scale, void* ds_a, float* dgamma_a, float* dbeta_a, float* dbias_a,
                                const void* DS, int d_ff, const unsigned long long* relu_bits, float mask_scale, void* dH,
                                const void* xhat_b,
                                const float* rstd_b, const float* gamma_b, void* ds_b, float* dgamma_b, float* dbeta_b,
                                float* dbias_b, const void* O, const void* Ores, int ldo, void* dctx, int lddc, float* delta,
                                void* split_work, long long split_bytes, float* colsum_ws, long long colsum_bytes) {
  if (M <= 0) return 0;
  // HEAD is present iff xhat_a is given; with head_blocks == 0 it is the bare LayerNorm backward of G (the gradient that
  // reaches the LAST sublayer of a stack from outside)
  const bool head = xhat_a != nullptr, ffn = d_ff > 0, tail = O != nullptr;
  if (!wfrag || head_blocks < 0 || (!head && !ffn && !tail)) return -1;
  if (head && ((head_blocks > 0 && (!dP || (ldp & 7) || ldp < 256 * head_blocks)) || (head_blocks == 0 && !G) || (G && (ldg & 7)) ||
               !rstd_a || !gamma_a || !ds_a))
    return -2;
  if (!head && head_blocks > 0) return -2;
  if (!head && !DS) return -2;
  if (ffn && ((d_ff & 255) || !relu_bits || !dH || !xhat_b || !rstd_b || !gamma_b || !ds_b)) return -3;
  if (tail && ((ldo & 7) || !dctx || (lddc & 7) || !delta)) return -4;
  if (n_blocks != head_blocks + (ffn ? 2 * (d_ff / 256) : 0) + (tail ? 1 : 0)) return -5;
  ChainBwdArgs a;
  a.M = M; a.wfrag = (const bf16x8*)wfrag; a.wave_frags = n_blocks * 16 + DEPTH;
  a.next_frags = next_blocks > 0 ? next_blocks * 16 + DEPTH : 0;
  a.nb = head_blocks; a.dP = (const bf16*)dP; a.ldp = ldp; a.G = (const bf16*)G; a.ldg = ldg; a.xhat_a = (const bf16*)xhat_a;
  a.rstd_a = rstd_a; a.gamma_a = gamma_a;
  const bool drop = head && drop_seed != nullptr && drop_thresh > 0;
  a.drop_a.seed = drop ? drop_seed : nullptr; a.drop_a.salt = drop_salt; a.drop_a.thresh = drop ? drop_thresh : 0;
  a.drop_a.scale = drop ? drop_scale : 1.f;
  a.ds_a = (bf16*)ds_a; a.dgamma_a = dgamma_a; a.dbeta_a = dbeta_a; a.dbias_a = dbias_a; a.DS = (const bf16*)DS;
  a.nc = d_ff / 256; a.relu_bits = relu_bits; a.mask_scale = mask_scale > 0.f ? mask_scale : 1.f; a.dH = (bf16*)dH;
  a.xhat_b = (const bf16*)xhat_b; a.rstd_b = rstd_b; a.gamma_b = gamma_b; a.ds_b = (bf16*)ds_b; a.dgamma_b = dgamma_b;
  a.dbeta_b = dbeta_b; a.dbias_b = dbias_b;
  a.O = (const bf16*)O; a.Ores = (const bf16*)Ores; a.ldo = ldo; a.dctx = (bf16*)dctx; a.lddc = lddc; a.delta = delta;
  const int mt = row_tiles(M);
  const dim3 grid((M + 32 * mt - 1) / (32 * mt)), blk(512);
  a.split_ws = nullptr; a.split_tickets = nullptr; a.split_parts = 0; a.colsum_ws = nullptr;
  const int sp = split_parts_for((int)grid.x, a.nc);
  if (split_work && ffn && mt == 1 && sp >= 2) {      // see st_row_chain
    const size_t words = (size_t)grid.x * sp * 512 * 16;
    if (split_bytes < (long long)((256 + words) * 4)) return -6;
    a.split_tickets = (unsigned*)split_work;
    a.split_ws = (float*)split_work + 256;
    a.split_parts = sp;
    const dim3 sgrid(grid.x * sp);
#define ST_BSPLIT(HEAD_, TAIL_)                                                                                        \
  do {                                                                                                                 \
    if (drop) hipLaunchKernelGGL((row_chain_bwd_split_kernel<HEAD_, TAIL_, true>), sgrid, blk, 0, stream, a);          \
    else hipLaunchKernelGGL((row_chain_bwd_split_kernel<HEAD_, TAIL_, false>), sgrid, blk, 0, stream, a);              \
  } while (0)
    if (head && tail) ST_BSPLIT(true, true);
    else if (head) ST_BSPLIT(true, false);
    else if (tail) ST_BSPLIT(false, true);
    else ST_BSPLIT(false, false);
#undef ST_BSPLIT
    ST_CHECK_LAUNCH();
    return 0;
  }
  static const bool pipe_on = [] { const char* e = getenv("ST_CHAIN_PIPE"); return !(e && (e[0] == '0' || e[0] == 'f')); }();      // development switch
  if (pipe_on && mt >= 2 && head && ffn && tail) {      // the encoder's shape: st_rowchain_pipe_bwd.cuh
    if (colsum_ws) {
      if (colsum_bytes < (long long)grid.x * 1536 * 4) return -6;
      a.colsum_ws = colsum_ws;
    }
    if (mt == 3) {
      if (drop) hipLaunchKernelGGL((row_chain_bwd_pipe_kernel<true, 3>), grid, blk, 0, stream, a);
      else hipLaunchKernelGGL((row_chain_bwd_pipe_kernel<false, 3>), grid, blk, 0, stream, a);
    } else {
      if (drop) hipLaunchKernelGGL((row_chain_bwd_pipe_kernel<true, 2>), grid, blk, 0, stream, a);
      else hipLaunchKernelGGL((row_chain_bwd_pipe_kernel<false, 2>), grid, blk, 0, stream, a);
    }
    ST_CHECK_LAUNCH();
    return 0;
  }
#define ST_BWD(HEAD_, FFN_, TAIL_)                                                                                     \
  do {                                                                                                                 \
    if (mt == 3) {                                                                                                     \
      if (drop) hipLaunchKernelGGL((row_chain_bwd_kernel<HEAD_, FFN_, TAIL_, true, 3>), grid, blk, 0, stream, a);      \
      else hipLaunchKernelGGL((row_chain_bwd_kernel<HEAD_, FFN_, TAIL_, false, 3>), grid, blk, 0, stream, a);          \
    } else if (mt == 2) {                                                                                              \
      if (drop) hipLaunchKernelGGL((row_chain_bwd_kernel<HEAD_, FFN_, TAIL_, true, 2>), grid, blk, 0, stream, a);      \
      else hipLaunchKernelGGL((row_chain_bwd_kernel<HEAD_, FFN_, TAIL_, false, 2>), grid, blk, 0, stream, a);          \
    } else {                                                                                                           \
      if (drop) hipLaunchKernelGGL((row_chain_bwd_kernel<HEAD_, FFN_, TAIL_, true, 1>), grid, blk, 0, stream, a);      \
      else hipLaunchKernelGGL((row_chain_bwd_kernel<HEAD_, FFN_, TAIL_, false, 1>), grid, blk, 0, stream, a);          \
    }                                                                                                                  \
  } while (0)
  if (head && ffn && tail) ST_BWD(true, true, true);
  else if (head && ffn) ST_BWD(true, true, false);
  else if (head && tail) ST_BWD(true, false, true);
  else if (ffn && tail) ST_BWD(false, true, true);
  else if (ffn) ST_BWD(false, true, false);
  else if (head) ST_BWD(true, false, false);
  else ST_BWD(false, false, true);
#undef ST_BWD
  ST_CHECK_LAUNCH();
  return 0;
}

// ---- d_model = 512 (st_rowchain_pipe512_bwd.cuh): HEAD + FFN + TAIL, 64-row workgroups
extern "C" int st_row_chain512_bwd(hipStream_t stream, int M, const void* wfrag, int n_blocks, int next_blocks, int head_blocks,
                                   const void* dP, int ldp, const void* G, int ldg, const void* xhat_a, const float* rstd_a,
                                   const float* gamma_a, const unsigned* drop_seed, unsigned drop_salt, int drop_thresh,
                                   float drop_scale, void* ds_a, float* dgamma_a, float* dbeta_a, float* dbias_a, int d_ff,
                                   const unsigned long long* relu_bits, float mask_scale, void* dH, const void* xhat_b,
                                   const float* rstd_b, const float* gamma_b, void* ds_b, float* dgamma_b, float* dbeta_b,
                                   float* dbias_b, const void* O, const void* Ores, int ldo, void* dctx, int lddc, float* delta) {
  if (M <= 0) return 0;
  if (!wfrag || (head_blocks != 0 && head_blocks != 6)) return -1;
  if (!xhat_a || (head_blocks > 0 && (!dP || (ldp & 7) || ldp < 256 * head_blocks)) || (head_blocks == 0 && !G) || (G && (ldg & 7)) ||
      !rstd_a || !gamma_a || !ds_a)
    return -2;
  if (d_ff <= 0 || (d_ff & 255) || !relu_bits || !dH || !xhat_b || !rstd_b || !gamma_b || !ds_b) return -3;
  if (!O || (ldo & 7) || !dctx || (lddc & 7) || !delta) return -4;
  if (n_blocks != 2 * head_blocks + 4 * (d_ff / 256) + 4) return -5;
  ChainBwdArgs a;
  a.M = M; a.wfrag = (const bf16x8*)wfrag; a.wave_frags = n_blocks * 16 + DEPTH;
  a.next_frags = next_blocks > 0 ? next_blocks * 16 + DEPTH : 0;
  a.nb = head_blocks; a.dP = (const bf16*)dP; a.ldp = ldp; a.G = (const bf16*)G; a.ldg = ldg; a.xhat_a = (const bf16*)xhat_a;
  a.rstd_a = rstd_a; a.gamma_a = gamma_a;
  const bool drop = drop_seed != nullptr && drop_thresh > 0;
  a.drop_a.seed = drop ? drop_seed : nullptr; a.drop_a.salt = drop_salt; a.drop_a.thresh = drop ? drop_thresh : 0;
  a.drop_a.scale = drop ? drop_scale : 1.f;
  a.ds_a = (bf16*)ds_a; a.dgamma_a = dgamma_a; a.dbeta_a = dbeta_a; a.dbias_a = dbias_a; a.DS = nullptr;
  a.nc = d_ff / 256; a.relu_bits = relu_bits; a.mask_scale = mask_scale > 0.f ? mask_scale : 1.f; a.dH = (bf16*)dH;
  a.xhat_b = (const bf16*)xhat_b; a.rstd_b = rstd_b; a.gamma_b = gamma_b; a.ds_b = (bf16*)ds_b; a.dgamma_b = dgamma_b;
  a.dbeta_b = dbeta_b; a.dbias_b = dbias_b;
  a.O = (const bf16*)O; a.Ores = (const bf16*)Ores; a.ldo = ldo; a.dctx = (bf16*)dctx; a.lddc = lddc; a.delta = delta;
  a.split_ws = nullptr; a.split_tickets = nullptr; a.split_parts = 0;
  const dim3 grid((M + 63) / 64), blk(512);
  if (drop) hipLaunchKernelGGL((row_chain512_bwd_kernel<true>), grid, blk, 0, stream, a);
  else hipLaunchKernelGGL((row_chain512_bwd_kernel<false>), grid, blk, 0, stream, a);
  ST_CHECK_LAUNCH();
  return 0;
}

// ---- column sums of the pipelined backward chain: workspace rows -> gradients
// st_row_chain_bwd_colsum_rows: workgroups (= workspace rows of 6 x 256 floats) of the launch st_row_chain_bwd would make for this
// shape if it takes a column-sum workspace, 0 if that launch adds its column sums atomically (decoder-sized M, chains without
// HEAD / FFN / TAIL).
extern "C" int st_row_chain_bwd_colsum_rows(int M, int has_head, int d_ff, int has_tail) {
  static const bool pipe_on = [] { const char* e = getenv("ST_CHAIN_PIPE"); return !(e && (e[0] == '0' || e[0] == 'f')); }();
  static const bool ws_on = [] { const char* e = getenv("ST_COLSUM_WS"); return !(e && e[0] == '0'); }();      // development switch
  if (M <= 0 || !pipe_on || !ws_on || !has_head || d_ff <= 0 || !has_tail) return 0;
  const int mt = row_tiles(M);
  return mt >= 2 ? (M + 32 * mt - 1) / (32 * mt) : 0;
}

namespace {
constexpr int FOLD_MAX = 16;
struct FoldItem { const float* ws; int rows; float* dst[6]; };
struct FoldArgs { int n; FoldItem it[FOLD_MAX]; };
// block = (item, vector v, 32-column eighth): 8 row groups x 32 columns, dst[v][col] += sum over the workspace rows
__global__ __launch_bounds__(256) void colsum_fold_kernel(FoldArgs g) {
  __shared__ float part[7][32];
  const int b = blockIdx.x, i = b / 48, v = (b % 48) >> 3, q = b & 7;
  const FoldItem& it = g.it[i];
  float* dst = it.dst[v];
  if (dst == nullptr) return;
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const float* p = it.ws + (size_t)v * 256 + q * 32 + c;
  // (fixed order: the result does not depend on the launch; sixteen loads in flight per thread - the kernel is latency, not bytes)
  float acc[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] = 0.f;
  int w = rg;
  for (; w + 8 * 15 < it.rows; w += 8 * 16) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] += p[(size_t)(w + 8 * u) * 1536];
  }
#pragma unroll
  for (int u = 0; u < 16; ++u)
    if (w + 8 * u < it.rows) acc[u] += p[(size_t)(w + 8 * u) * 1536];
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < 16; ++u) s += acc[u];
  if (rg) part[rg - 1][c] = s;
  __syncthreads();
  if (rg == 0) {
    float t = s;
#pragma unroll
    for (int k = 0; k < 7; ++k) t += part[k][c];
    atomicAdd(dst + q * 32 + c, t);
  }
}
}  // namespace

// dst[i][v][0..255] += sum over the `rows[i]` workspace rows of ws[i] ([rows][6][256] floats, written by st_row_chain_bwd), for n
// workspaces; dst: 6 n pointers (dgamma_a, dbeta_a, dbias_a, dgamma_b, dbeta_b, dbias_b per workspace; NULL entries are skipped).
// Deterministic (fixed summation order; one atomic per destination float, which nothing else writes at that time).
extern "C" int st_colsum_fold(hipStream_t stream, int n, const float* const* ws, const int* rows, float* const* dst) {
  for (int base = 0; base < n; base += FOLD_MAX) {
    FoldArgs g;
    g.n = 0;
    for (int i = base; i < n && i < base + FOLD_MAX; ++i) {
      if (ws[i] == nullptr || rows[i] <= 0) continue;
      FoldItem& it = g.it[g.n++];
      it.ws = ws[i]; it.rows = rows[i];
      for (int v = 0; v < 6; ++v) it.dst[v] = dst[6 * i + v];
    }
    if (g.n == 0) continue;
    hipLaunchKernelGGL(colsum_fold_kernel, dim3(g.n * 48), dim3(256), 0, stream, g);
    ST_CHECK_LAUNCH();
  }
  return 0;
}

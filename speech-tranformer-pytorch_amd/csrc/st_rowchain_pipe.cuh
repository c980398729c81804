// Pipelined row chains for encoder-sized row counts (64- and 96-row workgroups; included by st_rowchain.hip).
//
// The plain chain kernels (st_rowchain.hip) run a workgroup's phases back to back: a weight block's MFMAs, then its epilogue
// (VALU + LDS stores), then the copy of the finished tile to HBM, each behind a workgroup barrier with all eight waves in the
// same phase.  At 24,060 rows (251 workgroups of 96 rows: one per CU, all in step) the matrix pipes idle through every epilogue,
// the vector-memory path idles through every block, and the whole chip writes its saved tensors in bursts (round 3's phase
// stamps: 12 blocks x 2.3 us + 21 us of epilogues and copies + 5 us of prologue = 54 us for 21 us of matrix work).  One
// workgroup per CU is forced (three 50 KB activation tiles; any finer row split multiplies the weight stream past the CU's
// 64 B/clk), so the overlap has to come from inside a wave:
//
//   * every block's MFMA loop carries SIDE WORK of other blocks, dealt over its eight two-k-step groups: the epilogue of the
//     block before (accumulators -> bias / ReLU / mask bits / bf16 -> LDS) or copies of finished tiles (LDS -> HBM), so that
//     their instructions issue in the shadow of the MFMAs (a 32 x 32 x 16 MFMA holds the matrix pipe for 32 clocks; the two
//     waves of a SIMD alternate on it, which leaves each wave ~12 issue slots per MFMA of its own);
//   * the feed-forward blocks run in the order W1_0, W1_1, W2_0, W1_2, W2_1, .. : chunk c + 1's first GEMM is multiplied BEFORE
//     chunk c's second, so the epilogue of W1_(c+1) has an independent block (W2_c) to hide under, and the copy of hidden chunk
//     c - 1 hides under W1_(c+1).  The streams stay in chunk order (the 32-row kernels read them front to back): this kernel
//     addresses blocks by position;
//   * the bias enters through the accumulator's initial value (requested one block ahead), the ReLU mask bits are the sign
//     bits of the fp32 sums gathered with one v_alignbit each, LayerNorm's mean and variance come from ONE pass (sum and sum
//     of squares; fp32, 256 values of order one) and one exchange, the residual tile of the first LayerNorm is stored to LDS
//     behind the first block (it arrives under its MFMAs);
//   * tile copies leave through buffer descriptors whose range ends at the workgroup's last valid row: no per-piece
//     predicate, no branch inside a block.
#pragma once
#include "st_rowchain_common.cuh"

namespace {

#ifndef ST_PIPE_PRIO
#define ST_PIPE_PRIO 0
#endif
#ifndef ST_PIPE_R_LATE
#define ST_PIPE_R_LATE 1
#endif
#ifndef ST_PIPE_OUT_AUX
#define ST_PIPE_OUT_AUX 16     // cache-policy bits of the saved-tensor stores: sc1 = write-through, no line left in the L2 (see tile_out);
                               // 0 = plain: 57.4 vs 51.7 us at 24,060 rows, same box (nt 56.0, sc0 sc1 52.8, nt sc1 52.8)
#endif

#ifdef ST_DEV_TRACE
// development: wall-clock stamps per phase, thread 0 of every workgroup (tools/dev/chain_pipe_trace.py)
__device__ long long* g_chain_trace = nullptr;
#define TRP(i) do { if (g_chain_trace && threadIdx.x == 0) g_chain_trace[blockIdx.x * 32 + (i)] = wall_clock64(); } while (0)
#else
#define TRP(i) ((void)0)
#endif

struct BiasRegs { f32x4 v[4]; };
template <int MT> __device__ __forceinline__ void bias_load(const Ctx<MT>& c, const float* b, BiasRegs& r) {
#pragma unroll
  for (int g = 0; g < 4; ++g) r.v[g] = *reinterpret_cast<const f32x4*>(b + c.wave * 32 + 8 * g + 4 * c.hi);
}
template <int MT> __device__ __forceinline__ void acc_init(f32x16 (&acc)[MT], const BiasRegs& r) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mt][4 * g + e] = r.v[g][e];
}

// ---- a finished LDS tile on its way to HBM: piece p = rows (tid >> 5) + 16 p, 16 bytes at column (tid & 31) * 8
struct TileOut {
  __amdgpu_buffer_rsrc_t rs;
  const bf16* t;      // LDS tile (256 columns of it: a 512-wide tile leaves as two of these)
  unsigned ld2;       // row pitch of the destination in bytes
  int pitch;          // row pitch of the LDS tile in elements
};
template <int MT>
__device__ __forceinline__ TileOut tile_out_desc(const Ctx<MT>& c, const bf16* t, bf16* g, int ld, int pitch = AS) {
  TileOut o;
  o.t = t;
  o.pitch = pitch;
  o.ld2 = (unsigned)ld * 2u;
  // (null destination: an empty range - every store is dropped)
  const unsigned long long base = g ? (unsigned long long)(g + (size_t)c.row0 * ld) : 0ull;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base), hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
  const unsigned bytes = __builtin_amdgcn_readfirstlane(g ? (unsigned)(c.nvalid - 1) * o.ld2 + 512u : 0u);
  o.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
  return o;
}
template <int MT> struct Pieces { bf16x8 v[2]; };
__device__ __forceinline__ bf16x8 piece_read(const TileOut& o, int tid, int p) {
  return *reinterpret_cast<const bf16x8*>(o.t + ((tid >> 5) + 16 * p) * o.pitch + (tid & 31) * 8);
}
__device__ __forceinline__ void piece_write(const TileOut& o, int tid, int p, bf16x8 v) {
  // The piece's row offset goes into the VECTOR offset, the scalar offset stays 0.  With a register in the soffset field hipcc
  // (ROCm 7.2) lets the next VALU instruction overwrite the store's data registers at once, and gfx950 does not interlock that for
  // a 16-byte store: st_row_chain512's exposed tail copied LDS ADDRESSES into out1 (buffer_store_dwordx4 v[12:15], .., s10 offen
  // directly followed by v_add_u32 v12, ..: round 6, found by the dropout variant of tests/test_kernels_gpu.py::test_row_chain512_*).
  // With soffset = 0 the compiler's hazard recogniser knows the store-data hazard and pads it.
  const unsigned voff = ((unsigned)(tid >> 5) + 16u * (unsigned)p) * o.ld2 + (unsigned)(tid & 31) * 16u;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o.rs, voff, 0, ST_PIPE_OUT_AUX);
}
// exposed copy of a whole tile
template <int MT> __device__ __forceinline__ void tile_out_now(const Ctx<MT>& c, const TileOut& o) {
  bf16x8 v[2 * MT];
#pragma unroll
  for (int p = 0; p < 2 * MT; ++p) v[p] = piece_read(o, c.tid, p);
#pragma unroll
  for (int p = 0; p < 2 * MT; ++p) piece_write(o, c.tid, p, v[p]);
}

// ---- side work of a block, by group (k2 = 0 .. 7): a() in front of the group's MFMAs, b() behind them
struct NoSide {
  __device__ __forceinline__ void a(int) {}
  __device__ __forceinline__ void b(int) {}
};
// copies of up to two tiles: NT * 2 MT pieces over the first groups, reads in a(), stores in b()
template <int MT, int NT> struct CopySide {
  const Ctx<MT>& c;
  TileOut o[NT];
  static constexpr int NP = NT * 2 * MT;            // pieces
  static constexpr int PG = (NP + 7) / 8;           // per group
  bf16x8 v[PG];
  __device__ __forceinline__ void a(int k2) {
#pragma unroll
    for (int i = 0; i < PG; ++i) {
      const int p = k2 * PG + i;
      if (p < NP) v[i] = piece_read(o[p / (2 * MT)], c.tid, p % (2 * MT));
    }
  }
  __device__ __forceinline__ void b(int k2) {
#pragma unroll
    for (int i = 0; i < PG; ++i) {
      const int p = k2 * PG + i;
      if (p < NP) piece_write(o[p / (2 * MT)], c.tid, p % (2 * MT), v[i]);
    }
  }
};

__device__ __forceinline__ uint32_t f2u(float v) { return __builtin_bit_cast(uint32_t, v); }

// one (row tile, column group) unit of an epilogue: 4 accumulator registers -> 4 bf16 in the LDS tile.
// RELU: max(v, 0) and the mask bits (sign of the fp32 value; units and elements are visited in DESCENDING bit order, every bit
// enters at the bottom of its word through v_alignbit, so bit 16 mt + 4 g + e ends where relu_bits_from() expects it; the
// words are complemented at the end: set = kept)
template <bool RELU, bool DROP, int MT, int PT = AS>
__device__ __forceinline__ void epi_unit(const Ctx<MT>& c, const f32x16 (&acc)[MT], int mt, int g, bf16* t, const Drop& d, int gcol0, int ncols,
                                         float oscale, uint32_t& w_lo, uint32_t& w_hi) {
  const int row = mt * 32 + c.r, jl = c.wave * 32 + 8 * g + 4 * c.hi;
  uint32_t db = 0;
  if (DROP) db = d.bits(drop_counter_rc(c.row0 + row, gcol0 + jl, ncols));
  bf16x4 o;
#pragma unroll
  for (int e = 3; e >= 0; --e) {
    float v = acc[mt][4 * g + e];
    if (!RELU) v *= oscale;
    float s = v;                      // the value whose sign says "masked"
    if (DROP && d.on()) {
      const bool keep = d.keep(db, e);
      s = keep ? v : -1.f;
      v = keep ? v * d.scale : 0.f;
    }
    if (RELU) {
      if (mt * 16 + 4 * g + e >= 32) w_hi = __builtin_amdgcn_alignbit(w_hi, f2u(s), 31);
      else w_lo = __builtin_amdgcn_alignbit(w_lo, f2u(s), 31);
    }
    o[e] = (bf16)v;
  }
  if (RELU) {
    // max(v, 0) on the rounded pairs: a bf16 is negative exactly when it is negative as a 16-bit integer (v_pk_max_i16; an fp32
    // maximum costs two instructions per value here - the compiler canonicalises what comes out of an MFMA first)
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    o = __builtin_bit_cast(bf16x4, __builtin_elementwise_max(__builtin_bit_cast(s16x4, o), s16x4{0, 0, 0, 0}));
    // (pins the mask words here: left alone the compiler sinks all 16 MT v_alignbit behind the block's last MFMA)
    asm volatile("" : "+v"(w_lo), "+v"(w_hi));
  }
  *reinterpret_cast<bf16x4*>(t + row * PT + jl) = o;
}

// the epilogue of a finished accumulator set as side work: 4 MT units over the eight groups, the mask words stored behind the last
template <bool RELU, bool DROP, int MT, int PT = AS> struct EpiSide {
  const Ctx<MT>& c;
  const f32x16 (&acc)[MT];
  bf16* t;
  const Drop& d;
  int gcol0, ncols;
  float oscale;
  unsigned long long* bits;      // this lane's word (RELU; may be null)
  uint32_t w_lo = 0, w_hi = 0;
  static constexpr int NU = 4 * MT;
  __device__ __forceinline__ void unit(int j) {      // j ascending = bits descending
    epi_unit<RELU, DROP, MT, PT>(c, acc, MT - 1 - j / 4, 3 - j % 4, t, d, gcol0, ncols, oscale, w_lo, w_hi);
  }
  __device__ __forceinline__ void a(int) {}
  __device__ __forceinline__ void b(int k2) {
#pragma unroll
    for (int j = (k2 * NU) / 8; j < ((k2 + 1) * NU) / 8; ++j) unit(j);
    if (RELU && k2 == 7 && bits) *bits = finish();
  }
  __device__ __forceinline__ unsigned long long finish() const {
    const uint32_t lo = ~w_lo, hi = MT * 16 > 32 ? (~w_hi & ((1u << (MT * 16 - 32 > 0 ? MT * 16 - 32 : 1)) - 1u)) : 0u;
    return ((unsigned long long)hi << 32) | (MT * 16 >= 32 ? lo : (lo & ((1u << (MT * 16 % 32)) - 1u)));
  }
  __device__ __forceinline__ void all() {      // exposed
#pragma unroll
    for (int j = 0; j < NU; ++j) unit(j);
    if (RELU && bits) *bits = finish();
  }
};
// two side jobs in one block
template <class S0, class S1> struct Both {
  S0& s0; S1& s1;
  __device__ __forceinline__ void a(int k2) { s0.a(k2); s1.a(k2); }
  __device__ __forceinline__ void b(int k2) { s0.b(k2); s1.b(k2); }
};

#ifndef ST_PIPE_RING_BUF
#define ST_PIPE_RING_BUF 1      // (ring_load: st_rowchain_common.cuh)
#endif
// One 256 x 256 weight block as block_mma, the ring refilled from two places: the second half of THIS block (cur_blk), then the
// first half of the block that is multiplied NEXT (nxt_blk) - the blocks are not visited in stream order.  Ring<MT>::D == 8.
template <int PT, int MT, class Side>      // PT: row pitch of the activation tile in elements
__device__ __forceinline__ void block_mma_pt(Ctx<MT>& c, const bf16x8* cur_blk, const bf16x8* nxt_blk, const bf16* act, f32x16 (&acc)[MT],
                                             Side&& side) {
  static_assert(Ring<MT>::D == 8, "block_mma_p: half a block on request");
#if ST_PIPE_RING_BUF
  // the ring's refills as buffer loads: wave-uniform stream base in the descriptor, the block's byte offset in the scalar offset, the
  // fragment's in the immediate - no address arithmetic per load (flat loads took one v_lshl_add_u64 each: 16 VALU instructions per
  // block in a loop that is issue-bound, and eight registers of 64-bit lane indices)
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)c.ws, 0, 0x7fffffff, 0x00020000);
  const unsigned lane16 = (unsigned)c.l * 16u;
  const unsigned cur_off = __builtin_amdgcn_readfirstlane((unsigned)((const char*)cur_blk - (const char*)c.ws));
  const unsigned nxt_off = __builtin_amdgcn_readfirstlane((unsigned)((const char*)nxt_blk - (const char*)c.ws));
#endif
#ifndef ST_PIPE_LDS_AHEAD
#define ST_PIPE_LDS_AHEAD 0      // (1: the activation fragments of group k2 + 1 read while group k2 multiplies - measured at nothing, 24 registers)
#endif
#if ST_PIPE_LDS_AHEAD
  // the activation fragments of group k2 + 1 are read while group k2 multiplies (two register sets, alternating)
  bf16x8 xf[2][2][MT];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xf[0][u][mt] = frag_nat(act, PT, mt * 32 + c.r, u * 16 + c.hi * 8);
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) {
    if (k2 < 7) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[(k2 + 1) & 1][u][mt] = frag_nat(act, PT, mt * 32 + c.r, (2 * k2 + 2 + u) * 16 + c.hi * 8);
    }
    side.a(k2);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#ifdef ST_DEV_NO_MFMA      // development: the kernel without its matrix work (operands consumed by an empty asm)
        asm volatile("" :: "v"(c.ring[(2 * k2 + u) % 8]), "v"(xf[k2 & 1][u][mt]));
#else
        acc[mt] = mfma32(c.ring[(2 * k2 + u) % 8], xf[k2 & 1][u][mt], acc[mt]);
#endif
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f = 2 * k2 + u + 8;
      c.ring[(2 * k2 + u) % 8] = f < 16 ? cur_blk[f * 64 + c.l] : nxt_blk[(f - 16) * 64 + c.l];
    }
    side.b(k2);
    __builtin_amdgcn_sched_barrier(0);
  }
#else
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) {
    bf16x8 xf[2][MT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xf[u][mt] = frag_nat(act, PT, mt * 32 + c.r, (2 * k2 + u) * 16 + c.hi * 8);
    side.a(k2);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#ifdef ST_DEV_NO_MFMA
        asm volatile("" :: "v"(c.ring[(2 * k2 + u) % 8]), "v"(xf[u][mt]));
#else
        acc[mt] = mfma32(c.ring[(2 * k2 + u) % 8], xf[u][mt], acc[mt]);
#endif
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f = 2 * k2 + u + 8;
#if ST_PIPE_RING_BUF
      c.ring[(2 * k2 + u) % 8] = ring_load(wrs, lane16, f < 16 ? cur_off : nxt_off, f & 15);
#else
      c.ring[(2 * k2 + u) % 8] = f < 16 ? cur_blk[f * 64 + c.l] : nxt_blk[(f - 16) * 64 + c.l];
#endif
    }
    side.b(k2);
    __builtin_amdgcn_sched_barrier(0);
  }
#endif
}

template <int MT, class Side>
__device__ __forceinline__ void block_mma_p(Ctx<MT>& c, const bf16x8* cur_blk, const bf16x8* nxt_blk, const bf16* act, f32x16 (&acc)[MT],
                                            Side&& side) {
  block_mma_pt<AS>(c, cur_blk, nxt_blk, act, acc, side);
}

// v = acc (bias inside) + res; LayerNorm over the 256 columns held by the 8 waves from ONE pass (sum, sum of squares) and one
// exchange; xhat -> t_xhat, (dropped) output -> t_out (may be `res` itself: every residual read precedes the barrier).  One
// workgroup barrier inside; the caller places the one behind.  red2: [MT][32 rows][8 waves x (sum, sq) + pad] floats, row
// pitch 80 bytes (conflict-free 16-byte reads).
constexpr int RED2_PITCH = 20;
// x + the value of the lane 32 away: one v_permlane32_swap (VALU) instead of the ds_bpermute __shfl_xor becomes (an LDS round trip)
__device__ __forceinline__ float wave_sum32(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));      // a = {lo, lo}, b = {hi, hi}
  return a + b;
}
template <bool DROP, int MT>
__device__ __forceinline__ void epi_ln_p(const Ctx<MT>& c, f32x16 (&acc)[MT], const bf16* res, const BiasRegs& gamma, const BiasRegs& beta,
                                         float eps, const Drop& d, bf16* t_xhat, bf16* t_out, float* red2, float* g_rstd) {
  const int j0 = c.wave * 32;
  // (written without a branch and with every LDS read of a pass in front of its arithmetic: the phase is a chain of latencies -
  // 500 instructions in 2.4 us when each row tile waited for its own reads)
  bf16x4 rr[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) rr[mt][g] = *reinterpret_cast<const bf16x4*>(res + (mt * 32 + c.r) * AS + j0 + 8 * g + 4 * c.hi);
  float s[MT], q[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const float v0 = acc[mt][4 * g + e] + (float)rr[mt][g][e], v1 = acc[mt][4 * g + e + 1] + (float)rr[mt][g][e + 1];
        acc[mt][4 * g + e] = v0;
        acc[mt][4 * g + e + 1] = v1;
        s0 += v0; s1 += v1;
        q0 = fmaf(v0, v0, q0); q1 = fmaf(v1, v1, q1);
      }
    s[mt] = s0 + s1;
    q[mt] = q0 + q1;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    s[mt] = wave_sum32(s[mt]);
    q[mt] = wave_sum32(q[mt]);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)      // (both halves of the wave hold the same sums and write them to the same place)
    *reinterpret_cast<f32x2*>(red2 + (mt * 32 + c.r) * RED2_PITCH + 2 * c.wave) = f32x2{s[mt], q[mt]};
  __syncthreads();
  f32x4 pp[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4) pp[mt][w4] = *reinterpret_cast<const f32x4*>(red2 + (mt * 32 + c.r) * RED2_PITCH + 4 * w4);
  float rstd[MT], nm[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const float ss = (pp[mt][0][0] + pp[mt][0][2]) + (pp[mt][1][0] + pp[mt][1][2]) + ((pp[mt][2][0] + pp[mt][2][2]) + (pp[mt][3][0] + pp[mt][3][2]));
    const float qq = (pp[mt][0][1] + pp[mt][0][3]) + (pp[mt][1][1] + pp[mt][1][3]) + ((pp[mt][2][1] + pp[mt][2][3]) + (pp[mt][3][1] + pp[mt][3][3]));
    const float mean = ss * (1.f / DM);
    const float var = fmaxf(qq * (1.f / DM) - mean * mean, 0.f);
    rstd[mt] = rsqrtf(var + eps);
    nm[mt] = -mean * rstd[mt];
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int jl = j0 + 8 * g + 4 * c.hi, row = mt * 32 + c.r;
      uint32_t bits = 0;
      if (DROP) bits = d.bits(drop_counter_rc(c.row0 + row, jl, DM));
      bf16x4 xh, o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float n = fmaf(acc[mt][4 * g + e], rstd[mt], nm[mt]);
        float v = fmaf(n, gamma.v[g][e], beta.v[g][e]);
        if (DROP && d.on()) v = d.keep(bits, e) ? v * d.scale : 0.f;
        xh[e] = (bf16)n;
        o[e] = (bf16)v;
      }
      *reinterpret_cast<bf16x4*>(t_xhat + row * AS + jl) = xh;
      *reinterpret_cast<bf16x4*>(t_out + row * AS + jl) = o;
    }
  if (g_rstd && c.wave == 0 && c.hi == 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      if (mt * 32 + c.r < c.nvalid) g_rstd[c.row0 + mt * 32 + c.r] = rstd[mt];
  }
}

// PRE + FFN [+ POST]  or  POST alone (the encoder's layer-0 projection); MT = 2, 3
template <bool PRE, bool FFN, int NB, bool DROP, int MT>
__global__ __launch_bounds__(512, 1) void row_chain_pipe_kernel(ChainArgs a) {
  constexpr bool POST = NB > 0;
  static_assert(NB == 0 || NB == 1 || NB == 3, "pipelined chains: no, one or three projection blocks");
  static_assert((PRE && FFN) || (!PRE && !FFN && POST), "pipelined chains: PRE + FFN [+ POST] or POST alone");
  constexpr int RB = 32 * MT, TE = RB * AS;
  __shared__ __attribute__((aligned(16))) bf16 tiles[3 * TE];
  __shared__ __attribute__((aligned(16))) float red2[MT * 32 * RED2_PITCH];
  Ctx<MT> c;
  c.tid = threadIdx.x; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6); c.l = c.tid & 63; c.hi = c.l >> 5; c.r = c.l & 31;
  c.row0 = blockIdx.x * RB; c.nvalid = min(RB, a.M - c.row0);
  const bf16x8* sbase = a.wfrag + (size_t)c.wave * a.wave_frags * 64;      // wave-uniform
  auto blk = [&](int b) { return sbase + (size_t)b * 16 * 64; };
  c.ws = sbase;
#pragma unroll
  for (int i = 0; i < 8; ++i) c.ring[i] = sbase[i * 64 + c.l];
  int touched;
  {       // one line per thread warms this XCD's L2 with the chain's streams (see row_chain_kernel)
    const int nlines = NW * a.wave_frags * 8;
    const int ln = min(((int)blockIdx.x >> 3) * 512 + c.tid, nlines - 1);
    __builtin_amdgcn_sched_barrier(0);
    touched = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(a.wfrag) + (size_t)ln * 128);
    __builtin_amdgcn_sched_barrier(0);
  }
  TRP(0);
#ifdef ST_DEV_TRACE
  if (g_chain_trace && threadIdx.x == 0) g_chain_trace[blockIdx.x * 32 + 30] = clock64();
#endif
#if ST_PIPE_PRIO
  if (c.wave >= 4) __builtin_amdgcn_s_setprio(1);      // the later-dispatched half loses every arbitration against its SIMD partner otherwise
#endif
  bf16* T0 = tiles; bf16* T1 = tiles + TE; bf16* T2 = tiles + 2 * TE;
  const Drop d1 = make_drop(a.drop1), d2 = make_drop(a.drop2), off = make_drop(DropArgs{nullptr, 0u, 0, 1.f});
  const int p0 = PRE ? 1 : 0, nc = FFN ? a.nc : 0, q0 = p0 + 2 * nc;      // stream positions: PRE 0 | W1_c p0 + 2c, W2_c p0 + 2c + 1 | POST q0 + u
  bf16* cur;             // the running activation (MFMA operand)
  bf16 *X, *Y;           // the two other tiles
  TileOut pend[3];       // copies waiting for a block to hide under
  int npend = 0;
  BiasRegs nb1, nb2;     // biases on request one block ahead of the accumulator they initialise

  if (PRE) {
    TileRegs<MT> ra, rr;
    BiasRegs bo, g0, be0;
    tile_load(c, a.A, a.lda, ra);
    bias_load(c, a.bo, bo);
#if ST_PIPE_R_LATE
    // the residual is asked for only once A is here: both tiles on request together share the HBM burst of the launch's first
    // microseconds (every CU asks at once), and the first block needs A alone
    tile_store(c, ra, T0);
    __builtin_amdgcn_sched_barrier(0);
    tile_load(c, a.R, a.ldr, rr);
#else
    tile_load(c, a.R, a.ldr, rr);
    tile_store(c, ra, T0);
#endif
    bias_load(c, a.g0, g0);
    bias_load(c, a.be0, be0);
    __syncthreads();
    TRP(1);
    f32x16 acc[MT];
    acc_init(acc, bo);
    NoSide ns;
    block_mma_p(c, blk(0), blk(1), T0, acc, ns);
    TRP(2);
    tile_store(c, rr, T1);             // the residual arrived under the block
    if (FFN) {                         // (requested here, used behind the LayerNorm: a bias asked for where it is needed drains the
      bias_load(c, a.b2, nb2);         // whole in-order queue - ring, copies - in front of the block: 0.65 us per block, measured)
      bias_load(c, a.b1, nb1);
    }
    __syncthreads();                   // residual visible; every wave is past its MFMAs on the A tile
    epi_ln_p<false>(c, acc, T1, g0, be0, a.eps, off, T0, T2, red2, a.rstd0);      // xhat0 -> the A tile, out0 -> T2
    __syncthreads();
    TRP(3);
    cur = T2; X = T1; Y = T0;          // X: free now (the residual's last read precedes epi_ln_p's barrier); Y: xhat0, waiting for its copy
    pend[0] = tile_out_desc(c, T0, a.xhat0, DM);
    pend[1] = tile_out_desc(c, T2, a.out0, DM);
    npend = 2;
  } else {
    tile_in(c, a.A, a.lda, T0);
    __syncthreads();
    cur = T0; X = T1; Y = T2;
  }

  if (FFN) {
    const int dff = nc * 256;
    f32x16 acc1[MT], acc2[MT];
    acc_init(acc2, nb2);
    acc_init(acc1, nb1);
    if (nc > 1) bias_load(c, a.b1 + 256, nb1);
    auto bits_at = [&](int ch) {
      return a.relu_bits ? a.relu_bits + ((size_t)(blockIdx.x * nc + ch) * NW + c.wave) * 64 + c.l : nullptr;
    };
    // W1_0 with PRE's two copies beside it
    {
      CopySide<MT, 2> cs{c, {pend[0], pend[1]}};
      block_mma_p(c, blk(p0), nc > 1 ? blk(p0 + 2) : blk(p0 + 1), cur, acc1, cs);
      npend = 0;
    }
    TRP(4);
    {       // chunk 0's epilogue is the one nothing hides
      EpiSide<true, DROP, MT> e0{c, acc1, X, d1, 0, dff, 1.f, bits_at(0)};
      e0.all();
    }
    __syncthreads();      // hidden chunk 0 complete in X; every wave is past its copies of Y (xhat0)
    TRP(5);
    // from here: hidden chunk c lives in (c even ? X : Y); its copy to H rides under W1_(c+1), the block in front of the W2_c that
    // multiplies it (both only read the tile)
    for (int ch = 0; ch + 1 < nc; ++ch) {
      bf16* hc = (ch & 1) ? Y : X;          // chunk ch
      bf16* hn = (ch & 1) ? X : Y;          // chunk ch + 1, written beside W2_ch (chunk ch - 1 left it one barrier ago)
      acc_init(acc1, nb1);
      if (ch + 2 < nc) bias_load(c, a.b1 + (ch + 2) * 256, nb1);
      {
        CopySide<MT, 1> cs{c, {tile_out_desc(c, hc, a.H ? a.H + ch * 256 : nullptr, dff)}};
        block_mma_p(c, blk(p0 + 2 * (ch + 1)), blk(p0 + 2 * ch + 1), cur, acc1, cs);
      }
      if (ch < 3) TRP(6 + 2 * ch);
      {       // W2_ch with the epilogue of W1_(ch+1) beside it
        EpiSide<true, DROP, MT> es{c, acc1, hn, d1, (ch + 1) * 256, dff, 1.f, bits_at(ch + 1)};
        block_mma_p(c, blk(p0 + 2 * ch + 1), ch + 2 < nc ? blk(p0 + 2 * (ch + 2)) : blk(p0 + 2 * (ch + 1) + 1), hc, acc2, es);
      }
      __syncthreads();
      if (ch < 3) TRP(7 + 2 * ch);
    }
    BiasRegs g1, be1;
    bf16* hl = ((nc - 1) & 1) ? Y : X;      // the last chunk
    bf16* tx = ((nc - 1) & 1) ? X : Y;      // free
    {       // the last chunk's W2 with its own copy beside it; the LayerNorm vectors are requested in front
      bias_load(c, a.g1, g1);
      bias_load(c, a.be1, be1);
      if (POST) bias_load(c, a.bp, nb1);
      CopySide<MT, 1> cs{c, {tile_out_desc(c, hl, a.H ? a.H + (nc - 1) * 256 : nullptr, dff)}};
      block_mma_p(c, blk(p0 + 2 * (nc - 1) + 1), blk(q0), hl, acc2, cs);
      TRP(12);
    }
    // (no barrier: the LayerNorm's first pass reads only cur; its own barrier lies in front of the writes to tx and cur)
    epi_ln_p<DROP>(c, acc2, cur, g1, be1, a.eps, d2, tx, cur, red2, a.rstd1);      // xhat1 -> tx, out1 replaces cur in place
    __syncthreads();
    TRP(13);
    pend[0] = tile_out_desc(c, tx, a.xhat1, DM);
    pend[1] = tile_out_desc(c, cur, a.out1, DM);
    npend = 2;
    X = hl; Y = tx;
  }

  if (POST) {
    // NB = 1 or 3 projection blocks, straight-line: block u's accumulators are finished beside block u + 1 (two sets, alternating)
    // into the staging tiles X, Y (alternating), a staged block leaves beside block u + 2; the copies of the sublayer in front
    // ride under block 0.  X (the last hidden chunk) is free: its readers are behind the LayerNorm's barriers; Y (xhat1) is free
    // one barrier after block 0.
    f32x16 accA[MT], accB[MT];
    auto pdesc = [&](int u, bf16* st) { return tile_out_desc(c, st, a.P ? a.P + u * 256 : nullptr, a.ldp); };
    if (!FFN) bias_load(c, a.bp, nb1);
    acc_init(accA, nb1);
    if (NB > 1) bias_load(c, a.bp + 256, nb1);
    if (npend == 2) {
      CopySide<MT, 2> cs{c, {pend[0], pend[1]}};
      block_mma_p(c, blk(q0), blk(q0 + 1), cur, accA, cs);
    } else {
      NoSide ns;
      block_mma_p(c, blk(q0), blk(q0 + 1), cur, accA, ns);
    }
    TRP(14);
    if (NB == 3) {
      {       // block 1 -> B beside the epilogue of block 0 (A -> X)
        acc_init(accB, nb1);
        bias_load(c, a.bp + 512, nb1);
        EpiSide<false, false, MT> es{c, accA, X, off, 0, 0, 1.f, nullptr};
        block_mma_p(c, blk(q0 + 1), blk(q0 + 2), cur, accB, es);
        __syncthreads();
        TRP(15);
      }
      {       // block 2 -> A beside the epilogue of block 1 (B -> Y, the key block: pre-scaled) and the copy of block 0 (X)
        acc_init(accA, nb1);
        EpiSide<false, false, MT> es{c, accB, Y, off, 0, 0, a.post_kscale, nullptr};
        CopySide<MT, 1> cs{c, {pdesc(0, X)}};
        Both<EpiSide<false, false, MT>, CopySide<MT, 1>> both{es, cs};
        block_mma_p(c, blk(q0 + 2), blk(q0 + 3), cur, accA, both);
        __syncthreads();
        TRP(16);
      }
      // the last block's epilogue and the last two copies are exposed
      EpiSide<false, false, MT> es{c, accA, X, off, 0, 0, 1.f, nullptr};
      es.all();
      tile_out_now(c, pdesc(1, Y));
      __syncthreads();
      tile_out_now(c, pdesc(2, X));
    } else {
      EpiSide<false, false, MT> es{c, accA, X, off, 0, 0, 1.f, nullptr};
      es.all();
      __syncthreads();
      tile_out_now(c, pdesc(0, X));
    }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (i < npend) tile_out_now(c, pend[i]);
  }
  TRP(17);
#ifdef ST_DEV_TRACE
  if (g_chain_trace && threadIdx.x == 0) g_chain_trace[blockIdx.x * 32 + 31] = clock64();
#endif
  if (touched == 0x5a5a5a5a && a.M < 0) red2[0] = 1.f;      // (never true: keeps the warm-up load alive)
}

}  // namespace

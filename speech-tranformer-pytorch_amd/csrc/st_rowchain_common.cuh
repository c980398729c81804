// Shared pieces of the row-chain kernels (st_rowchain.hip) and of the kernels that run a chain stage in their own prologue
// (st_attn_xs.hip: the decoder-encoder attention computes its own output_linear + LayerNorm + q projection): the activation
// tiles in LDS, the streamed weight-block product, the bias / LayerNorm epilogues.
#pragma once
#include "st_common.cuh"

namespace {

constexpr int DM = 256;      // d_model = block edge
constexpr int AS = DM + 8;   // LDS activation row stride in elements (528 B: conflict-free ds_read_b128 over 16 rows)
constexpr int DEPTH = 16;    // weight fragments in flight per wave (16 KB): exactly one block ahead
constexpr int NW = 8;        // waves per workgroup
constexpr int TOUCH = 8;     // warm-up lines per thread (covers a 12-block chain and the 2-block one behind it from 4 workgroups per XCD up)

struct ChainArgs {
  int M;
  const bf16x8* wfrag;               // this chain's streams: [8 waves][nblocks * 16 + DEPTH fragments][64 lanes]
  int wave_frags;                    // nblocks * 16 + DEPTH
  int next_frags;                    // the same of the chain stored right behind this one (0: none): warmed for its launch
  float eps;
  const bf16* A; int lda;            // [M, 256]: PRE's GEMM operand (attention context); without PRE the chain input
  // PRE
  const bf16* R; int ldr;            // residual [M, 256]
  const float* bo; const float* g0; const float* be0;
  bf16* out0; bf16* xhat0; float* rstd0;     // ld 256
  // FFN
  int nc;                            // d_ff / 256
  const float* b1; const float* b2; const float* g1; const float* be1;
  bf16* H;                           // [M, d_ff] hidden activation (after ReLU and dropout1): the weight gradient's operand
  unsigned long long* relu_bits;     // optional: which hidden values are > 0, for st_row_chain_bwd - one word per lane, chunk
                                     // and workgroup in the accumulator layout both kernels share (bit 16 mt + 4 g + e), so the
                                     // backward chain reads 8 bytes per lane and chunk instead of H (49 MB per encoder layer,
                                     // as 8-byte pieces scattered over 32 rows per load instruction)
  bf16* out1; bf16* xhat1; float* rstd1;     // ld 256
  DropArgs drop1, drop2;
  // POST
  int nb; const float* bp; bf16* P; int ldp;
  float post_kscale;                 // projection block 1 (the keys of a q | k | v projection) leaves as (acc + bias) * post_kscale:
                                     // the attention kernels' k_prescaled operand (st_attn_common.cuh); 1 = plain
  // split feed-forward (row_chain_split_kernel): `split_parts` partial-sum slots of 32 x 256 fp32 per row block, then one ticket
  // per block; a part owns nc / split_parts consecutive hidden chunks (1 at decoder size; 2 when M / 32 x nc workgroups would
  // not fit one round of the chip - a 4-utterance shard's encoder: 98 row blocks x 2)
  float* split_ws; unsigned* split_tickets; int split_parts;
};

// fragments in flight per wave: with three row tiles a fragment feeds three MFMAs (it is consumed a third as often), and
// the registers are needed for the accumulators
template <int MT> struct Ring { static constexpr int D = MT == 1 ? DEPTH : DEPTH / 2; };

template <int MT> struct Ctx {
  int tid, wave, l, hi, r, row0, nvalid;
  const bf16x8* ws;      // wave-uniform stream cursor: the fragment Ring<MT>::D ahead of the next one to be multiplied
  bf16x8 ring[Ring<MT>::D];
};

// One 256 x 256 weight block: acc[mt][n = wave*32 + ...][m] (+)= W_block x act^T for the MT row tiles of the workgroup
// (every weight fragment feeds MT MFMAs), refilling the ring DEPTH fragments ahead.
// A weight fragment through a buffer descriptor (the pipelined chains, st_rowchain_pipe.cuh: stream base in the descriptor, block
// offset in the scalar offset, fragment offset in the immediate - no address arithmetic per load in their issue-bound block loop;
// the same in block_mma below measured nothing on the decoder-sized chains, which are latency, and was taken out again: round 6).
// Fragment f (0 .. 15) of the block at byte offset `blk_off` behind the descriptor's base
__device__ __forceinline__ bf16x8 ring_load(__amdgpu_buffer_rsrc_t rs, unsigned lane16, unsigned blk_off, int f) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + (unsigned)f * 1024u, blk_off, 0));
}

template <int MT>
__device__ __forceinline__ void block_mma(Ctx<MT>& c, const bf16* act, f32x16 (&acc)[MT]) {
  // two k-steps per group: the 2 MT activation reads, then the 2 MT MFMAs, then the two refills (one k-step per group
  // measured 1-3 % slower on the backward chains; without the scheduling barrier 5-12 % slower: hoisted refills double the
  // live registers)
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) {
    bf16x8 xf[2][MT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xf[u][mt] = frag_nat(act, AS, mt * 32 + c.r, (2 * k2 + u) * 16 + c.hi * 8);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma32(c.ring[(2 * k2 + u) % Ring<MT>::D], xf[u][mt], acc[mt]);
#pragma unroll
    for (int u = 0; u < 2; ++u) c.ring[(2 * k2 + u) % Ring<MT>::D] = c.ws[(2 * k2 + u) * 64 + c.l];
    __builtin_amdgcn_sched_barrier(0);
  }
  c.ws += 16 * 64;
}

template <int MT> __device__ __forceinline__ void zero_acc(f32x16 (&acc)[MT]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = zero16();
}

// [32 MT][256] tile: global (rows past M as zeros) -> LDS
template <int MT>
__device__ __forceinline__ void tile_in(const Ctx<MT>& c, const bf16* g, int ld, bf16* t) {
#pragma unroll
  for (int p = 0; p < 2 * MT; ++p) {
    const int id = c.tid + p * 512, rr = id >> 5, cc = id & 31;
    *reinterpret_cast<bf16x8*>(t + rr * AS + cc * 8) = gload8(g + (size_t)(c.row0 + rr) * ld + cc * 8, rr < c.nvalid);
  }
}
// the same in two halves, so that the loads fly under a block of MFMAs: global -> registers now, registers -> LDS later
template <int MT> struct TileRegs { bf16x8 v[2 * MT]; };
template <int MT>
__device__ __forceinline__ void tile_load(const Ctx<MT>& c, const bf16* g, int ld, TileRegs<MT>& t) {
#pragma unroll
  for (int p = 0; p < 2 * MT; ++p) {
    const int id = c.tid + p * 512, rr = id >> 5, cc = id & 31;
    t.v[p] = gload8(g + (size_t)(c.row0 + rr) * ld + cc * 8, rr < c.nvalid);
  }
}
template <int MT>
__device__ __forceinline__ void tile_store(const Ctx<MT>& c, const TileRegs<MT>& t, bf16* lds) {
#pragma unroll
  for (int p = 0; p < 2 * MT; ++p) {
    const int id = c.tid + p * 512, rr = id >> 5, cc = id & 31;
    *reinterpret_cast<bf16x8*>(lds + rr * AS + cc * 8) = t.v[p];
  }
}
// LDS -> global as 512-byte row segments.  64- and 96-row workgroups (encoder-sized launches: 135 MB of saved tensors per forward
// chain, read again by LATER kernels only) store write-through (sc1: the line does not stay in the XCD's L2).  Round 6, same
// box, 24,060 rows: forward chain 58.7 -> 51.9 us, backward 66.2 -> 64.0 - with plain stores the dirty lines of a launch
// (17 MB per XCD through a 4 MB write-back L2) sit between the weight stream and its readers.  Decoder-sized launches
// (32-row workgroups) keep plain stores: their few hundred KB ARE read back from the L2 by the next kernel.
#ifndef ST_TILE_OUT_SC1
#define ST_TILE_OUT_SC1 1
#endif
template <int MT>
__device__ __forceinline__ void tile_out(const Ctx<MT>& c, const bf16* t, bf16* g, int ld) {
#pragma unroll
  for (int p = 0; p < 2 * MT; ++p) {
    const int id = c.tid + p * 512, rr = id >> 5, cc = id & 31;
    if (rr < c.nvalid) {
      if (MT >= 2 && ST_TILE_OUT_SC1) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(t + rr * AS + cc * 8);
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(g + (size_t)(c.row0 + rr) * ld + cc * 8), "v"(v) : "memory");
      } else {
        *reinterpret_cast<bf16x8*>(g + (size_t)(c.row0 + rr) * ld + cc * 8) = *reinterpret_cast<const bf16x8*>(t + rr * AS + cc * 8);
      }
    }
  }
}

// The epilogues' per-column vectors (bias / gamma / beta), from global memory or from a copy in LDS.  vmcnt is in-order: a vector load
// issued in an epilogue waits behind every weight fragment the ring has on request - at 32 rows per workgroup (ring = one block ahead)
// that is a full drain of the stream in front of EVERY epilogue (round 5: the 12-block chain at 1,206 rows 29.1 -> 25.6 us with the
// loads replaced by constants).  32-row workgroups therefore copy the vectors into LDS once, in the prologue; 96-row workgroups have no
// LDS to spare and lose 1.5 % to it.
struct LVec {
  const ST_LDS float* p;
  __device__ __forceinline__ LVec operator+(int o) const { return LVec{p + o}; }
};
__device__ __forceinline__ f32x4 ld4(const float* p, int j) { return *reinterpret_cast<const f32x4*>(p + j); }
__device__ __forceinline__ f32x4 ld4(LVec v, int j) { return *reinterpret_cast<const ST_LDS f32x4*>(v.p + j); }

// acc + bias (+ReLU, dropout) -> bf16 into this wave's 32 columns of an LDS tile
// bits (optional): this wave's 64 words of ChainArgs::relu_bits for the block
template <bool RELU, bool DROP, int MT, class V>
__device__ __forceinline__ void epi_store(const Ctx<MT>& c, const f32x16 (&acc)[MT], V bias, bf16* t, const Drop& d,
                                          int gcol0, int ncols, unsigned long long* bits = nullptr, float oscale = 1.f) {
  uint32_t pos_lo = 0, pos_hi = 0;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int jl = c.wave * 32 + 8 * g + 4 * c.hi;
    const f32x4 bb = ld4(bias, jl);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = mt * 32 + c.r;
      uint32_t db = 0;
      if (DROP) db = d.bits(drop_counter_rc(c.row0 + row, gcol0 + jl, ncols));
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = (acc[mt][4 * g + e] + bb[e]) * oscale;
        if (RELU) v = fmaxf(v, 0.f);
        if (DROP && d.on()) v = d.keep(db, e) ? v * d.scale : 0.f;
        o[e] = (bf16)v;
        if (RELU) {
          const int b = mt * 16 + 4 * g + e;
          if (b < 32) pos_lo |= ((float)o[e] > 0.f ? 1u : 0u) << b;
          else pos_hi |= ((float)o[e] > 0.f ? 1u : 0u) << (b - 32);
        }
      }
      *reinterpret_cast<bf16x4*>(t + row * AS + jl) = o;
    }
  }
  if (RELU && bits != nullptr) bits[c.l] = ((unsigned long long)pos_hi << 32) | pos_lo;
}

// v = acc + bias + res; LayerNorm over the 256 columns held by the 8 waves; xhat -> t_xhat, (dropped) output -> t_out,
// both then leave for HBM.  Two workgroup barriers inside, one before the copies out: on return t_out is complete.
template <bool DROP, int MT, class V>
__device__ __forceinline__ void epi_ln(const Ctx<MT>& c, f32x16 (&acc)[MT], V bias, const bf16* res, V gamma,
                                       V beta, float eps, const Drop& d, bf16* t_xhat, bf16* t_out,
                                       float (*red)[NW * 32 * MT], bf16* g_out, bf16* g_xhat, float* g_rstd) {
  const int j0 = c.wave * 32;
  float sum[MT], sq[MT], mean[MT], rstd[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) sum[mt] = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int jl = j0 + 8 * g + 4 * c.hi;
    const f32x4 bb = ld4(bias, jl);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bf16x4 rr = *reinterpret_cast<const bf16x4*>(res + (mt * 32 + c.r) * AS + jl);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = acc[mt][4 * g + e] + bb[e] + (float)rr[e];
        acc[mt][4 * g + e] = v;
        sum[mt] += v;
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    sum[mt] += wave_xor32(sum[mt]);
    if (c.hi == 0) red[0][(c.wave * MT + mt) * 32 + c.r] = sum[mt];
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[0][(w * MT + mt) * 32 + c.r];
    mean[mt] = s * (1.f / DM);
    sq[mt] = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float dv = acc[mt][q] - mean[mt];
      sq[mt] += dv * dv;
    }
    sq[mt] += wave_xor32(sq[mt]);
    if (c.hi == 0) red[1][(c.wave * MT + mt) * 32 + c.r] = sq[mt];
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[1][(w * MT + mt) * 32 + c.r];
    rstd[mt] = rsqrtf(s * (1.f / DM) + eps);
    if (g_rstd && c.wave == 0 && c.hi == 0 && mt * 32 + c.r < c.nvalid) g_rstd[c.row0 + mt * 32 + c.r] = rstd[mt];
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int jl = j0 + 8 * g + 4 * c.hi;
    const f32x4 g4 = ld4(gamma, jl);
    const f32x4 b4 = ld4(beta, jl);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = mt * 32 + c.r;
      uint32_t bits = 0;
      if (DROP) bits = d.bits(drop_counter_rc(c.row0 + row, jl, DM));
      bf16x4 xh, o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float n = (acc[mt][4 * g + e] - mean[mt]) * rstd[mt];
        float v = n * g4[e] + b4[e];
        if (DROP && d.on()) v = d.keep(bits, e) ? v * d.scale : 0.f;
        xh[e] = (bf16)n;
        o[e] = (bf16)v;
      }
      *reinterpret_cast<bf16x4*>(t_xhat + row * AS + jl) = xh;
      *reinterpret_cast<bf16x4*>(t_out + row * AS + jl) = o;
    }
  }
  __syncthreads();
  if (g_xhat) tile_out(c, t_xhat, g_xhat, DM);
  if (g_out) tile_out(c, t_out, g_out, DM);
}

}  // namespace

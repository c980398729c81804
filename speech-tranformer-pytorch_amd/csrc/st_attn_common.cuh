// Shared pieces of the attention kernels (st_attn.hip: 128-row workgroups, every head width; st_attn64.hip: the
// 64-rows-per-wave kernels for long problems with 64-wide heads): argument block, work-list decoding, the streamed
// LDS tiles, row stores through LDS patches, dropout keep decisions.
#pragma once
#include "st_common.cuh"

namespace {

constexpr int TILE = 64;      // rows (keys or queries) per streamed tile
constexpr int WG_ROWS = 128;  // rows owned by a workgroup (4 waves x 32)

struct AttnArgs {
  const bf16* Q; int ldq;
  const bf16* K; int ldk;
  const bf16* V; int ldv;
  bf16* O; int ldo;               // forward: output; backward: forward output (for delta)
  bf16* Ores;                     // forward, optional: bf16(O_fp32 - bf16(O_fp32)), same layout as O - with it the
                                  // backward's delta = rowsum(dO * (O + Ores)) sees O to ~16 mantissa bits
  const bf16* dO; int lddo;
  bf16* dQ; int lddq;
  bf16* dK; int lddk;
  bf16* dV; int lddv;
  float* lse;                     // [H][q_rows_total], log2 domain: m + log2(l)
  float* delta;                   // [H][q_rows_total]
  const int* q_off; const int* q_len;
  const int* k_off; const int* k_len;
  const int* work;                // (b << 16) | tile, sorted by decreasing cost; or null
  int tiles_max;                  // without a work list: tiles per utterance enumerated
  int H;
  int q_rows_total;
  int causal;
  int psplit;                     // forward: P enters the P V product as two bf16 terms (hi + lo): the context, and with it
                                  // the backward's delta = rowsum(dO * O), is then consistent with the fp32 P the backward
                                  // recomputes - sum_k dS(q, k) = 0 to ~2^-16 instead of ~2^-9 (matters where the keys
                                  // are nearly identical and dQ / dK are differences of almost equal terms); small-Lq only
  float scale;                    // 1/sqrt(d_k)
  // K as the kernels read it is either the key projection itself (k_prescaled = 0) or K~ = scale * log2(e) * K, scaled ONCE in
  // the fp32 epilogue of the GEMM that produced it (k_prescaled = 1: st_row_chain's post_kscale): q . k~ is then the score in
  // the log2 domain as it leaves the matrix pipe - no multiply per score in any kernel, and forward and both backward bodies
  // exponentiate bit-identical scores.  (Pre-multiplying a register-resident operand and rounding it to bf16 a second time -
  // rounds 3 / 4 - perturbs every score by ~2^-9 |q||k|, differently in each kernel: profiles/r05_bwd64_accuracy_before.txt.)
  // c2: what a raw score q . k is multiplied by on its way into exp2 (scale * log2 e, or 1); dq_scale: what the dQ body's
  // sum dS K is multiplied by (scale, or ln 2 when the K it multiplied was K~).  dK = scale * dS^T Q is the gradient of the
  // UNSCALED key projection either way (d/dk = scale log2e * d/dk~, and dS_log2 = ln2 dS).
  float c2, dq_scale;
  DropArgs drop;                  // attention-probability dropout (Attention.py:89), training mode only
  // backward, few queries against many keys (st_attn.hip, KS = 2 dQ body): the keys of an item cut over `xsplit` workgroups,
  // fp32 partial dQ in xs_ws, one ticket per item - the last arriver adds the parts in order (st_common.cuh: last-arriver merges)
  float* xs_ws; unsigned* xs_tickets; int xsplit;
};

inline void set_score_scales(AttnArgs& a, float scale, int k_prescaled) {
  a.scale = scale;
  a.c2 = k_prescaled ? 1.f : scale * 1.4426950408889634f;
  a.dq_scale = k_prescaled ? 0.6931471805599453f : scale;
}

// blockIdx.x -> (utterance, head, tile)
__device__ __forceinline__ void decode_item(const AttnArgs& a, int bid, int& b, int& h, int& tile) {
  const int idx = bid / a.H;
  h = bid % a.H;
  if (a.work) {
    const int w = a.work[idx];
    b = w >> 16;
    tile = w & 0xffff;
  } else {
    b = idx / a.tiles_max;
    tile = idx % a.tiles_max;
  }
}

// ---- streamed [64 x DK] tiles ----------------------------------------------------------------------
// LDS image: natural rows, stride DK + 8 elements (144 B / 80 B): ds_read_b128 row fragments over 16 rows
// and the 4-row groups of ds_read_b64_tr_b16 are both bank-conflict free (DK = 64; 272 B for DK = 128 shifts 16 B per row likewise).
// DK = 128 (the reference's config/character.yaml: d_model 512, 4 heads) needs > 256 registers per lane: one workgroup per CU.
template <int DK, int ROWS = TILE> struct TileGeo {
  static constexpr int STR = DK + 8, E = ROWS * STR, CPR = DK / 8, CH = ROWS * CPR / 256;   // CH chunks per thread
};

template <int DK, int ROWS = TILE>
struct Stage {
  using G = TileGeo<DK, ROWS>;
  bf16x8 v[G::CH];
  // chunk id = tid + p*256 -> tile row id / CPR, 16-byte chunk id % CPR
  static __device__ __forceinline__ void offsets(uint32_t (&off)[G::CH], int ld) {
#pragma unroll
    for (int p = 0; p < G::CH; ++p) {
      const int id = threadIdx.x + p * 256;
      off[p] = ((uint32_t)(id / G::CPR) * (uint32_t)ld + (id % G::CPR) * 8) * 2u;
    }
  }
  // rows r0 .. r0+ROWS-1 of the utterance's column slice `base`; rows >= nvalid read row nvalid-1 - or, ZERO_TAIL, arrive as
  // zeros: a zero dO row (with a zero delta) or a zero K row takes a masked pair out of the backward's sums by itself
  // (dP = 0 and dS = P (0 - 0) = 0 for a query past the end; dQ += dS K with K = 0 for a key past the end), so the tile
  // that crosses a sequence end runs the unmasked path
  template <bool ZERO_TAIL = false>
  __device__ __forceinline__ void load(const uint32_t (&off)[G::CH], const bf16* __restrict__ base, int ld, int r0,
                                       int nvalid) {
    if (r0 + ROWS <= nvalid) {
      const char* tb = reinterpret_cast<const char*>(base + (size_t)r0 * ld);
#pragma unroll
      for (int p = 0; p < G::CH; ++p) v[p] = *reinterpret_cast<const bf16x8*>(tb + off[p]);
    } else {
#pragma unroll
      for (int p = 0; p < G::CH; ++p) {
        const int id = threadIdx.x + p * 256;
        const int row = min(r0 + id / G::CPR, nvalid - 1);
        v[p] = *reinterpret_cast<const bf16x8*>(base + (size_t)row * ld + (id % G::CPR) * 8);
        if (ZERO_TAIL && r0 + id / G::CPR >= nvalid) v[p] = zero_bf8();
      }
    }
  }
  __device__ __forceinline__ void store(bf16* tile) const {
#pragma unroll
    for (int p = 0; p < G::CH; ++p) {
      const int id = threadIdx.x + p * 256;
      *reinterpret_cast<bf16x8*>(tile + (id / G::CPR) * G::STR + (id % G::CPR) * 8) = v[p];
    }
  }
};

// Row fragment: elements t16*16 + hi*8 .. +7 of tile row R (A or B operand, contraction along DK).
template <int DK>
__device__ __forceinline__ bf16x8 rd_nat(const bf16* tile, int R, int t16) {
  const int hi = (threadIdx.x & 63) >> 5;
  return frag_nat(tile, TileGeo<DK>::STR, R, t16 * 16 + hi * 8);
}
// Transposing fragment: for column d0 + (lane & 31), the 8 tile rows base+0..3 and base+8..11
// (base already includes 4*hi) - the contraction runs over the tile's ROWS.
template <int DK>
__device__ __forceinline__ bf16x8 rd_tr(const bf16* tile, int d0, int base) {
  return frag_tr(tile, TileGeo<DK>::STR, d0, base, base + 8);
}

// The software pipeline shared by the three kernels.  load(set, tile) fills register set `set`,
// store(set) writes it to LDS buffer `set`, compute(buf, tile) consumes LDS buffer `buf`.
// Steady state has no conditionals, so the compiler's s_waitcnt vmcnt() stays counted: the loads of
// tile it+2 remain in flight across the LDS store of tile it+1.
template <typename L, typename S, typename C>
__device__ __forceinline__ void stream_tiles(int ntiles, L load, S store, C compute) {
  if (ntiles <= 0) return;   // (workgroup-uniform) nothing visible: the accumulators stay zero
  load(0, 0);
  if (ntiles > 1) load(1, 1);
  store(0);
  __syncthreads();
  int it = 0;
  for (; it + 3 < ntiles; it += 2) {
    load(0, it + 2);
    compute(0, it);
    store(1);
    __syncthreads();
    load(1, it + 3);
    compute(1, it + 1);
    store(0);
    __syncthreads();
  }
  // tail: 1..3 tiles left; tile `it` is in LDS buffer 0, tile it+1 (if any) in register set 1
  if (it + 2 < ntiles) load(0, it + 2);
  compute(0, it);
  if (it + 1 < ntiles) {
    store(1);
    __syncthreads();
    compute(1, it + 1);
    if (it + 2 < ntiles) {
      store(0);
      __syncthreads();
      compute(0, it + 2);
    }
  }
  __syncthreads();   // the epilogue reuses the tile buffers
}

#ifndef ST_ATTN_SC1
#define ST_ATTN_SC1 1      // the attention kernels' row stores write-through (store16_wt): whole step 2.844 -> 2.826 ms, same box (round 6); 0 = plain
#endif
// Store a transposed accumulator tile (lane = row, registers = DK columns) as coalesced rows:
// through a wave-private [32][DK] LDS patch so HBM sees whole DK*2-byte row segments instead of
// 64 scattered 8-byte writes per instruction.  `patch` is this wave's private 32*DK elements.
template <int DK, bool RESID = false>
__device__ __forceinline__ void store_rows(bf16* patch, const f32x16* acc, float mul, bf16* gbase, int ld, int row0,
                                           int nvalid_rows) {
  constexpr int ND = DK / 32, CPR = DK / 8;
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = acc[d][4 * g + e] * mul;
        v[e] = RESID ? (bf16)(x - (float)(bf16)x) : (bf16)x;      // RESID: what the bf16 rounding of x dropped
      }
      const int col = d * 32 + 8 * g + 4 * hi;
      *reinterpret_cast<bf16x4*>(patch + r * DK + (((col >> 3) ^ (r & (CPR - 1))) << 3) + (col & 7)) = v;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int p = 0; p < 32 * CPR / 64; ++p) {
    const int id = p * 64 + l, rr = id / CPR, c = id % CPR;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(patch + rr * DK + ((c ^ (rr & (CPR - 1))) << 3));
    if (rr < nvalid_rows) store16<ST_ATTN_SC1>(gbase + (size_t)(row0 + rr) * ld + c * 8, v);
  }
}

// The same for a value AND what its bf16 rounding dropped (O, Ores), in ONE pass: both images are written to two
// patches, one LDS wait, both are read back and stored - half the dependent LDS round trips of two store_rows calls
// (the attention epilogue is a latency tail: every wave of the workgroup is in it at the same time).
template <int DK>
__device__ __forceinline__ void store_rows_pair(bf16* patch_hi, bf16* patch_lo, const f32x16* acc, float mul, bf16* g_hi,
                                                bf16* g_lo, int ld, int row0, int nvalid_rows) {
  constexpr int ND = DK / 32, CPR = DK / 8;
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 vh, vl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = __fmul_rn(acc[d][4 * g + e], mul);      // (explicitly rounded product and difference: with the default
        vh[e] = (bf16)x;                                        // contraction the compiler is free to form fma(acc, mul, -vh) in one
        vl[e] = (bf16)__fsub_rn(x, (float)vh[e]);               // kernel and not in another - the residual then differs by an ulp)
      }
      const int col = d * 32 + 8 * g + 4 * hi;
      const int at = r * DK + (((col >> 3) ^ (r & (CPR - 1))) << 3) + (col & 7);
      *reinterpret_cast<bf16x4*>(patch_hi + at) = vh;
      *reinterpret_cast<bf16x4*>(patch_lo + at) = vl;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int p = 0; p < 32 * CPR / 64; ++p) {
    const int id = p * 64 + l, rr = id / CPR, c = id % CPR;
    const int at = rr * DK + ((c ^ (rr & (CPR - 1))) << 3);
    const bf16x8 vh = *reinterpret_cast<const bf16x8*>(patch_hi + at);
    const bf16x8 vl = *reinterpret_cast<const bf16x8*>(patch_lo + at);
    if (rr < nvalid_rows) {
      store16<ST_ATTN_SC1>(g_hi + (size_t)(row0 + rr) * ld + c * 8, vh);
      store16<ST_ATTN_SC1>(g_lo + (size_t)(row0 + rr) * ld + c * 8, vl);
    }
  }
}

// ---- pieces of the hand-scheduled streams' kernels (st_attn_bwd64.hip, st_attn64.hip) ------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const int __attribute__((address_space(4)))* sptr_t;      // constant address space: uniform addresses load on the scalar unit

template <typename T> __device__ __forceinline__ const T* uniform_ptr(const T* p) {
  const uint64_t u = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return (const T*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int sload(const int* p, int i) { return reinterpret_cast<sptr_t>((__UINTPTR_TYPE__)p)[i]; }

// blockIdx -> (utterance, head, tile) and the utterance's lengths / offsets, all through SCALAR loads: the general kernels'
// decode_item goes through three dependent vector-memory round trips (work list, lengths, offsets) before the first useful
// load can be asked for - 1.7 us of every item's ~20 (tools/dev/attn_bwd64_trace.py)
struct Item { int b, h, tile, lq, lk, qo, ko; };
__device__ __forceinline__ Item decode_scalar(const AttnArgs& a, int bid) {
  Item it;
  const int idx = bid / a.H;
  it.h = bid % a.H;
  if (a.work) {
    const int w = sload(a.work, idx);
    it.b = w >> 16;
    it.tile = w & 0xffff;
  } else {
    it.b = idx / a.tiles_max;
    it.tile = idx % a.tiles_max;
  }
  it.lq = sload(a.q_len, it.b); it.lk = sload(a.k_len, it.b); it.qo = sload(a.q_off, it.b); it.ko = sload(a.k_off, it.b);
  return it;
}

__device__ __forceinline__ bf16x8 scaled8(const bf16x8 v, float c) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (bf16)((float)v[e] * c);
  return r;
}

// this thread's two 16-byte chunks (rows tid >> 3 and 32 + (tid >> 3), chunk tid & 7) of the first 64-row tile of a streamed
// matrix whose range ends with row `rows - 1` (rows past it read as zeros): asked for BEFORE the register-resident fragments,
// so that one memory round trip serves both (the streams' own loads start at tile 1)
__device__ __forceinline__ void tile0(const bf16* base, int ld2, int rows, u32x4& c0, u32x4& c1) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (rows - 1) * ld2 + 128, 0x00020000);
  const unsigned off = (threadIdx.x >> 3) * (unsigned)ld2 + (threadIdx.x & 7) * 16u;
  c0 = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  c1 = __builtin_amdgcn_raw_buffer_load_b128(r, off + 32u * (unsigned)ld2, 0, 0);
}


// ---------------------------------------------------------------------------------------------
// Forward.  Each wave owns 32 query rows (lane & 31); key / value tiles are streamed.
// ---------------------------------------------------------------------------------------------
// Keep decisions of one lane's 16 accumulator registers: register r <-> pair (fixed, var0 + acc_row(r, hi)).
// FIXED_IS_Q: the lane's own index is the query (forward / dQ: registers run over keys), else it is the key.
template <bool FIXED_IS_Q>
__device__ __forceinline__ void keep16(const Drop& dr, int bh, int fixed, int var0, int hi, bool (&keep)[16]) {
  // Lanes l and l ^ 1 hold neighbouring fixed indices (2f, 2f + 1: tile origins are even) and therefore need the SAME
  // eight hashes (a 2 x 2 block serves both).  Each computes four - the even lane register groups g = 0, 1, the odd
  // lane g = 2, 3 - and fetches the other four from its neighbour with a DPP move (the hash costs two quarter-rate
  // v_mul_lo_u32; this halves what dropout adds to the VALU-bound softmax).
  const int par = threadIdx.x & 1, fx = fixed & 1;
  uint32_t own[4], nb[4];
#pragma unroll
  for (int gi = 0; gi < 2; ++gi)
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const int v = var0 + 8 * (2 * par + gi) + 4 * hi + 2 * hb;
      own[gi * 2 + hb] = FIXED_IS_Q ? dr.bits(drop_counter_qk(bh, fixed, v)) : dr.bits(drop_counter_qk(bh, v, fixed));
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) nb[j] = (uint32_t)__builtin_amdgcn_mov_dpp((int)own[j], 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const int j = (g & 1) * 2 + hb;
      uint32_t bits = ((g >> 1) == par) ? own[j] : nb[j];
      // byte of (q, k) inside its block = 2 * (q & 1) + (k & 1); this lane's two elements differ in the variable index
      bits >>= FIXED_IS_Q ? 16 * fx : 8 * fx;
#pragma unroll
      for (int e = 0; e < 2; ++e)
        keep[4 * g + 2 * hb + e] = (int)((bits >> (FIXED_IS_Q ? 8 * e : 16 * e)) & 0xffu) >= dr.thresh;
    }
}


// two accumulator tiles (lane = row) -> coalesced rows through two wave-private LDS patches: ONE wait for both images
template <int DK>
__device__ __forceinline__ void store_rows_2(bf16* pa, bf16* pb, const f32x16* acca, const f32x16* accb, float mula, float mulb,
                                             bf16* ga, int lda, bf16* gb, int ldb, int row0, int nvalid_rows) {
  constexpr int ND = DK / 32, CPR = DK / 8;
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 va, vb;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        va[e] = (bf16)(acca[d][4 * g + e] * mula);
        vb[e] = (bf16)(accb[d][4 * g + e] * mulb);
      }
      const int col = d * 32 + 8 * g + 4 * hi;
      const int at = r * DK + (((col >> 3) ^ (r & (CPR - 1))) << 3) + (col & 7);
      *reinterpret_cast<bf16x4*>(pa + at) = va;
      *reinterpret_cast<bf16x4*>(pb + at) = vb;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int p = 0; p < 32 * CPR / 64; ++p) {
    const int id = p * 64 + l, rr = id / CPR, c = id % CPR;
    const int at = rr * DK + ((c ^ (rr & (CPR - 1))) << 3);
    const bf16x8 va = *reinterpret_cast<const bf16x8*>(pa + at);
    const bf16x8 vb = *reinterpret_cast<const bf16x8*>(pb + at);
    if (rr < nvalid_rows) {
      store16<ST_ATTN_SC1>(ga + (size_t)(row0 + rr) * lda + c * 8, va);
      store16<ST_ATTN_SC1>(gb + (size_t)(row0 + rr) * ldb + c * 8, vb);
    }
  }
}


}  // namespace

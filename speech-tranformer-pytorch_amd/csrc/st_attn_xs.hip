// Attention forward for FEW queries against MANY keys with 64-wide heads: the decoder-encoder attention of a training step
// (<= 64 target positions against ~1000 encoder frames, Attention.py:82-90 as called from Layers.py:41) and of a beam-search
// step (the `beam` hypotheses of an utterance against its encoder keys, Decode.py:75-112).
//
// Such a launch is ~1 GFLOP in total: its duration is launch + prologue + the SERIAL chain of key tiles + epilogue, not
// arithmetic, and the chain is memory LATENCY: a workgroup must pull its (utterance, head)'s K and V (192 KB for 750 keys) through
// one CU, and a load issued when its data is needed costs ~2 us.  The general kernel (st_attn.hip, KS = 2) walks the chain with
// 4 waves - two halves of a 128-key stage per step behind one workgroup barrier, ~6 steps for 750 keys.  Here:
//   * one workgroup = 8 waves per (utterance, head, 32-QUERY tile) - 256 workgroups at config 2 (two tiles per utterance),
//     128 in a beam-10 decode step; wave w owns the 32-key blocks w, w + 8, w + 16, ... and keeps THREE blocks in flight in
//     registers: for <= 768 keys every load of the problem is issued in the prologue and the latency is paid once;
//   * K fragments go straight from global memory to the MFMA A operand (lane = key row: one 16-byte load per k-step);
//     V needs the transposing LDS read, so each wave stages ITS 32 x 64 V block in a private 4.5 KB LDS patch -
//     LDS operations of one wave execute in order, so there is NO barrier and no wait between waves inside the loop;
//   * each wave keeps its own running (m, l, O); after the loop the eight partial states meet in LDS (70 KB, aliasing the V
//     patches behind one barrier), wave w combines four of the 32 output registers of all eight, and the rows leave as whole
//     128-byte segments.  The exchange with the lane's other half (l ^ 32) is a v_permlane32_swap, not a ds_bpermute.
//     Two-term P (AttnArgs::psplit), Ores, the LSE and attention dropout as in the general kernel.
// Served through st_attn_fwd (same C-ABI entry, same arguments; st_attn_tile_rows tells the host the 32-row tile).
// Measured on the way (same box, config 2's decoder-encoder shape, general kernel 14.7 us): 64 queries per workgroup with one
// register stage 14.0-14.7 us - prologue 3.1, merge + stores 2.5, first block 3.7, every further block 2.35 us whatever its
// arithmetic (reduction trees and lockstep query blocks: no change): each block waited for loads issued half a block earlier.
#include "st_attn_common.cuh"
#include "st_rowchain_common.cuh"

namespace {

constexpr int XW = 8;                 // waves per workgroup
constexpr int XQ = 32;                // query rows per workgroup
constexpr int XNS = 3;                // 32-key blocks in flight per wave
constexpr int XVS = 72;               // V patch row stride (elements): as TileGeo<64>::STR
constexpr int XSLOTS = 34;            // exchange slots per (wave, query block): m, l, 32 output registers

// max / sum of a value with its partner lane (l ^ 32: the other half of the same query's scores) as ONE v_permlane32_swap
// instead of a ds_bpermute round trip through the LDS crossbar: swap(v, v) yields {lower half's v in both halves, upper
// half's v in both halves}
__device__ __forceinline__ float half_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

struct XStage {                       // one 32-key block in flight: K fragments + this lane's four V chunks
  bf16x8 k[4];
  bf16x8 v[4];
};

// F1: the chain stage in front of the decoder-encoder attention - cur = LN(ctx Wo^T + bo + x) (the self-attention's output_linear +
// residual + LayerNorm, Attention.py:92-94) and q = cur Wq^T + bq (Attention.py:74) - computed HERE for the workgroup's 32 rows from
// the same weight-fragment streams st_row_chain reads (csrc/st_rowchain_common.cuh), instead of by a launch of its own: the four
// heads' workgroups of a row tile repeat it (2 x 128 KB of weights from L2 each; head 0 writes out / xhat / rstd / q for the
// backward), the K / V requests of the attention are already in flight underneath, and q never leaves the chip on its way to the
// scores.  A decoder-sized st_row_chain of two blocks is 14 us of launch + prologue + epilogue for 1.2 us of matrix work.
struct XF1Args {
  const bf16* A; int lda;            // the self-attention context [rows, 256]
  const bf16* R; int ldr;            // the residual (the layer input)
  const bf16x8* wfrag; int wave_frags; int next_frags;
  float eps;
  const float* bo; const float* g0; const float* be0;
  bf16* out0; bf16* xhat0; float* rstd0;
  const float* bp; bf16* P; int ldp; // q projection: bias, output [rows, 256]
};

// SELF: the decoder's causal self-attention (Layers.py:39, Attention.py:82-90 on <= 64 target positions) in front of that chain
// stage, for the workgroup's 32 query rows and ALL heads (waves 0 .. H-1 take a head each: <= 64 keys, one tile) - the context goes
// straight into the stage's LDS tile instead of through a launch of its own and HBM.  The arithmetic is attn_fwd_kernel's
// (st_attn.hip: one 64-key tile, two-term P, the same order of operations): the saved O and lse are bit-identical, Ores (the
// bf16 residual of O) agrees to a few 1e-6 of O - inside the 2^-16 the pair promises.
struct XSelfArgs {
  const bf16* Q; const bf16* K; const bf16* V; int ld;      // the layer's q | k | v projection [rows, ld], head h at columns h * 64
  bf16* O; bf16* Ores; int ldo;                              // context and its bf16 residual (Ores may be null: no backward)
  float* lse;                                                // [H][q_rows_total], log2 domain
  DropArgs drop;
};

template <bool DROP, bool PS, bool F1, bool SELF = false, bool SDROP = false>
__global__ __launch_bounds__(512, 1) void attn_xs_fwd_kernel(AttnArgs a, XF1Args f, XSelfArgs sa) {
  constexpr int DK = 64;
  __shared__ __attribute__((aligned(16))) float xch[XW * XSLOTS * 64 + 1024];   // 73,728 B (the V patches alias its head; SELF: 4 x 64-key V patches behind two tiles)
  __shared__ __attribute__((aligned(16))) bf16 patch[2 * 32 * DK];              // O rows (hi), Ores rows (lo)
  __shared__ __attribute__((aligned(16))) float f1vec[F1 ? 4 * 256 : 4];         // the chain stage's bo | g0 | be0 | bq: epilogues read them from LDS (LVec)

  int b, h, tile;
  decode_item(a, blockIdx.x, b, h, tile);
  const int lq = a.q_len[b], lk = a.k_len[b];
  const int q0 = tile * XQ;
  if (q0 >= lq) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hi = l >> 5, r32 = l & 31;
  const float c2 = a.c2;
  const Drop dr = make_drop(a.drop);
  const int bh = b * a.H + h;
  const bf16* kbase = a.K + (size_t)a.k_off[b] * a.ldk + h * DK;
  const bf16* vbase = a.V + (size_t)a.k_off[b] * a.ldv + h * DK;
  const size_t qrow0 = (size_t)a.q_off[b];
  const int q = q0 + r32;

  bf16* vpatch = reinterpret_cast<bf16*>(xch) + wave * 32 * XVS;
  const int nblk = (lk + 31) >> 5;
  auto fetch = [&](XStage& st, int blk) {       // rows past the last key are clamped onto it (finite data; masked below)
    const int k0 = blk * 32;
    const size_t krow = (size_t)min(k0 + r32, lk - 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) st.k[t] = *reinterpret_cast<const bf16x8*>(kbase + krow * a.ldk + t * 16 + hi * 8);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int id = l + p * 64;
      st.v[p] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)min(k0 + (id >> 3), lk - 1) * a.ldv + (id & 7) * 8);
    }
  };

  XStage st0, st1, st2;
  int blk = wave;
  auto fetch_all = [&]() {
    if (blk < nblk) fetch(st0, blk);
    if (blk + XW < nblk) fetch(st1, blk + XW);
    if (blk + 2 * XW < nblk) fetch(st2, blk + 2 * XW);
  };
  if (!SELF) fetch_all();      // (SELF: behind the self-attention - its registers are needed there)

  bf16x8 qf[4];
  if (F1) {
    constexpr int TE = 32 * AS;
    bf16* cur = reinterpret_cast<bf16*>(xch);       // three activation tiles + the LayerNorm exchange: 52,736 B of the 69,632
    bf16* f0 = cur + TE;
    bf16* f1 = cur + 2 * TE;
    float (*red)[NW * 32] = reinterpret_cast<float (*)[NW * 32]>(xch + 3 * TE / 2);
    Ctx<1> c;
    c.tid = threadIdx.x; c.wave = wave; c.l = l; c.hi = hi; c.r = r32;
    c.row0 = (int)qrow0 + q0; c.nvalid = min(32, lq - q0);
    c.ws = f.wfrag + (size_t)wave * f.wave_frags * 64;
    auto ring_start = [&]() {
#pragma unroll
      for (int i = 0; i < Ring<1>::D; ++i) c.ring[i] = c.ws[i * 64 + l];
      c.ws += Ring<1>::D * 64;
    };
    if (!SELF) ring_start();
    // the chain stored behind this one (the layer's feed-forward chain) runs right after this launch: warm its streams as the
    // st_row_chain launch this stage replaces did (the workgroups of an XCD deal the 128-byte lines among their threads)
    int touched[TOUCH];
    auto touch_all = [&]() {
      const int nlines = NW * (f.wave_frags + f.next_frags) * 8;
      const int xcd = blockIdx.x & 7, nr = ((int)gridDim.x - xcd + 7) >> 3;
      const char* sb = reinterpret_cast<const char*>(f.wfrag);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TOUCH; ++t) {
        const int ln = min(((int)blockIdx.x >> 3) * 512 + (int)threadIdx.x + t * nr * 512, nlines - 1);
        touched[t] = *reinterpret_cast<const int*>(sb + (size_t)ln * 128);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if (!SELF) touch_all();
    const bool writer = h == 0;                     // one of the four heads' workgroups stores the stage's outputs
    // the stage's four epilogue vectors -> LDS (wave w < 4: vector w), requested with the residual tile: a vector load inside an
    // epilogue would wait behind every K / V piece and weight fragment on request (vmcnt is in-order)
    const float* vsrc = wave == 0 ? f.bo : wave == 1 ? f.g0 : wave == 2 ? f.be0 : f.bp;
    f32x4 vreg = {0.f, 0.f, 0.f, 0.f};
    if (wave < 4) vreg = *reinterpret_cast<const f32x4*>(vsrc + 4 * l);
    auto vec_store = [&]() { if (wave < 4) *reinterpret_cast<f32x4*>(f1vec + wave * 256 + 4 * l) = vreg; };
    const LVec lbo{(const ST_LDS float*)f1vec}, lg0{(const ST_LDS float*)f1vec + 256}, lbe0{(const ST_LDS float*)f1vec + 512},
        lbp{(const ST_LDS float*)f1vec + 768};
    if (SELF) {
      TileRegs<1> rr;
      tile_load(c, f.R, f.ldr, rr);
      if (wave < a.H) {      // ---- causal self-attention of head `wave` for the 32 queries: attn_fwd_kernel<64, *, 1, true>'s arithmetic
        const int hs = wave;
        const Drop sdr = make_drop(sa.drop);
        const int sbh = b * a.H + hs;
        const size_t srow = qrow0 + min(q, lq - 1);
        bf16x8 sq[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) sq[t] = *reinterpret_cast<const bf16x8*>(sa.Q + srow * sa.ld + hs * DK + t * 16 + hi * 8);
        bf16* vp = reinterpret_cast<bf16*>(xch) + 2 * TE + wave * 64 * XVS;        // this wave's 64-key V patch (f1 and behind)
        f32x16 s2[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const size_t krow = qrow0 + min(kb * 32 + r32, lq - 1);               // keys past the end: clamped (finite), masked below
          bf16x8 kf[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) kf[t] = *reinterpret_cast<const bf16x8*>(sa.K + krow * sa.ld + hs * DK + t * 16 + hi * 8);
#pragma unroll
          for (int p2 = 0; p2 < 4; ++p2) {
            const int id = l + p2 * 64;
            const bf16x8 vv = *reinterpret_cast<const bf16x8*>(sa.V + (qrow0 + min(kb * 32 + (id >> 3), lq - 1)) * sa.ld + hs * DK + (id & 7) * 8);
            *reinterpret_cast<bf16x8*>(vp + (kb * 32 + (id >> 3)) * XVS + (id & 7) * 8) = vv;
          }
          s2[kb] = zero16();
#pragma unroll
          for (int t = 0; t < 4; ++t) s2[kb] = mfma32(kf[t], sq[t], s2[kb]);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + acc_row(r, hi);
            if (key >= lq || key > q) s2[kb][r] = -INFINITY;
          }
        }
        asm volatile("" ::: "memory");
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s2[kb][r]);
        mx = fmaxf(mx, wave_xor32(mx));
        const float sm = mx * c2;                     // (key 0 is visible to every query: finite)
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(fmaf(s2[kb][r], c2, -sm));
            s2[kb][r] = pv;
            psum += pv;
          }
        if (SDROP) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            bool keep[16];
            keep16<true>(sdr, sbh, q, kb * 32, hi, keep);
#pragma unroll
            for (int r = 0; r < 16; ++r) s2[kb][r] = keep[r] ? s2[kb][r] : 0.f;
          }
        }
        f32x16 so[2];
        so[0] = zero16();
        so[1] = zero16();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const bf16x8 pf = pack_acc8(s2[kb], 8 * hf);
            bf16x8 pl;
#pragma unroll
            for (int j = 0; j < 8; ++j) pl[j] = (bf16)(s2[kb][8 * hf + j] - (float)pf[j]);
            const int base = kb * 32 + 16 * hf + 4 * hi;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              const bf16x8 vf = frag_tr(vp, XVS, d * 32, base, base + 8);
              so[d] = mfma32(vf, pf, so[d]);
              so[d] = mfma32(vf, pl, so[d]);
            }
          }
        const float ltot = psum + wave_xor32(psum);
        const float inv = ltot > 0.f ? (SDROP ? sdr.scale : 1.f) / ltot : 0.f;
        if (writer && q < lq && hi == 0) sa.lse[(size_t)hs * a.q_rows_total + qrow0 + q] = sm + log2f(ltot);
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bf16x4 vh, vl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float xv = __fmul_rn(so[d][4 * g + e], inv);
              vh[e] = (bf16)xv;
              vl[e] = (bf16)__fsub_rn(xv, (float)vh[e]);
            }
            const int col = hs * DK + d * 32 + 8 * g + 4 * hi;
            *reinterpret_cast<bf16x4*>(cur + r32 * AS + col) = vh;             // the chain stage's operand tile
            if (writer && q < lq) {
              *reinterpret_cast<bf16x4*>(sa.O + (qrow0 + q) * sa.ldo + col) = vh;
              if (sa.Ores) *reinterpret_cast<bf16x4*>(sa.Ores + (qrow0 + q) * sa.ldo + col) = vl;
            }
          }
      }
      tile_store(c, rr, f0);
      vec_store();
      fetch_all();
      ring_start();
      touch_all();
    } else {
      TileRegs<1> ra, rr;
      tile_load(c, f.A, f.lda, ra);
      tile_load(c, f.R, f.ldr, rr);
      tile_store(c, ra, cur);
      tile_store(c, rr, f0);
      vec_store();
    }
    const Drop off = make_drop(DropArgs{nullptr, 0u, 0, 1.f});
    __syncthreads();
    f32x16 acc[1];
    zero_acc(acc);
    block_mma(c, cur, acc);
    epi_ln<false>(c, acc, lbo, f0, lg0, lbe0, f.eps, off, cur, f1, red, writer ? f.out0 : nullptr, writer ? f.xhat0 : nullptr,
                  writer ? f.rstd0 : nullptr);
    zero_acc(acc);
    block_mma(c, f1, acc);                          // q = cur Wq^T (+ bq): staged in f0 (the residual: last read before epi_ln's barriers)
    epi_store<false, false>(c, acc, lbp, f0, off, 0, 0);
    __syncthreads();
    if (writer) tile_out(c, f0, f.P, f.ldp);
#pragma unroll
    for (int t = 0; t < 4; ++t) qf[t] = frag_nat(f0, AS, r32, h * DK + t * 16 + hi * 8);
    {
      int tsum = 0;
#pragma unroll
      for (int t = 0; t < TOUCH; ++t) tsum ^= touched[t];
      if (tsum == 0x5a5a5a5a && lq < 0) red[0][0] = 1.f;      // (never true: keeps the warm-up loads alive)
    }
    __syncthreads();                                // the tiles are dead: the waves' V patches take their place
  } else {
    const size_t row = qrow0 + min(q, lq - 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) qf[t] = *reinterpret_cast<const bf16x8*>(a.Q + row * a.ldq + h * DK + t * 16 + hi * 8);
  }

  f32x16 o[2];
  o[0] = zero16();
  o[1] = zero16();
  float m = -INFINITY, lsum = 0.f;

  // one 32-key block: V -> the wave's LDS patch, the scores (that frees the K fragments), THEN the request for the block three
  // ahead into the same registers, then softmax + P V
  auto block = [&](XStage& st, int blk) {
    const int k0 = blk * 32;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int id = l + p * 64;
      *reinterpret_cast<bf16x8*>(vpatch + (id >> 3) * XVS + (id & 7) * 8) = st.v[p];
    }
    asm volatile("" ::: "memory");     // (compiler order only: the hardware runs one wave's LDS operations in issue order)
    f32x16 s = zero16();
#pragma unroll
    for (int t = 0; t < 4; ++t) s = mfma32(st.k[t], qf[t], s);
    if (blk + XNS * XW < nblk) fetch(st, blk + XNS * XW);
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] *= c2;
    if (k0 + 32 > lk) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (k0 + acc_row(r, hi) >= lk) s[r] = -INFINITY;
    }
    const float a0 = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), a1 = fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7]));
    const float a2 = fmaxf(fmaxf(s[8], s[9]), fmaxf(s[10], s[11])), a3 = fmaxf(fmaxf(s[12], s[13]), fmaxf(s[14], s[15]));
    const float m_new = fmaxf(m, half_max(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3))));      // finite: key k0 itself is visible
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    m = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
    const float b0 = (s[0] + s[1]) + (s[2] + s[3]), b1 = (s[4] + s[5]) + (s[6] + s[7]);
    const float b2 = (s[8] + s[9]) + (s[10] + s[11]), b3 = (s[12] + s[13]) + (s[14] + s[15]);
    lsum = lsum * alpha + ((b0 + b1) + (b2 + b3));
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    if (DROP) {   // dropped probabilities leave the normaliser untouched; the 1/(1-p) scale is folded into the final 1/l
      bool keep[16];
      keep16<true>(dr, bh, q, k0, hi, keep);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = keep[r] ? s[r] : 0.f;
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const bf16x8 pf = pack_acc8(s, 8 * hf);
      bf16x8 pl;
      if (PS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pl[j] = (bf16)(s[8 * hf + j] - (float)pf[j]);
      }
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const bf16x8 vf = frag_tr(vpatch, XVS, d * 32, 16 * hf + 4 * hi, 16 * hf + 4 * hi + 8);
        o[d] = mfma32(vf, pf, o[d]);
        if (PS) o[d] = mfma32(vf, pl, o[d]);
      }
    }
    asm volatile("" ::: "memory");     // the next block's V store stays behind this block's transposing reads
  };

  while (blk < nblk) {
    block(st0, blk);
    blk += XW;
    if (blk >= nblk) break;
    block(st1, blk);
    blk += XW;
    if (blk >= nblk) break;
    block(st2, blk);
    blk += XW;
  }
  __syncthreads();            // every wave is done with its V patch: the exchange area may overwrite them

  // ---- the eight partial states -> LDS ------------------------------------------------------------------------------
  {
    float* x = xch + (size_t)(wave * XSLOTS) * 64 + l;
    x[0] = m;
    x[64] = half_sum(lsum);
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[(2 + d * 16 + r) * 64] = o[d][r];
  }
  __syncthreads();
  // ---- wave w combines output registers [4 w, 4 w + 4): column block w >> 2, register group w & 3 -------------------------
  {
    float mu[XW], lu[XW];
    float M = -INFINITY;
#pragma unroll
    for (int u = 0; u < XW; ++u) {
      const float* x = xch + (size_t)(u * XSLOTS) * 64 + l;
      mu[u] = x[0];
      lu[u] = x[64];
      M = fmaxf(M, mu[u]);
    }
    float L = 0.f, au[XW];
#pragma unroll
    for (int u = 0; u < XW; ++u) {
      au[u] = (mu[u] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mu[u] - M);
      L += au[u] * lu[u];
    }
    float val[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < XW; ++u) {
      const float* x = xch + (size_t)(u * XSLOTS + 2 + 4 * wave) * 64 + l;
#pragma unroll
      for (int e = 0; e < 4; ++e) val[e] += au[u] * x[e * 64];
    }
    const float inv = L > 0.f ? (DROP ? dr.scale : 1.f) / L : 0.f;
    bf16x4 vh, vl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xv = __fmul_rn(val[e], inv);
      vh[e] = (bf16)xv;
      vl[e] = (bf16)__fsub_rn(xv, (float)vh[e]);
    }
    const int col = (wave >> 2) * 32 + 8 * (wave & 3) + 4 * hi;
    *reinterpret_cast<bf16x4*>(patch + r32 * DK + col) = vh;
    *reinterpret_cast<bf16x4*>(patch + 32 * DK + r32 * DK + col) = vl;
    if (wave == 0 && hi == 0 && q < lq && a.lse) a.lse[(size_t)h * a.q_rows_total + qrow0 + q] = M + log2f(L);
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    const int row = threadIdx.x >> 3, c8 = threadIdx.x & 7;
    if (q0 + row < lq) {
      const size_t at = (qrow0 + q0 + row) * a.ldo + h * DK + c8 * 8;
      *reinterpret_cast<bf16x8*>(a.O + at) = *reinterpret_cast<const bf16x8*>(patch + row * DK + c8 * 8);
      if (a.Ores) *reinterpret_cast<bf16x8*>(a.Ores + at) = *reinterpret_cast<const bf16x8*>(patch + 32 * DK + row * DK + c8 * 8);
    }
  }
}

}  // namespace

extern "C" int st_attn_xs_tile_rows() { return XQ; }

extern "C" int st_attn_xs_fwd_launch(hipStream_t stream, const void* args_, int grid_x, int drop, const void* f1_, const void* self_) {
  const AttnArgs& a = *static_cast<const AttnArgs*>(args_);
  dim3 grid(grid_x), block(512);
  XF1Args f = {};
  XSelfArgs sa = {};
  if (f1_) f = *static_cast<const XF1Args*>(f1_);
  if (self_) sa = *static_cast<const XSelfArgs*>(self_);
  const bool sdrop = self_ && sa.drop.seed != nullptr && sa.drop.thresh > 0;
#define ST_XS(DR, PSS) \
  do { if (self_ && sdrop) hipLaunchKernelGGL((attn_xs_fwd_kernel<DR, PSS, true, true, true>), grid, block, 0, stream, a, f, sa); \
       else if (self_) hipLaunchKernelGGL((attn_xs_fwd_kernel<DR, PSS, true, true, false>), grid, block, 0, stream, a, f, sa); \
       else if (f1_) hipLaunchKernelGGL((attn_xs_fwd_kernel<DR, PSS, true>), grid, block, 0, stream, a, f, sa); \
       else hipLaunchKernelGGL((attn_xs_fwd_kernel<DR, PSS, false>), grid, block, 0, stream, a, f, sa); } while (0)
  if (drop && a.psplit) ST_XS(true, true);
  else if (drop) ST_XS(true, false);
  else if (a.psplit) ST_XS(false, true);
  else ST_XS(false, false);
#undef ST_XS
  return (int)hipGetLastError();
}

extern "C" void st_attn_xs_self_args(void* out, const void* Q, const void* K, const void* V, int ld, void* O, void* Ores, int ldo,
                                     float* lse, const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale) {
  XSelfArgs s;
  s.Q = (const bf16*)Q; s.K = (const bf16*)K; s.V = (const bf16*)V; s.ld = ld; s.O = (bf16*)O; s.Ores = (bf16*)Ores; s.ldo = ldo; s.lse = lse;
  const bool on = drop_seed != nullptr && drop_thresh > 0;
  s.drop.seed = on ? drop_seed : nullptr; s.drop.salt = drop_salt; s.drop.thresh = on ? drop_thresh : 0; s.drop.scale = on ? drop_scale : 1.f;
  *static_cast<XSelfArgs*>(out) = s;
}
extern "C" int st_attn_xs_self_args_size() { return (int)sizeof(XSelfArgs); }

// the F1 stage's arguments as st_attn.hip hands them over (the struct is local to this translation unit)
extern "C" void st_attn_xs_f1_args(void* out, const void* A, int lda, const void* R, int ldr, const void* wfrag, int n_blocks,
                                   int next_blocks, float eps, const float* bo, const float* g0, const float* be0, void* out0,
                                   void* xhat0, float* rstd0, const float* bq, void* Qout, int ldq) {
  XF1Args f;
  f.A = (const bf16*)A; f.lda = lda; f.R = (const bf16*)R; f.ldr = ldr; f.wfrag = (const bf16x8*)wfrag;
  f.wave_frags = n_blocks * 16 + DEPTH; f.next_frags = next_blocks > 0 ? next_blocks * 16 + DEPTH : 0; f.eps = eps;
  f.bo = bo; f.g0 = g0; f.be0 = be0; f.out0 = (bf16*)out0; f.xhat0 = (bf16*)xhat0; f.rstd0 = rstd0; f.bp = bq; f.P = (bf16*)Qout;
  f.ldp = ldq;
  *static_cast<XF1Args*>(out) = f;
}
extern "C" int st_attn_xs_f1_args_size() { return (int)sizeof(XF1Args); }

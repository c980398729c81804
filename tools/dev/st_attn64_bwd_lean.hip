// Attention forward for LONG non-causal problems with 64-wide heads (the encoder's self-attention).
//
// Measured on the MI355X (tools/dev/issue_probe.hip, profiles/r03_issue_probe.txt): one wave issues one instruction per
// ~4.5-5 clocks; a 32x32x16 MFMA occupies its SIMD's matrix pipe for 32; ~5 VALU instructions hide under one MFMA of the
// same wave, every further one costs its full issue slot; and two waves of a SIMD overlap each other's MFMA and VALU work
// by only 10-30 % (a SIMD issues about one instruction per 5 clocks whichever wave it comes from, once MFMAs are in the
// mix).  At d_k = 64 a 32 x 32 score block is 8 MFMAs against 16 scores per lane, so the general kernel of st_attn.hip
// (scale + running maximum + rescale test + exp + sum + convert: 12.6 VALU per MFMA by PMC) is bound by instruction
// issue at 2.5x its MFMA time.  Here the softmax is cut to what cannot be avoided:
//   * Q is multiplied by scale * log2(e) once, so the scores leave the matrix pipe in the log2 domain;
//   * NO maximum is subtracted.  softmax(s) = exp2(s) / sum exp2(s) whatever constant is subtracted from s; the
//     subtraction only keeps fp32 in range, and |s| < ~100 (69 nats) needs no help there.  The row sum l tells whether
//     that held: a workgroup that finds an l outside [1e-30, 1e30] (or inf / nan) repeats its item with the classical
//     running-maximum loop.  Per score that leaves exp + sum + half a convert (8.2 VALU per MFMA by PMC, prologue and
//     epilogue included): 43.1 -> 37.8 us on the encoder shape of config 2.
//   * a LEAN body: <= 168 registers per lane, i.e. three workgroups (12 waves) per CU instead of two - with waits at
//     35-40 % of every wave's time a third wave per SIMD is what fills the issue slots (37.8 -> 35.2 us).  One register
//     stage, one 32-key block live at a time; the K / V tiles are fetched with buffer loads (a descriptor per tile whose
//     range ends at the utterance's last key: rows past it read as zeros - no clamping, no address arithmetic, no
//     branches in the loop); only the last tile runs the masking path; the hot tile is 165 instructions for 16 MFMAs.
// Work decomposition as in the general kernel: 4 waves x 32 query rows per (utterance, head, 128-row tile), 64-key
// tiles through a padded LDS double buffer.  Tried on the same box and NOT kept (tools/dev/st_attn64_variants.hip,
// DESIGN.md section 5): 64 query rows per wave with one workgroup per CU (44-46 us: nothing hides a single wave's
// s_waitcnt time), with two (spills at 256 registers: 59 us), a hand-staggered instruction stream pinned with
// sched_group_barrier (44 us), 128-key stages (54 us), all fragment reads of a tile issued up front (39 us), packed adds
// for the row sums written by hand (38.4 vs 37.8 us), waves without a valid row skipping the tile body (54 us: the
// early return re-shuffled the compiler's register assignment).
#include "st_attn_common.cuh"
#include <type_traits>

namespace {

constexpr float F64_BIG = 1e30f, F64_SMALL = 1e-30f;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// one 64-key tile, plain exponentials (EXACT = false) or the classical running-maximum update (EXACT = true)
template <bool DROP, bool MASK, bool EXACT>
__device__ __forceinline__ void lean_tile(const bf16* ks, const bf16* vs, const bf16x8 (&qf)[4], f32x16 (&o)[2], float& m, float& lsum,
                                          int kt, int lk, int q, const Drop& dr, int bh) {
  constexpr int DK = 64;
  const int l = threadIdx.x & 63, hi = l >> 5;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    f32x16 s = zero16();
#pragma unroll
    for (int t = 0; t < 4; ++t) s = mfma32(rd_nat<DK>(ks, kb * 32 + (l & 31), t), qf[t], s);
    if (MASK) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt + kb * 32 + acc_row(r, hi) >= lk) s[r] = -INFINITY;
    }
    if (EXACT) {
      float mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, wave_xor32(mx));
      const float m_new = fmaxf(m, mx);
      const float m_fin = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m - m_fin);
      lsum *= alpha;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      m = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] -= m_fin;
    }
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r]);
      ps += s[r];
    }
    lsum += ps;
    if (DROP) {   // dropped probabilities leave the normaliser untouched; the 1/(1-p) scale is folded into the final 1/l
      bool keep[16];
      keep16<true>(dr, bh, q, kt + kb * 32, hi, keep);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = keep[r] ? s[r] : 0.f;
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const bf16x8 pf = pack_acc8(s, 8 * hf);
      const int base = kb * 32 + 16 * hf + 4 * hi;
#pragma unroll
      for (int d = 0; d < 2; ++d) o[d] = mfma32(rd_tr<DK>(vs, d * 32, base), pf, o[d]);
    }
    __builtin_amdgcn_sched_barrier(0);      // one block live at a time: hoisting the next block's reads costs the third workgroup
  }
}

template <bool DROP>
__global__ __launch_bounds__(256, 3) void attn_fwd64_kernel(AttnArgs a) {
  constexpr int DK = 64, NT = 4, ND = 2;
  using G = TileGeo<DK>;
  __shared__ __attribute__((aligned(16))) bf16 smem[4 * G::E];   // 2 buffers x (K tile, V tile); the row patches afterwards

  int b, h, tile;
  decode_item(a, blockIdx.x, b, h, tile);
  const int lq = a.q_len[b], lk = a.k_len[b];
  const int q0 = tile * WG_ROWS;
  if (q0 >= lq) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hi = l >> 5;
  const int q = q0 + wave * 32 + (l & 31);
  const size_t qrow = (size_t)a.q_off[b] + min(q, lq - 1);
  const float c2 = a.scale * 1.4426950408889634f;
  const Drop dr = make_drop(a.drop);
  const int bh = b * a.H + h;
  const int ntiles = (lk + TILE - 1) / TILE;
  const char* kbase = reinterpret_cast<const char*>(a.K + (size_t)a.k_off[b] * a.ldk + h * DK);
  const char* vbase = reinterpret_cast<const char*>(a.V + (size_t)a.k_off[b] * a.ldv + h * DK);

  bf16x8 qf[NT];     // log2 domain: q * scale * log2(e), rounded to bf16 once more (the scores then need no multiply)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.Q + qrow * a.ldq + h * DK + t * 16 + hi * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[t][e] = (bf16)((float)v[e] * c2);
  }
  // staging: thread -> two 16-byte chunks of the K tile and two of the V tile (chunk id = tid + p * 256: row id / 8)
  uint32_t vk[2], vv[2];
  int lds_at[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int id = threadIdx.x + p * 256, row = id >> 3, c8 = id & 7;
    vk[p] = (uint32_t)(row * a.ldk + c8 * 8) * 2u;
    vv[p] = (uint32_t)(row * a.ldv + c8 * 8) * 2u;
    lds_at[p] = row * G::STR + c8 * 8;
  }
  u32x4 rk[2], rv[2];
  auto load = [&](int it) {     // rows >= lk lie beyond the descriptor's range: they arrive as zeros
    const int left = lk - it * TILE;          // >= 1
    const __amdgpu_buffer_rsrc_t dk = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(kbase + (size_t)it * TILE * a.ldk * 2), 0, ((left - 1) * a.ldk + DK) * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t dv = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(vbase + (size_t)it * TILE * a.ldv * 2), 0, ((left - 1) * a.ldv + DK) * 2, 0x00020000);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      rk[p] = __builtin_amdgcn_raw_buffer_load_b128(dk, vk[p], 0, 0);
      rv[p] = __builtin_amdgcn_raw_buffer_load_b128(dv, vv[p], 0, 0);
    }
  };
  auto store = [&](bf16* ks) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      *reinterpret_cast<u32x4*>(ks + lds_at[p]) = rk[p];
      *reinterpret_cast<u32x4*>(ks + G::E + lds_at[p]) = rv[p];
    }
  };

  f32x16 o[ND];
  float m = 0.f, lsum, ltot;
  // (two instantiations of the whole loop, not one loop with a run-time switch: the two sides of such a switch keep the
  // accumulators in different registers and the compiler reconciles them with ~50 moves per tile)
  auto run = [&](auto exact_tag) {
    constexpr bool EXACT = decltype(exact_tag)::value;
    o[0] = zero16();
    o[1] = zero16();
    lsum = 0.f;
    load(0);
    for (int it = 0; it + 1 < ntiles; ++it) {       // every tile but the last: no key is masked
      bf16* ks = smem + (it & 1) * 2 * G::E;
      store(ks);
      load(it + 1);
      __syncthreads();
      lean_tile<DROP, false, EXACT>(ks, ks + G::E, qf, o, m, lsum, it * TILE, lk, q, dr, bh);
    }
    {
      const int it = ntiles - 1;
      bf16* ks = smem + (it & 1) * 2 * G::E;
      store(ks);
      __syncthreads();
      lean_tile<DROP, true, EXACT>(ks, ks + G::E, qf, o, m, lsum, it * TILE, lk, q, dr, bh);
    }
    __syncthreads();           // the tile buffers are free (epilogue patches, or the second attempt)
    ltot = lsum + wave_xor32(lsum);
  };
  run(std::false_type{});
  if (__syncthreads_or(q < lq && !(ltot > F64_SMALL && ltot < F64_BIG))) {      // left the plain-exponential range
    m = -INFINITY;
    run(std::true_type{});
  }
  const float inv = ltot > 0.f ? (DROP ? dr.scale : 1.f) / ltot : 0.f;
  if (q < lq && hi == 0 && a.lse) a.lse[(size_t)h * a.q_rows_total + qrow] = m + log2f(ltot);
  if (a.Ores)
    store_rows_pair<DK>(smem + wave * 32 * DK, smem + 4 * 32 * DK + wave * 32 * DK, o, inv, a.O + (size_t)a.q_off[b] * a.ldo + h * DK,
                        a.Ores + (size_t)a.q_off[b] * a.ldo + h * DK, a.ldo, q0 + wave * 32, min(32, lq - (q0 + wave * 32)));
  else
    store_rows<DK>(smem + wave * 32 * DK, o, inv, a.O + (size_t)a.q_off[b] * a.ldo + h * DK, a.ldo, q0 + wave * 32,
                   min(32, lq - (q0 + wave * 32)));
}

// ---------------------------------------------------------------------------------------------
// Backward of the same problems (64-wide heads, non-causal, delta produced with dO): the two bodies of st_attn.hip's merged
// launch in the same lean form - one register stage, buffer loads whose range ends at the sequence end (rows past it are
// zeros), one 32 x 32 block live at a time, masks only in the last streamed tile, no run-time switches inside the loops.
//   dQ body   (lane = query):  S^T = K Q^T,  dP^T = V dO^T,  dS^T = P^T (dP^T - delta),  dQ^T += K^T dS^T
//   dK/dV body (lane = key):   S = Q K^T,  dP = dO V^T,  dV^T += dO^T P,  dK^T += Q^T dS
// ---------------------------------------------------------------------------------------------
struct Stage64 {       // a [64 x 64] bf16 tile: 2 chunks of 16 bytes per thread, buffer-loaded, stored to the padded LDS image
  uint32_t voff[2];
  int lds_at[2];
  u32x4 r[2];
  __device__ __forceinline__ void init(int ld) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int id = threadIdx.x + p * 256, row = id >> 3, c8 = id & 7;
      voff[p] = (uint32_t)(row * ld + c8 * 8) * 2u;
      lds_at[p] = row * TileGeo<64>::STR + c8 * 8;
    }
  }
  // rows row0 .. row0 + 63 of the column slice at `base` (bytes); rows >= nrows lie beyond the descriptor: zeros
  __device__ __forceinline__ void load(const char* base, int ld, int row0, int nrows) {
    const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(base + (size_t)row0 * ld * 2), 0, ((nrows - row0 - 1) * ld + 64) * 2, 0x00020000);
#pragma unroll
    for (int p = 0; p < 2; ++p) r[p] = __builtin_amdgcn_raw_buffer_load_b128(d, voff[p], 0, 0);
  }
  __device__ __forceinline__ void store(bf16* tile) const {
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(tile + lds_at[p]) = r[p];
  }
};

template <bool DROP>
__device__ __forceinline__ void bwd64_dq_body(const AttnArgs& a, int bid, bf16* smem) {
  constexpr int DK = 64, NT = 4, ND = 2;
  using G = TileGeo<DK>;
  int b, h, tile;
  decode_item(a, bid, b, h, tile);
  const int lq = a.q_len[b], lk = a.k_len[b];
  const int q0 = tile * WG_ROWS;
  if (q0 >= lq) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hi = l >> 5;
  const int q = q0 + wave * 32 + (l & 31);
  const size_t qrow = (size_t)a.q_off[b] + min(q, lq - 1);
  const float c2 = a.scale * 1.4426950408889634f;
  const Drop dr = make_drop(a.drop);
  const int bh = b * a.H + h;
  const int ntiles = (lk + TILE - 1) / TILE;
  const char* kbase = reinterpret_cast<const char*>(a.K + (size_t)a.k_off[b] * a.ldk + h * DK);
  const char* vbase = reinterpret_cast<const char*>(a.V + (size_t)a.k_off[b] * a.ldv + h * DK);

  bf16x8 qf[NT], dof[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = h * DK + t * 16 + hi * 8;
    qf[t] = *reinterpret_cast<const bf16x8*>(a.Q + qrow * a.ldq + col);
    dof[t] = *reinterpret_cast<const bf16x8*>(a.dO + qrow * a.lddo + col);
  }
  const float dl = a.delta[(size_t)h * a.q_rows_total + qrow];
  const float lse = a.lse[(size_t)h * a.q_rows_total + qrow];
  Stage64 sk, sv;
  sk.init(a.ldk);
  sv.init(a.ldv);
  f32x16 dq[ND];
  dq[0] = zero16();
  dq[1] = zero16();

  auto tile_body = [&](const bf16* ks, const bf16* vs, int kt, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        s = mfma32(rd_nat<DK>(ks, kb * 32 + (l & 31), t), qf[t], s);
        dp = mfma32(rd_nat<DK>(vs, kb * 32 + (l & 31), t), dof[t], dp);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = fmaf(s[r], c2, -lse);
      if (MASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt + kb * 32 + acc_row(r, hi) >= lk) s[r] = -INFINITY;
      }
      if (DROP) {   // dS = P (M dP / (1-p) - delta)
        bool keep[16];
        keep16<true>(dr, bh, q, kt + kb * 32, hi, keep);
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = keep[r] ? dp[r] * dr.scale : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]) * (dp[r] - dl);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const bf16x8 dsf = pack_acc8(s, 8 * hf);
        const int base = kb * 32 + 16 * hf + 4 * hi;
#pragma unroll
        for (int d = 0; d < ND; ++d) dq[d] = mfma32(rd_tr<DK>(ks, d * 32, base), dsf, dq[d]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  sk.load(kbase, a.ldk, 0, lk);
  sv.load(vbase, a.ldv, 0, lk);
  for (int it = 0; it + 1 < ntiles; ++it) {
    bf16* ks = smem + (it & 1) * 2 * G::E;
    sk.store(ks);
    sv.store(ks + G::E);
    sk.load(kbase, a.ldk, (it + 1) * TILE, lk);
    sv.load(vbase, a.ldv, (it + 1) * TILE, lk);
    __syncthreads();
    tile_body(ks, ks + G::E, it * TILE, std::false_type{});
  }
  {
    const int it = ntiles - 1;
    bf16* ks = smem + (it & 1) * 2 * G::E;
    sk.store(ks);
    sv.store(ks + G::E);
    __syncthreads();
    tile_body(ks, ks + G::E, it * TILE, std::true_type{});
  }
  __syncthreads();
  store_rows<DK>(smem + wave * 32 * DK, dq, a.scale, a.dQ + (size_t)a.q_off[b] * a.lddq + h * DK, a.lddq, q0 + wave * 32,
                 min(32, lq - (q0 + wave * 32)));
}

template <bool DROP>
__device__ __forceinline__ void bwd64_dkv_body(const AttnArgs& a, int bid, bf16* smem) {
  constexpr int DK = 64, NT = 4, ND = 2;
  using G = TileGeo<DK>;
  constexpr int BUF = 2 * G::E + 256;   // Q tile, dO tile, lse[64] + delta[64] (fp32, counted in bf16 elements)
  int b, h, tile;
  decode_item(a, bid, b, h, tile);
  const int lq = a.q_len[b], lk = a.k_len[b];
  const int k0 = tile * WG_ROWS;
  if (k0 >= lk) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hi = l >> 5;
  const int key = k0 + wave * 32 + (l & 31);
  const size_t krow = (size_t)a.k_off[b] + min(key, lk - 1);
  const float c2 = a.scale * 1.4426950408889634f;
  const Drop dr = make_drop(a.drop);
  const int bh = b * a.H + h;
  const char* qbase = reinterpret_cast<const char*>(a.Q + (size_t)a.q_off[b] * a.ldq + h * DK);
  const char* dobase = reinterpret_cast<const char*>(a.dO + (size_t)a.q_off[b] * a.lddo + h * DK);
  // threads 0..63 carry the tile's lse values, 64..127 its delta values (128.. duplicate them)
  const float* statsrc = ((threadIdx.x & 64) ? a.delta : a.lse) + (size_t)h * a.q_rows_total + a.q_off[b];
  const int ntiles = (lq + TILE - 1) / TILE;

  bf16x8 kf[NT], vf[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = h * DK + t * 16 + hi * 8;
    kf[t] = *reinterpret_cast<const bf16x8*>(a.K + krow * a.ldk + col);
    vf[t] = *reinterpret_cast<const bf16x8*>(a.V + krow * a.ldv + col);
  }
  Stage64 sq, so;
  sq.init(a.ldq);
  so.init(a.lddo);
  float sst;
  f32x16 dk[ND], dv[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) { dk[d] = zero16(); dv[d] = zero16(); }

  auto tile_body = [&](const bf16* qs, int qt, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const bf16* dos = qs + G::E;
    const float* stat = reinterpret_cast<const float*>(qs + 2 * G::E);   // [0..63] lse, [64..127] delta
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        s = mfma32(rd_nat<DK>(qs, qb * 32 + (l & 31), t), kf[t], s);
        dp = mfma32(rd_nat<DK>(dos, qb * 32 + (l & 31), t), vf[t], dp);
      }
      bool keep[16];
      if (DROP) {   // dS = P (M dP / (1-p) - delta), and dV takes the dropped, rescaled P
        keep16<false>(dr, bh, key, qt + qb * 32, hi, keep);
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = keep[r] ? dp[r] * dr.scale : 0.f;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ql = qb * 32 + 8 * g + 4 * hi;
        const f32x4 ls = *reinterpret_cast<const f32x4*>(stat + ql);
        const f32x4 dl = *reinterpret_cast<const f32x4*>(stat + 64 + ql);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[4 * g + e] = fmaf(s[4 * g + e], c2, -ls[e]);
          dp[4 * g + e] -= dl[e];
        }
      }
      if (MASK) {   // queries past lq (keys past lk need none: lane = key, such a lane only fills its own, never stored, rows)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (qt + qb * 32 + acc_row(r, hi) >= lq) s[r] = -INFINITY;
      }
      f32x16 p;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = __builtin_amdgcn_exp2f(s[r]);
        s[r] = p[r] * dp[r];
        if (DROP) p[r] = keep[r] ? p[r] * dr.scale : 0.f;
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const bf16x8 pf = pack_acc8(p, 8 * hf);
        const bf16x8 dsf = pack_acc8(s, 8 * hf);
        const int base = qb * 32 + 16 * hf + 4 * hi;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
          dv[d] = mfma32(rd_tr<DK>(dos, d * 32, base), pf, dv[d]);
          dk[d] = mfma32(rd_tr<DK>(qs, d * 32, base), dsf, dk[d]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto load = [&](int it) {
    sq.load(qbase, a.ldq, it * TILE, lq);
    so.load(dobase, a.lddo, it * TILE, lq);
    sst = statsrc[min(it * TILE + (int)(threadIdx.x & 63), lq - 1)];
  };
  auto store = [&](bf16* base) {
    sq.store(base);
    so.store(base + G::E);
    reinterpret_cast<float*>(base + 2 * G::E)[threadIdx.x & 127] = sst;
  };
  load(0);
  for (int it = 0; it + 1 < ntiles; ++it) {
    bf16* qs = smem + (it & 1) * BUF;
    store(qs);
    load(it + 1);
    __syncthreads();
    tile_body(qs, it * TILE, std::false_type{});
  }
  {
    const int it = ntiles - 1;
    bf16* qs = smem + (it & 1) * BUF;
    store(qs);
    __syncthreads();
    tile_body(qs, it * TILE, std::true_type{});
  }
  __syncthreads();
  const int nrows = min(32, lk - (k0 + wave * 32));
  store_rows<DK>(smem + wave * 64 * DK, dk, a.scale, a.dK + (size_t)a.k_off[b] * a.lddk + h * DK, a.lddk, k0 + wave * 32, nrows);
  store_rows<DK>(smem + wave * 64 * DK + 32 * DK, dv, 1.f, a.dV + (size_t)a.k_off[b] * a.lddv + h * DK, a.lddv, k0 + wave * 32, nrows);
}

// workgroups [0, n_k): dK/dV items (the heavier ones: four contractions per block), then the dQ items
template <bool DROP, int WPS>
__global__ __launch_bounds__(256, WPS) void attn_bwd64_kernel(AttnArgs a, AttnArgs ak, int n_k) {
  constexpr int EK = 2 * (2 * TileGeo<64>::E + 256);
  __shared__ __attribute__((aligned(16))) bf16 smem[EK];
  if ((int)blockIdx.x < n_k) bwd64_dkv_body<DROP>(ak, blockIdx.x, smem);
  else bwd64_dq_body<DROP>(a, blockIdx.x - n_k, smem);
}
template <bool DROP, int WPS>
__global__ __launch_bounds__(256, WPS) void attn_bwd64_dq_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) bf16 smem[4 * TileGeo<64>::E];
  bwd64_dq_body<DROP>(a, blockIdx.x, smem);
}
template <bool DROP, int WPS>
__global__ __launch_bounds__(256, WPS) void attn_bwd64_dkv_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) bf16 smem[2 * (2 * TileGeo<64>::E + 256)];
  bwd64_dkv_body<DROP>(a, blockIdx.x, smem);
}

}  // namespace

extern "C" int st_attn64_fwd_launch(hipStream_t stream, const void* args_, int grid_x, int drop) {
  const AttnArgs& a = *static_cast<const AttnArgs*>(args_);
  dim3 grid(grid_x), block(256);
  if (drop) hipLaunchKernelGGL((attn_fwd64_kernel<true>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((attn_fwd64_kernel<false>), grid, block, 0, stream, a);
  return (int)hipGetLastError();
}

// which: 3 = both bodies in one launch (grid = n_q + n_k workgroups, the dK/dV items first), 1 = dQ only, 2 = dK/dV only
extern "C" int st_attn64_bwd_launch(hipStream_t stream, const void* aq_, const void* ak_, int n_q, int n_k, int drop, int which) {
  const AttnArgs& a = *static_cast<const AttnArgs*>(aq_);
  const AttnArgs& ak = *static_cast<const AttnArgs*>(ak_);
  dim3 block(256);
  static const int wps = [] { const char* e = getenv("ST_ATTN_BWD_WPS"); return e ? atoi(e) : 2; }();
#define ST_B(K, G, ...) do { if (drop) hipLaunchKernelGGL((K<true, 2>), dim3(G), block, 0, stream, __VA_ARGS__); \
                             else if (wps == 3) hipLaunchKernelGGL((K<false, 3>), dim3(G), block, 0, stream, __VA_ARGS__); \
                             else hipLaunchKernelGGL((K<false, 2>), dim3(G), block, 0, stream, __VA_ARGS__); } while (0)
  if (which == 3) ST_B(attn_bwd64_kernel, n_q + n_k, a, ak, n_k);
  else if (which == 1) { if (drop) hipLaunchKernelGGL((attn_bwd64_dq_kernel<true, 2>), dim3(n_q), block, 0, stream, a);
                         else hipLaunchKernelGGL((attn_bwd64_dq_kernel<false, 3>), dim3(n_q), block, 0, stream, a); }
  else { if (drop) hipLaunchKernelGGL((attn_bwd64_dkv_kernel<true, 2>), dim3(n_k), block, 0, stream, ak);
         else hipLaunchKernelGGL((attn_bwd64_dkv_kernel<false, 2>), dim3(n_k), block, 0, stream, ak); }
#undef ST_B
  return (int)hipGetLastError();
}

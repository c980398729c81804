// Attention for long non-causal problems with 64-wide heads, 64 query rows per wave (see the comment below).
#include "st_attn_common.cuh"

namespace {

// ---------------------------------------------------------------------------------------------
// Forward for LONG non-causal problems with 64-wide heads (the encoder's self-attention: 41 % of the round-2 step
// together with its backward).  Measured on the MI355X (tools/dev/issue_probe.hip): one wave issues one instruction per
// ~4.5-5 clocks, a 32x32x16 MFMA occupies its SIMD's matrix pipe for 32, ~5 VALU instructions hide under one MFMA of
// the same wave and every further one costs its full issue slot; two waves on a SIMD overlap each other's VALU and MFMA
// work by only 10-15 %.  At d_k = 64 a 32 x 32 score block is 8 MFMAs against 16 scores per lane, so the kernel above
// (scale + running max + exp + sum + convert: 12.6 VALU per MFMA by PMC) is bound by instruction issue at 2.5x its MFMA
// time.  This kernel is built around the instruction count instead:
//   * 64 query rows per wave (two 32-row blocks): every K / V^T fragment read from LDS feeds two MFMAs;
//   * Q is multiplied by scale * log2(e) once, so the scores leave the matrix pipe in the log2 domain;
//   * NO maximum is subtracted.  softmax(s) = exp2(s) / sum exp2(s) whatever constant is subtracted from s; the
//     subtraction only keeps fp32 in range, and |s| < ~100 (69 nats) needs no help there.  The row sum l tells whether
//     that held: a workgroup that finds any l outside [1e-30, 1e30] (or inf / nan) repeats its item with the classical
//     running-maximum loop (MODE 2).  Per score that leaves exp + sum + half a convert: 5 VALU per MFMA.
// ---------------------------------------------------------------------------------------------
constexpr float F64_BIG = 1e30f, F64_SMALL = 1e-30f;

// One 64-key tile for this wave's 64 queries.  MODE 0: plain exponentials, every key valid; 1: plain, keys >= lk masked;
// 2: exact online softmax (running maximum, rescale; masks keys >= lk).
template <int MODE, bool DROP>
__device__ __forceinline__ void fwd64_tile(const bf16* ks, const bf16* vs, const bf16x8 (&qf)[2][4], float (&m)[2],
                                           float (&lsum)[2], f32x16 (&o)[2][2], int kt, int lk, int q_first, const Drop& dr,
                                           int bh) {
  constexpr int DK = 64;
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
  const bool full = kt + TILE <= lk;     // wave-uniform
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    bf16x8 kf[4], vf[2][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) kf[t] = rd_nat<DK>(ks, kb * 32 + r, t);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) vf[k2][dt] = rd_tr<DK>(vs, dt * 32, kb * 32 + 16 * k2 + 4 * hi);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 s = zero16();
#pragma unroll
      for (int t = 0; t < 4; ++t) s = mfma32(kf[t], qf[qb][t], s);
      if (MODE >= 1 && !full) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (kt + kb * 32 + acc_row(i, hi) >= lk) s[i] = -INFINITY;
      }
      if (MODE == 2) {
        float mx = s[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
        mx = fmaxf(mx, wave_xor32(mx));
        const float m_new = fmaxf(m[qb], mx);
        if (__any(m_new != m[qb])) {
          const float m_fin = (m_new == -INFINITY) ? 0.f : m_new;
          const float alpha = __builtin_amdgcn_exp2f(m[qb] - m_fin);
          lsum[qb] *= alpha;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[qb][dt][i] *= alpha;
          m[qb] = m_new;
        }
        const float m_use = (m[qb] == -INFINITY) ? 0.f : m[qb];
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] -= m_use;
      }
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s[i] = __builtin_amdgcn_exp2f(s[i]);
        ps += s[i];
      }
      lsum[qb] += ps;
      if (DROP) {   // dropped probabilities leave the normaliser untouched; the 1/(1-p) scale is folded into the final 1/l
        bool keep[16];
        keep16<true>(dr, bh, q_first + qb * 32 + r, kt + kb * 32, hi, keep);
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = keep[i] ? s[i] : 0.f;
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const bf16x8 pf = pack_acc8(s, 8 * k2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o[qb][dt] = mfma32(vf[k2][dt], pf, o[qb][dt]);
      }
    }
  }
}

// ---- the same tile as ONE software-pipelined instruction stream --------------------------------------------------
// A wave's MFMAs and VALU work only overlap inside the wave (above), and only when they alternate in program order.  The
// tile's four 32 x 32 score blocks u0..u3 = (kb, qb) are therefore staggered by hand: while block u is exponentiated,
// summed and packed (VALU), the matrix pipe multiplies block u+1's scores and block u-1's P V product - regions R0..R5
// below, each a scheduling region whose order is pinned with sched_group_barrier (one MFMA, then its share of the VALU).
#define ST_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int SG_MFMA = 0x8, SG_VALU = 0x2 | 0x400, SG_DSR = 0x100;    // (VALU | TRANS: v_exp_f32 is a transcendental)

template <int ABL>
__device__ __forceinline__ f32x16 qk_block(const bf16x8 (&kf)[4], const bf16x8 (&qf)[4]) {
  f32x16 s = zero16();
#pragma unroll
  for (int t = 0; t < 4; ++t) s = mfma32(kf[t], qf[t], s);
  return s;
}

template <int ABL>
__device__ __forceinline__ void pv_block(f32x16 (&o)[2], const bf16x8 (&vf)[2][2], const bf16x8 (&p)[2]) {
  if (ABL == 3) return;
#pragma unroll
  for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) o[dt] = mfma32(vf[k2][dt], p[k2], o[dt]);
}

// exponentials in place, row sum, pack into the two B fragments of the P V product
template <int MODE, bool DROP, int ABL>
__device__ __forceinline__ void sm_block(f32x16& s, float& lsum, bf16x8 (&p)[2], bool full, int key0, int lk, const Drop& dr, int bh,
                                         int q) {
  const int hi = (threadIdx.x & 63) >> 5;
  if (MODE >= 1 && !full) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (key0 + acc_row(i, hi) >= lk) s[i] = -INFINITY;
  }
  float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    if (ABL != 1) {
      s[i] = __builtin_amdgcn_exp2f(s[i]);
      s[i + 1] = __builtin_amdgcn_exp2f(s[i + 1]);
    }
    if (ABL != 2) {
      ps0 += s[i];
      ps1 += s[i + 1];
    }
  }
  lsum += ps0 + ps1;
  if (DROP) {
    bool keep[16];
    keep16<true>(dr, bh, q, key0, hi, keep);
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = keep[i] ? s[i] : 0.f;
  }
  p[0] = pack_acc8(s, 0);
  p[1] = pack_acc8(s, 8);
}

template <int MODE, bool DROP, int ABL>
__device__ __forceinline__ void fwd64_tile_p(const bf16* ks, const bf16* vs, const bf16x8 (&qf)[2][4], float (&lsum)[2],
                                             f32x16 (&o)[2][2], int kt, int lk, int q_first, const Drop& dr, int bh) {
  constexpr int DK = 64;
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
  const bool full = kt + TILE <= lk;     // wave-uniform
  const int q0 = q_first + r, q1 = q0 + 32;
  bf16x8 kf[4], vf[2][2], kg[4], vg[2][2], pa[2], pb[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) kf[t] = rd_nat<DK>(ks, r, t);
  // R0
  f32x16 sa = qk_block<ABL>(kf, qf[0]);
#pragma unroll
  for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) vf[k2][dt] = rd_tr<DK>(vs, dt * 32, 16 * k2 + 4 * hi);
  __builtin_amdgcn_sched_barrier(0);
  // R1: scores of u1 | softmax numerator of u0; the second key block's fragments are requested
  f32x16 sb = qk_block<ABL>(kf, qf[1]);
#pragma unroll
  for (int t = 0; t < 4; ++t) kg[t] = rd_nat<DK>(ks, 32 + r, t);
#pragma unroll
  for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) vg[k2][dt] = rd_tr<DK>(vs, dt * 32, 32 + 16 * k2 + 4 * hi);
  sm_block<MODE, DROP, ABL>(sa, lsum[0], pa, full, kt, lk, dr, bh, q0);
  ST_SGB(SG_DSR, 12);
#pragma unroll
  for (int i = 0; i < 4; ++i) { ST_SGB(SG_MFMA, 1); ST_SGB(SG_VALU, 10); }
  __builtin_amdgcn_sched_barrier(0);
  // R2: scores of u2, P V of u0 | softmax numerator of u1
  sa = qk_block<ABL>(kg, qf[0]);
  pv_block<ABL>(o[0], vf, pa);
  sm_block<MODE, DROP, ABL>(sb, lsum[1], pb, full, kt, lk, dr, bh, q1);
#pragma unroll
  for (int i = 0; i < 8; ++i) { ST_SGB(SG_MFMA, 1); ST_SGB(SG_VALU, 5); }
  __builtin_amdgcn_sched_barrier(0);
  // R3: scores of u3, P V of u1 | softmax numerator of u2
  sb = qk_block<ABL>(kg, qf[1]);
  pv_block<ABL>(o[1], vf, pb);
  sm_block<MODE, DROP, ABL>(sa, lsum[0], pa, full, kt + 32, lk, dr, bh, q0);
#pragma unroll
  for (int i = 0; i < 8; ++i) { ST_SGB(SG_MFMA, 1); ST_SGB(SG_VALU, 5); }
  __builtin_amdgcn_sched_barrier(0);
  // R4: P V of u2 | softmax numerator of u3
  pv_block<ABL>(o[0], vg, pa);
  sm_block<MODE, DROP, ABL>(sb, lsum[1], pb, full, kt + 32, lk, dr, bh, q1);
#pragma unroll
  for (int i = 0; i < 4; ++i) { ST_SGB(SG_MFMA, 1); ST_SGB(SG_VALU, 10); }
  __builtin_amdgcn_sched_barrier(0);
  // R5: P V of u3
  pv_block<ABL>(o[1], vg, pb);
}

// ---- the tile for TWO workgroups per CU (256 registers): one 32-key block at a time, its fragments shared by the wave's
// two query blocks; only the block in flight is live (scheduling barriers keep the next block's reads from being hoisted)
template <int MODE, bool DROP, int ABL>
__device__ __forceinline__ void fwd64_tile_e(const bf16* ks, const bf16* vs, const bf16x8 (&qf)[2][4], float (&lsum)[2],
                                             f32x16 (&o)[2][2], int kt, int lk, int q_first, const Drop& dr, int bh) {
  constexpr int DK = 64;
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
  const bool full = kt + TILE <= lk;     // wave-uniform
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    bf16x8 kf[4], vf[2][2], pa[2], pb[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) kf[t] = rd_nat<DK>(ks, kb * 32 + r, t);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) vf[k2][dt] = rd_tr<DK>(vs, dt * 32, kb * 32 + 16 * k2 + 4 * hi);
    f32x16 sa = qk_block<0>(kf, qf[0]);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 sb = qk_block<0>(kf, qf[1]);
    sm_block<MODE, DROP, 0>(sa, lsum[0], pa, full, kt + kb * 32, lk, dr, bh, q_first + r);
    if (ABL == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { ST_SGB(SG_MFMA, 1); ST_SGB(SG_VALU, 10); }
    }
    __builtin_amdgcn_sched_barrier(0);
    pv_block<0>(o[0], vf, pa);
    sm_block<MODE, DROP, 0>(sb, lsum[1], pb, full, kt + kb * 32, lk, dr, bh, q_first + 32 + r);
    if (ABL == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { ST_SGB(SG_MFMA, 1); ST_SGB(SG_VALU, 10); }
    }
    __builtin_amdgcn_sched_barrier(0);
    pv_block<0>(o[1], vf, pb);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <bool DROP, int WPS, int VAR = 0, int ABL = 0>
__global__ __launch_bounds__(256, WPS) void attn_fwd64_kernel(AttnArgs a) {
  constexpr int DK = 64;
  using G = TileGeo<DK>;
  __shared__ __attribute__((aligned(16))) bf16 smem[4 * G::E];   // 2 buffers x (K tile, V tile); the epilogue's row patches after the loop

  int b, h, tile;
  decode_item(a, blockIdx.x, b, h, tile);
  const int lq = a.q_len[b], lk = a.k_len[b];
  const int q0 = tile * F64_WG;
  if (q0 >= lq) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
  const int qw0 = q0 + wave * F64_QW;
  const bool active = qw0 < lq;          // (wave-uniform) a wave past the sequence end only helps staging the tiles
  const float c2 = a.scale * 1.4426950408889634f;
  const Drop dr = make_drop(a.drop);
  const int bh = b * a.H + h;
  const int ntiles = (lk + TILE - 1) / TILE;
  const bf16* kbase = a.K + (size_t)a.k_off[b] * a.ldk + h * DK;
  const bf16* vbase = a.V + (size_t)a.k_off[b] * a.ldv + h * DK;

  // Q fragments in the log2 domain: q * scale * log2(e), rounded to bf16 once more (the scores then need no multiply)
  bf16x8 qf[2][4];
  size_t qrow[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    qrow[qb] = (size_t)a.q_off[b] + min(qw0 + qb * 32 + r, lq - 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.Q + qrow[qb] * a.ldq + h * DK + t * 16 + hi * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[qb][t][e] = (bf16)((float)v[e] * c2);
    }
  }

  uint32_t offk[G::CH], offv[G::CH];
  Stage<DK>::offsets(offk, a.ldk);
  Stage<DK>::offsets(offv, a.ldv);
  Stage<DK> sk, sv;                      // ONE register stage: tile it+1 is in flight while tile it is multiplied
  auto load = [&](int it) {
    sk.load(offk, kbase, a.ldk, it * TILE, lk);
    sv.load(offv, vbase, a.ldv, it * TILE, lk);
  };
  auto store = [&](int buf) {
    sk.store(smem + buf * 2 * G::E);
    sv.store(smem + buf * 2 * G::E + G::E);
  };

  f32x16 o[2][2];
  float m[2], lsum[2], ltot[2];
  bool exact = false;                    // second attempt: a row sum left the fp32 comfort zone
  for (;;) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      m[qb] = exact ? -INFINITY : 0.f;
      lsum[qb] = 0.f;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) o[qb][dt] = zero16();
    }
    load(0);
    for (int it = 0; it < ntiles; ++it) {
      const int buf = it & 1;
      if (ABL != 4 || it == 0) {
        store(buf);                            // tile `it`; its buffer was last read two tiles ago, i.e. before the previous barrier
        if (it + 1 < ntiles) load(it + 1);
      }
      if (ABL != 5 || it == 0) __syncthreads();
      const bf16* ks = smem + (ABL == 4 ? 0 : buf) * 2 * G::E;
      if (active) {
        if (exact) fwd64_tile<2, DROP>(ks, ks + G::E, qf, m, lsum, o, it * TILE, lk, qw0, dr, bh);
        else if (VAR == 1) {
          if (it + 1 < ntiles) fwd64_tile_p<0, DROP, ABL>(ks, ks + G::E, qf, lsum, o, it * TILE, lk, qw0, dr, bh);
          else fwd64_tile_p<1, DROP, ABL>(ks, ks + G::E, qf, lsum, o, it * TILE, lk, qw0, dr, bh);
        } else if (VAR == 2) {
          if (it + 1 < ntiles) fwd64_tile_e<0, DROP, 0>(ks, ks + G::E, qf, lsum, o, it * TILE, lk, qw0, dr, bh);
          else fwd64_tile_e<1, DROP, 0>(ks, ks + G::E, qf, lsum, o, it * TILE, lk, qw0, dr, bh);
        } else if (it + 1 < ntiles) fwd64_tile<0, DROP>(ks, ks + G::E, qf, m, lsum, o, it * TILE, lk, qw0, dr, bh);
        else fwd64_tile<1, DROP>(ks, ks + G::E, qf, m, lsum, o, it * TILE, lk, qw0, dr, bh);
      }
    }
    __syncthreads();                           // the tile buffers are free (epilogue patches, or the second attempt)
    bool bad = false;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      ltot[qb] = lsum[qb] + wave_xor32(lsum[qb]);
      bad |= !(ltot[qb] > F64_SMALL && ltot[qb] < F64_BIG);
    }
    if (exact || !__syncthreads_or(active && bad)) break;
    exact = true;
  }
  if (!active) return;

#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qfirst = qw0 + qb * 32;
    if (qfirst >= lq) break;                   // (wave-uniform)
    const float inv = ltot[qb] > 0.f ? (DROP ? dr.scale : 1.f) / ltot[qb] : 0.f;
    if (qfirst + r < lq && hi == 0 && a.lse) a.lse[(size_t)h * a.q_rows_total + qrow[qb]] = m[qb] + log2f(ltot[qb]);
    bf16* p_hi = smem + wave * 2 * 32 * DK;    // wave-private patches, reused by the second block (same wave: program order)
    if (a.Ores)
      store_rows_pair<DK>(p_hi, p_hi + 32 * DK, o[qb], inv, a.O + (size_t)a.q_off[b] * a.ldo + h * DK,
                          a.Ores + (size_t)a.q_off[b] * a.ldo + h * DK, a.ldo, qfirst, min(32, lq - qfirst));
    else
      store_rows<DK>(p_hi, o[qb], inv, a.O + (size_t)a.q_off[b] * a.ldo + h * DK, a.ldo, qfirst, min(32, lq - qfirst));
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Variant B: 32 query rows per wave, 128-row workgroups, two workgroups per CU (everything in VGPRs), 128-key stages.
// A stage is four 32-key score blocks u0..u3, staggered as above: block u is exponentiated / summed / packed while the
// matrix pipe multiplies block u+1's scores and block u-1's P V product.
// ---------------------------------------------------------------------------------------------
constexpr int FB_KEYS = 128;

template <int MODE, bool DROP, int ABL>
__device__ __forceinline__ void fwd64b_stage(const bf16* ks, const bf16* vs, const bf16x8 (&qf)[4], float& lsum, f32x16 (&o)[2], int kt,
                                             int lk, int q, const Drop& dr, int bh) {
  constexpr int DK = 64;
  using G = TileGeo<DK, FB_KEYS>;
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
  const bool full = kt + FB_KEYS <= lk;     // wave-uniform
  bf16x8 kf[4], vf[2][2], kg[4], vg[2][2], pa[2], pb[2];
  auto rdk = [&](bf16x8 (&f)[4], int u) {
#pragma unroll
    for (int t = 0; t < 4; ++t) f[t] = frag_nat(ks, G::STR, u * 32 + r, t * 16 + hi * 8);
  };
  auto rdv = [&](bf16x8 (&f)[2][2], int u) {
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) f[k2][dt] = frag_tr(vs, G::STR, dt * 32, u * 32 + 16 * k2 + 4 * hi, u * 32 + 16 * k2 + 4 * hi + 8);
  };
  // R0: scores of u0
  rdk(kf, 0);
  f32x16 sa = qk_block<ABL>(kf, qf);
  rdv(vf, 0);
  rdk(kg, 1);
  __builtin_amdgcn_sched_barrier(0);
  // R1: scores of u1 | numerator of u0
  f32x16 sb = qk_block<ABL>(kg, qf);
  rdv(vg, 1);
  rdk(kf, 2);
  sm_block<MODE, DROP, ABL>(sa, lsum, pa, full, kt, lk, dr, bh, q);
  ST_SGB(SG_DSR, 12);
#pragma unroll
  for (int i = 0; i < 4; ++i) { ST_SGB(SG_MFMA, 1); ST_SGB(SG_VALU, 10); }
  __builtin_amdgcn_sched_barrier(0);
  // R2: scores of u2, P V of u0 | numerator of u1
  sa = qk_block<ABL>(kf, qf);
  pv_block<ABL>(o, vf, pa);
  sm_block<MODE, DROP, ABL>(sb, lsum, pb, full, kt + 32, lk, dr, bh, q);
#pragma unroll
  for (int i = 0; i < 8; ++i) { ST_SGB(SG_MFMA, 1); ST_SGB(SG_VALU, 5); }
  __builtin_amdgcn_sched_barrier(0);
  rdv(vf, 2);
  rdk(kg, 3);
  // R3: scores of u3, P V of u1 | numerator of u2
  sb = qk_block<ABL>(kg, qf);
  pv_block<ABL>(o, vg, pb);
  sm_block<MODE, DROP, ABL>(sa, lsum, pa, full, kt + 64, lk, dr, bh, q);
  ST_SGB(SG_DSR, 12);
#pragma unroll
  for (int i = 0; i < 8; ++i) { ST_SGB(SG_MFMA, 1); ST_SGB(SG_VALU, 5); }
  __builtin_amdgcn_sched_barrier(0);
  rdv(vg, 3);
  // R4: P V of u2 | numerator of u3
  pv_block<ABL>(o, vf, pa);
  sm_block<MODE, DROP, ABL>(sb, lsum, pb, full, kt + 96, lk, dr, bh, q);
  ST_SGB(SG_DSR, 8);
#pragma unroll
  for (int i = 0; i < 4; ++i) { ST_SGB(SG_MFMA, 1); ST_SGB(SG_VALU, 10); }
  __builtin_amdgcn_sched_barrier(0);
  // R5: P V of u3
  pv_block<ABL>(o, vg, pb);
}

// exact stage (running maximum): the fall-back when a row sum left fp32's comfort zone
template <bool DROP>
__device__ __forceinline__ void fwd64b_stage_exact(const bf16* ks, const bf16* vs, const bf16x8 (&qf)[4], float& m, float& lsum,
                                                   f32x16 (&o)[2], int kt, int lk, int q, const Drop& dr, int bh) {
  constexpr int DK = 64;
  using G = TileGeo<DK, FB_KEYS>;
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
#pragma unroll 1
  for (int u = 0; u < 4; ++u) {
    if (kt + u * 32 >= lk) break;
    f32x16 s = zero16();
#pragma unroll
    for (int t = 0; t < 4; ++t) s = mfma32(frag_nat(ks, G::STR, u * 32 + r, t * 16 + hi * 8), qf[t], s);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (kt + u * 32 + acc_row(i, hi) >= lk) s[i] = -INFINITY;
    float mx = s[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
    mx = fmaxf(mx, wave_xor32(mx));
    const float m_new = fmaxf(m, mx);      // finite from the first block on (key 0 is visible to every query)
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    lsum *= alpha;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
    m = m_new;
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s[i] = __builtin_amdgcn_exp2f(s[i] - m);
      ps += s[i];
    }
    lsum += ps;
    if (DROP) {
      bool keep[16];
      keep16<true>(dr, bh, q, kt + u * 32, hi, keep);
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = keep[i] ? s[i] : 0.f;
    }
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const bf16x8 pf = pack_acc8(s, 8 * k2);
      const int base = u * 32 + 16 * k2 + 4 * hi;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) o[dt] = mfma32(frag_tr(vs, G::STR, dt * 32, base, base + 8), pf, o[dt]);
    }
  }
}

template <bool DROP, int ABL>
__global__ __launch_bounds__(256, 2) void attn_fwd64b_kernel(AttnArgs a) {
  constexpr int DK = 64;
  using G = TileGeo<DK, FB_KEYS>;
  __shared__ __attribute__((aligned(16))) bf16 smem[4 * G::E];   // 2 buffers x (K stage, V stage) = 73.7 KB: two workgroups per CU

  int b, h, tile;
  decode_item(a, blockIdx.x, b, h, tile);
  const int lq = a.q_len[b], lk = a.k_len[b];
  const int q0 = tile * WG_ROWS;
  if (q0 >= lq) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
  const int qw0 = q0 + wave * 32;
  const bool active = qw0 < lq;          // (wave-uniform) a wave past the sequence end only helps staging
  const int q = qw0 + r;
  const size_t qrow = (size_t)a.q_off[b] + min(q, lq - 1);
  const float c2 = a.scale * 1.4426950408889634f;
  const Drop dr = make_drop(a.drop);
  const int bh = b * a.H + h;
  const int nst = (lk + FB_KEYS - 1) / FB_KEYS;
  const bf16* kbase = a.K + (size_t)a.k_off[b] * a.ldk + h * DK;
  const bf16* vbase = a.V + (size_t)a.k_off[b] * a.ldv + h * DK;

  bf16x8 qf[4];      // log2 domain: q * scale * log2(e), rounded to bf16 once more (the scores then need no multiply)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.Q + qrow * a.ldq + h * DK + t * 16 + hi * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[t][e] = (bf16)((float)v[e] * c2);
  }

  uint32_t offk[G::CH], offv[G::CH];
  Stage<DK, FB_KEYS>::offsets(offk, a.ldk);
  Stage<DK, FB_KEYS>::offsets(offv, a.ldv);
  Stage<DK, FB_KEYS> sk, sv;             // ONE register stage: stage it+1 is in flight while stage it is multiplied
  auto load = [&](int it) {
    sk.load(offk, kbase, a.ldk, it * FB_KEYS, lk);
    sv.load(offv, vbase, a.ldv, it * FB_KEYS, lk);
  };
  auto store = [&](int buf) {
    sk.store(smem + buf * 2 * G::E);
    sv.store(smem + buf * 2 * G::E + G::E);
  };

  f32x16 o[2];
  float m = 0.f, lsum, ltot;
  bool exact = false;                    // second attempt: a row sum left the fp32 comfort zone
  for (;;) {
    lsum = 0.f;
    o[0] = zero16();
    o[1] = zero16();
    load(0);
    for (int it = 0; it < nst; ++it) {
      const int buf = it & 1;
      if (ABL != 4 || it == 0) {
        store(buf);                            // stage `it`; its buffer was last read two stages ago, i.e. before the previous barrier
        if (it + 1 < nst) load(it + 1);
      }
      if (ABL != 5 || it == 0) __syncthreads();
      const bf16* ks = smem + (ABL == 4 ? 0 : buf) * 2 * G::E;
      if (active) {
        if (exact) fwd64b_stage_exact<DROP>(ks, ks + G::E, qf, m, lsum, o, it * FB_KEYS, lk, q, dr, bh);
        else if (it + 1 < nst) fwd64b_stage<0, DROP, ABL>(ks, ks + G::E, qf, lsum, o, it * FB_KEYS, lk, q, dr, bh);
        else fwd64b_stage<1, DROP, ABL>(ks, ks + G::E, qf, lsum, o, it * FB_KEYS, lk, q, dr, bh);
      }
    }
    __syncthreads();                           // the stage buffers are free (epilogue patches, or the second attempt)
    ltot = lsum + wave_xor32(lsum);
    const bool bad = !(ltot > F64_SMALL && ltot < F64_BIG);
    if (exact || !__syncthreads_or(active && bad)) break;
    exact = true;
    m = -INFINITY;
  }
  if (!active) return;
  const float inv = ltot > 0.f ? (DROP ? dr.scale : 1.f) / ltot : 0.f;
  if (q < lq && hi == 0 && a.lse) a.lse[(size_t)h * a.q_rows_total + qrow] = m + log2f(ltot);
  bf16* p_hi = smem + wave * 2 * 32 * DK;
  if (a.Ores)
    store_rows_pair<DK>(p_hi, p_hi + 32 * DK, o, inv, a.O + (size_t)a.q_off[b] * a.ldo + h * DK,
                        a.Ores + (size_t)a.q_off[b] * a.ldo + h * DK, a.ldo, qw0, min(32, lq - qw0));
  else
    store_rows<DK>(p_hi, o, inv, a.O + (size_t)a.q_off[b] * a.ldo + h * DK, a.ldo, qw0, min(32, lq - qw0));
}

// ---------------------------------------------------------------------------------------------
// Variant C: the 128-row kernel of st_attn.hip (4 waves x 32 query rows, 64-key tiles, two register stages, two
// workgroups per CU) with nothing changed but the softmax: plain exponentials of pre-scaled scores, no maximum.
// ---------------------------------------------------------------------------------------------
template <bool DROP, int ABL>
__global__ __launch_bounds__(256, 2) void attn_fwd64c_kernel(AttnArgs a) {
  constexpr int DK = 64, NT = 4, ND = 2;
  using G = TileGeo<DK>;
  __shared__ __attribute__((aligned(16))) bf16 smem[4 * G::E];   // 2 buffers x (K tile, V tile)

  int b, h, tile;
  decode_item(a, blockIdx.x, b, h, tile);
  const int lq = a.q_len[b], lk = a.k_len[b];
  const int q0 = tile * WG_ROWS;
  if (q0 >= lq) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hi = l >> 5;
  const int q = q0 + wave * 32 + (l & 31);
  const bool q_ok = q < lq;
  const size_t qrow = (size_t)a.q_off[b] + min(q, lq - 1);
  const float c2 = a.scale * 1.4426950408889634f;
  const Drop dr = make_drop(a.drop);
  const int bh = b * a.H + h;
  const int ntiles = (lk + TILE - 1) / TILE;
  const bf16* kbase = a.K + (size_t)a.k_off[b] * a.ldk + h * DK;
  const bf16* vbase = a.V + (size_t)a.k_off[b] * a.ldv + h * DK;

  bf16x8 qf[NT];     // log2 domain
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.Q + qrow * a.ldq + h * DK + t * 16 + hi * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[t][e] = (bf16)((float)v[e] * c2);
  }
  uint32_t offk[G::CH], offv[G::CH];
  Stage<DK>::offsets(offk, a.ldk);
  Stage<DK>::offsets(offv, a.ldv);
  Stage<DK> sk[2], sv[2];
  f32x16 o[ND];
  float m = 0.f, lsum = 0.f;

  auto load = [&](int set, int it) {
    sk[set].load(offk, kbase, a.ldk, it * TILE, lk);
    sv[set].load(offv, vbase, a.ldv, it * TILE, lk);
  };
  auto store = [&](int set) {
    sk[set].store(smem + set * 2 * G::E);
    sv[set].store(smem + set * 2 * G::E + G::E);
  };
  auto fast = [&](int buf, int it) {
    const bf16* ks = smem + buf * 2 * G::E;
    const bf16* vs = ks + G::E;
    const int kt = it * TILE;
    const bool full = kt + TILE <= lk;
    if (ABL >= 10) {
      // every fragment of the tile is requested up front (one LDS latency per tile instead of one per MFMA pair), then the
      // two 32-key blocks are staggered: block 1's scores | block 0's numerator, block 0's P V | block 1's numerator
      const int r = l & 31;
      bf16x8 kf[2][4], vf[2][2][2], p0[2], p1[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < 4; ++t) kf[kb][t] = rd_nat<DK>(ks, kb * 32 + r, t);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) vf[kb][k2][dt] = rd_tr<DK>(vs, dt * 32, kb * 32 + 16 * k2 + 4 * hi);
      __builtin_amdgcn_sched_barrier(0);
      f32x16 s0 = qk_block<0>(kf[0], qf);
      __builtin_amdgcn_sched_barrier(0);
      f32x16 s1 = qk_block<0>(kf[1], qf);
      if (full) sm_block<0, DROP, 0>(s0, lsum, p0, true, kt, lk, dr, bh, q);
      else sm_block<1, DROP, 0>(s0, lsum, p0, false, kt, lk, dr, bh, q);
      if (ABL == 11) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { ST_SGB(SG_MFMA, 1); ST_SGB(SG_VALU, 10); }
      }
      __builtin_amdgcn_sched_barrier(0);
      pv_block<0>(o, vf[0], p0);
      if (full) sm_block<0, DROP, 0>(s1, lsum, p1, true, kt + 32, lk, dr, bh, q);
      else sm_block<1, DROP, 0>(s1, lsum, p1, false, kt + 32, lk, dr, bh, q);
      if (ABL == 11) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { ST_SGB(SG_MFMA, 1); ST_SGB(SG_VALU, 10); }
      }
      __builtin_amdgcn_sched_barrier(0);
      pv_block<0>(o, vf[1], p1);
      return;
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s = zero16();
#pragma unroll
      for (int t = 0; t < NT; ++t) s = mfma32(rd_nat<DK>(ks, kb * 32 + (l & 31), t), qf[t], s);
      if (!full) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt + kb * 32 + acc_row(r, hi) >= lk) s[r] = -INFINITY;
      }
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r]);
        ps += s[r];
      }
      lsum += ps;
      if (DROP) {
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
        for (int d = 0; d < ND; ++d) o[d] = mfma32(rd_tr<DK>(vs, d * 32, base), pf, o[d]);
      }
    }
  };
  auto exact = [&](int buf, int it) {
    const bf16* ks = smem + buf * 2 * G::E;
    const bf16* vs = ks + G::E;
    const int kt = it * TILE;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s = zero16();
#pragma unroll
      for (int t = 0; t < NT; ++t) s = mfma32(rd_nat<DK>(ks, kb * 32 + (l & 31), t), qf[t], s);
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt + kb * 32 + acc_row(r, hi) >= lk) s[r] = -INFINITY;
      float mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, wave_xor32(mx));
      const float m_new = fmaxf(m, mx);
      const float m_fin = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m - m_fin);
      lsum *= alpha;
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      m = m_new;
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - m_fin);
        ps += s[r];
      }
      lsum += ps;
      if (DROP) {
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
        for (int d = 0; d < ND; ++d) o[d] = mfma32(rd_tr<DK>(vs, d * 32, base), pf, o[d]);
      }
    }
  };
#pragma unroll
  for (int d = 0; d < ND; ++d) o[d] = zero16();
  stream_tiles(ntiles, load, store, fast);
  float ltot = lsum + wave_xor32(lsum);
  if (__syncthreads_or(!(ltot > F64_SMALL && ltot < F64_BIG))) {      // leave the plain-exponential range: classical loop
#pragma unroll
    for (int d = 0; d < ND; ++d) o[d] = zero16();
    m = -INFINITY;
    lsum = 0.f;
    stream_tiles(ntiles, load, store, exact);
    ltot = lsum + wave_xor32(lsum);
  }
  const float inv = ltot > 0.f ? (DROP ? dr.scale : 1.f) / ltot : 0.f;
  if (q_ok && hi == 0 && a.lse) a.lse[(size_t)h * a.q_rows_total + qrow] = m + log2f(ltot);
  if (a.Ores)
    store_rows_pair<DK>(smem + wave * 32 * DK, smem + 4 * 32 * DK + wave * 32 * DK, o, inv, a.O + (size_t)a.q_off[b] * a.ldo + h * DK,
                        a.Ores + (size_t)a.q_off[b] * a.ldo + h * DK, a.ldo, q0 + wave * 32, min(32, lq - (q0 + wave * 32)));
  else
    store_rows<DK>(smem + wave * 32 * DK, o, inv, a.O + (size_t)a.q_off[b] * a.ldo + h * DK, a.ldo, q0 + wave * 32,
                   min(32, lq - (q0 + wave * 32)));
}

extern "C" int st_attn64_fwd_launch(hipStream_t stream, const void* args_, int grid_x, int drop, int var) {
  const AttnArgs& a = *static_cast<const AttnArgs*>(args_);
  dim3 grid(grid_x), block(256);
#define ST_L(...) hipLaunchKernelGGL((attn_fwd64_kernel<__VA_ARGS__>), grid, block, 0, stream, a)
  if (drop) hipLaunchKernelGGL((attn_fwd64c_kernel<true, 0>), grid, block, 0, stream, a);
  else switch (var) {      // development variants (ST_ATTN_IMPL): 2x = pipelined tile, ablations 21..25
    case 3: ST_L(false, 1, 0); break;
#define ST_B(...) hipLaunchKernelGGL((attn_fwd64b_kernel<__VA_ARGS__>), grid, block, 0, stream, a)
    case 30: ST_B(false, 0); break;
    case 31: ST_B(false, 1); break;
    case 32: ST_B(false, 2); break;
    case 33: ST_B(false, 3); break;
    case 34: ST_B(false, 4); break;
    case 35: ST_B(false, 5); break;
#undef ST_B
    case 40: hipLaunchKernelGGL((attn_fwd64c_kernel<false, 0>), grid, block, 0, stream, a); break;
    case 41: hipLaunchKernelGGL((attn_fwd64c_kernel<false, 10>), grid, block, 0, stream, a); break;
    case 42: hipLaunchKernelGGL((attn_fwd64c_kernel<false, 11>), grid, block, 0, stream, a); break;
    case 21: ST_L(false, 1, 1, 1); break;
    case 22: ST_L(false, 1, 1, 2); break;
    case 23: ST_L(false, 1, 1, 3); break;
    case 24: ST_L(false, 1, 1, 4); break;
    case 25: ST_L(false, 1, 1, 5); break;
    case 2: ST_L(false, 1, 1); break;
    case 50: ST_L(false, 2, 2, 0); break;
    case 51: ST_L(false, 1, 2, 0); break;
    default: hipLaunchKernelGGL((attn_fwd64c_kernel<false, 0>), grid, block, 0, stream, a); break;
  }
#undef ST_L
  return (int)hipGetLastError();
}

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
// Everything else is the general kernel's: 4 waves x 32 query rows, 64-key tiles through two register stages and a
// padded LDS double buffer, two workgroups per CU.  Tried on the same box and NOT kept (tools/dev/st_attn64_variants.hip,
// DESIGN.md section 5): 64 query rows per wave with one workgroup per CU (44-46 us: nothing hides a single wave's
// s_waitcnt time), with two (spills at 256 registers: 59 us), a hand-staggered instruction stream pinned with
// sched_group_barrier (44 us), 128-key stages (54 us), all fragment reads of a tile issued up front (39 us).
#include "st_attn_common.cuh"

namespace {

constexpr float F64_BIG = 1e30f, F64_SMALL = 1e-30f;

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_fwd64_kernel(AttnArgs a) {
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

}  // namespace

extern "C" int st_attn64_fwd_launch(hipStream_t stream, const void* args_, int grid_x, int drop) {
  const AttnArgs& a = *static_cast<const AttnArgs*>(args_);
  dim3 grid(grid_x), block(256);
  if (drop) hipLaunchKernelGGL((attn_fwd64_kernel<true>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((attn_fwd64_kernel<false>), grid, block, 0, stream, a);
  return (int)hipGetLastError();
}

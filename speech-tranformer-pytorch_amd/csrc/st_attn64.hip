// Attention forward for LONG non-causal problems with 64-wide heads (the encoder's self-attention).
//
// Measured on the MI355X (tools/dev/issue_probe.hip, profiles/r03_issue_probe.txt): one wave issues one instruction per
// ~4.5-5 clocks; a 32x32x16 MFMA occupies its SIMD's matrix pipe for 32; ~5 VALU instructions hide under one MFMA of the
// same wave, every further one costs its full issue slot; and two waves of a SIMD overlap each other's MFMA and VALU work
// by only 10-30 % (a SIMD issues about one instruction per 5 clocks whichever wave it comes from, once MFMAs are in the
// mix).  At d_k = 64 a 32 x 32 score block is 8 MFMAs against 16 scores per lane, so the general kernel of st_attn.hip
// (scale + running maximum + rescale test + exp + sum + convert: 12.6 VALU per MFMA by PMC) is bound by instruction
// issue at 2.5x its MFMA time.  Here the softmax is cut to what cannot be avoided:
//   * (rounds 3-4) the scores are multiplied by scale * log2(e) in fp32, one multiply per score.  Round 3 pre-multiplied Q and
//     rounded it to bf16 once more (the scores then left the matrix pipe in the log2 domain: 1.5 us per launch cheaper) - round 4's
//     bisection of the full-size parity table (tests/test_fullsize_gpu.py, ST_ATTN_FWD64=0 against the general kernel) showed
//     what that costs: the logits do not notice (rel-L2 7.4e-3 either way), the GRADIENTS do - global rel-L2 against the fp64
//     oracle 4.28e-2 with the re-rounded Q, 3.86e-2 without (the reference arithmetic under bf16 autocast: 4.01e-2), per-tensor
//     median 5.48e-2 -> 4.84e-2 (5.24e-2).  The second rounding perturbs every score by ~2^-9 of |q||k|, i.e. the forward then
//     differentiates a slightly different function than the oracle's;
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
//   * (round 3, late) ROW SUMS ON THE MATRIX PIPE and a HAND-LAID hot tile.  The normaliser l = sum_k p(q, k) was 32 adds
//     per lane and tile (the compiler packs them: 16 v_pk_add_f32); one extra MFMA per P fragment against a fragment of
//     ones (A = 1: every accumulator row of lane q then holds the row sum of the bf16-rounded P - the very weights the
//     P V product uses) takes them off the VALU: 4 MFMAs more, 16 packed adds + the cross-half exchange less per tile.
//     The hot tile (no key masked, plain exponentials, no dropout) is written as ONE straight-line stream: the
//     exponentials + converts of a 32-key block sit between the matrix instructions of the next contraction (Q K1^T under
//     exp(S0); P0 V0 + row sums under exp(S1)), groups fenced with sched_barrier(0) so the order in the source is the order
//     in the binary (168 registers, no spill, still three workgroups per CU).  Same box, A/B: 34.8 -> 34.3 (sums) ->
//     32.9 us.  What this kernel's time is made of (same-box ablations of this very body, profiles/r03_attn_fwd_ablation.txt):
//     an empty kernel of the same grid 2.8 us; + prologue / epilogue 7.2; + the last (masked) tile 10.3; the loop with
//     ONLY its 20 MFMAs per tile (no loads, stores, barrier, LDS reads, VALU) 25.7; with the staging back 29.0; with
//     everything 33.7.  Replacing every v_exp_f32 by a v_fma_f32, or every LDS read by a broadcast read of one address,
//     changes nothing (+-0.5 us): no single unit is the limiter - 30 % of the launch is fixed cost per item, and the loop is
//     the sum of a matrix-pipe part at ~70 % efficiency, a staging part and an issue part of comparable size.
// Work decomposition as in the general kernel: 4 waves x 32 query rows per (utterance, head, 128-row tile), 64-key
// tiles through a padded LDS double buffer.  Tried on the same box and NOT kept (LABNOTES.md; the variant sources
// are in the git history): 64 query rows per wave with one workgroup per CU (44-46 us: nothing hides a single wave's
// s_waitcnt time), with two (spills at 256 registers: 59 us), a hand-staggered instruction stream pinned with
// sched_group_barrier (44 us; the sched_barrier(0) fences above are what finally held an order), 128-key stages (54 us),
// all fragment reads of a tile issued up front (39 us), packed adds
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
                                          f32x16& lacc, const bf16x8& ones, int kt, int lk, int q, const Drop& dr, int bh, float c2) {
  constexpr bool MSUM = !DROP && !EXACT;   // row sums on the matrix pipe (see the kernel's header)
  constexpr int DK = 64;
  const int l = threadIdx.x & 63, hi = l >> 5;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    f32x16 s = zero16();
#pragma unroll
    for (int t = 0; t < 4; ++t) s = mfma32(rd_nat<DK>(ks, kb * 32 + (l & 31), t), qf[t], s);
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] *= c2;      // log2 domain, in fp32 (a pre-scaled, re-rounded Q costs gradient accuracy: see the header)
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
    if (MSUM) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
    } else {
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r]);
        ps += s[r];
      }
      lsum += ps;
    }
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
      if (MSUM) lacc = mfma32(ones, pf, lacc);
    }
    __builtin_amdgcn_sched_barrier(0);      // one block live at a time: hoisting the next block's reads costs the third workgroup
  }
}


// The hot tile (no key masked, plain exponentials, no dropout) with the instruction stream laid out by hand: the
// exponentials + converts of one 32-key block sit BETWEEN the matrix instructions of the next contraction (a wave's own
// VALU work hides under its own MFMAs - up to ~5 issues per 32-clock MFMA; VALU work of another wave of the SIMD does
// not), every group fenced with sched_barrier(0) so the order below is the order in the binary.
#define SB() __builtin_amdgcn_sched_barrier(0)
// LAST: the utterance's last tile - `kleft` of its 64 keys exist.  The rows past them arrived as zeros (the descriptor's range
// ends at the last key): such a key scores 0, weighs exp2(0) = 1 and multiplies a zero V row, so the context needs no mask; the
// ROW SUM does, and gets it for free - the fragment of ones it is contracted with carries zeros at those keys.  (Subtracting
// their count from an unmasked sum was tried first: with row sums below ~1 - config 3's heads - the cancellation sent two items
// in three to the exact fall-back loop, attention forward 0.93 -> 1.54 ms.)
// KPRE: the keys arrive pre-scaled (AttnArgs::c2 == 1: K~ = scale log2e K from the projection's fp32 epilogue) - the scores
// leave the matrix pipe in the log2 domain and are exponentiated as they are (no multiply per score: -1.5 us per launch)
template <bool LAST, bool KPRE>
__device__ __forceinline__ void lean_tile_pipe(const bf16* ks, const bf16* vs, const bf16x8 (&qf)[4], f32x16 (&o)[2], f32x16& lacc,
                                               const bf16x8& ones, float c2, int kleft = 64) {
  constexpr int DK = 64;
  auto ex = [&](float s) { return __builtin_amdgcn_exp2f(KPRE ? s : s * c2); };
  const int l = threadIdx.x & 63, hi = l >> 5;
  auto ones_of = [&](int half_block) {      // keys half_block * 16 + 8 (j >> 2) + 4 hi + (j & 3): pack_acc8's contraction order
    if constexpr (!LAST) return ones;
    else {
      bf16x8 f;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (half_block * 16 + 8 * (j >> 2) + 4 * hi + (j & 3) < kleft) ? (bf16)1.f : (bf16)0.f;
      return f;
    }
  };
  bf16x8 kf[4];
  f32x16 s0 = zero16(), s1 = zero16();
#pragma unroll
  for (int t = 0; t < 4; ++t) kf[t] = rd_nat<DK>(ks, (l & 31), t);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    s0 = mfma32(kf[t], qf[t], s0);
    kf[t] = rd_nat<DK>(ks, 32 + (l & 31), t);
  }
  SB();
  // V fragments of key block 0, first half
  bf16x8 va0 = rd_tr<DK>(vs, 0, 4 * hi), va1 = rd_tr<DK>(vs, 32, 4 * hi);
  bf16x8 vb0 = rd_tr<DK>(vs, 0, 16 + 4 * hi), vb1 = rd_tr<DK>(vs, 32, 16 + 4 * hi);
  SB();
#pragma unroll
  for (int t = 0; t < 4; ++t) {      // S1 = Q K1^T under exp(S0)
    s1 = mfma32(kf[t], qf[t], s1);
#pragma unroll
    for (int r = 4 * t; r < 4 * t + 4; ++r) s0[r] = ex(s0[r]);
    SB();
  }
  bf16x8 pa = pack_acc8(s0, 0), pb = pack_acc8(s0, 8);
  SB();
  // P0 V0 (+ row sums) under exp(S1)
  o[0] = mfma32(va0, pa, o[0]);
#pragma unroll
  for (int r = 0; r < 3; ++r) s1[r] = ex(s1[r]);
  SB();
  o[1] = mfma32(va1, pa, o[1]);
#pragma unroll
  for (int r = 3; r < 6; ++r) s1[r] = ex(s1[r]);
  SB();
  lacc = mfma32(ones_of(0), pa, lacc);
  va0 = rd_tr<DK>(vs, 0, 32 + 4 * hi);
  va1 = rd_tr<DK>(vs, 32, 32 + 4 * hi);
#pragma unroll
  for (int r = 6; r < 8; ++r) s1[r] = ex(s1[r]);
  SB();
  o[0] = mfma32(vb0, pb, o[0]);
#pragma unroll
  for (int r = 8; r < 11; ++r) s1[r] = ex(s1[r]);
  SB();
  o[1] = mfma32(vb1, pb, o[1]);
#pragma unroll
  for (int r = 11; r < 14; ++r) s1[r] = ex(s1[r]);
  SB();
  lacc = mfma32(ones_of(1), pb, lacc);
  vb0 = rd_tr<DK>(vs, 0, 48 + 4 * hi);
  vb1 = rd_tr<DK>(vs, 32, 48 + 4 * hi);
#pragma unroll
  for (int r = 14; r < 16; ++r) s1[r] = ex(s1[r]);
  SB();
  pa = pack_acc8(s1, 0);
  pb = pack_acc8(s1, 8);
  SB();
  o[0] = mfma32(va0, pa, o[0]);
  o[1] = mfma32(va1, pa, o[1]);
  lacc = mfma32(ones_of(2), pa, lacc);
  o[0] = mfma32(vb0, pb, o[0]);
  o[1] = mfma32(vb1, pb, o[1]);
  lacc = mfma32(ones_of(3), pb, lacc);
  SB();
}

template <bool DROP, bool KPRE>
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
  const float c2 = a.c2;      // scale * log2 e, or 1 when the keys arrive pre-scaled (KPRE)
  const Drop dr = make_drop(a.drop);
  const int bh = b * a.H + h;
  const int ntiles = (lk + TILE - 1) / TILE;
  const char* kbase = reinterpret_cast<const char*>(a.K + (size_t)a.k_off[b] * a.ldk + h * DK);
  const char* vbase = reinterpret_cast<const char*>(a.V + (size_t)a.k_off[b] * a.ldv + h * DK);

  bf16x8 qf[NT];     // q as it is: the scores are scaled by scale * log2(e) in fp32 where they are exponentiated
#pragma unroll
  for (int t = 0; t < NT; ++t) qf[t] = *reinterpret_cast<const bf16x8*>(a.Q + qrow * a.ldq + h * DK + t * 16 + hi * 8);
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

  f32x16 o[ND], lacc;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.f;
  float m = 0.f, lsum, ltot;
  // (two instantiations of the whole loop, not one loop with a run-time switch: the two sides of such a switch keep the
  // accumulators in different registers and the compiler reconciles them with ~50 moves per tile)
  auto run = [&](auto exact_tag) {
    constexpr bool EXACT = decltype(exact_tag)::value;
    o[0] = zero16();
    o[1] = zero16();
    lacc = zero16();
    lsum = 0.f;
    load(0);
    for (int it = 0; it + 1 < ntiles; ++it) {       // every tile but the last: no key is masked
      bf16* ks = smem + (it & 1) * 2 * G::E;
      store(ks);
      load(it + 1);
      __syncthreads();
      if constexpr (!DROP && !EXACT) lean_tile_pipe<false, KPRE>(ks, ks + G::E, qf, o, lacc, ones, c2);
      else lean_tile<DROP, false, EXACT>(ks, ks + G::E, qf, o, m, lsum, lacc, ones, it * TILE, lk, q, dr, bh, c2);
    }
    {
      const int it = ntiles - 1;
      bf16* ks = smem + (it & 1) * 2 * G::E;
      store(ks);
      __syncthreads();
      // the last tile: the hot tile again, its row-sum fragments masked (see lean_tile_pipe); the other paths mask per score
      if constexpr (!DROP && !EXACT) lean_tile_pipe<true, KPRE>(ks, ks + G::E, qf, o, lacc, ones, c2, lk - it * TILE);
      else lean_tile<DROP, true, EXACT>(ks, ks + G::E, qf, o, m, lsum, lacc, ones, it * TILE, lk, q, dr, bh, c2);
    }
    __syncthreads();           // the tile buffers are free (epilogue patches, or the second attempt)
    // matrix-pipe sums: every accumulator row of lane q holds the whole row sum (both key halves: the contraction spans them)
    ltot = (!DROP && !EXACT) ? lacc[0] : lsum + wave_xor32(lsum);
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

}  // namespace

extern "C" int st_attn64_fwd_launch(hipStream_t stream, const void* args_, int grid_x, int drop, int kpre) {
  const AttnArgs& a = *static_cast<const AttnArgs*>(args_);
  dim3 grid(grid_x), block(256);
  // (the dropout variant and the exact fall-back loop multiply by a.c2 = 1 when the keys are pre-scaled: one instantiation)
  if (drop) hipLaunchKernelGGL((attn_fwd64_kernel<true, false>), grid, block, 0, stream, a);
  else if (kpre) hipLaunchKernelGGL((attn_fwd64_kernel<false, true>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((attn_fwd64_kernel<false, false>), grid, block, 0, stream, a);
  return (int)hipGetLastError();
}

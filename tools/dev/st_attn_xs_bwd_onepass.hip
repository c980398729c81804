// (dev, NOT built.)  One-pass backward of the few-queries attention (dQ, dK, dV of an (utterance, head) by one workgroup; appended
// to csrc/st_attn_xs.hip and dispatched from st_attn_bwd's merged-launch branch it builds and passes the 35 attention tests).
// Measured at config 2's decoder-encoder shape, same box: 42-46 us against 21.4 us for the merged launch of the general kernels
// (10.5 us with one key block per wave, 4.5 us without the loop: ~6 us per 32-key block and wave).  First reading - the in-loop
// dK / dV stores drag every counted wait to vmcnt(0) - was wrong: this version has a fixed-trip loop with unconditional (clamped)
// fetches and stores, its waits are counted (vmcnt(16) .. vmcnt(39) in the ISA) and it still takes 46.0 us; the "15.7 us without
// the stores" of the first version was the compiler deleting the whole dK / dV half as dead code.  With 4 waves per workgroup (one
// per SIMD, nothing to overlap with) the ~800 instructions of a block (56 MFMAs in dependent chains of 4, both score layouts, 192
// accumulator moves between AGPRs and VGPRs) simply run at ~18 clocks each.  Not pursued.
// =====================================================================================================================
// Backward of the same shape in ONE pass: dQ, dK and dV of an (utterance, head) by one workgroup.
//
// The general backward needs two bodies because dQ wants the scores with lane = query and dK / dV want them with lane = key,
// and their sums run over different workgroups.  With <= 64 queries the whole query side of an (utterance, head) fits ONE
// workgroup, so a wave that owns a 32-key block can finish that block's dK / dV itself and keep a partial dQ:
//   * 4 waves (one per SIMD, the whole register file each: three 32-key blocks in flight per wave as in the forward); wave w
//     owns key blocks w, w + 4, ...; K and V fragments straight from global memory - the SAME registers are the A operand of
//     S^T = K Q^T / dP^T = V dO^T (lane = query: feeds dQ) and the B operand of S = Q K^T / dP = dO V^T (lane = key: feeds
//     dK, dV), and likewise the Q / dO fragments read from the workgroup's LDS tiles: both layouts cost matrix instructions,
//     not loads (this launch is latency, not arithmetic);
//   * the transposing operands come from LDS: Q^T and dO^T from the shared query tiles, K^T from a per-wave patch written
//     from the K fragments (one wave's LDS operations run in order: no barrier in the loop);
//   * a block's dK / dV leave through a per-wave patch as whole 128-byte rows; the four partial dQ meet in LDS at the end.
// Replaces the merged launch of attn_bwd_kernel<64, *, 2> (128 key-split dQ workgroups + 816 one-tile dK/dV workgroups, 24 us
// at config 2) when delta = rowsum(dO * O) comes with dO (the training step's case).
// =====================================================================================================================
namespace {

constexpr int BW = 4;                 // waves per workgroup
constexpr int QS = 72;                // row stride of the Q / dO tiles (as TileGeo<64>::STR)

struct BStage {                       // one 32-key block in flight
  bf16x8 k[4];
  bf16x8 v[4];
};

template <bool DROP>
__global__ __launch_bounds__(256, 1) void attn_xs_bwd_kernel(AttnArgs a) {
  constexpr int DK = 64;
  __shared__ __attribute__((aligned(16))) bf16 q_s[64 * QS];                 // Q rows of the utterance (later: the dQ rows)
  __shared__ __attribute__((aligned(16))) bf16 do_s[64 * QS];                // dO rows (rows past the last query: zeros)
  __shared__ __attribute__((aligned(16))) float stat[128];                   // lse[64], delta[64]
  __shared__ __attribute__((aligned(16))) bf16 kpatch[BW * 32 * XVS];        // per wave: its K block, for the transposing read
  __shared__ __attribute__((aligned(16))) bf16 opatch[BW * 2 * 32 * DK];     // per wave: dK rows, dV rows on their way out
  __shared__ __attribute__((aligned(16))) float xq[BW * 64 * 64];            // the partial dQ of the four waves

  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int lq = a.q_len[b], lk = a.k_len[b];
  if (lq <= 0 || lk <= 0) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hi = l >> 5, r32 = l & 31;
  const float c2 = a.scale * 1.4426950408889634f;
  const Drop dr = make_drop(a.drop);
  const int bh = b * a.H + h;
  const size_t qrow0 = (size_t)a.q_off[b], krow0 = (size_t)a.k_off[b];
  const bf16* kbase = a.K + krow0 * a.ldk + h * DK;
  const bf16* vbase = a.V + krow0 * a.ldv + h * DK;

  // ---- the query side -> LDS: Q (rows past the end clamped: finite), dO and delta (rows past the end: ZERO, so that they
  //      contribute nothing to dK / dV; their dQ rows are never stored)
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int id = threadIdx.x + p * 256, row = id >> 3, c8 = id & 7;
    const bf16x8 qv = *reinterpret_cast<const bf16x8*>(a.Q + (qrow0 + min(row, lq - 1)) * a.ldq + h * DK + c8 * 8);
    const bf16x8 dv = row < lq ? *reinterpret_cast<const bf16x8*>(a.dO + (qrow0 + row) * a.lddo + h * DK + c8 * 8) : zero_bf8();
    *reinterpret_cast<bf16x8*>(q_s + row * QS + c8 * 8) = qv;
    *reinterpret_cast<bf16x8*>(do_s + row * QS + c8 * 8) = dv;
  }
  if (threadIdx.x < 128) {
    const int row = threadIdx.x & 63;
    const float* src = (threadIdx.x & 64) ? a.delta : a.lse;
    const float v = src[(size_t)h * a.q_rows_total + qrow0 + min(row, lq - 1)];
    stat[threadIdx.x] = ((threadIdx.x & 64) && row >= lq) ? 0.f : v;
  }

  const int nblk = (lk + 31) >> 5;
  auto fetch = [&](BStage& st, int blk) {       // blocks past the last one and rows past the last key are clamped (finite data)
    const size_t krow = (size_t)min(min(blk, nblk - 1) * 32 + r32, lk - 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      st.k[t] = *reinterpret_cast<const bf16x8*>(kbase + krow * a.ldk + t * 16 + hi * 8);
      st.v[t] = *reinterpret_cast<const bf16x8*>(vbase + krow * a.ldv + t * 16 + hi * 8);
    }
  };
  // A FIXED-TRIP loop with unconditional (clamped) fetches and stores: on gfx9 a store is a vmcnt event like a load, and only
  // in straight-line code does the compiler keep the wait for a block's fragments COUNTED - with conditional fetches / stores
  // and early exits it fell back to vmcnt(0), i.e. every block waited for the write acknowledgement of the one before (41.9 us
  // instead of ~16).  Slots past a wave's last block repeat the utterance's last block: their dK / dV stores rewrite the same
  // values, their dQ contribution is zeroed.
  BStage st0, st1, st2;
  int blk = wave;
  fetch(st0, blk);
  fetch(st1, blk + BW);
  fetch(st2, blk + 2 * BW);
  const int n_it = ((nblk + BW - 1) / BW + 2) / 3;
  __syncthreads();                      // the query tiles are in place

  float lse_q[2], dl_q[2];              // this lane's query in each query block (lane = query layout)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    lse_q[qb] = stat[qb * 32 + r32];
    dl_q[qb] = stat[64 + qb * 32 + r32];
  }
  f32x16 dq[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    dq[qb][0] = zero16();
    dq[qb][1] = zero16();
  }
  bf16* kp = kpatch + wave * 32 * XVS;
  bf16* op = opatch + wave * 2 * 32 * DK;
  const int nqb = lq > 32 ? 2 : 1;

  auto block = [&](BStage& st, int slot) {
    const bool valid = slot < nblk;
    const int kblk = min(slot, nblk - 1);
    const int k0 = kblk * 32;
    const int key = k0 + r32;
    // this wave's K block -> its patch (row = key, as the fragments hold it), for the transposing read of dQ's product
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<bf16x8*>(kp + r32 * XVS + t * 16 + hi * 8) = st.k[t];
    asm volatile("" ::: "memory");
    f32x16 dk[2], dv[2];
    dk[0] = zero16(); dk[1] = zero16(); dv[0] = zero16(); dv[1] = zero16();
    const bool partial = k0 + 32 > lk;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {        // (unrolled: a run-time index would send dq[][] to scratch)
      if (qb >= nqb) break;
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 qf[4], dof[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        qf[t] = frag_nat(q_s, QS, qb * 32 + r32, t * 16 + hi * 8);
        dof[t] = frag_nat(do_s, QS, qb * 32 + r32, t * 16 + hi * 8);
      }
      // ---- lane = query: S^T, dP^T -> dS^T -> dQ^T += K^T dS^T
      {
        f32x16 s = zero16(), dp = zero16();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s = mfma32(st.k[t], qf[t], s);
          dp = mfma32(st.v[t], dof[t], dp);
        }
        const float nl = -lse_q[qb], dl = dl_q[qb];
        if (DROP) {   // dS = P (M dP / (1-p) - delta)
          bool keep[16];
          keep16<true>(dr, bh, qb * 32 + r32, k0, hi, keep);
#pragma unroll
          for (int r = 0; r < 16; ++r) dp[r] = keep[r] ? dp[r] * dr.scale : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, nl)) * (dp[r] - dl);
        if (partial || !valid) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (!valid || k0 + acc_row(r, hi) >= lk) s[r] = 0.f;
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const bf16x8 dsf = pack_acc8(s, 8 * hf);
#pragma unroll
          for (int d = 0; d < 2; ++d)
            dq[qb][d] = mfma32(frag_tr(kp, XVS, d * 32, 16 * hf + 4 * hi, 16 * hf + 4 * hi + 8), dsf, dq[qb][d]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- lane = key: S, dP -> P, dS -> dV^T += dO^T P, dK^T += Q^T dS  (the same fragments, operands swapped)
      {
        f32x16 s = zero16(), dp = zero16();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s = mfma32(qf[t], st.k[t], s);
          dp = mfma32(dof[t], st.v[t], dp);
        }
        bool keep[16];
        if (DROP) {
          keep16<false>(dr, bh, key, qb * 32, hi, keep);
#pragma unroll
          for (int r = 0; r < 16; ++r) dp[r] = keep[r] ? dp[r] * dr.scale : 0.f;
        }
        f32x16 p;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 ls = *reinterpret_cast<const f32x4*>(stat + qb * 32 + 8 * g + 4 * hi);
          const f32x4 dl = *reinterpret_cast<const f32x4*>(stat + 64 + qb * 32 + 8 * g + 4 * hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            p[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -ls[e]));
            s[r] = p[r] * (dp[r] - dl[e]);
            if (DROP) p[r] = keep[r] ? p[r] * dr.scale : 0.f;
          }
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const bf16x8 pf = pack_acc8(p, 8 * hf), dsf = pack_acc8(s, 8 * hf);
          const int base = qb * 32 + 16 * hf + 4 * hi;
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            dv[d] = mfma32(frag_tr(do_s, QS, d * 32, base, base + 8), pf, dv[d]);
            dk[d] = mfma32(frag_tr(q_s, QS, d * 32, base, base + 8), dsf, dk[d]);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- the block's dK / dV rows (lane = key) -> the wave's patch -> whole 128-byte row segments
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 vk, vv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vk[e] = (bf16)(dk[d][4 * g + e] * a.scale);
          vv[e] = (bf16)dv[d][4 * g + e];
        }
        const int col = d * 32 + 8 * g + 4 * hi;
        *reinterpret_cast<bf16x4*>(op + r32 * DK + col) = vk;
        *reinterpret_cast<bf16x4*>(op + 32 * DK + r32 * DK + col) = vv;
      }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int id = l + p * 64, row = id >> 3, c8 = id & 7;
      const bf16x8 rk = *reinterpret_cast<const bf16x8*>(op + row * DK + c8 * 8);
      const bf16x8 rv = *reinterpret_cast<const bf16x8*>(op + 32 * DK + row * DK + c8 * 8);
      // rows past the last key hold that key's own dK / dV (their K / V rows were clamped onto it): written onto its row
      const size_t orow = krow0 + min(k0 + row, lk - 1);
      *reinterpret_cast<bf16x8*>(a.dK + orow * a.lddk + h * DK + c8 * 8) = rk;
      *reinterpret_cast<bf16x8*>(a.dV + orow * a.lddv + h * DK + c8 * 8) = rv;
    }
    asm volatile("" ::: "memory");
    fetch(st, slot + 3 * BW);
    __builtin_amdgcn_sched_barrier(0);
  };

  for (int it = 0; it < n_it; ++it) {
    block(st0, blk);
    block(st1, blk + BW);
    block(st2, blk + 2 * BW);
    blk += 3 * BW;
  }

  // ---- the four partial dQ -> LDS; wave w adds up registers of (query block w >> 1, column block w & 1) ------------------
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) xq[(size_t)(wave * 64 + qb * 32 + d * 16 + r) * 64 + l] = dq[qb][d][r];
  __syncthreads();              // (also: every wave is done with the Q tile, which now receives the dQ rows)
  {
    const int qb = wave >> 1, d = wave & 1;
    float sum[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sum[r] = 0.f;
#pragma unroll
      for (int u = 0; u < BW; ++u) sum[r] += xq[(size_t)(u * 64 + qb * 32 + d * 16 + r) * 64 + l];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (bf16)(sum[4 * g + e] * a.scale);
      *reinterpret_cast<bf16x4*>(q_s + (qb * 32 + r32) * DK + d * 32 + 8 * g + 4 * hi) = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int id = threadIdx.x + p * 256, row = id >> 3, c8 = id & 7;
    if (row < lq)
      *reinterpret_cast<bf16x8*>(a.dQ + (qrow0 + row) * a.lddq + h * DK + c8 * 8) = *reinterpret_cast<const bf16x8*>(q_s + row * DK + c8 * 8);
  }
}

}  // namespace

extern "C" int st_attn_xs_bwd_launch(hipStream_t stream, const void* args_, int B, int drop) {
  const AttnArgs& a = *static_cast<const AttnArgs*>(args_);
  dim3 grid(B * a.H), block(256);
  if (drop) hipLaunchKernelGGL((attn_xs_bwd_kernel<true>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((attn_xs_bwd_kernel<false>), grid, block, 0, stream, a);
  return (int)hipGetLastError();
}

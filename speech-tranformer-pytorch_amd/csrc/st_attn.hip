// Fused masked scaled-dot-product attention, forward and backward (gfx950).
//
// Replaces the score / mask / softmax / context chain of MultiHeadAttention.forward
// (reference transformer/Attention.py:82-90): S = QK^T / sqrt(d_k), key-padding and
// causal masking, softmax over keys, context = P V - without ever materialising the
// [B, h, Lq, Lk] tensors.  The dense masks of transformer/Utils.py:41-70 are replaced
// by per-utterance lengths: key j of utterance b is masked iff j >= k_len[b]
// (padding_info_mask) or, when `causal`, j > i (feature_info_mask).
//
// Layout: activations are row matrices [rows, ld] (bf16); utterance b owns rows
// off[b] .. off[b] + len[b] - 1 (packed or padded, the kernel does not care); head h
// is the column slice [h*DK, (h+1)*DK).  Q, K, V may live in one fused [rows, 3d]
// projection buffer - each has its own base pointer and leading dimension.
//
// All three kernels keep the "row statistics" index on the LANE: scores are computed
// transposed (S^T = K Q^T, lane = query) in the forward and dQ kernels and as
// S = Q K^T (lane = key) in the dK/dV kernel, so softmax max/sum, LSE and delta are
// lane-local and the second MFMA of every pair consumes the first one's accumulator
// registers directly (pack_acc8) - no P / dS round trip through LDS.
#include "st_common.cuh"

namespace {

constexpr int KV_TILE = 64;   // keys per LDS tile (forward / dQ kernels)
constexpr int Q_TILE = 64;    // queries per LDS tile (dK/dV kernel)
constexpr int WG_ROWS = 128;  // rows owned by a workgroup (4 waves x 32)

struct AttnArgs {
  const bf16* Q; int ldq;
  const bf16* K; int ldk;
  const bf16* V; int ldv;
  bf16* O; int ldo;               // forward: output; backward: forward output (for delta)
  const bf16* dO; int lddo;
  bf16* dQ; int lddq;
  bf16* dK; int lddk;
  bf16* dV; int lddv;
  float* lse;                     // [H][q_rows_total], log2 domain: m + log2(l)
  float* delta;                   // [H][q_rows_total]
  const int* q_off; const int* q_len;
  const int* k_off; const int* k_len;
  int q_rows_total;
  int causal;
  float scale;                    // 1/sqrt(d_k)
};

template <int DK> __host__ __device__ constexpr int nat_stride() { return DK + 8; }
template <int DK> __host__ __device__ constexpr int tr_stride() { return DK == 64 ? 96 : DK; }

// Stage ROWS x DK bf16 rows (global row r0+row, predicated on row < nvalid) into LDS.
template <int DK, int ROWS>
__device__ __forceinline__ void stage_rows(bf16* tile, int stride, const bf16* base, int ld, int r0, int nvalid) {
  constexpr int CPR = DK / 8;  // 16-byte chunks per row
  constexpr int CHUNKS = ROWS * CPR;
#pragma unroll
  for (int p = 0; p < (CHUNKS + 255) / 256; ++p) {
    const int id = threadIdx.x + p * 256;
    if (CHUNKS % 256 != 0 && id >= CHUNKS) break;
    const int r = id / CPR, ch = id % CPR;
    const bf16x8 v = gload8(base + (size_t)(r0 + r) * ld + ch * 8, (r0 + r) < nvalid);
    *reinterpret_cast<bf16x8*>(tile + r * stride + ch * 8) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Forward.  grid = (ceil(max_q / 128), H, B).  Each wave owns 32 query rows (lane & 31).
// ---------------------------------------------------------------------------------------------
template <int DK>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  constexpr int NT = DK / 16;   // k-steps of the QK^T contraction
  constexpr int ND = DK / 32;   // 32-wide output column tiles
  constexpr int KS = nat_stride<DK>(), VS = tr_stride<DK>();
  __shared__ __attribute__((aligned(16))) bf16 smem[KV_TILE * KS + KV_TILE * VS];
  bf16* ks = smem;
  bf16* vs = smem + KV_TILE * KS;

  const int b = blockIdx.z, h = blockIdx.y;
  const int lq = a.q_len[b], lk = a.k_len[b];
  const int q0 = blockIdx.x * WG_ROWS;
  if (q0 >= lq) return;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, hi = l >> 5;
  const int q = q0 + wave * 32 + (l & 31);
  const bool q_ok = q < lq;
  const size_t qrow = (size_t)a.q_off[b] + q;
  const float c2 = a.scale * 1.4426950408889634f;  // scores -> log2 domain

  bf16x8 qf[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) qf[t] = gload8(a.Q + qrow * a.ldq + h * DK + t * 16 + hi * 8, q_ok);

  f32x16 o[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) o[d] = zero16();
  float m = -INFINITY, lsum = 0.f;

  const int k_hi = a.causal ? min(lk, q0 + WG_ROWS) : lk;   // keys this workgroup can see
  const bf16* kbase = a.K + (size_t)a.k_off[b] * a.ldk + h * DK;
  const bf16* vbase = a.V + (size_t)a.k_off[b] * a.ldv + h * DK;

  for (int kt = 0; kt < k_hi; kt += KV_TILE) {
    __syncthreads();
    stage_rows<DK, KV_TILE>(ks, KS, kbase, a.ldk, kt, lk);
    stage_rows<DK, KV_TILE>(vs, VS, vbase, a.ldv, kt, lk);
    __syncthreads();

    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      s[kb] = zero16();
#pragma unroll
      for (int t = 0; t < NT; ++t)
        s[kb] = mfma32(frag_nat(ks, KS, kb * 32 + (l & 31), t * 16 + hi * 8), qf[t], s[kb]);
    }
    // mask + running max (log2 domain)
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt + kb * 32 + acc_row(r, hi);
        const bool dead = key >= lk || (a.causal && key > q);
        const float v = dead ? -INFINITY : s[kb][r] * c2;
        s[kb][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, wave_xor32(mx));
    const float m_new = fmaxf(m, mx);
    // m_new is finite from the first tile on (key 0 is visible to every query); guard anyway
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = exp2f(m - m_use);
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = exp2f(s[kb][r] - m_use);
        s[kb][r] = p;
        psum += p;
      }
    lsum = lsum * alpha + psum;
    m = m_new;
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    // O^T += V^T P^T : A operand = V^T (transposing LDS read), B operand = P^T (own registers)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const bf16x8 pf = pack_acc8(s[kb], 8 * hf);
        const int base = kb * 32 + 16 * hf + 4 * hi;
#pragma unroll
        for (int d = 0; d < ND; ++d) o[d] = mfma32(frag_tr(vs, VS, d * 32, base, base + 8), pf, o[d]);
      }
  }

  const float ltot = lsum + wave_xor32(lsum);
  const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
  if (!q_ok) return;
  if (hi == 0 && a.lse) a.lse[(size_t)h * a.q_rows_total + qrow] = m + log2f(ltot);
  bf16* orow = a.O + qrow * a.ldo + h * DK;
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (bf16)(o[d][4 * g + e] * inv);
      *reinterpret_cast<bf16x4*>(orow + d * 32 + 8 * g + 4 * hi) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Backward, part 1: dQ (and delta = rowsum(dO * O)).  Same decomposition as the forward.
//   P^T = exp2(S^T c2 - lse),  dP^T = V dO^T,  dS^T = P^T (dP^T - delta),  dQ^T += K^T dS^T
// ---------------------------------------------------------------------------------------------
template <int DK>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs a) {
  constexpr int NT = DK / 16, ND = DK / 32;
  constexpr int KS = nat_stride<DK>();
  __shared__ __attribute__((aligned(16))) bf16 smem[2 * KV_TILE * KS];
  bf16* ks = smem;
  bf16* vs = smem + KV_TILE * KS;

  const int b = blockIdx.z, h = blockIdx.y;
  const int lq = a.q_len[b], lk = a.k_len[b];
  const int q0 = blockIdx.x * WG_ROWS;
  if (q0 >= lq) return;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, hi = l >> 5;
  const int q = q0 + wave * 32 + (l & 31);
  const bool q_ok = q < lq;
  const size_t qrow = (size_t)a.q_off[b] + q;
  const float c2 = a.scale * 1.4426950408889634f;

  bf16x8 qf[NT], dof[NT];
  float dl = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = h * DK + t * 16 + hi * 8;
    qf[t] = gload8(a.Q + qrow * a.ldq + col, q_ok);
    dof[t] = gload8(a.dO + qrow * a.lddo + col, q_ok);
    const bf16x8 of = gload8(a.O + qrow * a.ldo + col, q_ok);
#pragma unroll
    for (int e = 0; e < 8; ++e) dl += (float)dof[t][e] * (float)of[e];
  }
  dl += wave_xor32(dl);
  const float lse = q_ok ? a.lse[(size_t)h * a.q_rows_total + qrow] : INFINITY;
  if (q_ok && hi == 0) a.delta[(size_t)h * a.q_rows_total + qrow] = dl;

  f32x16 dq[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) dq[d] = zero16();

  const int k_hi = a.causal ? min(lk, q0 + WG_ROWS) : lk;
  const bf16* kbase = a.K + (size_t)a.k_off[b] * a.ldk + h * DK;
  const bf16* vbase = a.V + (size_t)a.k_off[b] * a.ldv + h * DK;

  for (int kt = 0; kt < k_hi; kt += KV_TILE) {
    __syncthreads();
    stage_rows<DK, KV_TILE>(ks, KS, kbase, a.ldk, kt, lk);
    stage_rows<DK, KV_TILE>(vs, KS, vbase, a.ldv, kt, lk);
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        s = mfma32(frag_nat(ks, KS, kb * 32 + (l & 31), t * 16 + hi * 8), qf[t], s);
        dp = mfma32(frag_nat(vs, KS, kb * 32 + (l & 31), t * 16 + hi * 8), dof[t], dp);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt + kb * 32 + acc_row(r, hi);
        const bool dead = key >= lk || (a.causal && key > q);
        const float p = dead ? 0.f : exp2f(s[r] * c2 - lse);
        s[r] = p * (dp[r] - dl);
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const bf16x8 dsf = pack_acc8(s, 8 * hf);
        const int base = kb * 32 + 16 * hf + 4 * hi;
#pragma unroll
        for (int d = 0; d < ND; ++d) dq[d] = mfma32(frag_tr(ks, KS, d * 32, base, base + 8), dsf, dq[d]);
      }
    }
  }
  if (!q_ok) return;
  bf16* drow = a.dQ + qrow * a.lddq + h * DK;
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (bf16)(dq[d][4 * g + e] * a.scale);
      *reinterpret_cast<bf16x4*>(drow + d * 32 + 8 * g + 4 * hi) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Backward, part 2: dK, dV.  grid = (ceil(max_k / 128), H, B); each wave owns 32 keys (lane & 31)
// and loops over 64-query tiles staged in LDS.
//   S = Q K^T (lane = key, registers = queries),  P = exp2(S c2 - lse[q])
//   dV^T += dO^T P,   dP = dO V^T,   dS = P (dP - delta[q]),   dK^T += Q^T dS
// ---------------------------------------------------------------------------------------------
template <int DK>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnArgs a) {
  constexpr int NT = DK / 16, ND = DK / 32;
  constexpr int QS = nat_stride<DK>();
  __shared__ __attribute__((aligned(16))) bf16 smem[2 * Q_TILE * QS];
  __shared__ __attribute__((aligned(16))) float stat[2][Q_TILE];
  bf16* qs = smem;
  bf16* dos = smem + Q_TILE * QS;

  const int b = blockIdx.z, h = blockIdx.y;
  const int lq = a.q_len[b], lk = a.k_len[b];
  const int k0 = blockIdx.x * WG_ROWS;
  if (k0 >= lk) return;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, hi = l >> 5;
  const int key = k0 + wave * 32 + (l & 31);
  const bool k_ok = key < lk;
  const size_t krow = (size_t)a.k_off[b] + key;
  const float c2 = a.scale * 1.4426950408889634f;

  bf16x8 kf[NT], vf[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = h * DK + t * 16 + hi * 8;
    kf[t] = gload8(a.K + krow * a.ldk + col, k_ok);
    vf[t] = gload8(a.V + krow * a.ldv + col, k_ok);
  }
  f32x16 dk[ND], dv[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) { dk[d] = zero16(); dv[d] = zero16(); }

  const bf16* qbase = a.Q + (size_t)a.q_off[b] * a.ldq + h * DK;
  const bf16* dobase = a.dO + (size_t)a.q_off[b] * a.lddo + h * DK;
  const float* lse = a.lse + (size_t)h * a.q_rows_total + a.q_off[b];
  const float* delta = a.delta + (size_t)h * a.q_rows_total + a.q_off[b];
  const int q_begin = a.causal ? (k0 / Q_TILE) * Q_TILE : 0;  // queries before the first key see none of them

  for (int qt = q_begin; qt < lq; qt += Q_TILE) {
    __syncthreads();
    stage_rows<DK, Q_TILE>(qs, QS, qbase, a.ldq, qt, lq);
    stage_rows<DK, Q_TILE>(dos, QS, dobase, a.lddo, qt, lq);
    if (threadIdx.x < Q_TILE) {
      const int qq = qt + threadIdx.x;
      stat[0][threadIdx.x] = qq < lq ? lse[qq] : INFINITY;   // +inf -> P = 0 for rows past the end
      stat[1][threadIdx.x] = qq < lq ? delta[qq] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        s = mfma32(frag_nat(qs, QS, qb * 32 + (l & 31), t * 16 + hi * 8), kf[t], s);
        dp = mfma32(frag_nat(dos, QS, qb * 32 + (l & 31), t * 16 + hi * 8), vf[t], dp);
      }
      f32x16 p;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 ls = *reinterpret_cast<const f32x4*>(&stat[0][qb * 32 + 8 * g + 4 * hi]);
        const f32x4 dl = *reinterpret_cast<const f32x4*>(&stat[1][qb * 32 + 8 * g + 4 * hi]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const int qq = qt + qb * 32 + acc_row(r, hi);
          const bool dead = !k_ok || (a.causal && key > qq);
          const float pv = dead ? 0.f : exp2f(s[r] * c2 - ls[e]);
          p[r] = pv;
          s[r] = pv * (dp[r] - dl[e]);
        }
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const bf16x8 pf = pack_acc8(p, 8 * hf);
        const bf16x8 dsf = pack_acc8(s, 8 * hf);
        const int base = qb * 32 + 16 * hf + 4 * hi;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
          dv[d] = mfma32(frag_tr(dos, QS, d * 32, base, base + 8), pf, dv[d]);
          dk[d] = mfma32(frag_tr(qs, QS, d * 32, base, base + 8), dsf, dk[d]);
        }
      }
    }
  }
  if (!k_ok) return;
  bf16* dkrow = a.dK + krow * a.lddk + h * DK;
  bf16* dvrow = a.dV + krow * a.lddv + h * DK;
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 vk, vv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        vk[e] = (bf16)(dk[d][4 * g + e] * a.scale);
        vv[e] = (bf16)dv[d][4 * g + e];
      }
      *reinterpret_cast<bf16x4*>(dkrow + d * 32 + 8 * g + 4 * hi) = vk;
      *reinterpret_cast<bf16x4*>(dvrow + d * 32 + 8 * g + 4 * hi) = vv;
    }
}

int check_common(int d_k, int ldq, int ldk, int ldv) {
  if (d_k != 32 && d_k != 64) return -1;
  if ((ldq & 7) || (ldk & 7) || (ldv & 7)) return -2;
  return 0;
}

}  // namespace

extern "C" int st_attn_fwd(hipStream_t stream, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                           void* O, int ldo, float* lse, const int* q_off, const int* q_len, const int* k_off,
                           const int* k_len, int B, int H, int d_k, int max_q, int q_rows_total, int causal,
                           float scale) {
  if (B <= 0 || H <= 0 || max_q <= 0) return 0;
  int rc = check_common(d_k, ldq, ldk, ldv);
  if (rc) return rc;
  if (ldo & 3) return -3;
  AttnArgs a = {};
  a.Q = (const bf16*)Q; a.ldq = ldq; a.K = (const bf16*)K; a.ldk = ldk; a.V = (const bf16*)V; a.ldv = ldv;
  a.O = (bf16*)O; a.ldo = ldo; a.lse = lse; a.q_off = q_off; a.q_len = q_len; a.k_off = k_off; a.k_len = k_len;
  a.q_rows_total = q_rows_total; a.causal = causal; a.scale = scale;
  dim3 grid((max_q + WG_ROWS - 1) / WG_ROWS, H, B), block(256);
  if (d_k == 64) hipLaunchKernelGGL((attn_fwd_kernel<64>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((attn_fwd_kernel<32>), grid, block, 0, stream, a);
  ST_CHECK_LAUNCH();
  return 0;
}

extern "C" int st_attn_bwd(hipStream_t stream, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                           const void* O, int ldo, const void* dO, int lddo, const float* lse, float* delta,
                           void* dQ, int lddq, void* dK, int lddk, void* dV, int lddv, const int* q_off,
                           const int* q_len, const int* k_off, const int* k_len, int B, int H, int d_k, int max_q,
                           int max_k, int q_rows_total, int causal, float scale, int parts) {
  if (B <= 0 || H <= 0 || max_q <= 0 || max_k <= 0) return 0;
  int rc = check_common(d_k, ldq, ldk, ldv);
  if (rc) return rc;
  if ((ldo & 7) || (lddo & 7) || (lddq & 3) || (lddk & 3) || (lddv & 3)) return -3;
  AttnArgs a = {};
  a.Q = (const bf16*)Q; a.ldq = ldq; a.K = (const bf16*)K; a.ldk = ldk; a.V = (const bf16*)V; a.ldv = ldv;
  a.O = (bf16*)O; a.ldo = ldo; a.dO = (const bf16*)dO; a.lddo = lddo; a.lse = (float*)lse; a.delta = delta;
  a.dQ = (bf16*)dQ; a.lddq = lddq; a.dK = (bf16*)dK; a.lddk = lddk; a.dV = (bf16*)dV; a.lddv = lddv;
  a.q_off = q_off; a.q_len = q_len; a.k_off = k_off; a.k_len = k_len;
  a.q_rows_total = q_rows_total; a.causal = causal; a.scale = scale;
  dim3 block(256);
  dim3 gq((max_q + WG_ROWS - 1) / WG_ROWS, H, B), gk((max_k + WG_ROWS - 1) / WG_ROWS, H, B);
  if (d_k == 64) {
    if (parts & 1) hipLaunchKernelGGL((attn_bwd_dq_kernel<64>), gq, block, 0, stream, a);
    if (parts & 2) hipLaunchKernelGGL((attn_bwd_dkv_kernel<64>), gk, block, 0, stream, a);
  } else {
    if (parts & 1) hipLaunchKernelGGL((attn_bwd_dq_kernel<32>), gq, block, 0, stream, a);
    if (parts & 2) hipLaunchKernelGGL((attn_bwd_dkv_kernel<32>), gk, block, 0, stream, a);
  }
  ST_CHECK_LAUNCH();
  return 0;
}

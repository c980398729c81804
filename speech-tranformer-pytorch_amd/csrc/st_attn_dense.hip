// Attention under an ARBITRARY dense mask, and with keys and values projected from different tensors: the general form of
// reference transformer/Attention.py:64-96 (mask [B, Lq, Lk], any pattern; k and v any two tensors of equal length).  Every
// call site of the reference passes one of two mask families (key padding, key padding | causal: Utils.py:41-70) and k == v,
// and those take the fused kernels of st_attn*.hip through (offset, length, causal) predicates; this file is the SLOW path
// that closes the boundary for everything else - one workgroup per (utterance, head, query) forward and for the query-side
// gradients, one per (utterance, head, key) for the key-side gradients, plain fp32 FMAs, scores in LDS, no MFMA, no atomics.
//   forward   s = scale q.k (masked: -inf), P = softmax(s), Pd = dropout(P), O = Pd V        lse = log-sum-exp (natural log)
//   backward  dP = dO V^T, delta = sum_k Pd dP, dS = P (keep/(1-p) dP - delta), dQ = scale dS K, dK = scale dS^T Q, dV = Pd^T dO
// A row whose keys are ALL masked gets a zero context and zero gradients (the reference: softmax over -inf = NaN).
// Layout: padded - utterance b owns query rows b Lq .. and key rows b Lk .. of the row matrices; head h = columns h d_k ..
#include "st_common.cuh"

namespace {

struct DenseArgs {
  const bf16* Q; int ldq; const bf16* K; int ldk; const bf16* V; int ldv;
  const unsigned char* mask;      // [B, Lq, Lk], nonzero = masked (NULL: nothing masked)
  bf16* O; int ldo; float* lse;   // lse [H, B Lq]
  float* P;                       // optional [B, H, Lq, Lk]: the probabilities (before dropout), Attention.py:96
  const bf16* dO; int lddo; float* delta;      // delta [H, B Lq]: written by the dQ pass, read by the dK/dV pass
  bf16* dQ; int lddq; bf16* dK; int lddk; bf16* dV; int lddv;
  int B, H, d_k, Lq, Lk;
  float scale;
  DropArgs drop;
};

__device__ __forceinline__ float block_sum(float v, float* red) {      // 256 threads; red: 4 floats
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__device__ __forceinline__ float dot_row(const float* a, const bf16* row, int d_k) {
  float acc = 0.f;
  for (int i = 0; i < d_k; i += 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(row + i);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = fmaf(a[i + e], (float)v[e], acc);
  }
  return acc;
}

// keep-scale of pair (q, k) of head slot bh: 0 (dropped) or 1 / (1 - p); 1 when dropout is off
__device__ __forceinline__ float keep_scale(const Drop& dr, int bh, int q, int k) {
  if (!dr.on()) return 1.f;
  return dr.keep(dr.bits(drop_counter_qk(bh, q, k)), 2 * (q & 1) + (k & 1)) ? dr.scale : 0.f;
}

// one workgroup per (b, h, q).  BWD = false: forward.  BWD = true: the query side of the backward (dQ, delta).
template <bool BWD>
__global__ __launch_bounds__(256) void dense_q_kernel(DenseArgs a) {
  extern __shared__ float sm_dense[];      // [Lk] scores / probabilities, [Lk] dS (BWD), [d_k] q, [d_k] dO (BWD), [4 d_k] partials, [4] red
  float* sc = sm_dense;
  float* ds = sc + a.Lk;
  float* qv = ds + (BWD ? a.Lk : 0);
  float* dov = qv + a.d_k;
  float* part = dov + (BWD ? a.d_k : 0);
  float* red = part + 4 * a.d_k;
  const int q = blockIdx.x % a.Lq, h = (blockIdx.x / a.Lq) % a.H, b = blockIdx.x / (a.Lq * a.H);
  const int tid = threadIdx.x, bh = b * a.H + h;
  const size_t qrow = (size_t)b * a.Lq + q, krow0 = (size_t)b * a.Lk;
  const Drop dr = make_drop(a.drop);
  const unsigned char* mrow = a.mask ? a.mask + ((size_t)b * a.Lq + q) * a.Lk : nullptr;
  for (int i = tid; i < a.d_k; i += 256) {
    qv[i] = (float)a.Q[qrow * a.ldq + h * a.d_k + i];
    if (BWD) dov[i] = (float)a.dO[qrow * a.lddo + h * a.d_k + i];
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int k = tid; k < a.Lk; k += 256) {
    float s = -INFINITY;
    if (!mrow || !mrow[k]) s = a.scale * dot_row(qv, a.K + (krow0 + k) * a.ldk + h * a.d_k, a.d_k);
    sc[k] = s;
    mx = fmaxf(mx, s);
  }
  float lse;
  if (!BWD) {
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int k = tid; k < a.Lk; k += 256) {
      const float e = mx == -INFINITY ? 0.f : __expf(sc[k] - mx);
      sc[k] = e;
      sum += e;
    }
    sum = block_sum(sum, red);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    lse = sum > 0.f ? mx + __logf(sum) : INFINITY;      // +inf: the backward recomputes P = exp(s - inf) = 0
    if (tid == 0) a.lse[(size_t)h * a.B * a.Lq + qrow] = lse;
    for (int k = tid; k < a.Lk; k += 256) {
      const float p = sc[k] * inv;
      if (a.P) a.P[((size_t)bh * a.Lq + q) * a.Lk + k] = p;
      sc[k] = p * keep_scale(dr, bh, q, k);
    }
  } else {
    lse = a.lse[(size_t)h * a.B * a.Lq + qrow];
    float dl = 0.f;
    for (int k = tid; k < a.Lk; k += 256) {
      const float p = sc[k] == -INFINITY ? 0.f : __expf(sc[k] - lse);
      const float dp = dot_row(dov, a.V + (krow0 + k) * a.ldv + h * a.d_k, a.d_k) * keep_scale(dr, bh, q, k);
      sc[k] = p;
      ds[k] = dp;
      dl += p * dp;
    }
    dl = block_sum(dl, red);
    if (tid == 0) a.delta[(size_t)h * a.B * a.Lq + qrow] = dl;
    for (int k = tid; k < a.Lk; k += 256) sc[k] = sc[k] * (ds[k] - dl);      // dS
  }
  __syncthreads();
  // out[i] = sum_k w[k] M[k][i]   (forward: Pd V -> O; backward: dS K -> dQ): thread (g, i) sums every fourth... key group g
  const bf16* M = BWD ? a.K : a.V;
  const int ldm = BWD ? a.ldk : a.ldv;
  for (int i0 = 0; i0 < a.d_k; i0 += 64) {
    const int i = i0 + (tid & 63), g = tid >> 6;
    float acc = 0.f;
    if (i < a.d_k)
      for (int k = g; k < a.Lk; k += 4) acc = fmaf(sc[k], (float)M[(krow0 + k) * ldm + h * a.d_k + i], acc);
    if (i < a.d_k) part[g * a.d_k + i] = acc;
  }
  __syncthreads();
  for (int i = tid; i < a.d_k; i += 256) {
    const float v = part[i] + part[a.d_k + i] + part[2 * a.d_k + i] + part[3 * a.d_k + i];
    if (!BWD) a.O[qrow * a.ldo + h * a.d_k + i] = (bf16)v;
    else a.dQ[qrow * a.lddq + h * a.d_k + i] = (bf16)(v * a.scale);
  }
}

// the key side of the backward: one workgroup per (b, h, k) walks the queries (delta from the dQ pass)
__global__ __launch_bounds__(256) void dense_kv_kernel(DenseArgs a) {
  extern __shared__ float sm_dense[];      // [Lq] dS, [Lq] Pd, [d_k] k, [d_k] v, [8 d_k] partials
  float* ds = sm_dense;
  float* pd = ds + a.Lq;
  float* kv = pd + a.Lq;
  float* vv = kv + a.d_k;
  float* part = vv + a.d_k;
  const int k = blockIdx.x % a.Lk, h = (blockIdx.x / a.Lk) % a.H, b = blockIdx.x / (a.Lk * a.H);
  const int tid = threadIdx.x, bh = b * a.H + h;
  const size_t krow = (size_t)b * a.Lk + k, qrow0 = (size_t)b * a.Lq;
  const Drop dr = make_drop(a.drop);
  for (int i = tid; i < a.d_k; i += 256) {
    kv[i] = (float)a.K[krow * a.ldk + h * a.d_k + i];
    vv[i] = (float)a.V[krow * a.ldv + h * a.d_k + i];
  }
  __syncthreads();
  for (int q = tid; q < a.Lq; q += 256) {
    float p = 0.f, dsv = 0.f, pdv = 0.f;
    if (!a.mask || !a.mask[((size_t)b * a.Lq + q) * a.Lk + k]) {
      const size_t st = (size_t)h * a.B * a.Lq + qrow0 + q;
      const float s = a.scale * dot_row(kv, a.Q + (qrow0 + q) * a.ldq + h * a.d_k, a.d_k);
      p = __expf(s - a.lse[st]);
      const float ks = keep_scale(dr, bh, q, k);
      const float dp = dot_row(vv, a.dO + (qrow0 + q) * a.lddo + h * a.d_k, a.d_k) * ks;
      dsv = p * (dp - a.delta[st]);
      pdv = p * ks;
    }
    ds[q] = dsv;
    pd[q] = pdv;
  }
  __syncthreads();
  for (int i0 = 0; i0 < a.d_k; i0 += 64) {
    const int i = i0 + (tid & 63), g = tid >> 6;
    float ak = 0.f, av = 0.f;
    if (i < a.d_k)
      for (int q = g; q < a.Lq; q += 4) {
        ak = fmaf(ds[q], (float)a.Q[(qrow0 + q) * a.ldq + h * a.d_k + i], ak);
        av = fmaf(pd[q], (float)a.dO[(qrow0 + q) * a.lddo + h * a.d_k + i], av);
      }
    if (i < a.d_k) { part[g * a.d_k + i] = ak; part[(4 + g) * a.d_k + i] = av; }
  }
  __syncthreads();
  for (int i = tid; i < a.d_k; i += 256) {
    a.dK[krow * a.lddk + h * a.d_k + i] =
        (bf16)((part[i] + part[a.d_k + i] + part[2 * a.d_k + i] + part[3 * a.d_k + i]) * a.scale);
    a.dV[krow * a.lddv + h * a.d_k + i] =
        (bf16)(part[4 * a.d_k + i] + part[5 * a.d_k + i] + part[6 * a.d_k + i] + part[7 * a.d_k + i]);
  }
}

int fill(DenseArgs& a, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const unsigned char* mask, int B, int H,
         int d_k, int Lq, int Lk, float scale, const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 1;
  if (!Q || !K || !V || d_k <= 0 || (d_k & 7) || (ldq & 7) || (ldk & 7) || (ldv & 7)) return -1;
  if ((long long)B * H * (Lq > Lk ? Lq : Lk) > 0x7fffffffLL) return -3;
  a = DenseArgs{};
  a.Q = (const bf16*)Q; a.ldq = ldq; a.K = (const bf16*)K; a.ldk = ldk; a.V = (const bf16*)V; a.ldv = ldv; a.mask = mask;
  a.B = B; a.H = H; a.d_k = d_k; a.Lq = Lq; a.Lk = Lk; a.scale = scale;
  const bool on = drop_seed != nullptr && drop_thresh > 0;
  a.drop.seed = on ? drop_seed : nullptr; a.drop.salt = drop_salt; a.drop.thresh = on ? drop_thresh : 0; a.drop.scale = on ? drop_scale : 1.f;
  return 0;
}

}  // namespace

extern "C" int st_attn_dense_fwd(hipStream_t stream, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                                 const unsigned char* mask, void* O, int ldo, float* lse, float* P, int B, int H, int d_k, int Lq,
                                 int Lk, float scale, const unsigned* drop_seed, unsigned drop_salt, int drop_thresh,
                                 float drop_scale) {
  DenseArgs a;
  const int rc = fill(a, Q, ldq, K, ldk, V, ldv, mask, B, H, d_k, Lq, Lk, scale, drop_seed, drop_salt, drop_thresh, drop_scale);
  if (rc) return rc > 0 ? 0 : rc;
  if (!O || !lse || (ldo & 7)) return -2;
  a.O = (bf16*)O; a.ldo = ldo; a.lse = lse; a.P = P;
  const size_t smem = (size_t)(Lk + 5 * d_k + 4) * sizeof(float);
  if (smem > 64 * 1024) return -3;
  hipLaunchKernelGGL(dense_q_kernel<false>, dim3(B * H * Lq), dim3(256), smem, stream, a);
  ST_CHECK_LAUNCH();
  return 0;
}

extern "C" int st_attn_dense_bwd(hipStream_t stream, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                                 const unsigned char* mask, const void* dO, int lddo, const float* lse, float* delta, void* dQ, int lddq,
                                 void* dK, int lddk, void* dV, int lddv, int B, int H, int d_k, int Lq, int Lk, float scale,
                                 const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale) {
  DenseArgs a;
  const int rc = fill(a, Q, ldq, K, ldk, V, ldv, mask, B, H, d_k, Lq, Lk, scale, drop_seed, drop_salt, drop_thresh, drop_scale);
  if (rc) return rc > 0 ? 0 : rc;
  if (!dO || !lse || !delta || !dQ || !dK || !dV || (lddo & 7) || (lddq & 7) || (lddk & 7) || (lddv & 7)) return -2;
  a.dO = (const bf16*)dO; a.lddo = lddo; a.lse = const_cast<float*>(lse); a.delta = delta;
  a.dQ = (bf16*)dQ; a.lddq = lddq; a.dK = (bf16*)dK; a.lddk = lddk; a.dV = (bf16*)dV; a.lddv = lddv;
  const size_t sq = (size_t)(2 * Lk + 6 * d_k + 4) * sizeof(float), sk = (size_t)(2 * Lq + 10 * d_k) * sizeof(float);
  if (sq > 64 * 1024 || sk > 64 * 1024) return -3;
  hipLaunchKernelGGL(dense_q_kernel<true>, dim3(B * H * Lq), dim3(256), sq, stream, a);
  hipLaunchKernelGGL(dense_kv_kernel, dim3(B * H * Lk), dim3(256), sk, stream, a);
  ST_CHECK_LAUNCH();
  return 0;
}

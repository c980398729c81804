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
//
// Work decomposition: one workgroup (4 waves x 32 rows) per (utterance, head, 128-row tile).  The host
// passes a WORK LIST of (utterance, tile) pairs sorted by decreasing cost (number of streamed tiles), so the
// hardware dispatcher - which hands out workgroups in blockIdx order as CUs free up - does longest-first
// list scheduling over the ragged batch; without a list the kernels enumerate utterance-major.
//
// Memory pipeline (same scheme as st_gemm_sym.hip; measured: the kernels are instruction-issue bound, and
// LDS-DMA tops out at ~10 B/clk/CU): every thread carries 16-byte chunks of the streamed 64-row tiles
// global -> registers (two tiles in flight) -> padded LDS double buffer, with byte offsets fixed for the whole
// kernel - the steady-state loop has no address arithmetic, no guards and one barrier per tile.  Rows past
// the end of a sequence are CLAMPED onto its last row (finite data); the mask path zeroes their weight.
#include "st_attn_common.cuh"

// st_attn64.hip (AttnArgs passed by address: the type is local to each translation unit, the layout is shared)
extern "C" int st_attn64_fwd_launch(hipStream_t stream, const void* args, int grid_x, int drop, int kpre);
// st_attn_xs.hip: few queries against many keys (the decoder-encoder attention), 64-wide heads
extern "C" int st_attn_xs_fwd_launch(hipStream_t stream, const void* args, int grid_x, int drop, const void* f1, const void* self);
extern "C" void st_attn_xs_self_args(void* out, const void* Q, const void* K, const void* V, int ld, void* O, void* Ores, int ldo,
                                     float* lse, const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale);
extern "C" int st_attn_xs_self_args_size();
extern "C" void st_attn_xs_f1_args(void* out, const void* A, int lda, const void* R, int ldr, const void* wfrag, int n_blocks,
                                   int next_blocks, float eps, const float* bo, const float* g0, const float* be0, void* out0,
                                   void* xhat0, float* rstd0, const float* bq, void* Qout, int ldq);
extern "C" int st_attn_xs_f1_args_size();
extern "C" int st_attn_xs_tile_rows();
// st_attn_bwd64.hip: the hand-scheduled backward for long non-causal problems with 64-wide heads
extern "C" int st_attn_bwd64_launch(hipStream_t stream, const void* a, const void* ak, int n_q, int n_k, int drop);

namespace {

// KS = 2 ("few queries, many keys": the decoder-encoder attention, <= 64 queries against ~1000 keys): the workgroup
// owns 64 query rows and streams 128-key stages; waves 0,1 take the first 64 keys of a stage, waves 2,3 the second
// and the two partial softmax states are merged through LDS at the end - half the serial tile chain per workgroup.
// One workgroup per CU for KS = 2 (its 128-key register stages + the two-term P need > 256 registers: 44 spilled at two
// per CU; the decoder-encoder attention is <= one workgroup per CU anyway): 21.4 -> 18.6 us, same box.
// PS: P enters the P V product as two bf16 terms (AttnArgs::psplit).  A template parameter, not a run-time branch: the two
// paths keep the O accumulators in different registers, and the compiler reconciled them with 64 v_mov_b64 per 64-key
// tile in the single-term path (a quarter of that loop's VALU instructions).
template <int DK, bool DROP, int KS, bool PS>
__global__ __launch_bounds__(256, (KS > 1 || DK > 64) ? 1 : 2) void attn_fwd_kernel(AttnArgs a) {
  using G = TileGeo<DK, TILE * KS>;
  constexpr int NT = DK / 16;   // k-steps of the QK^T contraction
  constexpr int ND = DK / 32;   // 32-wide output column tiles
  constexpr int ROWS = TILE * KS, QROWS = WG_ROWS / KS;
  __shared__ __attribute__((aligned(16))) bf16 smem[4 * G::E];   // 2 buffers x (K tile, V tile)

  int b, h, tile;
  decode_item(a, blockIdx.x, b, h, tile);
  const int lq = a.q_len[b], lk = a.k_len[b];
  const int q0 = tile * QROWS;
  if (q0 >= lq) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hi = l >> 5;
  const int qw = KS > 1 ? (wave & 1) : wave, kp = KS > 1 ? (wave >> 1) : 0;   // query block, key half
  const int q = q0 + qw * 32 + (l & 31);
  const bool q_ok = q < lq;
  const size_t qrow = (size_t)a.q_off[b] + min(q, lq - 1);
  const float c2 = a.c2;  // scores -> log2 domain (scale * log2 e; 1 when K arrives pre-scaled)
  const Drop dr = make_drop(a.drop);
  const int bh = b * a.H + h;

  const int k_hi = a.causal ? min(lk, q0 + QROWS) : lk;   // keys this workgroup can see
  const int ntiles = (k_hi + ROWS - 1) / ROWS;
  const bf16* kbase = a.K + (size_t)a.k_off[b] * a.ldk + h * DK;
  const bf16* vbase = a.V + (size_t)a.k_off[b] * a.ldv + h * DK;

  bf16x8 qf[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) qf[t] = *reinterpret_cast<const bf16x8*>(a.Q + qrow * a.ldq + h * DK + t * 16 + hi * 8);

  uint32_t offk[G::CH], offv[G::CH];
  Stage<DK, ROWS>::offsets(offk, a.ldk);
  Stage<DK, ROWS>::offsets(offv, a.ldv);
  Stage<DK, ROWS> sk[2], sv[2];

  f32x16 o[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) o[d] = zero16();
  float m = -INFINITY, lsum = 0.f;

  auto load = [&](int set, int it) {
    sk[set].load(offk, kbase, a.ldk, it * ROWS, lk);
    sv[set].load(offv, vbase, a.ldv, it * ROWS, lk);
  };
  auto store = [&](int set) {
    sk[set].store(smem + set * 2 * G::E);
    sv[set].store(smem + set * 2 * G::E + G::E);
  };
  auto compute = [&](int buf, int it) {
    const bf16* ks = smem + buf * 2 * G::E + kp * TILE * G::STR;   // this wave's 64 keys of the stage
    const bf16* vs = ks + G::E;
    const int kt = it * ROWS + kp * TILE;
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      s[kb] = zero16();
#pragma unroll
      for (int t = 0; t < NT; ++t) s[kb] = mfma32(rd_nat<DK>(ks, kb * 32 + (l & 31), t), qf[t], s[kb]);
    }
    // masks only on tiles that cross a sequence end or the diagonal (wave-uniform test)
    const bool full = (kt + TILE <= lk) && (!a.causal || kt + TILE - 1 <= q0 + qw * 32);
    if (!full) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt + kb * 32 + acc_row(r, hi);
          if (key >= lk || (a.causal && key > q)) s[kb][r] = -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, wave_xor32(mx));
    // log2 domain; m_new is finite from the first tile on (key 0 is visible to every query)
    const float m_new = fmaxf(m, mx * c2);
    if (__any(m_new != m)) {   // the running maximum settles after a few tiles: skip the rescale then
      const float m_fin = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m - m_fin);
      lsum *= alpha;
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      m = m_new;
    }
    const float m_use = (m == -INFINITY) ? 0.f : m;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][r], c2, -m_use));
        s[kb][r] = p;
        psum += p;
      }
    lsum += psum;
    if (DROP) {   // dropped probabilities leave the normaliser untouched; the 1/(1-p) scale is folded into `inv`
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        bool keep[16];
        keep16<true>(dr, bh, q, kt + kb * 32, hi, keep);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = keep[r] ? s[kb][r] : 0.f;
      }
    }
    // O^T += V^T P^T : A operand = V^T (transposing LDS read), B operand = P^T (own registers)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const bf16x8 pf = pack_acc8(s[kb], 8 * hf);
        const int base = kb * 32 + 16 * hf + 4 * hi;
        if (PS) {
          bf16x8 pl;
#pragma unroll
          for (int j = 0; j < 8; ++j) pl[j] = (bf16)(s[kb][8 * hf + j] - (float)pf[j]);
#pragma unroll
          for (int d = 0; d < ND; ++d) {
            const bf16x8 vf = rd_tr<DK>(vs, d * 32, base);
            o[d] = mfma32(vf, pf, o[d]);
            o[d] = mfma32(vf, pl, o[d]);
          }
        } else {
#pragma unroll
          for (int d = 0; d < ND; ++d) o[d] = mfma32(rd_tr<DK>(vs, d * 32, base), pf, o[d]);
        }
      }
  };
  stream_tiles(ntiles, load, store, compute);

  float ltot = lsum + wave_xor32(lsum);
  if (KS > 1) {   // merge the two key halves: (m, l, O) of waves 2,3 -> LDS -> waves 0,1
    float* xch = reinterpret_cast<float*>(smem + 2 * 32 * DK) + qw * (2 + ND * 16) * 64 + l;   // behind the 2 store patches
    if (kp == 1) {
      xch[0] = m;
      xch[64] = ltot;
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(2 + d * 16 + r) * 64] = o[d][r];
    }
    __syncthreads();
    if (kp == 1) return;
    const float m1 = xch[0], l1 = xch[64];
    const float mn = fmaxf(m, m1);
    const float a0 = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mn);
    const float a1 = (m1 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m1 - mn);
    ltot = ltot * a0 + l1 * a1;
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] = o[d][r] * a0 + xch[(2 + d * 16 + r) * 64] * a1;
    m = mn;
  }
  const float inv = ltot > 0.f ? (DROP ? dr.scale : 1.f) / ltot : 0.f;
  if (q_ok && hi == 0 && a.lse) a.lse[(size_t)h * a.q_rows_total + qrow] = m + log2f(ltot);
  if (a.Ores)      // (the tile buffers are free: the lo patches lie behind the hi patches / the key-split exchange area)
    store_rows_pair<DK>(smem + qw * 32 * DK, smem + (KS == 1 ? 4 * 32 * DK : 256 * DK) + qw * 32 * DK, o, inv, a.O + (size_t)a.q_off[b] * a.ldo + h * DK,
                        a.Ores + (size_t)a.q_off[b] * a.ldo + h * DK, a.ldo, q0 + qw * 32, min(32, lq - (q0 + qw * 32)));
  else
    store_rows<DK>(smem + qw * 32 * DK, o, inv, a.O + (size_t)a.q_off[b] * a.ldo + h * DK, a.ldo, q0 + qw * 32,
                   min(32, lq - (q0 + qw * 32)));
}

// ---------------------------------------------------------------------------------------------
// Backward, part 1: dQ (and delta = rowsum(dO * O)).  Same decomposition as the forward.
//   P^T = exp2(S^T c2 - lse),  dP^T = V dO^T,  dS^T = P^T (dP^T - delta),  dQ^T += K^T dS^T
// ---------------------------------------------------------------------------------------------
template <int DK, bool DROP, int KS>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnArgs& a, int bid, bf16* smem) {
  using G = TileGeo<DK, TILE * KS>;
  constexpr int NT = DK / 16, ND = DK / 32;
  constexpr int ROWS = TILE * KS, QROWS = WG_ROWS / KS;   // KS = 2: see attn_fwd_kernel

  // KS = 2 and a.xsplit > 1: the item's key tiles are cut over `xsplit` consecutive workgroups (round 6: one workgroup streaming
  // all ~750 keys of an utterance WAS the decoder-encoder attention's backward launch - 14 of its 21 us, and all of it at 4 utterances)
  const int xs = (KS > 1 && a.xsplit > 1) ? a.xsplit : 1;      // workgroup-uniform
  const int part = xs > 1 ? bid % xs : 0;
  if (xs > 1) bid /= xs;
  int b, h, tile;
  decode_item(a, bid, b, h, tile);
  const int lq = a.q_len[b], lk = a.k_len[b];
  const int q0 = tile * QROWS;
  if (q0 >= lq) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hi = l >> 5;
  const int qw = KS > 1 ? (wave & 1) : wave, kp = KS > 1 ? (wave >> 1) : 0;
  const int q = q0 + qw * 32 + (l & 31);
  const bool q_ok = q < lq;
  const size_t qrow = (size_t)a.q_off[b] + min(q, lq - 1);
  const float c2 = a.c2;
  const Drop dr = make_drop(a.drop);
  const int bh = b * a.H + h;

  const int k_hi = a.causal ? min(lk, q0 + QROWS) : lk;
  const int ntiles = (k_hi + ROWS - 1) / ROWS;
  const int nparts = min(xs, max(ntiles, 1));      // (an item with fewer tiles than parts: the spare workgroups leave at once)
  if (part >= nparts) return;
  const int it0 = part * ntiles / nparts, it1 = (part + 1) * ntiles / nparts;
  const bf16* kbase = a.K + (size_t)a.k_off[b] * a.ldk + h * DK;
  const bf16* vbase = a.V + (size_t)a.k_off[b] * a.ldv + h * DK;

  bf16x8 qf[NT], dof[NT];
  float dl = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = h * DK + t * 16 + hi * 8;
    qf[t] = *reinterpret_cast<const bf16x8*>(a.Q + qrow * a.ldq + col);
    dof[t] = *reinterpret_cast<const bf16x8*>(a.dO + qrow * a.lddo + col);
  }
  if (a.O != nullptr) {   // delta = rowsum(dO * O) computed (and published) here ...
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(a.O + qrow * a.ldo + h * DK + t * 16 + hi * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += (float)dof[t][e] * (float)of[e];
    }
    dl += wave_xor32(dl);
    if (q_ok && hi == 0 && kp == 0 && part == 0) a.delta[(size_t)h * a.q_rows_total + qrow] = dl;
  } else {                // ... or already produced by the launch that wrote dO (st_gemm, ST_EPI_BF16_DELTA)
    dl = a.delta[(size_t)h * a.q_rows_total + qrow];
  }
  const float lse = a.lse[(size_t)h * a.q_rows_total + qrow];

  uint32_t offk[G::CH], offv[G::CH];
  Stage<DK, ROWS>::offsets(offk, a.ldk);
  Stage<DK, ROWS>::offsets(offv, a.ldv);
  Stage<DK, ROWS> sk[2], sv[2];

  f32x16 dq[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) dq[d] = zero16();

  auto load = [&](int set, int it) {
    sk[set].load(offk, kbase, a.ldk, (it0 + it) * ROWS, lk);
    sv[set].load(offv, vbase, a.ldv, (it0 + it) * ROWS, lk);
  };
  auto store = [&](int set) {
    sk[set].store(smem + set * 2 * G::E);
    sv[set].store(smem + set * 2 * G::E + G::E);
  };
  auto compute = [&](int buf, int it) {
    const bf16* ks = smem + buf * 2 * G::E + kp * TILE * G::STR;
    const bf16* vs = ks + G::E;
    const int kt = (it0 + it) * ROWS + kp * TILE;
    const bool full = (kt + TILE <= lk) && (!a.causal || kt + TILE - 1 <= q0 + qw * 32);   // no masks needed
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
      if (!full) {   // masked pairs: exp2(-inf) = 0 (one wave-uniform branch; the exp chain stays straight-line)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt + kb * 32 + acc_row(r, hi);
          if (key >= lk || (a.causal && key > q)) s[r] = -INFINITY;
        }
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
    }
  };
  stream_tiles(it1 - it0, load, store, compute);
  if (KS > 1) {   // dQ of the two key halves: waves 2,3 -> LDS -> waves 0,1
    float* xch = reinterpret_cast<float*>(smem + 2 * 32 * DK) + qw * (ND * 16) * 64 + l;
    if (kp == 1) {
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(d * 16 + r) * 64] = dq[d][r];
    }
    __syncthreads();
    if (kp == 0) {
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] += xch[(d * 16 + r) * 64];
    }
    if (nparts > 1) {      // (workgroup-uniform) the parts of the item: fp32 partial -> scratch, ticket, the last arriver adds them in order
      __shared__ bool last;
      constexpr int WAVE8 = ND * 8 * 64;      // 8-byte words of one wave's 32 x DK partial
      unsigned long long* slot = reinterpret_cast<unsigned long long*>(a.xs_ws) + (size_t)bid * xs * (2 * WAVE8);
      const bool live = kp == 0 && q0 + qw * 32 < lq;      // (wave-uniform) this wave holds rows of the item
      auto pack2 = [](float lo, float hi_) { return ((unsigned long long)__float_as_uint(hi_) << 32) | __float_as_uint(lo); };
      if (live) {
        unsigned long long* mine = slot + (size_t)(part * 2 + qw) * WAVE8 + l;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
          for (int r = 0; r < 16; r += 2)
            __hip_atomic_store(mine + (d * 8 + r / 2) * 64, pack2(dq[d][r], dq[d][r + 1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      ST_PUBLISH_FENCE();
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned* tk = a.xs_tickets + bid;
        last = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nparts - 1);
        if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if (!last) return;
      ST_MERGER_FENCE();
      if (live) {
#pragma unroll
        for (int d = 0; d < ND; ++d) dq[d] = zero16();
        for (int p = 0; p < nparts; ++p) {      // part order: the sum does not depend on who is last
          const unsigned long long* src = slot + (size_t)(p * 2 + qw) * WAVE8 + l;
#pragma unroll
          for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const unsigned long long v = __hip_atomic_load(src + (d * 8 + r / 2) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              dq[d][r] += __uint_as_float((unsigned)v);
              dq[d][r + 1] += __uint_as_float((unsigned)(v >> 32));
            }
        }
      }
    }
    if (kp == 1) return;
  }
  store_rows<DK>(smem + qw * 32 * DK, dq, a.dq_scale, a.dQ + (size_t)a.q_off[b] * a.lddq + h * DK, a.lddq,
                 q0 + qw * 32, min(32, lq - (q0 + qw * 32)));
}

template <int DK, bool DROP, int KS>
__global__ __launch_bounds__(256, DK > 64 ? 1 : 2) void attn_bwd_dq_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) bf16 smem[4 * TileGeo<DK, TILE * KS>::E];
  attn_bwd_dq_body<DK, DROP, KS>(a, blockIdx.x, smem);
}

// ---------------------------------------------------------------------------------------------
// Backward, part 2: dK, dV.  Each wave owns 32 keys (lane & 31) and loops over 64-query tiles
// (Q rows, dO rows and the tile's 64 lse + 64 delta values).
//   S = Q K^T (lane = key, registers = queries),  P = exp2(S c2 - lse[q])
//   dV^T += dO^T P,   dP = dO V^T,   dS = P (dP - delta[q]),   dK^T += Q^T dS
// ---------------------------------------------------------------------------------------------
template <int DK, bool DROP>
__device__ __forceinline__ void attn_bwd_dkv_body(const AttnArgs& a, int bid, bf16* smem) {
  using G = TileGeo<DK>;
  constexpr int NT = DK / 16, ND = DK / 32;
  constexpr int BUF = 2 * G::E + 256;   // Q tile, dO tile, lse[64] + delta[64] (fp32, counted in bf16 elements)

  int b, h, tile;
  decode_item(a, bid, b, h, tile);
  const int lq = a.q_len[b], lk = a.k_len[b];
  const int k0 = tile * WG_ROWS;
  if (k0 >= lk) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, hi = l >> 5;
  const int key = k0 + wave * 32 + (l & 31);
  const size_t krow = (size_t)a.k_off[b] + min(key, lk - 1);
  const float c2 = a.c2;
  const Drop dr = make_drop(a.drop);
  const int bh = b * a.H + h;

  const bf16* qbase = a.Q + (size_t)a.q_off[b] * a.ldq + h * DK;
  const bf16* dobase = a.dO + (size_t)a.q_off[b] * a.lddo + h * DK;
  // threads 0..63 carry the tile's lse values, 64..127 its delta values (128.. duplicate them)
  const float* statsrc = ((threadIdx.x & 64) ? a.delta : a.lse) + (size_t)h * a.q_rows_total + a.q_off[b];
  const int q_begin = a.causal ? (k0 / TILE) * TILE : 0;  // queries before the first key see none of them
  const int ntiles = (lq - q_begin + TILE - 1) / TILE;

  bf16x8 kf[NT], vf[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = h * DK + t * 16 + hi * 8;
    kf[t] = *reinterpret_cast<const bf16x8*>(a.K + krow * a.ldk + col);
    vf[t] = *reinterpret_cast<const bf16x8*>(a.V + krow * a.ldv + col);
  }

  uint32_t offq[G::CH], offo[G::CH];
  Stage<DK>::offsets(offq, a.ldq);
  Stage<DK>::offsets(offo, a.lddo);
  Stage<DK> sq[2], so[2];
  float sst[2];

  f32x16 dk[ND], dv[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) { dk[d] = zero16(); dv[d] = zero16(); }

  auto load = [&](int set, int it) {
    const int qt = q_begin + it * TILE;
    sq[set].load(offq, qbase, a.ldq, qt, lq);
    so[set].template load<true>(offo, dobase, a.lddo, qt, lq);          // queries past the end: ZERO dO rows and a zero delta -
    sst[set] = statsrc[min(qt + (int)(threadIdx.x & 63), lq - 1)];    // dP = 0, dS = P (0 - 0) = 0, dV += 0: no mask needed
    if ((threadIdx.x & 64) && qt + (int)(threadIdx.x & 63) >= lq) sst[set] = 0.f;
  };
  auto store = [&](int set) {
    bf16* base = smem + set * BUF;
    sq[set].store(base);
    so[set].store(base + G::E);
    reinterpret_cast<float*>(base + 2 * G::E)[threadIdx.x & 127] = sst[set];
  };
  auto compute = [&](int buf, int it) {
    const bf16* qs = smem + buf * BUF;
    const bf16* dos = qs + G::E;
    const float* stat = reinterpret_cast<const float*>(qs + 2 * G::E);   // [0..63] lse, [64..127] delta
    const int qt = q_begin + it * TILE;
    // wave-uniform: every (query, key) pair of this tile x this wave's 32 keys is unmasked.  Keys past lk need no mask:
    // lane = key here, so such a lane (its K / V rows are clamped, finite) only fills its OWN dK / dV rows, which are never
    // stored - a partial last key block runs the plain path on every tile (it took the masked one on all of them)
    const bool full = !a.causal || k0 + wave * 32 + 31 <= qt;
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
      if (!full) {   // masked pairs: exp2(-inf) = 0 (one wave-uniform branch; the exp chain stays straight-line)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qq = qt + qb * 32 + acc_row(r, hi);
          if (qq >= lq || (a.causal && key > qq)) s[r] = -INFINITY;
        }
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
    }
  };
  stream_tiles(ntiles, load, store, compute);
  const int nrows = min(32, lk - (k0 + wave * 32));
  store_rows<DK>(smem + wave * 64 * DK, dk, a.scale, a.dK + (size_t)a.k_off[b] * a.lddk + h * DK, a.lddk,
                 k0 + wave * 32, nrows);
  store_rows<DK>(smem + wave * 64 * DK + 32 * DK, dv, 1.f, a.dV + (size_t)a.k_off[b] * a.lddv + h * DK, a.lddv,
                 k0 + wave * 32, nrows);
}

template <int DK, bool DROP>
__global__ __launch_bounds__(256, DK > 64 ? 1 : 2) void attn_bwd_dkv_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) bf16 smem[2 * (2 * TileGeo<DK>::E + 256)];
  attn_bwd_dkv_body<DK, DROP>(a, blockIdx.x, smem);
}

// dQ and dK/dV in ONE launch (possible when delta comes from the producer of dO: no kernel-to-kernel dependency
// is left).  Workgroups [0, n_k) run the dK/dV body on the key-tile work list (the heavier items: four
// contractions per tile), the rest the dQ body on the query-tile list, which fills in as the dK/dV items drain
// (the key-split variant KS = 2 orders them the other way round, see below).
template <int DK, bool DROP, int KS>
__global__ __launch_bounds__(256, DK > 64 ? 1 : 2) void attn_bwd_kernel(AttnArgs a, AttnArgs ak, int n_k) {
  constexpr int EQ = 4 * TileGeo<DK, TILE * KS>::E, EK = 2 * (2 * TileGeo<DK>::E + 256);
  __shared__ __attribute__((aligned(16))) bf16 smem[EQ > EK ? EQ : EK];
  if (KS > 1) {
    // few queries against many keys (decoder-encoder attention): the dQ items are the long serial chains here (one
    // workgroup streams all keys of an utterance), so they are dispatched first and the one-tile dK/dV items fill in
    // around them (27.2 -> 23.3 us at config 2)
    const int n_q = (int)gridDim.x - n_k;
    if ((int)blockIdx.x < n_q) attn_bwd_dq_body<DK, DROP, KS>(a, blockIdx.x, smem);
    else attn_bwd_dkv_body<DK, DROP>(ak, blockIdx.x - n_q, smem);
    return;
  }
  if ((int)blockIdx.x < n_k) attn_bwd_dkv_body<DK, DROP>(ak, blockIdx.x, smem);
  else attn_bwd_dq_body<DK, DROP, KS>(a, blockIdx.x - n_k, smem);
}

int check_common(int d_k, int ldq, int ldk, int ldv) {
  if (d_k != 32 && d_k != 64 && d_k != 128) return -1;
  if ((ldq & 7) || (ldk & 7) || (ldv & 7)) return -2;
  return 0;
}

bool set_drop(AttnArgs& a, const unsigned* seed, unsigned salt, int thresh, float scale) {
  const bool on = seed != nullptr && thresh > 0;
  a.drop.seed = on ? seed : nullptr;
  a.drop.salt = salt;
  a.drop.thresh = on ? thresh : 0;
  a.drop.scale = on ? scale : 1.f;
  return on;
}

// few queries against many keys (decoder-encoder attention): split the keys over the wave pairs
bool key_split(int max_q, int max_k, int causal) { return !causal && max_q <= 64 && max_k >= 256; }

// ... and, in the merged backward launch, over up to four workgroups per (utterance, head): whole 128-key tiles each.  Scratch:
// XS_TICKETS tickets (zero before the first launch; the kernel leaves them zero), then xs fp32 partials of 64 x d_k per item.
constexpr int XS_TICKETS = 4096;
int xsplit_for(int d_k, int max_q, int max_k, int causal) {
  static const bool off = [] { const char* e = getenv("ST_ATTN_XSPLIT"); return e && e[0] == '0'; }();      // development switch
  if (off || !key_split(max_q, max_k, causal)) return 1;
  const int nt = (max_k + 127) / 128;
  return nt < 4 ? nt : 4;
}
int device_cus() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cus;
  }();
  return n;
}
long long xsplit_bytes(int items, int xs, int d_k) { return (long long)XS_TICKETS * 4 + (long long)items * xs * 64 * d_k * 4; }

// grid size and enumeration mode for one family of workgroups
int plan(AttnArgs& a, const int* work, int n_work, int B, int H, int max_rows, int wg_rows = WG_ROWS) {
  a.work = work;
  a.H = H;
  a.tiles_max = (max_rows + wg_rows - 1) / wg_rows;
  return (work ? n_work : B * a.tiles_max) * H;
}

// development switch (ST_ATTN_IMPL=1: the general kernels everywhere), read once
int attn_impl() {
  static const int v = [] { const char* e = getenv("ST_ATTN_IMPL"); return e ? atoi(e) : 0; }();
  return v;
}

// development switches (same-process A/B runs of the specialised kernels against the general ones): read from the
// environment ONCE, and again only when a host asks (st_env_refresh: tests and tools/dev call it after changing a variable) -
// the production dispatch never calls getenv
struct AttnEnv {
  bool fwd64_off, xs_off;      // ST_ATTN_FWD64=0, ST_ATTN_XS=0
  int bwd64;                   // ST_ATTN_BWD64: 0 = default (streams), 1 = "0" (general kernels), 2 = "e" (streams in eval mode only)
};
AttnEnv read_attn_env() {
  auto off = [](const char* name) { const char* e = getenv(name); return e && e[0] == '0'; };
  const char* b = getenv("ST_ATTN_BWD64");
  return {off("ST_ATTN_FWD64"), off("ST_ATTN_XS"), (b && b[0] == '0') ? 1 : (b && b[0] == 'e') ? 2 : 0};
}
AttnEnv& attn_env() {
  static AttnEnv e = read_attn_env();
  return e;
}

// long non-causal problems with 64-wide heads take the plain-exponential forward of st_attn64.hip
bool fwd_long64(int d_k, int max_q, int max_k, int causal) {
  return d_k == 64 && !causal && max_q > 128 && attn_impl() != 1 && !attn_env().fwd64_off;
}

// ... and the hand-scheduled backward of st_attn_bwd64.hip (delta supplied by the producer of dO).  ST_ATTN_BWD64=0 keeps the
// general kernels, =e the streams in eval mode only (no dropout variant)
bool bwd_long64(int d_k, int max_q, int max_k, int causal, bool drop) {
  if (!(d_k == 64 && !causal && max_q > 128 && max_k > 128 && attn_impl() != 1)) return false;
  const int mode = attn_env().bwd64;
  return mode == 0 || (mode == 2 && !drop);
}

// few queries against many keys with 64-wide heads (the decoder-encoder attention) take the forward of st_attn_xs.hip
bool fwd_xs(int d_k, int max_q, int max_k, int causal) {
  return d_k == 64 && key_split(max_q, max_k, causal) && attn_impl() != 1 && !attn_env().xs_off;
}

}  // namespace

extern "C" int st_env_refresh(void) {
  // re-read the development switches ST_ATTN_FWD64 / ST_ATTN_XS / ST_ATTN_BWD64 (cached at first use otherwise)
  attn_env() = read_attn_env();
  return 0;
}

extern "C" int st_attn_tile_rows(int which, int d_k, int max_q, int max_k, int causal) {
  // rows per work-list tile of the kernel that will serve this problem: which = 0 forward (query tiles),
  // 1 backward dQ (query tiles), 2 backward dK/dV (key tiles).  128-row workgroups except the few-queries forward of
  // st_attn_xs.hip (32): hosts must ask (the answer depends on the shape).
  if (which == 0 && fwd_xs(d_k, max_q, max_k, causal)) return st_attn_xs_tile_rows();
  return WG_ROWS;
}

extern "C" int st_attn_fwd(hipStream_t stream, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                           void* O, int ldo, void* Ores, float* lse, const int* q_off, const int* q_len, const int* k_off,
                           const int* k_len, int B, int H, int d_k, int max_q, int max_k, int q_rows_total, int causal,
                           float scale, const int* work, int n_work, const unsigned* drop_seed, unsigned drop_salt,
                           int drop_thresh, float drop_scale, int k_prescaled) {
  if (B <= 0 || H <= 0 || max_q <= 0 || (work && n_work <= 0)) return 0;
  int rc = check_common(d_k, ldq, ldk, ldv);
  if (rc) return rc;
  if (ldo & 7) return -3;
  if (B > 32767) return -4;
  AttnArgs a = {};
  a.Q = (const bf16*)Q; a.ldq = ldq; a.K = (const bf16*)K; a.ldk = ldk; a.V = (const bf16*)V; a.ldv = ldv;
  a.O = (bf16*)O; a.ldo = ldo; a.Ores = (bf16*)Ores; a.lse = lse; a.q_off = q_off; a.q_len = q_len; a.k_off = k_off; a.k_len = k_len;
  a.q_rows_total = q_rows_total; a.causal = causal; set_score_scales(a, scale, k_prescaled);
  const bool drop = set_drop(a, drop_seed, drop_salt, drop_thresh, drop_scale);
  a.psplit = (Ores != nullptr && max_q <= 64) ? 1 : 0;     // the decoder's attentions, when a backward will follow
  if (fwd_long64(d_k, max_q, max_k, causal)) {
    dim3 grid(plan(a, work, n_work, B, H, max_q));
    return st_attn64_fwd_launch(stream, &a, (int)grid.x, drop ? 1 : 0, k_prescaled ? 1 : 0);
  }
  const bool ks2 = key_split(max_q, max_k, causal);   // one 64-row query tile per utterance == the 128-row tile 0
  if (fwd_xs(d_k, max_q, max_k, causal))
    return st_attn_xs_fwd_launch(stream, &a, plan(a, work, n_work, B, H, max_q, st_attn_xs_tile_rows()), drop ? 1 : 0, nullptr, nullptr);
  dim3 grid(plan(a, work, n_work, B, H, max_q)), block(256);
#define ST_FWD(DKK, DR) \
  do { if (ks2 && a.psplit) hipLaunchKernelGGL((attn_fwd_kernel<DKK, DR, 2, true>), grid, block, 0, stream, a); \
       else if (ks2) hipLaunchKernelGGL((attn_fwd_kernel<DKK, DR, 2, false>), grid, block, 0, stream, a); \
       else if (a.psplit) hipLaunchKernelGGL((attn_fwd_kernel<DKK, DR, 1, true>), grid, block, 0, stream, a); \
       else hipLaunchKernelGGL((attn_fwd_kernel<DKK, DR, 1, false>), grid, block, 0, stream, a); } while (0)
  if (d_k == 64 && !drop) ST_FWD(64, false);
  else if (d_k == 64) ST_FWD(64, true);
  else if (d_k == 128 && !drop) ST_FWD(128, false);
  else if (d_k == 128) ST_FWD(128, true);
  else if (!drop) ST_FWD(32, false);
  else ST_FWD(32, true);
#undef ST_FWD
  ST_CHECK_LAUNCH();
  return 0;
}

extern "C" int st_attn_f1_applicable(int d_model, int d_k, int max_q, int max_k) {
  // st_attn_f1_fwd serves the shapes the few-queries forward serves, at d_model 256 (the row chains' width)
  return (d_model == 256 && fwd_xs(d_k, max_q, max_k, 0)) ? 1 : 0;
}

namespace {
int attn_f1_impl(hipStream_t stream, const void* ctxA, int lda, const void* R, int ldr, const void* wfrag, int n_blocks,
                 int next_blocks, float eps, const float* bo, const float* g0, const float* be0, void* out0, void* xhat0,
                 float* rstd0, const float* bq, void* Qout, int ldq, const void* K, int ldk, const void* V, int ldv, void* O,
                 int ldo, void* Ores, float* lse, const int* q_off, const int* q_len, const int* k_off, const int* k_len, int B,
                 int H, int d_k, int max_q, int max_k, int q_rows_total, float scale, const int* work, int n_work,
                 const unsigned* drop_seed, unsigned drop_salt, int drop_thresh, float drop_scale, const void* self_args) {
  if (B <= 0 || H <= 0 || max_q <= 0 || (work && n_work <= 0)) return 0;
  if (!st_attn_f1_applicable(H * d_k, d_k, max_q, max_k)) return -10;
  if ((!ctxA && !self_args) || !R || !wfrag || n_blocks != 2 || !bo || !g0 || !be0 || !out0 || !bq || !Qout) return -11;
  if ((lda & 7) || (ldr & 7) || (ldq & 7) || ldq < 256 || (ldk & 7) || (ldv & 7) || (ldo & 7)) return -2;
  if (B > 32767) return -4;
  AttnArgs a = {};
  a.Q = (const bf16*)Qout; a.ldq = ldq; a.K = (const bf16*)K; a.ldk = ldk; a.V = (const bf16*)V; a.ldv = ldv;
  a.O = (bf16*)O; a.ldo = ldo; a.Ores = (bf16*)Ores; a.lse = lse; a.q_off = q_off; a.q_len = q_len; a.k_off = k_off; a.k_len = k_len;
  a.q_rows_total = q_rows_total; a.causal = 0; set_score_scales(a, scale, 0);
  const bool drop = set_drop(a, drop_seed, drop_salt, drop_thresh, drop_scale);
  a.psplit = (Ores != nullptr && max_q <= 64) ? 1 : 0;
  alignas(16) char f1[256];
  if (st_attn_xs_f1_args_size() > (int)sizeof(f1)) return -12;
  st_attn_xs_f1_args(f1, ctxA, lda, R, ldr, wfrag, n_blocks, next_blocks, eps, bo, g0, be0, out0, xhat0, rstd0, bq, Qout, ldq);
  return st_attn_xs_fwd_launch(stream, &a, plan(a, work, n_work, B, H, max_q, st_attn_xs_tile_rows()), drop ? 1 : 0, f1, self_args);
}
}  // namespace

extern "C" int st_attn_f1_fwd(hipStream_t stream, const void* ctxA, int lda, const void* R, int ldr, const void* wfrag, int n_blocks,
                              int next_blocks, float eps, const float* bo, const float* g0, const float* be0, void* out0,
                              void* xhat0, float* rstd0, const float* bq, void* Qout, int ldq, const void* K, int ldk,
                              const void* V, int ldv, void* O, int ldo, void* Ores, float* lse, const int* q_off,
                              const int* q_len, const int* k_off, const int* k_len, int B, int H, int d_k, int max_q, int max_k,
                              int q_rows_total, float scale, const int* work, int n_work, const unsigned* drop_seed,
                              unsigned drop_salt, int drop_thresh, float drop_scale) {
  return attn_f1_impl(stream, ctxA, lda, R, ldr, wfrag, n_blocks, next_blocks, eps, bo, g0, be0, out0, xhat0, rstd0, bq, Qout, ldq, K,
                      ldk, V, ldv, O, ldo, Ores, lse, q_off, q_len, k_off, k_len, B, H, d_k, max_q, max_k, q_rows_total, scale, work,
                      n_work, drop_seed, drop_salt, drop_thresh, drop_scale, nullptr);
}

// st_attn_f1_fwd with the decoder's causal SELF-attention in front of the chain stage (its context never leaves the chip on
// its way to output_linear): qkv_q / qkv_k / qkv_v = the layer's q | k | v projection (leading dimension ld_qkv, head h at
// columns h * 64), Os / Oress / lses = what st_attn_fwd(causal) would have written (bit-identical), the self-attention's dropout
// with its own salt / threshold / scale on the shared device seed.  The self-attention's keys are the utterance's own <= 64
// target positions (q_off / q_len describe both sides).
extern "C" int st_attn_sf1_fwd(hipStream_t stream, const void* qkv_q, const void* qkv_k, const void* qkv_v, int ld_qkv, void* Os,
                               void* Oress, int ldos, float* lses, unsigned sdrop_salt, int sdrop_thresh, float sdrop_scale,
                               const void* R, int ldr, const void* wfrag, int n_blocks, int next_blocks, float eps,
                               const float* bo, const float* g0, const float* be0, void* out0, void* xhat0, float* rstd0,
                               const float* bq, void* Qout, int ldq, const void* K, int ldk, const void* V, int ldv, void* O,
                               int ldo, void* Ores, float* lse, const int* q_off, const int* q_len, const int* k_off,
                               const int* k_len, int B, int H, int d_k, int max_q, int max_k, int q_rows_total, float scale,
                               const int* work, int n_work, const unsigned* drop_seed, unsigned drop_salt, int drop_thresh,
                               float drop_scale) {
  if (!qkv_q || !qkv_k || !qkv_v || !Os || !lses || (ld_qkv & 7) || (ldos & 7) || H > 8) return -13;
  alignas(16) char sa[128];
  if (st_attn_xs_self_args_size() > (int)sizeof(sa)) return -12;
  st_attn_xs_self_args(sa, qkv_q, qkv_k, qkv_v, ld_qkv, Os, Oress, ldos, lses, drop_seed, sdrop_salt, sdrop_thresh, sdrop_scale);
  return attn_f1_impl(stream, nullptr, 8, R, ldr, wfrag, n_blocks, next_blocks, eps, bo, g0, be0, out0, xhat0, rstd0, bq, Qout, ldq, K,
                      ldk, V, ldv, O, ldo, Ores, lse, q_off, q_len, k_off, k_len, B, H, d_k, max_q, max_k, q_rows_total, scale, work,
                      n_work, drop_seed, drop_salt, drop_thresh, drop_scale, sa);
}

// KiB of scratch st_attn_bwd wants for this shape (0: none - it then ignores split_work): the merged launch (parts = 3, O = NULL) of a
// few-queries / many-keys problem cuts every dQ item's keys over up to four workgroups.  The first 16 KiB (tickets) must be zero
// before the first launch; launches on one stream may share the scratch.
extern "C" int st_attn_bwd_split_kib(int B, int H, int d_k, int max_q, int max_k, int causal) {
  if (B <= 0 || H <= 0) return 0;
  const int xs = xsplit_for(d_k, max_q, max_k, causal);
  if (xs <= 1 || (long long)B * H > XS_TICKETS) return 0;
  return (int)((xsplit_bytes(B * H, xs, d_k) + 1023) / 1024);
}

extern "C" int st_attn_bwd(hipStream_t stream, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                           const void* O, int ldo, const void* dO, int lddo, const float* lse, float* delta,
                           void* dQ, int lddq, void* dK, int lddk, void* dV, int lddv, const int* q_off,
                           const int* q_len, const int* k_off, const int* k_len, int B, int H, int d_k, int max_q,
                           int max_k, int q_rows_total, int causal, float scale, int parts, const int* work_q,
                           int n_work_q, const int* work_k, int n_work_k, const unsigned* drop_seed,
                           unsigned drop_salt, int drop_thresh, float drop_scale, int k_prescaled, void* split_work,
                           long long split_bytes) {
  if (B <= 0 || H <= 0 || max_q <= 0 || max_k <= 0) return 0;
  int rc = check_common(d_k, ldq, ldk, ldv);
  if (rc) return rc;
  if ((O && (ldo & 7)) || (lddo & 7) || (lddq & 7) || (lddk & 7) || (lddv & 7)) return -3;
  if (B > 32767) return -4;
  AttnArgs a = {};
  a.Q = (const bf16*)Q; a.ldq = ldq; a.K = (const bf16*)K; a.ldk = ldk; a.V = (const bf16*)V; a.ldv = ldv;
  a.O = (bf16*)O; a.ldo = ldo; a.dO = (const bf16*)dO; a.lddo = lddo; a.lse = (float*)lse; a.delta = delta;
  a.dQ = (bf16*)dQ; a.lddq = lddq; a.dK = (bf16*)dK; a.lddk = lddk; a.dV = (bf16*)dV; a.lddv = lddv;
  a.q_off = q_off; a.q_len = q_len; a.k_off = k_off; a.k_len = k_len;
  a.q_rows_total = q_rows_total; a.causal = causal; set_score_scales(a, scale, k_prescaled);
  const bool drop = set_drop(a, drop_seed, drop_salt, drop_thresh, drop_scale);
  dim3 block(256);
  const bool ks2 = key_split(max_q, max_k, causal);
  const bool run_q = (parts & 1) && !(work_q && n_work_q <= 0), run_k = (parts & 2) && !(work_k && n_work_k <= 0);
  if (O == nullptr && (run_q || run_k) && !ks2 && bwd_long64(d_k, max_q, max_k, causal, drop) && lddq % 8 == 0) {
    AttnArgs ak = a;
    const int nq = run_q ? plan(a, work_q, n_work_q, B, H, max_q) : 0, nk = run_k ? plan(ak, work_k, n_work_k, B, H, max_k) : 0;
#ifdef ST_DEV_TRACE      // development builds only (ST_DEV_TRACE=1 python -m st_amd.build): per-workgroup clock stamps, tools/dev/attn_bwd64_trace.py
    if (const char* tp = getenv("ST_ATTN_TRACE_PTR")) a.Ores = (bf16*)strtoull(tp, nullptr, 0);
#endif
    return st_attn_bwd64_launch(stream, &a, &ak, nq, nk, drop ? 1 : 0);
  }
  if (run_q && run_k && O == nullptr) {
    // delta was produced together with dO (st_gemm, ST_EPI_BF16_DELTA): the two kernels are independent -> one launch
    AttnArgs ak = a;
    int nq = plan(a, work_q, n_work_q, B, H, max_q);
    const int nk = plan(ak, work_k, n_work_k, B, H, max_k);
    // the dQ items' keys over xs workgroups each (attn_bwd_dq_body) - as long as the whole launch still fits one round of the chip
    // (two workgroups per CU).  Measured, round 6 (tools/dev/xattn_bwd_parts.py): 4 utterances (16 + 96 items) 14.2 -> 10.7 us with
    // xs = 4; 32 utterances (128 + 768 items: more than a round already) 19.4 -> 27.3 us - there the launch is the sum of its items,
    // and every part pays an item's prologue again.
    int xs = split_work ? xsplit_for(d_k, max_q, max_k, causal) : 1;
    while (xs > 1 && nq * xs + nk > 2 * device_cus()) xs >>= 1;
    if (xs > 1 && nq <= XS_TICKETS) {
      if (split_bytes < xsplit_bytes(nq, xs, d_k)) return -6;
      a.xs_tickets = (unsigned*)split_work;
      a.xs_ws = (float*)((char*)split_work + XS_TICKETS * 4);
      a.xsplit = xs;
      nq *= xs;
    }
    dim3 grid(nq + nk);
#define ST_BWD(DKK, DR) \
  do { if (ks2) hipLaunchKernelGGL((attn_bwd_kernel<DKK, DR, 2>), grid, block, 0, stream, a, ak, nk); \
       else hipLaunchKernelGGL((attn_bwd_kernel<DKK, DR, 1>), grid, block, 0, stream, a, ak, nk); } while (0)
    if (d_k == 64 && !drop) ST_BWD(64, false);
    else if (d_k == 64) ST_BWD(64, true);
    else if (d_k == 128 && !drop) ST_BWD(128, false);
    else if (d_k == 128) ST_BWD(128, true);
    else if (!drop) ST_BWD(32, false);
    else ST_BWD(32, true);
#undef ST_BWD
    ST_CHECK_LAUNCH();
    return 0;
  }
  if (run_q) {
    dim3 gq(plan(a, work_q, n_work_q, B, H, max_q));
#define ST_DQ(DKK, DR) \
  do { if (ks2) hipLaunchKernelGGL((attn_bwd_dq_kernel<DKK, DR, 2>), gq, block, 0, stream, a); \
       else hipLaunchKernelGGL((attn_bwd_dq_kernel<DKK, DR, 1>), gq, block, 0, stream, a); } while (0)
    if (d_k == 64 && !drop) ST_DQ(64, false);
    else if (d_k == 64) ST_DQ(64, true);
    else if (d_k == 128 && !drop) ST_DQ(128, false);
    else if (d_k == 128) ST_DQ(128, true);
    else if (!drop) ST_DQ(32, false);
    else ST_DQ(32, true);
#undef ST_DQ
  }
  if (run_k) {
    dim3 gk(plan(a, work_k, n_work_k, B, H, max_k));
    if (d_k == 64 && !drop) hipLaunchKernelGGL((attn_bwd_dkv_kernel<64, false>), gk, block, 0, stream, a);
    else if (d_k == 64) hipLaunchKernelGGL((attn_bwd_dkv_kernel<64, true>), gk, block, 0, stream, a);
    else if (d_k == 128 && !drop) hipLaunchKernelGGL((attn_bwd_dkv_kernel<128, false>), gk, block, 0, stream, a);
    else if (d_k == 128) hipLaunchKernelGGL((attn_bwd_dkv_kernel<128, true>), gk, block, 0, stream, a);
    else if (!drop) hipLaunchKernelGGL((attn_bwd_dkv_kernel<32, false>), gk, block, 0, stream, a);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<32, true>), gk, block, 0, stream, a);
  }
  ST_CHECK_LAUNCH();
  return 0;
}

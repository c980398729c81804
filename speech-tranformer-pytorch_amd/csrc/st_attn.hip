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
// Memory pipeline: the streamed operand tiles (K/V, or Q/dO + their lse/delta) arrive by
// LDS-DMA (global_load_lds) into a 4-slot LDS ring, three tiles ahead of the MFMAs, with a
// counted s_waitcnt vmcnt(N) + one raw s_barrier per tile (the loop issues no other vector
// memory operation, so the in-order counter is exact).  Measured tile latency under load is
// ~2 us against ~0.3 us of MFMA work per tile: without the ring the kernels are latency-bound.
#include "st_common.cuh"

namespace {

#define ST_AS1 __attribute__((address_space(1)))

constexpr int TILE = 64;      // rows (keys or queries) per streamed tile
constexpr int RING = 4;       // LDS ring slots; RING-1 tiles in flight
constexpr int WG_ROWS = 128;  // rows owned by a workgroup (4 waves x 32)

__device__ __attribute__((aligned(16))) bf16 g_zero_row[8];  // zero source for rows past the end

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

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Wait until at most `younger` tiles (of IPI wave-instructions each) are still in flight.
template <int IPI>
__device__ __forceinline__ void wait_tiles(int younger) {
  if (younger >= 2) wait_vmcnt<2 * IPI>();
  else if (younger == 1) wait_vmcnt<IPI>();
  else wait_vmcnt<0>();
}

// ---- [64 x DK] bf16 tile images in LDS -----------------------------------------------------------
// Lane-linear (LDS-DMA) image: 16-byte chunk p of row r holds logical chunk p ^ swz(r).
//   SWZ_NAT: swz(r) = (r >> SH) & (CPR - 1)  -> ds_read_b128 row fragments of 16 consecutive rows are
//            conflict-free (transposing reads of the same tile are 2-way conflicted);
//   SWZ_TR : swz(r) = ((r >> 1) & 1) << 2 (DK = 64) -> the 4 rows of a ds_read_b64_tr_b16 fall into
//            4 different 64-byte bank groups (tile only read through the transposing read).
enum { SWZ_NAT = 0, SWZ_TR = 1 };

template <int DK, int MODE>
__device__ __forceinline__ int swz(int r) {
  constexpr int CPR = DK / 8;
  if (MODE == SWZ_NAT) return (r >> (DK == 64 ? 1 : 2)) & (CPR - 1);
  return DK == 64 ? ((r >> 1) & 1) << 2 : 0;
}

// Issue the LDS-DMA of rows r0 .. r0+63 (zero rows past nvalid) of a [*, ld] matrix, column slice
// starting at `base`, into `tile`.  DK/32 wave-instructions per wave.
template <int DK, int MODE>
__device__ __forceinline__ void issue_rows(bf16* tile, const bf16* __restrict__ base, int ld, int r0, int nvalid) {
  constexpr int CPR = DK / 8, RPI = 64 / CPR;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t < DK / 32; ++t) {
    const int I = t * 4 + w;
    const int r = I * RPI + i / CPR, p = i % CPR;
    const bf16* src = (r0 + r < nvalid) ? base + (size_t)(r0 + r) * ld + ((p ^ swz<DK, MODE>(r)) << 3) : g_zero_row;
    lds_dma16(src, lds_addr(tile + I * 512));
  }
}

// Row fragment: elements t16*16 + hi*8 .. +7 of tile row R (A or B operand, contraction along DK).
template <int DK, int MODE>
__device__ __forceinline__ bf16x8 rd_nat(const bf16* tile, int R, int t16) {
  const int hi = (threadIdx.x & 63) >> 5;
  return *reinterpret_cast<const bf16x8*>(tile + R * DK + (((t16 * 2 + hi) ^ swz<DK, MODE>(R)) << 3));
}

// Transposing fragment: for column d0 + (lane & 31), the 8 tile rows base+0..3 and base+8..11
// (base already includes 4*hi) - the contraction runs over the tile's ROWS.
template <int DK, int MODE>
__device__ __forceinline__ bf16x8 rd_tr(const bf16* tile, int d0, int base) {
  const int l = threadIdx.x & 63, t = l & 15;
  const int col = d0 + ((l >> 4) & 1) * 16 + 4 * (t & 3);
  const int ra = base + (t >> 2), rb = ra + 8;
  const bf16* pa = tile + ra * DK + (((col >> 3) ^ swz<DK, MODE>(ra)) << 3) + (col & 7);
  const bf16* pb = tile + rb * DK + (((col >> 3) ^ swz<DK, MODE>(rb)) << 3) + (col & 7);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ST_LDS bf16x4*)(pa));
  const bf16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ST_LDS bf16x4*)(pb));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = up[0]; f[5] = up[1]; f[6] = up[2]; f[7] = up[3];
  return f;
}

// Store a transposed accumulator tile (lane = row, registers = DK columns) as coalesced rows:
// through a wave-private [32][DK] LDS patch so HBM sees whole DK*2-byte row segments instead of
// 64 scattered 8-byte writes per instruction.  `patch` is this wave's private 32*DK elements.
template <int DK>
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
      for (int e = 0; e < 4; ++e) v[e] = (bf16)(acc[d][4 * g + e] * mul);
      const int col = d * 32 + 8 * g + 4 * hi;
      *reinterpret_cast<bf16x4*>(patch + r * DK + (((col >> 3) ^ (r & (CPR - 1))) << 3) + (col & 7)) = v;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int p = 0; p < 32 * CPR / 64; ++p) {
    const int id = p * 64 + l, rr = id / CPR, c = id % CPR;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(patch + rr * DK + ((c ^ (rr & (CPR - 1))) << 3));
    if (rr < nvalid_rows) *reinterpret_cast<bf16x8*>(gbase + (size_t)(row0 + rr) * ld + c * 8) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Forward.  grid = (ceil(max_q / 128), H, B).  Each wave owns 32 query rows (lane & 31).
// ---------------------------------------------------------------------------------------------
template <int DK>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  constexpr int NT = DK / 16;   // k-steps of the QK^T contraction
  constexpr int ND = DK / 32;   // 32-wide output column tiles
  constexpr int TE = TILE * DK; // elements per tile
  constexpr int IPI = 2 * (DK / 32);
  __shared__ __attribute__((aligned(1024))) bf16 smem[RING * 2 * TE];

  const int b = blockIdx.z, h = blockIdx.y;
  const int lq = a.q_len[b], lk = a.k_len[b];
  const int q0 = blockIdx.x * WG_ROWS;
  if (q0 >= lq) return;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, hi = l >> 5;
  const int q = q0 + wave * 32 + (l & 31);
  const bool q_ok = q < lq;
  const size_t qrow = (size_t)a.q_off[b] + q;
  const float c2 = a.scale * 1.4426950408889634f;  // scores -> log2 domain

  const int k_hi = a.causal ? min(lk, q0 + WG_ROWS) : lk;   // keys this workgroup can see
  const int ntiles = (k_hi + TILE - 1) / TILE;
  const bf16* kbase = a.K + (size_t)a.k_off[b] * a.ldk + h * DK;
  const bf16* vbase = a.V + (size_t)a.k_off[b] * a.ldv + h * DK;

  bf16x8 qf[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) qf[t] = gload8(a.Q + qrow * a.ldq + h * DK + t * 16 + hi * 8, q_ok);
  // the Q fragments are in registers before the counted waits below start counting ring tiles
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int t = 0; t < NT; ++t) touch(qf[t]);
#pragma unroll
  for (int t = 0; t < RING - 1; ++t)
    if (t < ntiles) {
      issue_rows<DK, SWZ_NAT>(smem + t * 2 * TE, kbase, a.ldk, t * TILE, lk);
      issue_rows<DK, SWZ_TR>(smem + t * 2 * TE + TE, vbase, a.ldv, t * TILE, lk);
    }

  f32x16 o[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) o[d] = zero16();
  float m = -INFINITY, lsum = 0.f;

  for (int it = 0; it < ntiles; ++it) {
    wait_tiles<IPI>(min(RING - 2, ntiles - 1 - it));
    __builtin_amdgcn_s_barrier();   // tile `it` landed for every wave; slot of tile it-1 is free
    if (it + RING - 1 < ntiles) {
      const int nt = it + RING - 1, slot = nt % RING;
      issue_rows<DK, SWZ_NAT>(smem + slot * 2 * TE, kbase, a.ldk, nt * TILE, lk);
      issue_rows<DK, SWZ_TR>(smem + slot * 2 * TE + TE, vbase, a.ldv, nt * TILE, lk);
    }
    const bf16* ks = smem + (it % RING) * 2 * TE;
    const bf16* vs = ks + TE;
    const int kt = it * TILE;

    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      s[kb] = zero16();
#pragma unroll
      for (int t = 0; t < NT; ++t) s[kb] = mfma32(rd_nat<DK, SWZ_NAT>(ks, kb * 32 + (l & 31), t), qf[t], s[kb]);
    }
    // The softmax arithmetic is the VALU critical path (the MFMAs of a tile take ~512 cycles, a
    // naive per-element mask/scale/exp chain ~3000): masks only on tiles that cross a sequence end or
    // the diagonal (wave-uniform test), scale folded into one fma feeding the raw v_exp_f32.
    const bool full = (kt + TILE <= lk) && (!a.causal || kt + TILE - 1 <= q0 + wave * 32);
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
    const float m_new = fmaxf(m, mx * c2);           // log2 domain
    // m_new is finite from the first tile on (key 0 is visible to every query); guard anyway
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m - m_use);
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][r], c2, -m_use));
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
        for (int d = 0; d < ND; ++d) o[d] = mfma32(rd_tr<DK, SWZ_TR>(vs, d * 32, base), pf, o[d]);
      }
  }

  const float ltot = lsum + wave_xor32(lsum);
  const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
  if (q_ok && hi == 0 && a.lse) a.lse[(size_t)h * a.q_rows_total + qrow] = m + log2f(ltot);
  __builtin_amdgcn_s_barrier();   // every wave is done with the ring: reuse it for the output patches
  store_rows<DK>(smem + wave * 32 * DK, o, inv, a.O + (size_t)a.q_off[b] * a.ldo + h * DK, a.ldo, q0 + wave * 32,
                 min(32, lq - (q0 + wave * 32)));
}

// ---------------------------------------------------------------------------------------------
// Backward, part 1: dQ (and delta = rowsum(dO * O)).  Same decomposition as the forward.
//   P^T = exp2(S^T c2 - lse),  dP^T = V dO^T,  dS^T = P^T (dP^T - delta),  dQ^T += K^T dS^T
// ---------------------------------------------------------------------------------------------
template <int DK>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs a) {
  constexpr int NT = DK / 16, ND = DK / 32, TE = TILE * DK, IPI = 2 * (DK / 32);
  __shared__ __attribute__((aligned(1024))) bf16 smem[RING * 2 * TE];

  const int b = blockIdx.z, h = blockIdx.y;
  const int lq = a.q_len[b], lk = a.k_len[b];
  const int q0 = blockIdx.x * WG_ROWS;
  if (q0 >= lq) return;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, hi = l >> 5;
  const int q = q0 + wave * 32 + (l & 31);
  const bool q_ok = q < lq;
  const size_t qrow = (size_t)a.q_off[b] + q;
  const float c2 = a.scale * 1.4426950408889634f;

  const int k_hi = a.causal ? min(lk, q0 + WG_ROWS) : lk;
  const int ntiles = (k_hi + TILE - 1) / TILE;
  const bf16* kbase = a.K + (size_t)a.k_off[b] * a.ldk + h * DK;
  const bf16* vbase = a.V + (size_t)a.k_off[b] * a.ldv + h * DK;

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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // prologue loads / stores retired before counting tiles
#pragma unroll
  for (int t = 0; t < NT; ++t) { touch(qf[t]); touch(dof[t]); }
  float lse_r = lse;
  touch(lse_r);
  touch(dl);
#pragma unroll
  for (int t = 0; t < RING - 1; ++t)
    if (t < ntiles) {
      issue_rows<DK, SWZ_NAT>(smem + t * 2 * TE, kbase, a.ldk, t * TILE, lk);
      issue_rows<DK, SWZ_NAT>(smem + t * 2 * TE + TE, vbase, a.ldv, t * TILE, lk);
    }

  f32x16 dq[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) dq[d] = zero16();

  for (int it = 0; it < ntiles; ++it) {
    wait_tiles<IPI>(min(RING - 2, ntiles - 1 - it));
    __builtin_amdgcn_s_barrier();
    if (it + RING - 1 < ntiles) {
      const int nt = it + RING - 1, slot = nt % RING;
      issue_rows<DK, SWZ_NAT>(smem + slot * 2 * TE, kbase, a.ldk, nt * TILE, lk);
      issue_rows<DK, SWZ_NAT>(smem + slot * 2 * TE + TE, vbase, a.ldv, nt * TILE, lk);
    }
    const bf16* ks = smem + (it % RING) * 2 * TE;
    const bf16* vs = ks + TE;
    const int kt = it * TILE;
    const bool full = (kt + TILE <= lk) && (!a.causal || kt + TILE - 1 <= q0 + wave * 32);   // no masks needed
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        s = mfma32(rd_nat<DK, SWZ_NAT>(ks, kb * 32 + (l & 31), t), qf[t], s);
        dp = mfma32(rd_nat<DK, SWZ_NAT>(vs, kb * 32 + (l & 31), t), dof[t], dp);
      }
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lse_r)) * (dp[r] - dl);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt + kb * 32 + acc_row(r, hi);
          const bool dead = key >= lk || (a.causal && key > q);
          const float p = dead ? 0.f : __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lse_r));
          s[r] = p * (dp[r] - dl);
        }
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const bf16x8 dsf = pack_acc8(s, 8 * hf);
        const int base = kb * 32 + 16 * hf + 4 * hi;
#pragma unroll
        for (int d = 0; d < ND; ++d) dq[d] = mfma32(rd_tr<DK, SWZ_NAT>(ks, d * 32, base), dsf, dq[d]);
      }
    }
  }
  __builtin_amdgcn_s_barrier();
  store_rows<DK>(smem + wave * 32 * DK, dq, a.scale, a.dQ + (size_t)a.q_off[b] * a.lddq + h * DK, a.lddq,
                 q0 + wave * 32, min(32, lq - (q0 + wave * 32)));
}

// ---------------------------------------------------------------------------------------------
// Backward, part 2: dK, dV.  grid = (ceil(max_k / 128), H, B); each wave owns 32 keys (lane & 31)
// and loops over 64-query tiles streamed through the LDS ring (Q rows, dO rows, and a per-wave
// copy of the tile's 64 lse + 64 delta values).
//   S = Q K^T (lane = key, registers = queries),  P = exp2(S c2 - lse[q])
//   dV^T += dO^T P,   dP = dO V^T,   dS = P (dP - delta[q]),   dK^T += Q^T dS
// ---------------------------------------------------------------------------------------------
template <int DK>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnArgs a) {
  constexpr int NT = DK / 16, ND = DK / 32, TE = TILE * DK, IPI = 2 * (DK / 32) + 2;
  constexpr int SLOT = 2 * TE + 4 * 256;   // Q tile, dO tile, 4 x 512 B per-wave statistics (bf16 elements)
  __shared__ __attribute__((aligned(1024))) bf16 smem[RING * SLOT];

  const int b = blockIdx.z, h = blockIdx.y;
  const int lq = a.q_len[b], lk = a.k_len[b];
  const int k0 = blockIdx.x * WG_ROWS;
  if (k0 >= lk) return;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, hi = l >> 5;
  const int wu = __builtin_amdgcn_readfirstlane(wave);
  const int key = k0 + wave * 32 + (l & 31);
  const bool k_ok = key < lk;
  const size_t krow = (size_t)a.k_off[b] + key;
  const float c2 = a.scale * 1.4426950408889634f;

  const bf16* qbase = a.Q + (size_t)a.q_off[b] * a.ldq + h * DK;
  const bf16* dobase = a.dO + (size_t)a.q_off[b] * a.lddo + h * DK;
  const float* lse = a.lse + (size_t)h * a.q_rows_total + a.q_off[b];
  const float* delta = a.delta + (size_t)h * a.q_rows_total + a.q_off[b];
  const int q_begin = a.causal ? (k0 / TILE) * TILE : 0;  // queries before the first key see none of them
  const int ntiles = (lq - q_begin + TILE - 1) / TILE;

  bf16x8 kf[NT], vf[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = h * DK + t * 16 + hi * 8;
    kf[t] = gload8(a.K + krow * a.ldk + col, k_ok);
    vf[t] = gload8(a.V + krow * a.ldv + col, k_ok);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int t = 0; t < NT; ++t) { touch(kf[t]); touch(vf[t]); }

  // one ring item = 64 Q rows, 64 dO rows and (per wave) lse[64], delta[64]: lane i fetches one
  // float of each (4-byte LDS-DMA; queries past lq read a zero and are masked by index below)
  auto issue_tile = [&](int slot, int qt) {
    bf16* base = smem + slot * SLOT;
    issue_rows<DK, SWZ_NAT>(base, qbase, a.ldq, qt, lq);
    issue_rows<DK, SWZ_NAT>(base + TE, dobase, a.lddo, qt, lq);
    const bool in = qt + l < lq;
    const void* sl = in ? (const void*)(lse + qt + l) : (const void*)g_zero_row;
    const void* sd = in ? (const void*)(delta + qt + l) : (const void*)g_zero_row;
    bf16* st = base + 2 * TE + wu * 256;
    lds_dma4(sl, lds_addr(st));
    lds_dma4(sd, lds_addr(st + 128));
  };
#pragma unroll
  for (int t = 0; t < RING - 1; ++t)
    if (t < ntiles) issue_tile(t, q_begin + t * TILE);

  f32x16 dk[ND], dv[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) { dk[d] = zero16(); dv[d] = zero16(); }

  for (int it = 0; it < ntiles; ++it) {
    wait_tiles<IPI>(min(RING - 2, ntiles - 1 - it));
    __builtin_amdgcn_s_barrier();
    if (it + RING - 1 < ntiles) issue_tile((it + RING - 1) % RING, q_begin + (it + RING - 1) * TILE);
    const bf16* qs = smem + (it % RING) * SLOT;
    const bf16* dos = qs + TE;
    const float* stat = reinterpret_cast<const float*>(qs + 2 * TE + wave * 256);   // [0..63] lse, [64..127] delta
    const int qt = q_begin + it * TILE;
    // wave-uniform: every (query, key) pair of this tile x this wave's 32 keys is unmasked
    const bool full = (qt + TILE <= lq) && (k0 + wave * 32 + 32 <= lk) && (!a.causal || k0 + wave * 32 + 31 <= qt);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        s = mfma32(rd_nat<DK, SWZ_NAT>(qs, qb * 32 + (l & 31), t), kf[t], s);
        dp = mfma32(rd_nat<DK, SWZ_NAT>(dos, qb * 32 + (l & 31), t), vf[t], dp);
      }
      f32x16 p;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ql = qb * 32 + 8 * g + 4 * hi;
        const f32x4 ls = *reinterpret_cast<const f32x4*>(stat + ql);
        const f32x4 dl = *reinterpret_cast<const f32x4*>(stat + 64 + ql);
        if (full) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -ls[e]));
            p[r] = pv;
            s[r] = pv * (dp[r] - dl[e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            const int qq = qt + ql + e;
            const bool dead = !k_ok || qq >= lq || (a.causal && key > qq);
            const float pv = dead ? 0.f : __builtin_amdgcn_exp2f(fmaf(s[r], c2, -ls[e]));
            p[r] = pv;
            s[r] = pv * (dp[r] - dl[e]);
          }
        }
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const bf16x8 pf = pack_acc8(p, 8 * hf);
        const bf16x8 dsf = pack_acc8(s, 8 * hf);
        const int base = qb * 32 + 16 * hf + 4 * hi;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
          dv[d] = mfma32(rd_tr<DK, SWZ_NAT>(dos, d * 32, base), pf, dv[d]);
          dk[d] = mfma32(rd_tr<DK, SWZ_NAT>(qs, d * 32, base), dsf, dk[d]);
        }
      }
    }
  }
  __builtin_amdgcn_s_barrier();
  const int nrows = min(32, lk - (k0 + wave * 32));
  store_rows<DK>(smem + wave * 64 * DK, dk, a.scale, a.dK + (size_t)a.k_off[b] * a.lddk + h * DK, a.lddk,
                 k0 + wave * 32, nrows);
  store_rows<DK>(smem + wave * 64 * DK + 32 * DK, dv, 1.f, a.dV + (size_t)a.k_off[b] * a.lddv + h * DK, a.lddv,
                 k0 + wave * 32, nrows);
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
  if (ldo & 7) return -3;
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
  if ((ldo & 7) || (lddo & 7) || (lddq & 7) || (lddk & 7) || (lddv & 7)) return -3;
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

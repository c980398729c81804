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
#include "st_common.cuh"

namespace {

constexpr int TILE = 64;      // rows (keys or queries) per streamed tile
constexpr int WG_ROWS = 128;  // rows owned by a workgroup (4 waves x 32)

struct AttnArgs {
  const bf16* Q; int ldq;
  const bf16* K; int ldk;
  const bf16* V; int ldv;
  bf16* O; int ldo;               // forward: output; backward: forward output (for delta)
  bf16* Ores;                     // forward, optional: bf16(O_fp32 - bf16(O_fp32)), same layout as O - with it the
                                  // backward's delta = rowsum(dO * (O + Ores)) sees O to ~16 mantissa bits
  const bf16* dO; int lddo;
  bf16* dQ; int lddq;
  bf16* dK; int lddk;
  bf16* dV; int lddv;
  float* lse;                     // [H][q_rows_total], log2 domain: m + log2(l)
  float* delta;                   // [H][q_rows_total]
  const int* q_off; const int* q_len;
  const int* k_off; const int* k_len;
  const int* work;                // (b << 16) | tile, sorted by decreasing cost; or null
  int tiles_max;                  // without a work list: tiles per utterance enumerated
  int H;
  int q_rows_total;
  int causal;
  int psplit;                     // forward: P enters the P V product as two bf16 terms (hi + lo): the context, and with it
                                  // the backward's delta = rowsum(dO * O), is then consistent with the fp32 P the backward
                                  // recomputes - sum_k dS(q, k) = 0 to ~2^-16 instead of ~2^-9 (matters where the keys
                                  // are nearly identical and dQ / dK are differences of almost equal terms); small-Lq only
  float scale;                    // 1/sqrt(d_k)
  DropArgs drop;                  // attention-probability dropout (Attention.py:89), training mode only
};

// blockIdx.x -> (utterance, head, tile)
__device__ __forceinline__ void decode_item(const AttnArgs& a, int bid, int& b, int& h, int& tile) {
  const int idx = bid / a.H;
  h = bid % a.H;
  if (a.work) {
    const int w = a.work[idx];
    b = w >> 16;
    tile = w & 0xffff;
  } else {
    b = idx / a.tiles_max;
    tile = idx % a.tiles_max;
  }
}

// ---- streamed [64 x DK] tiles ----------------------------------------------------------------------
// LDS image: natural rows, stride DK + 8 elements (144 B / 80 B): ds_read_b128 row fragments over 16 rows
// and the 4-row groups of ds_read_b64_tr_b16 are both bank-conflict free (DK = 64; 272 B for DK = 128 shifts 16 B per row likewise).
// DK = 128 (the reference's config/character.yaml: d_model 512, 4 heads) needs > 256 registers per lane: one workgroup per CU.
template <int DK, int ROWS = TILE> struct TileGeo {
  static constexpr int STR = DK + 8, E = ROWS * STR, CPR = DK / 8, CH = ROWS * CPR / 256;   // CH chunks per thread
};

template <int DK, int ROWS = TILE>
struct Stage {
  using G = TileGeo<DK, ROWS>;
  bf16x8 v[G::CH];
  // chunk id = tid + p*256 -> tile row id / CPR, 16-byte chunk id % CPR
  static __device__ __forceinline__ void offsets(uint32_t (&off)[G::CH], int ld) {
#pragma unroll
    for (int p = 0; p < G::CH; ++p) {
      const int id = threadIdx.x + p * 256;
      off[p] = ((uint32_t)(id / G::CPR) * (uint32_t)ld + (id % G::CPR) * 8) * 2u;
    }
  }
  // rows r0 .. r0+ROWS-1 of the utterance's column slice `base`; rows >= nvalid read row nvalid-1
  __device__ __forceinline__ void load(const uint32_t (&off)[G::CH], const bf16* __restrict__ base, int ld, int r0,
                                       int nvalid) {
    if (r0 + ROWS <= nvalid) {
      const char* tb = reinterpret_cast<const char*>(base + (size_t)r0 * ld);
#pragma unroll
      for (int p = 0; p < G::CH; ++p) v[p] = *reinterpret_cast<const bf16x8*>(tb + off[p]);
    } else {
#pragma unroll
      for (int p = 0; p < G::CH; ++p) {
        const int id = threadIdx.x + p * 256;
        const int row = min(r0 + id / G::CPR, nvalid - 1);
        v[p] = *reinterpret_cast<const bf16x8*>(base + (size_t)row * ld + (id % G::CPR) * 8);
      }
    }
  }
  __device__ __forceinline__ void store(bf16* tile) const {
#pragma unroll
    for (int p = 0; p < G::CH; ++p) {
      const int id = threadIdx.x + p * 256;
      *reinterpret_cast<bf16x8*>(tile + (id / G::CPR) * G::STR + (id % G::CPR) * 8) = v[p];
    }
  }
};

// Row fragment: elements t16*16 + hi*8 .. +7 of tile row R (A or B operand, contraction along DK).
template <int DK>
__device__ __forceinline__ bf16x8 rd_nat(const bf16* tile, int R, int t16) {
  const int hi = (threadIdx.x & 63) >> 5;
  return frag_nat(tile, TileGeo<DK>::STR, R, t16 * 16 + hi * 8);
}
// Transposing fragment: for column d0 + (lane & 31), the 8 tile rows base+0..3 and base+8..11
// (base already includes 4*hi) - the contraction runs over the tile's ROWS.
template <int DK>
__device__ __forceinline__ bf16x8 rd_tr(const bf16* tile, int d0, int base) {
  return frag_tr(tile, TileGeo<DK>::STR, d0, base, base + 8);
}

// The software pipeline shared by the three kernels.  load(set, tile) fills register set `set`,
// store(set) writes it to LDS buffer `set`, compute(buf, tile) consumes LDS buffer `buf`.
// Steady state has no conditionals, so the compiler's s_waitcnt vmcnt() stays counted: the loads of
// tile it+2 remain in flight across the LDS store of tile it+1.
template <typename L, typename S, typename C>
__device__ __forceinline__ void stream_tiles(int ntiles, L load, S store, C compute) {
  if (ntiles <= 0) return;   // (workgroup-uniform) nothing visible: the accumulators stay zero
  load(0, 0);
  if (ntiles > 1) load(1, 1);
  store(0);
  __syncthreads();
  int it = 0;
  for (; it + 3 < ntiles; it += 2) {
    load(0, it + 2);
    compute(0, it);
    store(1);
    __syncthreads();
    load(1, it + 3);
    compute(1, it + 1);
    store(0);
    __syncthreads();
  }
  // tail: 1..3 tiles left; tile `it` is in LDS buffer 0, tile it+1 (if any) in register set 1
  if (it + 2 < ntiles) load(0, it + 2);
  compute(0, it);
  if (it + 1 < ntiles) {
    store(1);
    __syncthreads();
    compute(1, it + 1);
    if (it + 2 < ntiles) {
      store(0);
      __syncthreads();
      compute(0, it + 2);
    }
  }
  __syncthreads();   // the epilogue reuses the tile buffers
}

// Store a transposed accumulator tile (lane = row, registers = DK columns) as coalesced rows:
// through a wave-private [32][DK] LDS patch so HBM sees whole DK*2-byte row segments instead of
// 64 scattered 8-byte writes per instruction.  `patch` is this wave's private 32*DK elements.
template <int DK, bool RESID = false>
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
      for (int e = 0; e < 4; ++e) {
        const float x = acc[d][4 * g + e] * mul;
        v[e] = RESID ? (bf16)(x - (float)(bf16)x) : (bf16)x;      // RESID: what the bf16 rounding of x dropped
      }
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

// The same for a value AND what its bf16 rounding dropped (O, Ores), in ONE pass: both images are written to two
// patches, one LDS wait, both are read back and stored - half the dependent LDS round trips of two store_rows calls
// (the attention epilogue is a latency tail: every wave of the workgroup is in it at the same time).
template <int DK>
__device__ __forceinline__ void store_rows_pair(bf16* patch_hi, bf16* patch_lo, const f32x16* acc, float mul, bf16* g_hi,
                                                bf16* g_lo, int ld, int row0, int nvalid_rows) {
  constexpr int ND = DK / 32, CPR = DK / 8;
  const int l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 vh, vl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = acc[d][4 * g + e] * mul;
        vh[e] = (bf16)x;
        vl[e] = (bf16)(x - (float)vh[e]);
      }
      const int col = d * 32 + 8 * g + 4 * hi;
      const int at = r * DK + (((col >> 3) ^ (r & (CPR - 1))) << 3) + (col & 7);
      *reinterpret_cast<bf16x4*>(patch_hi + at) = vh;
      *reinterpret_cast<bf16x4*>(patch_lo + at) = vl;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int p = 0; p < 32 * CPR / 64; ++p) {
    const int id = p * 64 + l, rr = id / CPR, c = id % CPR;
    const int at = rr * DK + ((c ^ (rr & (CPR - 1))) << 3);
    const bf16x8 vh = *reinterpret_cast<const bf16x8*>(patch_hi + at);
    const bf16x8 vl = *reinterpret_cast<const bf16x8*>(patch_lo + at);
    if (rr < nvalid_rows) {
      *reinterpret_cast<bf16x8*>(g_hi + (size_t)(row0 + rr) * ld + c * 8) = vh;
      *reinterpret_cast<bf16x8*>(g_lo + (size_t)(row0 + rr) * ld + c * 8) = vl;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Forward.  Each wave owns 32 query rows (lane & 31); key / value tiles are streamed.
// ---------------------------------------------------------------------------------------------
// Keep decisions of one lane's 16 accumulator registers: register r <-> pair (fixed, var0 + acc_row(r, hi)).
// FIXED_IS_Q: the lane's own index is the query (forward / dQ: registers run over keys), else it is the key.
template <bool FIXED_IS_Q>
__device__ __forceinline__ void keep16(const Drop& dr, int bh, int fixed, int var0, int hi, bool (&keep)[16]) {
  // Lanes l and l ^ 1 hold neighbouring fixed indices (2f, 2f + 1: tile origins are even) and therefore need the SAME
  // eight hashes (a 2 x 2 block serves both).  Each computes four - the even lane register groups g = 0, 1, the odd
  // lane g = 2, 3 - and fetches the other four from its neighbour with a DPP move (the hash costs two quarter-rate
  // v_mul_lo_u32; this halves what dropout adds to the VALU-bound softmax).
  const int par = threadIdx.x & 1, fx = fixed & 1;
  uint32_t own[4], nb[4];
#pragma unroll
  for (int gi = 0; gi < 2; ++gi)
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const int v = var0 + 8 * (2 * par + gi) + 4 * hi + 2 * hb;
      own[gi * 2 + hb] = FIXED_IS_Q ? dr.bits(drop_counter_qk(bh, fixed, v)) : dr.bits(drop_counter_qk(bh, v, fixed));
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) nb[j] = (uint32_t)__builtin_amdgcn_mov_dpp((int)own[j], 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const int j = (g & 1) * 2 + hb;
      uint32_t bits = ((g >> 1) == par) ? own[j] : nb[j];
      // byte of (q, k) inside its block = 2 * (q & 1) + (k & 1); this lane's two elements differ in the variable index
      bits >>= FIXED_IS_Q ? 16 * fx : 8 * fx;
#pragma unroll
      for (int e = 0; e < 2; ++e)
        keep[4 * g + 2 * hb + e] = (int)((bits >> (FIXED_IS_Q ? 8 * e : 16 * e)) & 0xffu) >= dr.thresh;
    }
}

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
  const float c2 = a.scale * 1.4426950408889634f;  // scores -> log2 domain
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
  const float c2 = a.scale * 1.4426950408889634f;
  const Drop dr = make_drop(a.drop);
  const int bh = b * a.H + h;

  const int k_hi = a.causal ? min(lk, q0 + QROWS) : lk;
  const int ntiles = (k_hi + ROWS - 1) / ROWS;
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
    if (q_ok && hi == 0 && kp == 0) a.delta[(size_t)h * a.q_rows_total + qrow] = dl;
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
    sk[set].load(offk, kbase, a.ldk, it * ROWS, lk);
    sv[set].load(offv, vbase, a.ldv, it * ROWS, lk);
  };
  auto store = [&](int set) {
    sk[set].store(smem + set * 2 * G::E);
    sv[set].store(smem + set * 2 * G::E + G::E);
  };
  auto compute = [&](int buf, int it) {
    const bf16* ks = smem + buf * 2 * G::E + kp * TILE * G::STR;
    const bf16* vs = ks + G::E;
    const int kt = it * ROWS + kp * TILE;
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
  stream_tiles(ntiles, load, store, compute);
  if (KS > 1) {   // dQ of the two key halves: waves 2,3 -> LDS -> waves 0,1
    float* xch = reinterpret_cast<float*>(smem + 2 * 32 * DK) + qw * (ND * 16) * 64 + l;
    if (kp == 1) {
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(d * 16 + r) * 64] = dq[d][r];
    }
    __syncthreads();
    if (kp == 1) return;
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[d][r] += xch[(d * 16 + r) * 64];
  }
  store_rows<DK>(smem + qw * 32 * DK, dq, a.scale, a.dQ + (size_t)a.q_off[b] * a.lddq + h * DK, a.lddq,
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
  const bool k_ok = key < lk;
  const size_t krow = (size_t)a.k_off[b] + min(key, lk - 1);
  const float c2 = a.scale * 1.4426950408889634f;
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
    so[set].load(offo, dobase, a.lddo, qt, lq);
    sst[set] = statsrc[min(qt + (int)(threadIdx.x & 63), lq - 1)];
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
    // wave-uniform: every (query, key) pair of this tile x this wave's 32 keys is unmasked
    const bool full = (qt + TILE <= lq) && (k0 + wave * 32 + 32 <= lk) && (!a.causal || k0 + wave * 32 + 31 <= qt);
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
          if (!k_ok || qq >= lq || (a.causal && key > qq)) s[r] = -INFINITY;
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

// grid size and enumeration mode for one family of workgroups
int plan(AttnArgs& a, const int* work, int n_work, int B, int H, int max_rows) {
  a.work = work;
  a.H = H;
  a.tiles_max = (max_rows + WG_ROWS - 1) / WG_ROWS;
  return (work ? n_work : B * a.tiles_max) * H;
}

}  // namespace

extern "C" int st_attn_fwd(hipStream_t stream, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                           void* O, int ldo, void* Ores, float* lse, const int* q_off, const int* q_len, const int* k_off,
                           const int* k_len, int B, int H, int d_k, int max_q, int max_k, int q_rows_total, int causal,
                           float scale, const int* work, int n_work, const unsigned* drop_seed, unsigned drop_salt,
                           int drop_thresh, float drop_scale) {
  if (B <= 0 || H <= 0 || max_q <= 0 || (work && n_work <= 0)) return 0;
  int rc = check_common(d_k, ldq, ldk, ldv);
  if (rc) return rc;
  if (ldo & 7) return -3;
  if (B > 32767) return -4;
  AttnArgs a = {};
  a.Q = (const bf16*)Q; a.ldq = ldq; a.K = (const bf16*)K; a.ldk = ldk; a.V = (const bf16*)V; a.ldv = ldv;
  a.O = (bf16*)O; a.ldo = ldo; a.Ores = (bf16*)Ores; a.lse = lse; a.q_off = q_off; a.q_len = q_len; a.k_off = k_off; a.k_len = k_len;
  a.q_rows_total = q_rows_total; a.causal = causal; a.scale = scale;
  const bool drop = set_drop(a, drop_seed, drop_salt, drop_thresh, drop_scale);
  a.psplit = (Ores != nullptr && max_q <= 64) ? 1 : 0;     // the decoder's attentions, when a backward will follow
  dim3 grid(plan(a, work, n_work, B, H, max_q)), block(256);
  const bool ks2 = key_split(max_q, max_k, causal);   // one 64-row query tile per utterance == the 128-row tile 0
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

extern "C" int st_attn_bwd(hipStream_t stream, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                           const void* O, int ldo, const void* dO, int lddo, const float* lse, float* delta,
                           void* dQ, int lddq, void* dK, int lddk, void* dV, int lddv, const int* q_off,
                           const int* q_len, const int* k_off, const int* k_len, int B, int H, int d_k, int max_q,
                           int max_k, int q_rows_total, int causal, float scale, int parts, const int* work_q,
                           int n_work_q, const int* work_k, int n_work_k, const unsigned* drop_seed,
                           unsigned drop_salt, int drop_thresh, float drop_scale) {
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
  a.q_rows_total = q_rows_total; a.causal = causal; a.scale = scale;
  const bool drop = set_drop(a, drop_seed, drop_salt, drop_thresh, drop_scale);
  dim3 block(256);
  const bool ks2 = key_split(max_q, max_k, causal);
  const bool run_q = (parts & 1) && !(work_q && n_work_q <= 0), run_k = (parts & 2) && !(work_k && n_work_k <= 0);
  if (run_q && run_k && O == nullptr) {
    // delta was produced together with dO (st_gemm, ST_EPI_BF16_DELTA): the two kernels are independent -> one launch
    AttnArgs ak = a;
    const int nq = plan(a, work_q, n_work_q, B, H, max_q), nk = plan(ak, work_k, n_work_k, B, H, max_k);
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

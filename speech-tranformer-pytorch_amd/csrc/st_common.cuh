// Shared device helpers for the gfx950 (CDNA4 / MI355X) speech-transformer kernels.
//
// Conventions used by every kernel in this directory
//   * wavefront = 64 lanes; workgroups are 256 threads = 4 waves (one per SIMD).
//   * matrix core op: v_mfma_f32_32x32x16_bf16.  For D = A*B with A[32 x 16], B[16 x 32]:
//       A operand, lane l : row  i = l & 31, k = (l >> 5) * 8 + 0..7   (8 bf16 = 4 VGPRs)
//       B operand, lane l : col  j = l & 31, k = (l >> 5) * 8 + 0..7
//       C/D,       lane l : col  j = l & 31, row i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), r = 0..15
//     Both operands are therefore "row fragments": 8 consecutive contraction
//     elements of one row.  We always put the operand whose rows we want to be
//     LANE-LOCAL in the output on the B side, so that per-row statistics
//     (softmax max/sum, LayerNorm mean/var) and 8-byte row stores need no
//     cross-lane traffic beyond one exchange with lane ^ 32.
//   * an operand stored contraction-major in LDS ([c][row], row contiguous) is read
//     with ds_read_b64_tr_b16 (frag_tr below), so no tile is ever transposed in
//     registers or re-written transposed to HBM.
#pragma once

// ---- last-arriver merges (st_grad_norm, the split row chains, st_gemm_splitk, the beam step) ---------------------------------
// Every contributing workgroup publishes its partial with system-scope (write-through) stores, waits until the memory system
// has acknowledged them (s_waitcnt 0), then draws a ticket with an agent-scope atomic; the workgroup that draws the last one
// reads the partials with agent-scope loads (which miss the non-coherent L2 lines) and merges.  This is the sequence gfx950's
// agent-scope release / acquire lower to, minus the whole-L2 write-back and invalidate that the language-level fences add
// (measured: +60-80 us per merge launch, tools/dev/merge_probe.hip) - the HIP memory model does not promise it, the hardware
// does (MI355X_MICROARCH.md, "cross-XCD visibility"), and tests/test_kernels_gpu.py::test_last_arriver_merges_under_uneven_load
// checks every merge under load against a fenced reference.  ONE switch turns all of them into the language-level form:
// build with ST_MERGE_FENCE=1 in the environment (st_amd/build.py adds -DST_MERGE_FENCE=1; a separately stamped library).
#ifdef ST_MERGE_FENCE
#define ST_PUBLISH_FENCE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define ST_MERGER_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define ST_PUBLISH_FENCE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront")
#define ST_MERGER_FENCE() ((void)0)
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define ST_WAVE 64
#define ST_LDS __attribute__((address_space(3)))

#define ST_CHECK_LAUNCH()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

__device__ __forceinline__ bf16x8 zero_bf8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16)0.f;
  return z;
}

// Row of accumulator register r for the lane half `hi` (C/D layout above).
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// 16-byte global load of 8 bf16 (predicated, zero fill).
__device__ __forceinline__ bf16x8 gload8(const bf16* p, bool ok) {
  if (ok) return *reinterpret_cast<const bf16x8*>(p);
  return zero_bf8();
}

// Natural row fragment from an LDS tile stored [row][c] with `stride` elements
// per row: 8 bf16 at (row, c0 .. c0+7).  One ds_read_b128.
__device__ __forceinline__ bf16x8 frag_nat(const bf16* tile, int stride, int row, int c0) {
  return *reinterpret_cast<const bf16x8*>(tile + row * stride + c0);
}

// Transposing fragment read from an LDS tile stored contraction-major
// [c][row] (`stride` elements per c-row).  Returns, for this lane's row
// `row0 + (lane & 31)`, the 8 elements c = ca + 0..3 and cb + 0..3.
// ds_read_b64_tr_b16 semantics (per 16-lane group): lane t supplies the address
// of 4 consecutive bf16 of row (t >> 2), chunk (t & 3); lane t receives, for
// j = 0..3, element (t & 3) of the chunk addressed by lane 4 * j + (t >> 2),
// i.e. column t of the 4 x 16 block.  (Verified on hardware by tests/test_probe_gpu.py.)
__device__ __forceinline__ bf16x8 frag_tr(const bf16* tile, int stride, int row0, int ca, int cb) {
  const int l = threadIdx.x & 63;
  const int t = l & 15;
  const int colblk = ((l >> 4) & 1) * 16;
  const bf16* pa = tile + (ca + (t >> 2)) * stride + row0 + colblk + 4 * (t & 3);
  const bf16* pb = tile + (cb + (t >> 2)) * stride + row0 + colblk + 4 * (t & 3);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ST_LDS bf16x4*)(pa));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ST_LDS bf16x4*)(pb));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

// Pack 8 consecutive accumulator registers (r0 .. r0+7) of one lane into the
// B-operand fragment for the follow-up MFMA whose contraction index is this
// accumulator's ROW index.  k-slot (hi, j) <-> row r0/8*16 + 8*(j>>2) + 4*hi + (j&3)
// relative to the 32-row block; the matching A operand must gather the same rows
// (frag_tr with ca = base + 4*hi, cb = base + 8 + 4*hi).
__device__ __forceinline__ bf16x8 pack_acc8(const f32x16& a, int r0) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (bf16)a[r0 + j];
  return f;
}

// ---- LDS-DMA (global_load_lds) issued from inline asm -----------------------------------------
// Lane i of the wave copies 16 (or 4) bytes from its own global address to LDS byte address
// lds_dst + 16*i (4*i); lds_dst must be wave-uniform.  Issued through asm on purpose: hipcc knows
// nothing about these loads, so it neither drains them (vmcnt(0)) in front of LDS reads / barriers nor
// counts them - the kernels place counted s_waitcnt vmcnt(N) themselves (cdna_hip_programming 5.7).
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(__UINTPTR_TYPE__)(ST_LDS const char*)p;
}
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
__device__ __forceinline__ void lds_dma4(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// Make the compiler retire its own wait for an ordinary load HERE (a register use), so that it does
// not place an s_waitcnt vmcnt(0) for it inside a loop that has asm LDS-DMA in flight.
template <typename T> __device__ __forceinline__ void touch(T& v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ float wave_xor32(float v) { return __shfl_xor(v, 32, 64); }

// 16-byte global store, write-through (sc1): the line does not stay in the XCD's L2.  For streams of output that only LATER
// kernels read (round 6: with plain stores the dirty lines of an encoder-sized launch crowd the write-back L2 - st_rowchain_common.cuh).
template <class V> __device__ __forceinline__ void store16_wt(void* p, V v) {
  static_assert(sizeof(V) == 16, "store16_wt: 16 bytes");
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
template <bool WT, class V> __device__ __forceinline__ void store16(void* p, V v) {
  if (WT) store16_wt(p, v);
  else *reinterpret_cast<V*>(p) = v;
}

// ---- dropout (training mode of nn.Dropout: Attention.py:89, SubLayers.py:25,27, Models.py:31) --------------
// Counter-based: one 32-bit hash per counter yields FOUR 8-bit keep decisions, so a mask is a pure function of
// (device seed, call-site salt, element index) - the backward kernels regenerate it instead of reading a saved
// mask, and a HIP-graph replay draws fresh masks because the seed lives in device memory (advanced per step).
// keep iff byte >= thresh, thresh = round(256 p); survivors are scaled by 256 / (256 - thresh) (exactly unbiased
// for the realised keep probability; 8-bit thresholds as in FlashAttention's dropout).
__device__ __forceinline__ uint32_t st_hash32(uint32_t x) {   // "lowbias32" integer finaliser
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
struct Drop {
  uint32_t key;     // st_hash32(seed + salt * 0x9e3779b9)
  int thresh;       // 0 = dropout off
  float scale;
  __device__ __forceinline__ bool on() const { return thresh > 0; }
  __device__ __forceinline__ uint32_t bits(uint32_t counter) const { return st_hash32(counter ^ key); }
  __device__ __forceinline__ bool keep(uint32_t bits, int byte) const {
    return (int)((bits >> (8 * byte)) & 0xffu) >= thresh;
  }
};
// host-side argument block shared by the entry points that take dropout
struct DropArgs {
  const unsigned* seed;   // device pointer (null = off)
  unsigned salt;
  int thresh;
  float scale;
};
__device__ __forceinline__ Drop make_drop(const DropArgs& d) {
  Drop r;
  r.thresh = d.seed ? d.thresh : 0;
  r.scale = d.scale;
  r.key = d.seed ? st_hash32(*d.seed + d.salt * 0x9e3779b9u) : 0u;
  return r;
}
// row-matrix elements: counter of the 4 consecutive columns col .. col+3 (col % 4 == 0) of `row`
__device__ __forceinline__ uint32_t drop_counter_rc(int row, int col, int ncols) {
  return (uint32_t)row * (uint32_t)(ncols >> 2) + (uint32_t)(col >> 2);
}
// attention probabilities: counter of the 2 x 2 block containing (query q, key k) of head slot bh;
// byte index of (q, k) inside it = 2 * (q & 1) + (k & 1)
__device__ __forceinline__ uint32_t drop_counter_qk(int bh, int q, int k) {
  return (((uint32_t)(q >> 1) << 15) | (uint32_t)(k >> 1)) + (uint32_t)bh * 0x85ebca6bu;
}

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }

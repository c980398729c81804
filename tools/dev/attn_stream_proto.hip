// Dev prototype (NOT part of the product): timing model of a "streaming" attention forward for the encoder shape.
// Hypothesis: with K and V^T stored as MFMA A-operand fragment streams per (utterance, head) - as the
// row chains store their weights - a wave can run QK^T / softmax / PV over its 64 queries with coalesced 1 KB loads, no LDS
// staging and no barriers.  This kernel streams synthetic fragment buffers of the right size and executes the real
// instruction mix (MFMAs, online softmax, P packing); it does not produce a checked result.
//   K stream : [bh][key tile of 32][4 k-steps][64 lanes] x 16 B        (S^T tile = K_tile Q^T)
//   V^T stream: [bh][key tile of 32][2 key-steps][2 dv tiles][64 lanes] x 16 B   (O^T += V^T_tile P^T)
#include "../../speech-tranformer-pytorch_amd/csrc/st_common.cuh"

constexpr int DEPTH = 16;   // fragments in flight per wave (two key tiles ahead)

// The fragment loads are issued from inline asm (the compiler neither drains them with vmcnt(0) at the loop edge nor
// counts them); the kernel places the counted waits itself and ties them to the registers about to be read.
__device__ __forceinline__ void frag_load(bf16x8& dst, const bf16x8* base_uniform, unsigned lane_off_bytes) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(lane_off_bytes), "s"(base_uniform) : "memory");
}
__device__ __forceinline__ void wait8(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d, bf16x8& e, bf16x8& f, bf16x8& g, bf16x8& h) {
  asm volatile("s_waitcnt vmcnt(8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}

struct ProtoArgs {
  const bf16x8* kv;        // per (b,h): tiles * 8 fragments * 64 lanes
  const bf16* q; int ldq;  // natural [rows, 256]
  bf16* o; int ldo;
  const int* off; const int* len; int H;
  const int* work; int n_work;     // (b << 16) | query block of 256
  const long long* kv_off;         // fragment offset of (b, h = 0); heads are consecutive
};

__global__ __launch_bounds__(256, 2) void attn_stream_kernel(ProtoArgs a) {
  const int item = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int w = a.work[item], b = w >> 16, qb = w & 0xffff;
  const int T = a.len[b], ntile = (T + 31) / 32;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, hi = l >> 5, r = l & 31;
  const int q0 = qb * 256 + wave * 64;
  if (q0 >= T) return;
  const float c2 = 0.125f * 1.4426950408889634f;
  // Q fragments of the wave's two query tiles (B operand): rows q0 + qt*32 + r
  bf16x8 qf[2][4];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = min(q0 + qt * 32 + r, T - 1);
      qf[qt][t] = *reinterpret_cast<const bf16x8*>(a.q + (size_t)(a.off[b] + row) * a.ldq + h * 64 + t * 16 + hi * 8);
    }
  const bf16x8* ws = a.kv + (a.kv_off[b] + (size_t)h * ntile * 8) * 64;     // wave-uniform
  bf16x8 ring[DEPTH];
  const unsigned loff = l * 16;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the Q fragments)
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) frag_load(ring[i], ws + i * 64, loff);
  ws += DEPTH * 64;
  f32x16 o[2][2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int d = 0; d < 2; ++d) o[qt][d] = zero16();
  float m[2] = {-INFINITY, -INFINITY}, lsum[2] = {0.f, 0.f};
  for (int kt = 0; kt < ntile; kt += 2) {          // two key tiles (16 fragments) per iteration: ring index static
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      wait8(ring[half * 8], ring[half * 8 + 1], ring[half * 8 + 2], ring[half * 8 + 3], ring[half * 8 + 4], ring[half * 8 + 5],
            ring[half * 8 + 6], ring[half * 8 + 7]);
      f32x16 s[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        s[qt] = zero16();
#pragma unroll
        for (int t = 0; t < 4; ++t) s[qt] = mfma32(ring[half * 8 + t], qf[qt][t], s[qt]);
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        float mx = s[qt][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[qt][i]);
        mx = fmaxf(mx, wave_xor32(mx));
        const float mn = fmaxf(m[qt], mx * c2);
        if (__any(mn != m[qt])) {
          const float alpha = __builtin_amdgcn_exp2f(m[qt] - mn);
          lsum[qt] *= alpha;
#pragma unroll
          for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[qt][d][i] *= alpha;
          m[qt] = mn;
        }
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[qt][i], c2, -m[qt]));
          s[qt][i] = p;
          ps += p;
        }
        lsum[qt] += ps;
      }
      // O^T += V^T P^T: two key steps of 16 x two dv tiles; the V^T fragments are ring[half*8 + 4 ..]
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const bf16x8 vf = ring[half * 8 + 4 + ks * 2 + d];
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) o[qt][d] = mfma32(vf, pack_acc8(s[qt], 8 * ks), o[qt][d]);
        }
      // refill this half of the ring (the tile two ahead); the stream is padded by DEPTH fragments
#pragma unroll
      for (int i = 0; i < 8; ++i) frag_load(ring[half * 8 + i], ws + (half * 8 + i) * 64, loff);
      __builtin_amdgcn_sched_barrier(0);
    }
    ws += 16 * 64;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float inv = 1.f / (lsum[qt] + wave_xor32(lsum[qt]));
    const int row = q0 + qt * 32 + r;
    if (row < T) {
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (bf16)(o[qt][d][4 * g + e] * inv);
          *reinterpret_cast<bf16x4*>(a.o + (size_t)(a.off[b] + row) * a.ldo + h * 64 + d * 32 + 8 * g + 4 * hi) = v;
        }
    }
  }
}

extern "C" int proto_attn_stream(hipStream_t stream, const void* kv, const void* q, int ldq, void* o, int ldo, const int* off,
                                 const int* len, int H, const int* work, int n_work, const long long* kv_off) {
  ProtoArgs a{(const bf16x8*)kv, (const bf16*)q, ldq, (bf16*)o, ldo, off, len, H, work, n_work, kv_off};
  hipLaunchKernelGGL(attn_stream_kernel, dim3(n_work * H), dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}

// Dev probe: the row chains' block_mma loop in isolation (csrc/st_rowchain_common.cuh) - per wave and group of two k-steps: 2 MT
// ds_read_b128 of activation fragments that all eight waves share, 2 MT MFMAs against two streamed weight fragments, two ring
// refills; 16 fragments per 256 x 256 block, ring depth 8 (MT = 3).  The chains measure 2.2-2.4 us per block against 1.62 of pure
// MFMA issue and 1.08 of weight stream.  Variants: no LDS reads; LDS reads as the chain issues them; reads issued one group AHEAD
// (software-pipelined inside the wave); waves 4-7 delayed by half a group at the start (do the two waves of a SIMD stay in step?).
//   hipcc --offload-arch=gfx950 -O3 -w tools/dev/block_mma_probe.hip -o /tmp/bmp && /tmp/bmp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int AS = 264, MT = 3, D = 8;

template <int MODE>   // 0: no LDS reads, 1: as the chain, 2: reads one group ahead, 3: as the chain + late waves 4-7
__global__ __launch_bounds__(512, 1) void k(const bf16x8* __restrict__ w, int wave_frags, int nb, float* out) {
  __shared__ __attribute__((aligned(16))) __bf16 tile[96 * AS];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63, r = l & 31, hi = l >> 5;
  for (int i = threadIdx.x; i < 96 * AS; i += 512) tile[i] = (__bf16)(0.001f * (i & 255));
  __syncthreads();
  const bf16x8* ws = w + (size_t)wave * wave_frags * 64;
  bf16x8 ring[D];
#pragma unroll
  for (int i = 0; i < D; ++i) ring[i] = ws[i * 64 + l];
  ws += D * 64;
  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
  bf16x8 xc;
#pragma unroll
  for (int e = 0; e < 8; ++e) xc[e] = (__bf16)(0.001f * (l + e));
  auto frag = [&](int mt, int ks) { return *reinterpret_cast<const bf16x8*>(tile + (mt * 32 + r) * AS + ks * 16 + hi * 8); };
  if (MODE == 3 && wave >= 4) {      // half a group late: three MFMAs of nothing
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xc, xc, acc[m], 0, 0, 0);
  }
  bf16x8 xn[2][MT];
  if (MODE == 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int m = 0; m < MT; ++m) xn[u][m] = frag(m, u);
  }
  for (int b = 0; b < nb; ++b) {
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
      bf16x8 xf[2][MT];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if (MODE == 0) xf[u][m] = xc;
          else if (MODE == 2) xf[u][m] = xn[u][m];
          else xf[u][m] = frag(m, 2 * k2 + u);
        }
      if (MODE == 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int m = 0; m < MT; ++m) xn[u][m] = frag(m, (2 * k2 + 2 + u) & 15);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[(2 * k2 + u) % D], xf[u][m], acc[m], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) ring[(2 * k2 + u) % D] = ws[(2 * k2 + u) * 64 + l];
      __builtin_amdgcn_sched_barrier(0);
    }
    ws += 16 * 64;
  }
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[m][e];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, const bf16x8* w, int wave_frags, int nb, float* out, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(512), 0, 0, w, wave_frags, nb, out);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(512), 0, 0, w, wave_frags, nb, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double us = best * 1000.0 / 20.0;
  printf("  %-44s grid %3d: %6.2f us per launch, %5.2f us per block\n", name, grid, us, us / nb);
}

int main() {
  const int nb = 48, wave_frags = nb * 16 + 64;
  const size_t bytes = (size_t)8 * wave_frags * 64 * 16;
  bf16x8* w;
  float* out;
  hipMalloc(&w, bytes);
  hipMalloc(&out, 4096);
  hipMemset(w, 0, bytes);
  for (int grid : {1, 251}) {
    run<0>("MFMAs + weight stream, no LDS reads", w, wave_frags, nb, out, grid);
    run<1>("+ activation reads as the chain issues them", w, wave_frags, nb, out, grid);
    run<2>("+ activation reads one group ahead", w, wave_frags, nb, out, grid);
    run<3>("as the chain, waves 4-7 half a group late", w, wave_frags, nb, out, grid);
  }
  return 0;
}

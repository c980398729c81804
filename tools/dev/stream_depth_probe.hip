// Dev probe: how fast can ONE workgroup pull a per-wave weight-fragment stream through its CU, as a function of the fragments
// it keeps in flight?  The decoder-sized row chains (csrc/st_rowchain.hip, MT = 1) multiply one MFMA per 1 KB fragment with
// a 16-deep register ring per wave (128 KB in flight per CU) and measure ~1.9-2.2 us per 128 KB weight block against 1.2 us
// at the CU's 64 B/clk vector-memory rate.  This kernel is that loop alone: 8 waves, NB blocks of 16 fragments per wave, ring
// depth D, one v_mfma_f32_32x32x16_bf16 per fragment against a fixed B operand, the streams L2-resident (warm-up launch) and
// shared by all workgroups like the chains' (grid = 38 / 152 / 256 workgroups).
//   hipcc --offload-arch=gfx950 -O3 -w tools/dev/stream_depth_probe.hip -o /tmp/sdp && /tmp/sdp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int D, int MT>
__global__ __launch_bounds__(512, 1) void k(const bf16x8* __restrict__ w, int wave_frags, int nb, float* out) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
  const bf16x8* ws = w + (size_t)wave * wave_frags * 64;
  bf16x8 ring[D];
#pragma unroll
  for (int i = 0; i < D; ++i) ring[i] = ws[i * 64 + l];
  ws += D * 64;
  bf16x8 x;
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = (__bf16)(0.001f * (l + e));
  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
  for (int b = 0; b < nb * 16 / D; ++b) {
#pragma unroll
    for (int g = 0; g < D / 2; ++g) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[2 * g + u], x, acc[m], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) ring[2 * g + u] = ws[(2 * g + u) * 64 + l];
      __builtin_amdgcn_sched_barrier(0);
    }
    ws += D * 64;
  }
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[m][e];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int D, int MT>
void run(const bf16x8* w, int wave_frags, int nb, float* out, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<D, MT>), dim3(grid), dim3(512), 0, 0, w, wave_frags, nb, out);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<D, MT>), dim3(grid), dim3(512), 0, 0, w, wave_frags, nb, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double us = best * 1000.0 / 20.0;
  printf("  D %2d MT %d grid %3d nb %2d: %6.2f us per launch, %5.2f us per 128 KB block, %5.1f GB/s per workgroup\n", D, MT, grid, nb, us,
         us / nb, nb * 131072.0 / us / 1000.0);
}

int main() {
  const int nb = 48;                            // long enough that launch overhead (~6 us back to back) is small; 6 MB of streams: L2 / MALL resident
  const int wave_frags = nb * 16 + 64;
  const size_t bytes = (size_t)8 * wave_frags * 64 * 16;
  bf16x8* w;
  float* out;
  hipMalloc(&w, bytes);
  hipMalloc(&out, 4096);
  hipMemset(w, 0, bytes);
  for (int grid : {1, 38, 152, 256}) {
    printf("grid %d\n", grid);
    run<8, 1>(w, wave_frags, nb, out, grid);
    run<16, 1>(w, wave_frags, nb, out, grid);
    run<32, 1>(w, wave_frags, nb, out, grid);
    run<48, 1>(w, wave_frags, nb, out, grid);
    run<16, 2>(w, wave_frags, nb, out, grid);
    run<16, 3>(w, wave_frags, nb, out, grid);
    run<8, 3>(w, wave_frags, nb, out, grid);
  }
  // the same with a short chain (12 blocks: what a decoder chain streams), launch overhead included
  const int nb2 = 12;
  for (int grid : {38, 152}) {
    printf("grid %d, 12 blocks\n", grid);
    run<16, 1>(w, nb2 * 16 + 64, nb2, out, grid);
    run<32, 1>(w, nb2 * 16 + 64, nb2, out, grid);
    run<48, 1>(w, nb2 * 16 + 64, nb2, out, grid);
  }
  return 0;
}

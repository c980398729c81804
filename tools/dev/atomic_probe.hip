// Dev probe: throughput and cross-XCD correctness of fp32 atomic adds of 64 x 64 tiles into a [rows, 64] buffer,
// the dQ accumulation pattern of a single-pass attention backward.  Each "region" (64 rows x 64 fp32 = 16 KB) receives
// `hits` tile adds.  Modes: agent-scope atomics from workgroups spread over all XCDs; workgroup-scope atomics with every
// adder of a region on ONE XCD (bid % 8 == region % 8); plain stores as the bandwidth reference.
// hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/dev/atomic_probe.hip -o tools/dev/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

template <int MODE>
__global__ __launch_bounds__(256) void add_tiles(float* buf, int regions, int hits, int spread) {
  // workgroup bid handles (region, hit); MODE 1 keeps all hits of a region on one XCD
  int region, hit;
  if (spread) { region = blockIdx.x % regions; hit = blockIdx.x / regions; }
  else {       // bid = 8 * (q * hits + hit) + x,  region = 8 q + x
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    hit = j % hits; region = (j / hits) * 8 + x;
  }
  if (region >= regions) return;
  float* base = buf + (size_t)region * 4096;
  const float v = 1.0f + 0.001f * hit;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float* p = base + i * 256 + threadIdx.x;
    if (MODE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else *p = v;
  }
}

int main() {
  const int regions = 24060 / 64 * 4, hits = 6;     // one encoder layer: dQ rows x 4 heads, ~6 key tiles of 128 per utterance
  const size_t n = (size_t)regions * 4096;
  float* d;
  hipMalloc(&d, n * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> h(n);
  for (int mode = 0; mode < 3; ++mode)
    for (int spread = 1; spread >= 0; --spread) {
      if (mode == 2 && !spread) continue;
      const int grid = spread ? regions * hits : ((regions + 7) / 8) * hits * 8;
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipMemset(d, 0, n * 4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(add_tiles<0>, dim3(grid), dim3(256), 0, 0, d, regions, hits, spread);
        else if (mode == 1) hipLaunchKernelGGL(add_tiles<1>, dim3(grid), dim3(256), 0, 0, d, regions, hits, spread);
        else hipLaunchKernelGGL(add_tiles<2>, dim3(grid), dim3(256), 0, 0, d, regions, hits, spread);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
      double want = 0; for (int k = 0; k < hits; ++k) want += 1.0 + 0.001 * k;
      size_t bad = 0;
      if (mode < 2) for (size_t i = 0; i < n; ++i) bad += std::fabs(h[i] - want) > 1e-3;
      const double mb = (double)grid * 16384 / 1e6;
      printf("%-22s %-28s %7.1f us  %6.1f MB of adds  %6.2f TB/s   wrong elements: %zu\n",
             mode == 0 ? "agent-scope atomics" : mode == 1 ? "workgroup-scope atomics" : "plain stores",
             spread ? "hits spread over all XCDs" : "hits of a region on one XCD", best * 1e3, mb, mb / best / 1e3 * 1e-3 * 1e3, bad);
    }
  return 0;
}

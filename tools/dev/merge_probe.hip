// Dev probe: what does an in-kernel "split the work over S workgroups, the last arriver merges" cost on a multi-XCD part?
// G groups x S workgroups; each workgroup first dirties `dirty_kb` of its own output (as a real kernel would have), spins
// `work_us`, writes a 16 KB fp32 partial, publishes it and takes a ticket; the last arriver of a group sums the S partials.
// Variants of publish / read:
//   0  no merge at all (baseline: partial written, nobody reads)
//   1  __threadfence() (agent-scope release: L2 write-back) + ticket; reader __threadfence() (acquire) + plain loads
//   2  write-through stores (system-scope atomic stores, no L2 write-back) + s_waitcnt + ticket; reader agent-scope loads
// Checks the merged sums.  hipcc -O3 --offload-arch=gfx950 tools/dev/merge_probe.hip -o tools/dev/merge_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

template <int MODE>
__global__ __launch_bounds__(256) void split_merge(float* partial, float* out, unsigned* ticket, float* dirty, int S, int work_us, int dirty_kb,
                                                   int same_xcd) {
  // group / split of this workgroup; same_xcd: the S splits of a group share bid % 8
  int grp, s;
  if (same_xcd) { const int x = blockIdx.x & 7, j = blockIdx.x >> 3; s = j % S; grp = (j / S) * 8 + x; }
  else { grp = blockIdx.x / S; s = blockIdx.x % S; }
  const int tid = threadIdx.x;
  float* mydirty = dirty + (size_t)blockIdx.x * dirty_kb * 256;
  for (int i = tid; i < dirty_kb * 256; i += 256) mydirty[i] = (float)i;          // other output of the kernel, dirty in L2
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < work_us * 100) __builtin_amdgcn_s_sleep(4);
  float* mine = partial + ((size_t)grp * S + s) * 4096;
  const float v = 1.0f + s;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (MODE == 2) __hip_atomic_store(mine + i * 256 + tid, v + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else mine[i * 256 + tid] = v + i;
  }
  if (MODE == 0) return;
  __shared__ bool last;
  if (MODE == 1) __threadfence();
  else __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (tid == 0) last = __hip_atomic_fetch_add(ticket + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)S - 1;
  __syncthreads();
  if (!last) return;
  if (MODE == 1) __threadfence();
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k = 0; k < S; ++k) {
    const float* p = partial + ((size_t)grp * S + k) * 4096;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      acc[i] += MODE == 2 ? __hip_atomic_load(p + i * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[i * 256 + tid];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) out[(size_t)grp * 4096 + i * 256 + tid] = acc[i];
  if (tid == 0) ticket[grp] = 0u;
}

int main() {
  const int G = 128, S = 4, N = G * S;
  float *partial, *out, *dirty; unsigned* ticket;
  hipMalloc(&partial, (size_t)N * 16384); hipMalloc(&out, (size_t)G * 16384); hipMalloc(&ticket, G * 4);
  hipMalloc(&dirty, (size_t)N * 256 * 1024);
  hipMemset(ticket, 0, G * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> h((size_t)G * 4096);
  for (int dirty_kb : {0, 64, 256})
    for (int same : {0, 1})
      for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
          hipMemset(out, 0, (size_t)G * 16384);
          hipDeviceSynchronize();
          hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL(split_merge<0>, dim3(N), dim3(256), 0, 0, partial, out, ticket, dirty, S, 5, dirty_kb, same);
          if (mode == 1) hipLaunchKernelGGL(split_merge<1>, dim3(N), dim3(256), 0, 0, partial, out, ticket, dirty, S, 5, dirty_kb, same);
          if (mode == 2) hipLaunchKernelGGL(split_merge<2>, dim3(N), dim3(256), 0, 0, partial, out, ticket, dirty, S, 5, dirty_kb, same);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          best = ms < best ? ms : best;
        }
        size_t bad = 0;
        if (mode) {
          hipMemcpy(h.data(), out, (size_t)G * 16384, hipMemcpyDeviceToHost);
          for (int g = 0; g < G; ++g) for (int i = 0; i < 16; ++i) for (int t = 0; t < 256; ++t) {
            float want = 0; for (int k = 0; k < S; ++k) want += 1.0f + k + i;
            bad += std::fabs(h[(size_t)g * 4096 + i * 256 + t] - want) > 1e-3;
          }
        }
        printf("dirty %3d KB/wg  %-9s  mode %d (%s): %6.1f us   wrong %zu\n", dirty_kb, same ? "same-XCD" : "spread", mode,
               mode == 0 ? "no merge" : mode == 1 ? "__threadfence" : "write-through + scoped loads", best * 1e3, bad);
      }
  return 0;
}

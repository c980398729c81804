// Dev probe: where does the dispatcher put workgroup `bid` when exactly (or slightly more than) one device-full of
// workgroups is launched?  Prints bid -> (xcc, se, cu) and checks candidate closed-form maps.
// hipcc -O2 --offload-arch=gfx950 tools/dev/placement_probe.hip -o /tmp/placement_probe && /tmp/placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <set>

template <int PER_CU>
__global__ __launch_bounds__(256) void probe(long long* out, int spin_us) {
  extern __shared__ char lds[];
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) {
    lds[0] = 1;
    out[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // XCC_ID
    out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_ID
    out[blockIdx.x * 4 + 2] = t0;
  }
  while (wall_clock64() - t0 < spin_us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x * 4 + 3] = wall_clock64();
}

int main() {
  for (int per_cu : {3, 2}) {
    const int lds = per_cu == 3 ? 50 * 1024 : 70 * 1024;
    for (int extra : {0, 48}) {
      const int n = per_cu * 256 + extra;
      long long* d;
      hipMalloc(&d, n * 32);
      hipMemset(d, 0, n * 32);
      hipFuncSetAttribute((const void*)probe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<3>, dim3(n), dim3(256), lds, 0, d, 20);
      hipDeviceSynchronize();
      std::vector<long long> h(n * 4);
      hipMemcpy(h.data(), d, n * 32, hipMemcpyDeviceToHost);
      long long t0 = h[2];
      for (int i = 0; i < n; ++i) t0 = h[i * 4 + 2] < t0 ? h[i * 4 + 2] : t0;
      std::map<long long, std::vector<int>> by_cu;
      int xcd_rr = 0, late = 0;
      for (int i = 0; i < n; ++i) {
        const long long xcc = h[i * 4] & 0xf, hw = h[i * 4 + 1];
        const long long key = (xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 0xf);
        by_cu[key].push_back(i);
        xcd_rr += (xcc == i % 8);
        late += (h[i * 4 + 2] - t0 > 500);
      }
      printf("== %d per CU (lds %d), %d workgroups: %zu distinct CUs, xcc == bid %% 8 for %d, started late (> 5 us): %d\n", per_cu, lds, n,
             by_cu.size(), xcd_rr, late);
      std::map<size_t, int> hist;
      for (auto& kv : by_cu) hist[kv.second.size()]++;
      for (auto& kv : hist) printf("   %d CUs host %zu workgroups\n", kv.second, kv.first);
      int shown = 0;
      for (auto& kv : by_cu) {
        if (shown++ >= 12) break;
        printf("   xcc %lld se %lld cu %lld:", kv.first >> 16, (kv.first >> 8) & 0xff, kv.first & 0xff);
        for (int b : kv.second) printf(" %d(j=%d)", b, b / 8);
        printf("\n");
      }
      // candidate maps over the XCD-local index j = bid / 8: breadth-first (j % 32) or depth-first (j / per_cu)
      int bf = 0, df = 0, tot = 0;
      for (auto& kv : by_cu) {
        std::set<int> a, b;
        for (int bid : kv.second) if (bid < per_cu * 256) { a.insert((bid / 8) % 32); b.insert((bid / 8) / per_cu); }
        bf += a.size() == 1; df += b.size() == 1; ++tot;
      }
      printf("   CUs whose workgroups share j %% 32: %d / %d;  share j / %d: %d / %d\n", bf, tot, per_cu, df, tot);
      hipFree(d);
    }
  }
  return 0;
}

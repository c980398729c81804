// Dev probe: LDS read / write throughput per CU (inline-asm DS instructions, addresses as the GEMM kernels form them:
// lane (r = l & 31, hi = l >> 5) -> row r of a tile with a padded row stride, 16-byte column hi; immediate offsets).
//   hipcc --offload-arch=gfx950 -O3 -w tools/dev/lds_bw.hip -o tools/dev/lds_bw && tools/dev/lds_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) float f2;

template <int MODE, int STRIDE_B, int WAVES>   // 0: ds_read_b128, 1: ds_read_b64, 2: ds_write_b128, 3: ds_write_b64, 4: ds_read_b64_tr_b16
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[32 * 1024];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, r = l & 31, hi = l >> 5;
  for (int i = threadIdx.x; i < 8 * 1024; i += 64 * WAVES) reinterpret_cast<float*>(smem)[i] = (float)i;
  __syncthreads();
  f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const unsigned addr = (unsigned)(size_t)(smem) + ((w & 3) * 32 + r) % 128 * STRIDE_B % (16 * 1024) + hi * 16;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      asm volatile("ds_read_b128 %0, %4 offset:0\n ds_read_b128 %1, %4 offset:32\n ds_read_b128 %2, %4 offset:4096\n ds_read_b128 %3, %4 offset:4128\n s_waitcnt lgkmcnt(0)"
                   : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(addr));
    } else if (MODE == 1) {
      f2 b0, b1, b2, b3;
      asm volatile("ds_read_b64 %0, %4 offset:0\n ds_read_b64 %1, %4 offset:32\n ds_read_b64 %2, %4 offset:4096\n ds_read_b64 %3, %4 offset:4128\n s_waitcnt lgkmcnt(0)"
                   : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(addr));
      a0[0] += b0[0] + b1[0] + b2[0] + b3[0];
    } else if (MODE == 2) {
      asm volatile("ds_write_b128 %0, %1 offset:0\n ds_write_b128 %0, %1 offset:32\n ds_write_b128 %0, %1 offset:4096\n ds_write_b128 %0, %1 offset:4128\n s_waitcnt lgkmcnt(0)"
                   :: "v"(addr), "v"(a0));
    } else if (MODE == 3) {
      f2 b = {1.f, 2.f};
      asm volatile("ds_write_b64 %0, %1 offset:0\n ds_write_b64 %0, %1 offset:32\n ds_write_b64 %0, %1 offset:4096\n ds_write_b64 %0, %1 offset:4128\n s_waitcnt lgkmcnt(0)"
                   :: "v"(addr), "v"(b));
    } else {
      f2 b0, b1, b2, b3;
      asm volatile("ds_read_b64_tr_b16 %0, %4 offset:0\n ds_read_b64_tr_b16 %1, %4 offset:32\n ds_read_b64_tr_b16 %2, %4 offset:4096\n ds_read_b64_tr_b16 %3, %4 offset:4128\n s_waitcnt lgkmcnt(0)"
                   : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(addr));
      a0[0] += b0[0] + b1[0] + b2[0] + b3[0];
    }
  }
  if (a0[0] == 123.456f) out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}

template <int MODE, int STRIDE_B, int WAVES>
void run(const char* name, int per_cu) {
  float* d;
  hipMalloc(&d, 8192);
  const int iters = 40000, grid = 256 * per_cu;
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL((k<MODE, STRIDE_B, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL((k<MODE, STRIDE_B, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, d, iters);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  const int bytes_per_lane = (MODE == 0 || MODE == 2) ? 16 : 8;
  const double bytes = (double)iters * 4 * 64 * WAVES * bytes_per_lane * per_cu;   // per CU
  printf("%-26s stride %3d B, %2d waves/CU: %7.1f GB/s per CU = %5.1f B/clk at 2.4 GHz\n", name, STRIDE_B, WAVES * per_cu,
         bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 2.4e9);
  hipFree(d);
}

int main() {
  run<0, 80, 4>("ds_read_b128", 1);
  run<0, 80, 4>("ds_read_b128", 2);
  run<0, 80, 4>("ds_read_b128", 4);
  run<0, 64, 4>("ds_read_b128 (conflicting)", 2);
  run<0, 144, 4>("ds_read_b128", 2);
  run<1, 80, 4>("ds_read_b64", 2);
  run<4, 256, 4>("ds_read_b64_tr_b16", 2);
  run<2, 80, 4>("ds_write_b128", 1);
  run<2, 80, 4>("ds_write_b128", 2);
  run<2, 80, 4>("ds_write_b128", 4);
  run<3, 80, 4>("ds_write_b64", 2);
  return 0;
}

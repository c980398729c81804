// Dev probe: what a gfx950 SIMD can issue per cycle - VALU / transcendental / MFMA alone and mixed, one wave or two per SIMD,
// in one instruction stream or in two.  Answers (for the attention rewrite): is the VALU pipe 2 or 4 cycles per wave64
// instruction, do two waves of a SIMD add up their issue rates, how many VALU fit under one 32x32x16 MFMA.
//   hipcc --offload-arch=gfx950 -O3 -w tools/dev/issue_probe.hip -o tools/dev/issue_probe && tools/dev/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)

// MODE 0: NV independent v_fma_f32 per group, no MFMA
// MODE 1: v_exp_f32
// MODE 2: 1 MFMA + NV v_fma per group (same wave)
// MODE 3: MFMA only
// MODE 4: role split: waves 0-3 MFMA only, waves 4-7 VALU (NV fma per group) only
// MODE 5: 1 MFMA + NV mixed softmax-like ops (max3, exp, add, cvt_pk)
// MODE 6: v_pk_mul_f32
// MODE 7: v_cvt_pk_bf16_f32
// MODE 8: v_max3_f32
template <int MODE, int NV>
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* cyc) {
  const int wave = threadIdx.x >> 6;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)(threadIdx.x + i) * 1e-3f;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * threadIdx.x); b[j] = (__bf16)(0.002f * j); }
  const float c = 1.0001f, d = 0.0001f;
  __syncthreads();
  const long long t0 = clock64();
  const bool do_mfma = (MODE == 2 || MODE == 3 || MODE == 5 || (MODE == 4 && wave < 4));
  const bool do_valu = (MODE == 0 || MODE == 1 || MODE == 2 || MODE == 5 || MODE >= 6 || (MODE == 4 && wave >= 4));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (do_mfma) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(a), "v"(b));
      if (do_valu) {
#pragma unroll
        for (int n = 0; n < NV; ++n) {
          float& x = v[(n + g * 3) & 7];
          if (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
          else if (MODE == 5) {
            const int kind = n & 7;   // per 8: 2 max3, 2 exp, 2 add, 1 cvt, 1 sub  (~ softmax mix)
            if (kind == 0 || kind == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
            else if (kind == 1 || kind == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
            else if (kind == 2 || kind == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(d));
            else if (kind == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(c));
            else asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(d));
          } else if (MODE == 6) {
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&v[((n + g) & 3) * 2]) : "v"(*(double*)&v[0]));
          } else if (MODE == 7) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(c));
          else if (MODE == 8) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
        }
      }
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][3];
  if (s == 123.456f) out[threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

template <int MODE, int NV>
void run(const char* name, int waves_per_simd) {
  float* d; long long* cyc;
  hipMalloc(&d, 8192); hipMalloc(&cyc, 64);
  const int iters = 4000, threads = 256 * waves_per_simd;
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, d, 10, cyc);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, d, iters, cyc);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  const double groups = (double)iters * 4;
  printf("%-34s NV=%2d waves/SIMD=%d : %7.1f clk/group (wave0, s_memtime)  wave4 %7.1f   wall %.3f ms -> %6.1f ns/group\n", name, NV,
         waves_per_simd, h[0] / groups, waves_per_simd > 1 ? h[4] / groups : 0.0, ms, ms * 1e6 / groups);
  hipFree(d); hipFree(cyc);
}

#define RUNV(M, name) \
  run<M, 1>(name, 1); run<M, 1>(name, 2); run<M, 4>(name, 1); run<M, 4>(name, 2); run<M, 8>(name, 1); run<M, 8>(name, 2);
#define RUNMIX(M, name) \
  run<M, 0>(name, 1); run<M, 2>(name, 1); run<M, 4>(name, 1); run<M, 5>(name, 1); run<M, 6>(name, 1); run<M, 7>(name, 1); run<M, 8>(name, 1); \
  run<M, 10>(name, 1); run<M, 12>(name, 1); run<M, 16>(name, 1); \
  run<M, 0>(name, 2); run<M, 2>(name, 2); run<M, 4>(name, 2); run<M, 5>(name, 2); run<M, 6>(name, 2); run<M, 7>(name, 2); run<M, 8>(name, 2); \
  run<M, 10>(name, 2); run<M, 12>(name, 2); run<M, 16>(name, 2);

int main() {
  printf("group = what one wave issues per inner step; s_memtime ticks; 256 workgroups (one per CU)\n");
  RUNV(0, "v_fma_f32 only");
  RUNV(1, "v_exp_f32 only");
  RUNV(6, "v_pk_mul_f32 only");
  RUNV(7, "v_cvt_pk_bf16_f32 only");
  RUNV(8, "v_max3_f32 only");
  run<3, 0>("MFMA only", 1); run<3, 0>("MFMA only", 2);
  RUNMIX(2, "1 MFMA + NV fma (same wave)");
  RUNMIX(5, "1 MFMA + NV softmax mix");
  run<4, 4>("roles: w0-3 MFMA, w4-7 NV fma", 2); run<4, 8>("roles: w0-3 MFMA, w4-7 NV fma", 2); run<4, 12>("roles: w0-3 MFMA, w4-7 NV fma", 2);
  run<4, 16>("roles: w0-3 MFMA, w4-7 NV fma", 2);
  return 0;
}

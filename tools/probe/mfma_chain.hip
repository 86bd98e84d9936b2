// cycles per v_mfma_f32_32x32x16_bf16 by dependency pattern (one wave per SIMD): NACC accumulators used round-robin (1 = one dependent
// chain, as the S^T phase of the generator sweep accumulates its 32 k-steps), accumulator in AGPRs (builtin) or VGPRs (inline asm)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, bool VG>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int slot, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      if (VG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i % NACC]) : "v"(a), "v"(b));
      else acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i % NACC], 0, 0, 0);
    }
  }
  if (VG) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64 * 8);
  const int iters = 200;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<1, false>), dim3(256), dim3(256), 0, 0, out, cyc, 0, iters);
    hipLaunchKernelGGL((k<2, false>), dim3(256), dim3(256), 0, 0, out, cyc, 1, iters);
    hipLaunchKernelGGL((k<4, false>), dim3(256), dim3(256), 0, 0, out, cyc, 2, iters);
    hipLaunchKernelGGL((k<1, true>), dim3(256), dim3(256), 0, 0, out, cyc, 3, iters);
    hipLaunchKernelGGL((k<2, true>), dim3(256), dim3(256), 0, 0, out, cyc, 4, iters);
    hipLaunchKernelGGL((k<4, true>), dim3(256), dim3(256), 0, 0, out, cyc, 5, iters);
  }
  hipDeviceSynchronize();
  unsigned long long h[8]; hipMemcpy(h, cyc, 8 * 8, hipMemcpyDeviceToHost);
  const double n = 32.0 * iters;
  printf("cycles per MFMA, AGPR accumulators: 1 chain %.1f | 2 alternating %.1f | 4 %.1f\n", h[0] / n, h[1] / n, h[2] / n);
  printf("cycles per MFMA, VGPR accumulators: 1 chain %.1f | 2 alternating %.1f | 4 %.1f\n", h[3] / n, h[4] / n, h[5] / n);
  return 0;
}

// Can a wave's VALU instructions issue in the shadow of its own MFMAs on gfx950?  One wave per SIMD (256 threads, one workgroup per CU):
// 32 back-to-back v_mfma_f32_32x32x16_bf16 (16 accumulators) with N independent VALU instructions behind each.
//   mode: NV = VALU per gap (v_fma_f32 on 4 rotating registers), NE = v_exp_f32 per gap, MF = with / without the MFMA
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/mfma_valu tools/probe/mfma_valu.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NV, int NE, bool MF>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int slot, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[16];
  for (int j = 0; j < 16; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
  float x[4] = {1.f + lane, 2.f, 3.f, 4.f}, y = 0.999f, z[2] = {0.5f, 0.25f};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      if (MF) acc[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i >> 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[v & 3]) : "v"(y));
#pragma unroll
      for (int v = 0; v < NE; ++v) asm volatile("v_exp_f32 %0, %0" : "+v"(z[v & 1]));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = x[0] + x[1] + x[2] + x[3] + z[0] + z[1];
  for (int j = 0; j < 16; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64 * 8);
  const int iters = 200;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<0, 0, true>), dim3(256), dim3(256), 0, 0, out, cyc, 0, iters);
    hipLaunchKernelGGL((k<2, 0, true>), dim3(256), dim3(256), 0, 0, out, cyc, 1, iters);
    hipLaunchKernelGGL((k<4, 0, true>), dim3(256), dim3(256), 0, 0, out, cyc, 2, iters);
    hipLaunchKernelGGL((k<6, 0, true>), dim3(256), dim3(256), 0, 0, out, cyc, 3, iters);
    hipLaunchKernelGGL((k<8, 0, true>), dim3(256), dim3(256), 0, 0, out, cyc, 4, iters);
    hipLaunchKernelGGL((k<12, 0, true>), dim3(256), dim3(256), 0, 0, out, cyc, 5, iters);
    hipLaunchKernelGGL((k<4, 0, false>), dim3(256), dim3(256), 0, 0, out, cyc, 6, iters);
    hipLaunchKernelGGL((k<8, 0, false>), dim3(256), dim3(256), 0, 0, out, cyc, 7, iters);
    hipLaunchKernelGGL((k<0, 1, true>), dim3(256), dim3(256), 0, 0, out, cyc, 8, iters);
    hipLaunchKernelGGL((k<0, 2, true>), dim3(256), dim3(256), 0, 0, out, cyc, 9, iters);
    hipLaunchKernelGGL((k<0, 1, false>), dim3(256), dim3(256), 0, 0, out, cyc, 10, iters);
    hipLaunchKernelGGL((k<3, 1, true>), dim3(256), dim3(256), 0, 0, out, cyc, 11, iters);
  }
  hipDeviceSynchronize();
  unsigned long long h[16]; hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
  const double n = 32.0 * iters;
  printf("cycles per step (one step = [MFMA] + N VALU), one wave per SIMD:\n");
  printf("  MFMA alone %.1f | + 2 fma %.1f | + 4 fma %.1f | + 6 fma %.1f | + 8 fma %.1f | + 12 fma %.1f\n", h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n);
  printf("  no MFMA: 4 fma %.1f | 8 fma %.1f | 1 exp %.1f\n", h[6] / n, h[7] / n, h[10] / n);
  printf("  MFMA + 1 exp %.1f | + 2 exp %.1f | + 3 fma + 1 exp %.1f\n", h[8] / n, h[9] / n, h[11] / n);
  return 0;
}

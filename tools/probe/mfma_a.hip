// phase A of gen2p_kernel in isolation: 32 v_mfma_f32_32x32x16_bf16 on ONE accumulator, each fed by a ds_read_b128 issued PD steps earlier
//   MODE bit 0: accumulator in VGPRs (inline asm) instead of AGPRs (builtin)      bit 1: the kernel's swizzled row addresses instead of linear
//   bit 2: all four waves read the SAME 32-KiB tile instead of 4 KiB each          bit 3: B operand from 28 different register groups
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f) { if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); } }
template <int OFF> __device__ __forceinline__ u32x4 rd128(unsigned a) { u32x4 r; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF)); return r; }
template <int MODE>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int slot, int iters) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, r31 = lane & 31;
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = (float)(i & 7);
  __syncthreads();
  f32x16 sT;
  for (int r = 0; r < 16; ++r) sT[r] = 0.f;
  bf16x8 xf[28];
  for (int j = 0; j < 28; ++j) for (int e = 0; e < 8; ++e) xf[j][e] = (__bf16)(float)(lane - e + j);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const int sw = ((r31 & 3) << 2) | ((r31 >> 2) & 3);
  unsigned uaq[8];
  for (int e = 0; e < 8; ++e) {
    if (MODE & 2) uaq[e] = (lds0 + r31 * 1024 + ((half ^ sw) * 16) + ((MODE & 4) ? 0 : 0)) ^ (e << 5);
    else uaq[e] = lds0 + lane * 16 + ((MODE & 4) ? 0 : wave * 4096) + e * 1024 * ((MODE & 4) ? 1 : 0);
  }
  constexpr int PD = 5;
  u32x4 fa[PD];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    sfor<0, PD>([&](auto kc) { constexpr int ks = decltype(kc)::value; fa[ks % PD] = rd128<(ks >> 3) * 256>(uaq[ks & 7]); });
    sfor<0, 32>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      constexpr int n = (31 - ks) < (PD - 1) ? (31 - ks) : (PD - 1);
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fa[ks % PD]) : "n"(n));
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 bop = (MODE & 8) ? xf[ks % 28] : xf[0];
      if (MODE & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sT) : "v"(__builtin_bit_cast(bf16x8, fa[ks % PD])), "v"(bop));
      else sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[ks % PD]), bop, sT, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks + PD < 32) fa[ks % PD] = rd128<((ks + PD) >> 3) * 256>(uaq[(ks + PD) & 7]);
    });
    if (MODE & 1) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(sT));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += sT[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
}
#define RUN(m) do { hipFuncSetAttribute((const void*)k<m>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
  hipLaunchKernelGGL((k<m>), dim3(256), dim3(256), 65536, 0, out, cyc, m, iters); } while (0)
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64 * 8);
  const int iters = 200;
  for (int rep = 0; rep < 2; ++rep) { RUN(0); RUN(1); RUN(2); RUN(3); RUN(6); RUN(7); RUN(8); RUN(9); RUN(15); RUN(14); RUN(4); RUN(5); }
  hipDeviceSynchronize();
  unsigned long long h[16]; hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
  const double n = 32.0 * iters;
  const int ms[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 14, 15};
  for (int m : ms) printf("mode %2d [%s acc, %s addresses, %s, %s B]: %.1f cycles per MFMA\n", m, (m & 1) ? "VGPR" : "AGPR", (m & 2) ? "swizzled" : "linear",
                          (m & 4) ? "shared tile" : "own 4 KiB", (m & 8) ? "rotating" : "fixed", h[m] / n);
  return 0;
}

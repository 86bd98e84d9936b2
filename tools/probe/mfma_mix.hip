// What a wave's instruction stream can hold per MFMA on gfx950, one wave per SIMD (4 waves per CU, all doing the same): per step
//   [s_waitcnt lgkmcnt] MFMA  [NF x v_fma] [NE x v_exp] [NT x ds_read_b64_tr_b16, waited for PD steps later] [NB x ds_read_b128]
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/mfma_mix tools/probe/mfma_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int NF, int NE, int NT, int NB, bool WAIT, bool USE, int TPB = 256>
__global__ void __launch_bounds__(TPB, 1) k(float* out, unsigned long long* cyc, int slot, int iters) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
  for (int i = threadIdx.x; i < 65536 / 4; i += TPB) reinterpret_cast<float*>(lds)[i] = (float)(i & 7);
  __syncthreads();
  constexpr int NACC = TPB == 512 ? 6 : 16;      // (two waves per SIMD: 256 registers per lane)
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
  float x[4] = {1.f + lane, 2.f, 3.f, 4.f}, y = 0.999f, z[2] = {0.5f, 0.25f};
  const unsigned a8 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + lane * 8 + wave * 2048;
  const unsigned a16 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + lane * 16 + wave * 4096;
  constexpr int PD = 4;
  s16x4 fl[PD], fh[PD];
  i32x4 q[PD];
  for (int i = 0; i < PD; ++i) { fl[i] = s16x4{1, 2, 3, 4}; fh[i] = s16x4{1, 2, 3, 4}; q[i] = i32x4{0, 0, 0, 0}; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (WAIT && NT + NB > 0) {
        constexpr int n = (NT + NB) * (PD - 1);
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fl[i % PD]), "+v"(fh[i % PD]), "+v"(q[i % PD]) : "v"(0), "n"(n > 15 ? 15 : n));
      }
      bf16x8 av = a;
      if (USE && NT == 2) av = __builtin_bit_cast(bf16x8, __builtin_shufflevector(fl[i % PD], fh[i % PD], 0, 1, 2, 3, 4, 5, 6, 7));
      __builtin_amdgcn_sched_barrier(0);
      acc[(i >> 1) % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b, acc[(i >> 1) % NACC], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < NF; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[v & 3]) : "v"(y));
#pragma unroll
      for (int v = 0; v < NE; ++v) asm volatile("v_exp_f32 %0, %0" : "+v"(z[v & 1]));
      if (NT >= 1) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fl[i % PD]) : "v"(a8), "n"(0));
      if (NT >= 2) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fh[i % PD]) : "v"(a8), "n"(16384));
      if (NB >= 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i % PD]) : "v"(a16), "n"(32768));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = x[0] + x[1] + x[2] + x[3] + z[0] + z[1];
  for (int i = 0; i < PD; ++i) s += (float)fl[i][0] + (float)fh[i][0] + (float)q[i][0];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
}
#define RUN2(slot, ...) do { hipFuncSetAttribute((const void*)k<__VA_ARGS__, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
  hipLaunchKernelGGL((k<__VA_ARGS__, 512>), dim3(256), dim3(512), 65536, 0, out, cyc, slot, iters); } while (0)
#define RUN(slot, ...) do { hipFuncSetAttribute((const void*)k<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
  hipLaunchKernelGGL((k<__VA_ARGS__>), dim3(256), dim3(256), 65536, 0, out, cyc, slot, iters); } while (0)
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64 * 8);
  const int iters = 200;
  for (int rep = 0; rep < 2; ++rep) {
    RUN(0, 0, 0, 0, 0, false, false);      // MFMA alone
    RUN(1, 0, 0, 2, 0, false, false);      // + 2 tr reads, never waited for
    RUN(2, 0, 0, 2, 0, true, false);       // + counted wait
    RUN(3, 0, 0, 2, 0, true, true);        // + the MFMA uses them
    RUN(4, 1, 1, 2, 0, true, true);        // + fma + exp (phase B's exp steps)
    RUN(5, 2, 1, 2, 0, true, true);
    RUN(6, 0, 0, 0, 1, true, false);       // phase A: one ds_read_b128 per MFMA
    RUN(7, 0, 0, 0, 1, false, false);
    RUN(8, 4, 0, 2, 0, true, true);
    RUN(9, 1, 1, 0, 0, false, false);
    RUN(10, 0, 0, 1, 0, true, false);
    RUN(11, 3, 1, 2, 0, true, true);
    RUN(12, 1, 0, 0, 1, true, false);      // phase A + VALU: b128 read per MFMA + 1 fma
    RUN(13, 2, 0, 0, 1, true, false);
    RUN(14, 4, 0, 0, 1, true, false);
    RUN(15, 0, 1, 0, 1, true, false);      // + 1 exp
    RUN(16, 1, 1, 0, 1, true, false);
    RUN(17, 2, 1, 0, 1, true, false);
    RUN(18, 6, 0, 0, 0, false, false);
    RUN(19, 2, 0, 2, 0, true, true);
    RUN(20, 1, 0, 2, 0, true, true);
    RUN(21, 0, 1, 2, 0, true, true);
    RUN2(22, 0, 0, 0, 0, false, false);     // TWO waves per SIMD (512 threads): MFMA alone
    RUN2(23, 0, 0, 2, 0, true, true);       // + 2 tr reads consumed
    RUN2(24, 1, 1, 2, 0, true, true);       // + 1 fma + 1 exp
    RUN2(25, 2, 1, 2, 0, true, true);
    RUN2(26, 4, 2, 2, 0, true, true);
    RUN2(27, 2, 1, 0, 1, true, false);      // phase A mix: b128 + 2 fma + 1 exp
    RUN2(28, 6, 2, 2, 1, true, true);
  }
  { hipError_t e_ = hipDeviceSynchronize(); if (e_ != hipSuccess) printf("sync error: %s\n", hipGetErrorString(e_)); e_ = hipGetLastError(); if (e_ != hipSuccess) printf("last error: %s\n", hipGetErrorString(e_)); }
  unsigned long long h[32]; hipMemcpy(h, cyc, 32 * 8, hipMemcpyDeviceToHost);
  const double n = 32.0 * iters;
  printf("cycles per step, one wave per SIMD, 4 waves per CU in lockstep:\n");
  printf("  MFMA alone %.1f | + 2 tr reads (no wait) %.1f | + counted wait %.1f | + MFMA consumes them %.1f | 1 tr read + wait %.1f\n", h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[10] / n);
  printf("  consume + 1 fma + 1 exp %.1f | + 2 fma + 1 exp %.1f | + 3 fma + 1 exp %.1f | + 4 fma %.1f | (1 fma + 1 exp, no LDS: %.1f)\n", h[4] / n, h[5] / n, h[11] / n, h[8] / n, h[9] / n);
  printf("  1 ds_read_b128 per MFMA: waited %.1f | not waited %.1f\n", h[6] / n, h[7] / n);
  printf("  b128 read per MFMA + 1 fma %.1f | + 2 fma %.1f | + 4 fma %.1f | + 1 exp %.1f | + 1 fma + 1 exp %.1f | + 2 fma + 1 exp %.1f | (6 fma, no LDS: %.1f)\n",
         h[12] / n, h[13] / n, h[14] / n, h[15] / n, h[16] / n, h[17] / n, h[18] / n);
  printf("  2 tr reads consumed + 1 fma %.1f | + 2 fma %.1f | + 1 exp %.1f\n", h[20] / n, h[19] / n, h[21] / n);
  printf("TWO waves per SIMD, cycles per step of ONE wave (the SIMD retires two MFMAs per step):\n");
  printf("  MFMA alone %.1f | + 2 tr consumed %.1f | + 1 fma + 1 exp %.1f | + 2 fma + 1 exp %.1f | + 4 fma + 2 exp %.1f | b128 + 2 fma + 1 exp %.1f | 2 tr + b128 + 6 fma + 2 exp %.1f\n",
         h[22] / n, h[23] / n, h[24] / n, h[25] / n, h[26] / n, h[27] / n, h[28] / n);
  return 0;
}

// One against two waves per SIMD (256 / 512 threads per workgroup, one workgroup per CU): cycles per step of a wave for
//   step = [counted wait] MFMA [NF x v_fma] [NE x v_exp] [2 x ds_read_b64_tr_b16 feeding the MFMA 4 steps later | 1 x ds_read_b128]
// With two waves the SIMD retires two MFMAs per step: "cycles per MFMA of the SIMD" = cycles per step / 2.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int TPB, int NF, int NE, int LD>      // LD: 0 none, 1 = two transposed reads (consumed), 2 = one b128 read (waited for)
__global__ void __launch_bounds__(TPB) k(float* out, unsigned long long* cyc, int slot, int iters) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 65536 / 4; i += TPB) reinterpret_cast<float*>(lds)[i] = (float)(i & 7);
  __syncthreads();
  constexpr int NACC = 6;
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
  float x[4] = {1.f + lane, 2.f, 3.f, 4.f}, y = 0.999f, z[2] = {0.5f, 0.25f};
  const unsigned a8 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + lane * 8 + (wave & 7) * 2048;
  const unsigned a16 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + lane * 16 + (wave & 7) * 4096;
  constexpr int PD = 4;
  s16x4 fl[PD], fh[PD];
  i32x4 q[PD];
  for (int i = 0; i < PD; ++i) { fl[i] = s16x4{1, 2, 3, 4}; fh[i] = s16x4{1, 2, 3, 4}; q[i] = i32x4{0, 0, 0, 0}; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (LD == 1) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fl[i % PD]), "+v"(fh[i % PD]));
      if (LD == 2) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(q[i % PD]));
      bf16x8 av = a;
      if (LD == 1) av = __builtin_bit_cast(bf16x8, __builtin_shufflevector(fl[i % PD], fh[i % PD], 0, 1, 2, 3, 4, 5, 6, 7));
      __builtin_amdgcn_sched_barrier(0);
      acc[(i >> 1) % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b, acc[(i >> 1) % NACC], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < NF; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[v & 3]) : "v"(y));
#pragma unroll
      for (int v = 0; v < NE; ++v) asm volatile("v_exp_f32 %0, %0" : "+v"(z[v & 1]));
      if (LD == 1) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fl[i % PD]) : "v"(a8), "n"(0));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fh[i % PD]) : "v"(a8), "n"(16384));
      }
      if (LD == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i % PD]) : "v"(a16), "n"(32768));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = x[0] + x[1] + x[2] + x[3] + z[0] + z[1];
  for (int i = 0; i < PD; ++i) s += (float)fl[i][0] + (float)fh[i][0] + (float)q[i][0];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * TPB + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
}
static int nslot = 0;
static const char* names[64];
#define RUN(name, ...) do { names[nslot] = name; \
  hipError_t e1 = hipFuncSetAttribute((const void*)k<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
  hipLaunchKernelGGL((k<__VA_ARGS__>), dim3(256), dim3(tpb), 65536, 0, out, cyc, nslot, iters); \
  hipError_t e2 = hipGetLastError(); if (e1 != hipSuccess || e2 != hipSuccess) printf("%s: %s / %s\n", name, hipGetErrorString(e1), hipGetErrorString(e2)); ++nslot; } while (0)
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64 * 8); hipMemset(cyc, 0, 64 * 8);
  const int iters = 200;
  for (int rep = 0; rep < 2; ++rep) {
    nslot = 0;
    { const int tpb = 256;
      RUN("1 wave : MFMA alone", 256, 0, 0, 0); RUN("1 wave : MFMA + 2 tr", 256, 0, 0, 1); RUN("1 wave : MFMA + 2 tr + 1 fma + 1 exp", 256, 1, 1, 1);
      RUN("1 wave : MFMA + 2 tr + 3 fma + 1 exp", 256, 3, 1, 1); RUN("1 wave : MFMA + b128 + 2 fma + 1 exp", 256, 2, 1, 2); RUN("1 wave : MFMA + b128 + 4 fma + 2 exp", 256, 4, 2, 2); }
    { const int tpb = 512;
      RUN("2 waves: MFMA alone", 512, 0, 0, 0); RUN("2 waves: MFMA + 2 tr", 512, 0, 0, 1); RUN("2 waves: MFMA + 2 tr + 1 fma + 1 exp", 512, 1, 1, 1);
      RUN("2 waves: MFMA + 2 tr + 3 fma + 1 exp", 512, 3, 1, 1); RUN("2 waves: MFMA + b128 + 2 fma + 1 exp", 512, 2, 1, 2); RUN("2 waves: MFMA + b128 + 4 fma + 2 exp", 512, 4, 2, 2); }
  }
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) printf("sync: %s\n", hipGetErrorString(e));
  unsigned long long h[64];
  e = hipMemcpy(h, cyc, 64 * 8, hipMemcpyDeviceToHost);
  if (e != hipSuccess) printf("copy: %s\n", hipGetErrorString(e));
  for (int i = 0; i < nslot; ++i) printf("%-44s %6.1f cycles per step of a wave\n", names[i], h[i] / (32.0 * iters));
  return 0;
}

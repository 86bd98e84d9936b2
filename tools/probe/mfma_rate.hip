// cycles per v_mfma_f32_32x32x16_bf16 in back-to-back sequences, one wave per SIMD (256 threads, one workgroup per CU):
//   a) 16 independent accumulators, operands fixed   b) the same with a ds_read_b64_tr_b16 pair in every gap
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/mfma_rate tools/probe/mfma_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = (float)(i & 7);
  __syncthreads();
  f32x16 acc[16];
  for (int j = 0; j < 16; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
  if (MODE == 7) {       // high-entropy operands (what a GEMM on real activations feeds the matrix unit): does the sustained rate depend on the data?
    unsigned x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int e = 0; e < 8; ++e) {
      x ^= x << 13; x ^= x >> 17; x ^= x << 5;
      a[e] = (__bf16)(((int)(x & 0xffff) - 32768) * (1.0f / 32768.0f));
      x ^= x << 13; x ^= x >> 17; x ^= x << 5;
      b[e] = (__bf16)(((int)(x & 0xffff) - 32768) * (1.0f / 32768.0f));
    }
  }
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (lane & 15) * 1024 + (lane >> 4) * 8;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ((MODE < 3 || MODE == 7) ? iters : 0); ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if (MODE == 1) {
          s16x4 lo, hi;
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(0));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(16384));
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi));
          s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          a = __builtin_bit_cast(bf16x8, v);
        }
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
        if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (MODE == 6) {       // the second product of gen2_kernel: 32 MFMAs, two transposed reads each, issued 6 MFMAs ahead, counted waits
    const unsigned a8 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + lane * 8;
    s16x4 fl[6], fh[6];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fl[i]) : "v"(a8), "n"(0));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fh[i]) : "v"(a8), "n"(16384));
      }
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        if (i <= 26) asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(fl[i % 6]), "+v"(fh[i % 6]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fl[i % 6]), "+v"(fh[i % 6]));
        s16x8 v = __builtin_shufflevector(fl[i % 6], fh[i % 6], 0, 1, 2, 3, 4, 5, 6, 7);
        __builtin_amdgcn_sched_barrier(0);
        acc[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v), b, acc[i >> 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i + 6 < 32) {
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fl[i % 6]) : "v"(a8), "n"(512));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fh[i % 6]) : "v"(a8), "n"(16384 + 512));
        }
      }
    }
  } else if (MODE >= 3 && MODE != 7) {       // LDS read throughput, four waves of the CU streaming: 3 = ds_read_b64_tr_b16, 4 = ds_read_b128, 5 = ds_read_b64
    const unsigned a8 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + lane * 8 + (threadIdx.x >> 6) * 4096;
    const unsigned a16 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + lane * 16 + (threadIdx.x >> 6) * 4096;
    s16x4 r[8];
    typedef int i4 __attribute__((ext_vector_type(4)));
    i4 q[8];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (MODE == 3) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r[u]) : "v"(a8), "n"(0));
          if (MODE == 5) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[u]) : "v"(a8), "n"(0));
          if (MODE == 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[u]) : "v"(a16), "n"(0));
        }
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 4) acc[0][0] += (float)q[0][0]; else acc[0][0] += (float)r[0][0];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 16; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64);
  const int iters = 200;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<6>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
  }
  hipDeviceSynchronize();
  unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  const double n = 32.0 * iters;
  printf("cycles per MFMA: plain %.1f   with 2 tr reads + wait(0) per gap %.1f   with sched_barrier per MFMA %.1f\n", h[0] / n, h[1] / n, h[2] / n);
  const double nr = 64.0 * iters;
  printf("gen2_kernel's second-product pattern (MFMA + 2 tr reads 6 ahead): %.1f cycles per MFMA\n", h[6] / n);
  printf("cycles per LDS read wave-instruction (4 waves per CU streaming): tr_b64 %.2f   b128 %.2f   b64 %.2f\n", h[3] / nr, h[4] / nr, h[5] / nr);
  // sustained matrix-unit rate, every SIMD of the chip issuing back-to-back MFMAs for ~10 ms: small-integer operands against high-entropy ones
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int long_iters = 20000;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, 0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, long_iters);
      else hipLaunchKernelGGL(k<7>, dim3(256), dim3(256), 0, 0, out, cyc, long_iters);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
    }
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const double nm = 32.0 * long_iters, flop = 256.0 * 4 * nm * 32768.0;
    printf("%s operands: %.1f cycles per MFMA (s_memtime), %.2f ms for %.0f MFMAs per SIMD = %.0f TFLOP/s, clock %.2f GHz\n",
           mode == 0 ? "small-integer" : "high-entropy ", h[mode == 0 ? 0 : 7] / nm, ms, nm, flop / (ms * 1e-3) / 1e12, h[mode == 0 ? 0 : 7] / (ms * 1e-3) / 1e9);
  }
  return 0;
}

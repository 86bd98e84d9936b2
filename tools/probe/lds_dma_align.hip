// Does the LDS-DMA (global_load_lds, 16 bytes per lane) take a source address that is only 8- or 4-byte aligned?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/lds_dma_align tools/probe/lds_dma_align.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_cvoid;
__global__ void k(const char* src, int shift, unsigned* out) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x;
  // lane l reads 16 bytes at src + shift + l * 1000 (rows of 1000 bytes: every other row start is 8 mod 16)
  __builtin_amdgcn_global_load_lds((glb_cvoid*)(src + shift + (long)lane * 1000), (lds_void*)lds, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = reinterpret_cast<unsigned*>(lds)[lane * 4 + i];
}
int main() {
  const int N = 64 * 1000 + 64;
  std::vector<unsigned char> h(N);
  for (int i = 0; i < N; ++i) h[i] = (unsigned char)(i * 7 + 3);
  char* d; unsigned* o;
  hipMalloc(&d, N); hipMalloc(&o, 64 * 16);
  hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
  for (int shift : {0, 8, 4, 2}) {
    hipMemset(o, 0, 64 * 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, shift, o);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned char> r(64 * 16);
    hipMemcpy(r.data(), o, 64 * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int b = 0; b < 16; ++b) bad += r[l * 16 + b] != h[shift + l * 1000 + b];
    printf("shift %d: %s, %d wrong bytes of 1024\n", shift, hipGetErrorString(e), bad);
  }
  return 0;
}

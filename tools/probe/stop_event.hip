// What an event between two dependent kernels costs the stream: hipEventRecord (a barrier packet of its own) against the event attached to the
// producing kernel's dispatch packet (hipExtLaunchKernelGGL's stopEvent), with and without another stream waiting for it.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/stop_event tools/probe/stop_event.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void spin(float* p, int iters) {
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x] = v;
}
int main() {
  float *a, *b;
  CK(hipMalloc(&a, 4096)); CK(hipMalloc(&b, 4096));
  hipStream_t s0, s1;
  CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  const int N = 100;
  std::vector<hipEvent_t> ev(N);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  const int A = 20000, Bi = 2000;       // ~10 us and ~1 us
  for (int mode = 0; mode < 5; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, a, 4000000);      // ~2 ms in front: everything below is enqueued meanwhile
      CK(hipEventRecord(t0, s0));
      for (int i = 0; i < N; ++i) {
        if (mode == 3 || mode == 4) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, nullptr, ev[i], 0, a, A);
        else hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, a, A);
        if (mode == 1 || mode == 2) CK(hipEventRecord(ev[i], s0));
        if (mode == 2 || mode == 4) { CK(hipStreamWaitEvent(s1, ev[i], 0)); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, b, Bi); }
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s0, a, Bi);
      }
      CK(hipEventRecord(t1, s0));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      if (rep == 1) {
        const char* names[5] = {"no event", "hipEventRecord between", "hipEventRecord + other stream waits", "stopEvent on the kernel", "stopEvent + other stream waits"};
        printf("%-40s %.2f us per [10-us kernel ; 1-us kernel]\n", names[mode], ms * 1e3 / N);
      }
    }
  }
  return 0;
}

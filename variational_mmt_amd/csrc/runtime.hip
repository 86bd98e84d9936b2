// Stream helpers (gfx950): a HIP stream restricted to a subset of the compute units.
// The training step runs latency-critical LSTM step kernels on the main stream while bulk weight-gradient GEMMs run on a
// side stream; long-lived GEMM workgroups that fill all 256 CUs make the step kernels' workgroups wait for a free slot.
// A CU mask on the SIDE stream keeps a share of the CUs permanently free for the main stream.
#include "common.hpp"
#include "vmmt.h"

extern "C" int vmmt_stream_create_masked(const uint32_t* mask, int words, int priority, void** stream) {
  if (!mask || words <= 0 || !stream) return VMMT_EINVAL;
  hipStream_t s = nullptr;
  (void)priority;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
  if (e != hipSuccess) return VMMT_ELAUNCH;
  *stream = (void*)s;
  return VMMT_OK;
}

extern "C" int vmmt_stream_destroy(void* stream) {
  if (!stream) return VMMT_EINVAL;
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? VMMT_OK : VMMT_ELAUNCH;
}

// Diagnostic: where do a stream's workgroups land?  Every workgroup writes (XCC id | HW_ID << 8) to out[blockIdx.x] and stays resident for
// `hold_us` microseconds (so that the grid spreads over the CUs the stream may use instead of draining through the first few).
// tools/probe_cu_mask.py reads the mapping of hipExtStreamCreateWithCUMask's bits to XCDs / shader engines / CUs off it.
__global__ void probe_where_kernel(unsigned* __restrict__ out, int hold_us) {
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[blockIdx.x] = (xcc & 15u) | (hw << 8);
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)hold_us * 100ull) __builtin_amdgcn_s_sleep(8);
  }
}

extern "C" int vmmt_probe_where(uint32_t* out, int n_workgroups, int threads, int hold_us, void* stream) {
  if (!out || n_workgroups <= 0 || threads <= 0 || threads > 1024 || hold_us < 0 || hold_us > 100000) return VMMT_EINVAL;
  hipLaunchKernelGGL(probe_where_kernel, dim3((unsigned)n_workgroups), dim3((unsigned)threads), 0, (hipStream_t)stream, out, hold_us);
  return hipGetLastError() == hipSuccess ? VMMT_OK : VMMT_ELAUNCH;
}

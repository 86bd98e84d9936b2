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

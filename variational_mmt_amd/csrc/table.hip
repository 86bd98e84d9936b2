// Kernels of the image-feature table's way into HBM (the data side of the training step's row gather).
//   train_img_feats = (train_img_feats - train_mean[None,:]) / train_std[None,:]        train_mm_vi_model1.py:499-501
// The table is uploaded slab by slab (engine.load_image_table) and standardised in place on the device, so the host never
// holds a second copy of a multi-GB array.  Pure streaming: 8 B per element, HBM bound.
#include "common.hpp"
#include "vmmt.h"

namespace vmmt {

// One lane = four consecutive columns (16-byte load/store); a workgroup row-loops so mean/std stay in registers.
__global__ void standardise_rows_kernel(float* __restrict__ X, long ld, const float* __restrict__ mean,
                                        const float* __restrict__ stdv, long R, int D, int rows_per_block) {
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (c >= D) return;
  const bool vec = (c + 4 <= D) && ((ld & 3) == 0) && ((((uintptr_t)X) & 15) == 0);
  if (vec) {
    float m[4], s[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { m[e] = mean[c + e]; s[e] = stdv[c + e]; }
    for (long r = r0; r < r1; ++r) {
      f32x4* p = reinterpret_cast<f32x4*>(X + r * ld + c);
      f32x4 v = *p;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (v[e] - m[e]) / s[e];       // IEEE fp32 subtract + divide, as numpy does
      *p = v;
    }
  } else {
    for (int e = 0; e < 4 && c + e < D; ++e) {
      const float m = mean[c + e], s = stdv[c + e];
      for (long r = r0; r < r1; ++r) X[r * ld + c + e] = (X[r * ld + c + e] - m) / s;
    }
  }
}

}  // namespace vmmt
using namespace vmmt;
#define ST ((hipStream_t)stream)

extern "C" int vmmt_standardise_rows(float* X, int64_t ld, const float* mean, const float* stdv, int64_t R, int D, void* stream) {
  if (!X || !mean || !stdv || R < 0 || D <= 0 || ld < D) return VMMT_EINVAL;
  if (R == 0) return VMMT_OK;
  const int rpb = 16;
  const long gy = (R + rpb - 1) / rpb;
  if (gy > 0x7fffffffL) return VMMT_EINVAL;
  // grid.y is limited to 65535: fold the excess into more rows per workgroup
  int rows = rpb;
  long ny = gy;
  while (ny > 65535) { rows *= 2; ny = (R + rows - 1) / rows; }
  dim3 grid((D + 4 * 256 - 1) / (4 * 256), (unsigned)ny), block(256);
  hipLaunchKernelGGL(standardise_rows_kernel, grid, block, 0, ST, X, (long)ld, mean, stdv, (long)R, D, rows);
  return check_launch();
}

// Fused q(z|x) forward for gfx950: masked mean of the encoder memory -> both 2-layer MLPs (location, scale) -> sample -> KL,
// ONE launch.
//
// Reference: GlobalInferenceNetwork (onmt/modules/NormalVariationalEncoder.py:65-110: encode_seq = masked mean over time of
// the DETACHED memory, LocationLayer / ScaleLayer :12-43 = Linear-ReLU-Linear [-Softplus]), the sample z = mu + sigma * eps
// (onmt/Models.py:930-933, modules/Dists.py:21-26) and the KL against the standard normal prior (onmt/VILoss.py:446-456).
//
// Every sentence is independent through the whole chain, so a workgroup takes 16 sentences through all of it without meeting
// another workgroup: mean (f32 accumulation in time order, rounded once to bf16) -> h1 = relu(hbar W1^T + b1) -> out = h1 W2^T + b2
// -> mu | softplus -> z, KL.  The chain is a few hundred MFLOP of matrix work cut into four dependent GEMMs plus two
// element-wise kernels when launched separately: 104-135 us of launch latencies between the encoder and the decoder
// (profiles/r1_step_timeline.txt); here the 16 x K activations stay in LDS and the weights stream once per workgroup from L2.
//   MFMA: v_mfma_f32_16x16x32_bf16, A = the workgroup's 16 rows from LDS, B = weight rows (K contiguous) straight from global
//   memory in fragment shape; four waves split the output columns.
#include "common.hpp"
#include "vmmt.h"

namespace vmmt {

struct QnetArgs {
  const bf16_t* ctx; long ldc; const long long* lens;
  const bf16_t* w1[2]; long ldw1; const float* b1[2];      // [0] location, [1] scale; W1 [Z][ldw1] (k = H contiguous)
  const bf16_t* w2[2]; long ldw2; const float* b2[2];      // W2 [Z][ldw2]
  const float* eps;                                        // [B][Z]
  bf16_t* hbar; long ldh;                                  // out [B][ldh]   (input of the networks: saved for backward)
  bf16_t* h1[2]; long ldh1;                                // out [B][ldh1]  (hidden activations: saved for backward)
  float* mu; float* sigma; float* z32; bf16_t* zT; long ldz;
  float* kl_b; float* stats;
  int B, S, H, Z, Zv, training, split;                            // Z: tiled latent size; Zv <= Z: the model's (row stride of eps / mu / sigma / z32)
};

typedef float f32x4_q __attribute__((ext_vector_type(4)));

// C[16 x NT*16 per wave] = A_lds[16 x K] * W[n][K]^T for this wave's column range; returns accumulators acc[NT] (col = lane & 15,
// row = 4 * (lane >> 4) + reg)
// (K % 128 == 0.)  The weight fragments of FOUR k-steps are requested before the first of their MFMAs: a workgroup streams ~0.4 MB
// of weights per network, and with one k-step in flight every step paid a full L2 / HBM latency (80 us for the whole kernel).
template <int NT>
__device__ __forceinline__ void mm16(const bf16_t* a_lds, int pitch, const bf16_t* w, long ldw, int n0, int K, f32x4_q (&acc)[NT]) {
  const int lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = f32x4_q{0.f, 0.f, 0.f, 0.f};
  const bf16_t* wl = w + (long)(n0 + n) * ldw + kg * 8;
  u32x4 bv[2][4][NT];
  auto fetch = [&](int buf, int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[buf][q][j] = *reinterpret_cast<const u32x4*>(wl + (long)j * 16 * ldw + k0 + q * 32);
  };
  fetch(0, 0);
  for (int k0 = 0; k0 < K; k0 += 256) {
    if (k0 + 128 < K) fetch(1, k0 + 128);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bf16x8 av = *reinterpret_cast<const bf16x8*>(a_lds + n * pitch + k0 + q * 32 + kg * 8);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, bv[0][q][j]), acc[j], 0, 0, 0);
    }
    if (k0 + 128 >= K) break;
    if (k0 + 256 < K) fetch(0, k0 + 256);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bf16x8 av = *reinterpret_cast<const bf16x8*>(a_lds + n * pitch + k0 + 128 + q * 32 + kg * 8);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, bv[1][q][j]), acc[j], 0, 0, 0);
    }
  }
}

template <int NT>        // NT = Z / 128: 16-column tiles per wave (8 waves)
__global__ void __launch_bounds__(512) qnet_fwd_kernel(QnetArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = a.H, Z = a.Z, Zv = a.Zv, B = a.B;
  const int PH = H + 8, PZ = Z + 8;
  bf16_t* hb = reinterpret_cast<bf16_t*>(smem);                       // [16][PH]
  bf16_t* h1s = hb + 16 * PH;                                         // [16][PZ]
  float* outs = reinterpret_cast<float*>(h1s + 16 * PZ);              // [2][16][Z]  mu | sigma
  const int r0 = blockIdx.x * 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
  // ---- 1. masked mean (time order, f32, one rounding): thread -> (row, 8-column chunk, chunks strided by 256)
  {
    const int r = threadIdx.x >> 5, cb = (threadIdx.x & 31) * 8;
    const int b = min(r0 + r, B - 1);
    int len = (int)a.lens[b];
    len = len < a.S ? len : a.S;
    for (int c0 = cb; c0 < H; c0 += 256) {
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
      // eight positions per round trip (the sum itself stays in time order: same bits as vmmt_masked_mean); one load per trip
      // made this phase ~40 us of back-to-back memory latencies
      for (int s0 = 0; s0 < len; s0 += 8) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int sp = s0 + u < len ? s0 + u : len - 1;
          v[u] = *reinterpret_cast<const u32x4*>(a.ctx + ((long)sp * B + b) * a.ldc + c0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (s0 + u < len) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc[2 * e] += __uint_as_float(v[u][e] << 16);
              acc[2 * e + 1] += __uint_as_float(v[u][e] & 0xffff0000u);
            }
          }
        }
      }
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = (unsigned)f2bf(acc[2 * e] / (float)len) | ((unsigned)f2bf(acc[2 * e + 1] / (float)len) << 16);
      *reinterpret_cast<u32x4*>(hb + r * PH + c0) = o;
      if (r0 + r < B && (!a.split || blockIdx.y == 0)) *reinterpret_cast<u32x4*>(a.hbar + (long)(r0 + r) * a.ldh + c0) = o;
    }
  }
  __syncthreads();
  const int n0 = wave * (NT * 16);
  // split: the location and the scale network of a sentence group run in two workgroups (blockIdx.y) on two CUs -- the chain is
  // latency-bound (each network streams its ~1 MB of weights through one workgroup), so the two halves take half the time; mu / sigma
  // then go to memory and vmmt_latent_fwd draws the sample (one more small launch).  Same bits in mu / sigma either way.
  const int br_lo = a.split ? (int)blockIdx.y : 0, br_hi = a.split ? (int)blockIdx.y + 1 : 2;
  for (int br = br_lo; br < br_hi; ++br) {
    // ---- 2. h1 = relu(hbar W1^T + b1)
    f32x4_q acc[NT];
    mm16<NT>(hb, PH, a.w1[br], a.ldw1, n0, H, acc);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + j * 16 + n;
      const float bias = col < Zv ? a.b1[br][col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = kg * 4 + r;
        const float v = fmaxf(acc[j][r] + bias, 0.f);
        const bf16_t hv = f2bf(v);
        h1s[row * PZ + col] = hv;
        if (r0 + row < B) a.h1[br][(long)(r0 + row) * a.ldh1 + col] = hv;
      }
    }
    __syncthreads();
    // ---- 3. out = h1 W2^T + b2 ; location: identity, scale: nn.Softplus(beta 1, threshold 20)
    mm16<NT>(h1s, PZ, a.w2[br], a.ldw2, n0, Z, acc);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + j * 16 + n;
      const float bias = col < Zv ? a.b2[br][col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = kg * 4 + r;
        float v = acc[j][r] + bias;
        if (br == 1) v = v > 20.f ? v : log1pf(__expf(v));
        if (a.split) {
          if (r0 + row < B && col < Zv) (br == 0 ? a.mu : a.sigma)[(long)(r0 + row) * Zv + col] = v;
        } else {
          outs[(br * 16 + row) * Z + col] = v;
        }
      }
    }
    __syncthreads();            // h1s is rewritten by the next branch; outs is read below
  }
  if (a.split) return;
  // ---- 4. z = mu + sigma * eps (training) | mu ; KL_b = sum_k 0.5 (mu^2 + sigma^2 - 1) - log sigma : wave w -> rows 2w, 2w+1
  for (int rr = 0; rr < 2; ++rr) {
    const int row = wave * 2 + rr, b = r0 + row;
    if (b >= B) continue;
    float kl = 0.f;
    for (int k = lane; k < Zv; k += 64) {
      const float m = outs[row * Z + k], s = outs[(16 + row) * Z + k];
      const float z = a.training ? m + s * a.eps[(long)b * Zv + k] : m;
      a.mu[(long)b * Zv + k] = m;
      a.sigma[(long)b * Zv + k] = s;
      a.z32[(long)b * Zv + k] = z;
      a.zT[(long)b * a.ldz + k] = f2bf(z);
      kl += 0.5f * (m * m + s * s - 1.f) - logf(s);
    }
    kl = wave_sum(kl);
    if (lane == 0) {
      a.kl_b[b] = kl;
      atomicAdd(a.stats + VMMT_STAT_KL_SUM, kl);
    }
  }
}

}  // namespace vmmt

// dtype must be VMMT_BF16; H % 256 == 0 and Z % 128 == 0 (8 waves x 16-column tiles, K in chunks of 128), LDS <= 64 KiB; all row
// starts 16-byte aligned.  Returns VMMT_EINVAL otherwise (the caller then issues the separate kernels).
extern "C" int vmmt_qnet_fwd(int dtype, const void* ctx, int64_t ldc, const int64_t* lens, const void* w1_loc, const void* w1_scale,
                             int64_t ldw1, const float* b1_loc, const float* b1_scale, const void* w2_loc, const void* w2_scale,
                             int64_t ldw2, const float* b2_loc, const float* b2_scale, const float* eps, void* hbar, int64_t ldh,
                             void* h1_loc, void* h1_scale, int64_t ldh1, float* mu, float* sigma, float* z32, void* zT, int64_t ldz,
                             float* kl_b, float* stats, int B, int S, int H, int Z, int Z_valid, int training, int split, void* stream) {
  using namespace vmmt;
  if (dtype != VMMT_BF16 || !ctx || !lens || !w1_loc || !w1_scale || !b1_loc || !b1_scale || !w2_loc || !w2_scale || !b2_loc || !b2_scale ||
      !hbar || !h1_loc || !h1_scale || !mu || !sigma || !z32 || !zT || !kl_b || !stats || (training && !eps) || B <= 0 || S <= 0)
    return VMMT_EINVAL;
  if (H % 256 != 0 || Z % 128 != 0 || Z > 512 || Z_valid <= 0 || Z_valid > Z) return VMMT_EINVAL;
  const uintptr_t al = (uintptr_t)ctx | (uintptr_t)w1_loc | (uintptr_t)w1_scale | (uintptr_t)w2_loc | (uintptr_t)w2_scale | (uintptr_t)hbar;
  if ((al & 15) || ldc % 8 || ldw1 % 8 || ldw2 % 8 || ldh % 8) return VMMT_EINVAL;
  QnetArgs a;
  a.ctx = (const bf16_t*)ctx; a.ldc = ldc; a.lens = (const long long*)lens;
  a.w1[0] = (const bf16_t*)w1_loc; a.w1[1] = (const bf16_t*)w1_scale; a.ldw1 = ldw1; a.b1[0] = b1_loc; a.b1[1] = b1_scale;
  a.w2[0] = (const bf16_t*)w2_loc; a.w2[1] = (const bf16_t*)w2_scale; a.ldw2 = ldw2; a.b2[0] = b2_loc; a.b2[1] = b2_scale;
  a.eps = eps; a.hbar = (bf16_t*)hbar; a.ldh = ldh; a.h1[0] = (bf16_t*)h1_loc; a.h1[1] = (bf16_t*)h1_scale; a.ldh1 = ldh1;
  a.mu = mu; a.sigma = sigma; a.z32 = z32; a.zT = (bf16_t*)zT; a.ldz = ldz; a.kl_b = kl_b; a.stats = stats;
  a.B = B; a.S = S; a.H = H; a.Z = Z; a.Zv = Z_valid; a.training = training; a.split = split ? 1 : 0;
  const size_t sm = (size_t)16 * (H + 8) * 2 + (size_t)16 * (Z + 8) * 2 + (size_t)2 * 16 * Z * 4;
  if (sm > 128 * 1024) return VMMT_EINVAL;
  const dim3 grid((B + 15) / 16, split ? 2 : 1);
  // (Z = 512 needs 100 KiB of LDS: above the 64-KiB default the limit is raised once per instantiation)
#define VMMT_QNET_LAUNCH(NT)                                                                                                    \
  {                                                                                                                             \
    static size_t allowed = 64 * 1024;                                                                                          \
    if (sm > allowed) {                                                                                                         \
      if (hipFuncSetAttribute((const void*)qnet_fwd_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) \
        return VMMT_ELAUNCH;                                                                                                    \
      allowed = sm;                                                                                                             \
    }                                                                                                                           \
    hipLaunchKernelGGL(qnet_fwd_kernel<NT>, grid, dim3(512), sm, (hipStream_t)stream, a);                                       \
  }
  switch (Z / 128) {
    case 1: VMMT_QNET_LAUNCH(1) break;
    case 2: VMMT_QNET_LAUNCH(2) break;
    case 3: VMMT_QNET_LAUNCH(3) break;
    case 4: VMMT_QNET_LAUNCH(4) break;
    default: return VMMT_EINVAL;
  }
#undef VMMT_QNET_LAUNCH
  return check_launch();
}

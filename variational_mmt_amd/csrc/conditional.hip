// Kernels that only the conditional-prior variant (`--conditional`) needs.
//   p(z|x)     = gen_net_global(mean_s context)                      onmt/Models.py:889
//   q(z|x,y,v) = GlobalFullInferenceNetwork([h_x ; h_y ; v])         onmt/modules/NormalVariationalEncoder.py:164-228
//   KL between two diagonal Gaussians                                onmt/VILoss.py:437-452
// Row layouts: "time-major" = row s*B + b (the encoder memory), "batch-major" = row b*T + t (the output of encoder_tgt, which
// the reference runs over the transposed target, hazard H5: its recurrence walks the BATCH axis).
#include "common.hpp"
#include "vmmt.h"

namespace vmmt {

// z = mu + sigma * eps (training: a sample of q) | mu_p (evaluation: the mean of p, Models.py:913);
// kl_b[b] = sum_k ((mu-mu_p)^2 + sigma^2 - sigma_p^2) / (2 sigma_p^2) + log sigma_p - log sigma; stats[KL_SUM] += sum_b kl_b
template <class T>
__global__ void latent_cond_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ sigma, const float* __restrict__ mu_p,
                                       const float* __restrict__ sigma_p, const float* __restrict__ eps, float* __restrict__ z32,
                                       T* __restrict__ zT, long ldz, float* __restrict__ kl_b, float* __restrict__ stats, int B, int Z,
                                       int training) {
  int b = blockIdx.x;
  float kl = 0.f;
  for (int k = threadIdx.x; k < Z; k += blockDim.x) {
    const long i = (long)b * Z + k;
    float m = mu[i], s = sigma[i], mp = mu_p[i], sp = sigma_p[i];
    float z = training ? m + s * eps[i] : mp;
    z32[i] = z;
    zT[(long)b * ldz + k] = from_f<T>(z);
    float d = m - mp;
    kl += (d * d + s * s - sp * sp) / (2.f * sp * sp) + logf(sp) - logf(s);
  }
  kl = wave_sum(kl);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = kl;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    kl_b[b] = t;
    atomicAdd(stats + VMMT_STAT_KL_SUM, t);
  }
}

// gradient of max(mult * KL_mean, margin) * inv_norm w.r.t. the outputs of the four MLPs (locations, pre-softplus scales)
template <class T>
__global__ void latent_cond_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ sigma, const float* __restrict__ mu_p,
                                       const float* __restrict__ sigma_p, const float* __restrict__ kl_sum, float batch_global,
                                       float mult, int use_freebits, float margin, float inv_norm, const float* __restrict__ dz,
                                       const float* __restrict__ eps, T* __restrict__ dmu, long ld1,
                                       T* __restrict__ dpre, long ld2, T* __restrict__ dmu_p, long ld3, T* __restrict__ dpre_p, long ld4,
                                       int B, int Z) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * Z) return;
  int b = i / Z, k = i - (long)b * Z;
  float gs = mult / batch_global * inv_norm;
  if (use_freebits && mult * (*kl_sum) / batch_global < margin) gs = 0.f;
  float m = mu[i], s = sigma[i], mp = mu_p[i], sp = sigma_p[i];
  float d = m - mp, isp2 = 1.f / (sp * sp);
  float gm = gs * d * isp2, gsg = gs * (s * isp2 - 1.f / s);
  if (dz) { gm += dz[i]; gsg += dz[i] * eps[i]; }               // reparameterised sample not detached (see latent_bwd_kernel)
  dmu[(long)b * ld1 + k] = from_f<T>(gm);
  dpre[(long)b * ld2 + k] = from_f<T>(gsg * (1.f - __expf(-s)));          // d softplus = 1 - exp(-y)
  dmu_p[(long)b * ld3 + k] = from_f<T>(-gs * d * isp2);
  dpre_p[(long)b * ld4 + k] = from_f<T>(gs * (1.f / sp - (d * d + s * s) * isp2 / sp) * (1.f - __expf(-sp)));
}

// batch-major masked mean: out[b] = sum_{t < len_b} x[b*T + t] / len_b
template <class T>
__global__ void masked_mean_bm_kernel(const T* __restrict__ x, long ldx, const long long* __restrict__ lens, T* __restrict__ out,
                                      long ldo, int B, int Tn, int H) {
  int b = blockIdx.x;
  int len = (int)lens[b];
  len = len < Tn ? len : Tn;
  for (int h = threadIdx.x; h < H; h += blockDim.x) {
    float a = 0.f;
    for (int t = 0; t < len; ++t) a += to_f<T>(x[((long)b * Tn + t) * ldx + h]);
    out[(long)b * ldo + h] = from_f<T>(a / (float)len);
  }
}

// backward of either masked mean: dx[row(s, b)] (+)= s < len_b ? dh[b] / len_b : 0
template <class T>
__global__ void masked_mean_bwd_kernel(const T* __restrict__ dh, long lddh, const long long* __restrict__ lens, T* __restrict__ dx,
                                       long lddx, int B, int S, int H, int batch_major, int accumulate) {
  const int b = blockIdx.x, s = blockIdx.y;
  int len = (int)lens[b];
  len = len < S ? len : S;
  const long row = batch_major ? (long)b * S + s : (long)s * B + b;
  const float sc = s < len ? 1.f / (float)len : 0.f;
  for (int h = threadIdx.x; h < H; h += blockDim.x) {
    float v = to_f<T>(dh[(long)b * lddh + h]) * sc;
    T* p = dx + row * lddx + h;
    if (accumulate) { if (sc != 0.f) *p = from_f<T>(to_f<T>(*p) + v); }
    else *p = from_f<T>(v);
  }
}

}  // namespace vmmt

#define ST (hipStream_t) stream

extern "C" int vmmt_latent_cond_fwd(int dtype, const float* mu, const float* sigma, const float* mu_p, const float* sigma_p,
                                    const float* eps, float* z32, void* zT, int64_t ldz, float* kl_b, float* stats, int B, int Z,
                                    int training, void* stream) {
  using namespace vmmt;
  if (!mu || !sigma || !mu_p || !sigma_p || !z32 || !zT || !kl_b || !stats || B <= 0 || Z <= 0 || (training && !eps)) return VMMT_EINVAL;
  if (dtype == VMMT_F32) hipLaunchKernelGGL(latent_cond_fwd_kernel<float>, dim3(B), dim3(256), 0, ST, mu, sigma, mu_p, sigma_p, eps, z32, (float*)zT, (long)ldz, kl_b, stats, B, Z, training);
  else if (dtype == VMMT_BF16) hipLaunchKernelGGL(latent_cond_fwd_kernel<bf16_t>, dim3(B), dim3(256), 0, ST, mu, sigma, mu_p, sigma_p, eps, z32, (bf16_t*)zT, (long)ldz, kl_b, stats, B, Z, training);
  else return VMMT_EINVAL;
  return check_launch();
}

extern "C" int vmmt_latent_cond_bwd(int dtype, const float* mu, const float* sigma, const float* mu_p, const float* sigma_p,
                                    const float* kl_sum, float batch_global, float mult, int use_freebits, float margin,
                                    float inv_norm, const float* dz, const float* eps, void* dmu, int64_t ld1, void* dpre, int64_t ld2,
                                    void* dmu_p, int64_t ld3, void* dpre_p, int64_t ld4, int B, int Z, void* stream) {
  using namespace vmmt;
  if (!mu || !sigma || !mu_p || !sigma_p || !kl_sum || !dmu || !dpre || !dmu_p || !dpre_p || B <= 0 || Z <= 0 || (dz && !eps)) return VMMT_EINVAL;
  long n = (long)B * Z;
  dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == VMMT_F32) hipLaunchKernelGGL(latent_cond_bwd_kernel<float>, grid, dim3(256), 0, ST, mu, sigma, mu_p, sigma_p, kl_sum, batch_global, mult, use_freebits, margin, inv_norm, dz, eps, (float*)dmu, (long)ld1, (float*)dpre, (long)ld2, (float*)dmu_p, (long)ld3, (float*)dpre_p, (long)ld4, B, Z);
  else if (dtype == VMMT_BF16) hipLaunchKernelGGL(latent_cond_bwd_kernel<bf16_t>, grid, dim3(256), 0, ST, mu, sigma, mu_p, sigma_p, kl_sum, batch_global, mult, use_freebits, margin, inv_norm, dz, eps, (bf16_t*)dmu, (long)ld1, (bf16_t*)dpre, (long)ld2, (bf16_t*)dmu_p, (long)ld3, (bf16_t*)dpre_p, (long)ld4, B, Z);
  else return VMMT_EINVAL;
  return check_launch();
}

extern "C" int vmmt_masked_mean_bm(int dtype, const void* x, int64_t ldx, const int64_t* lens, void* out, int64_t ldo, int B, int T,
                                   int H, void* stream) {
  using namespace vmmt;
  if (!x || !lens || !out || B <= 0 || T <= 0 || H <= 0) return VMMT_EINVAL;
  if (dtype == VMMT_F32) hipLaunchKernelGGL(masked_mean_bm_kernel<float>, dim3(B), dim3(256), 0, ST, (const float*)x, (long)ldx, (const long long*)lens, (float*)out, (long)ldo, B, T, H);
  else if (dtype == VMMT_BF16) hipLaunchKernelGGL(masked_mean_bm_kernel<bf16_t>, dim3(B), dim3(256), 0, ST, (const bf16_t*)x, (long)ldx, (const long long*)lens, (bf16_t*)out, (long)ldo, B, T, H);
  else return VMMT_EINVAL;
  return check_launch();
}

extern "C" int vmmt_masked_mean_bwd(int dtype, const void* dh, int64_t lddh, const int64_t* lens, void* dx, int64_t lddx, int B, int S,
                                    int H, int batch_major, int accumulate, void* stream) {
  using namespace vmmt;
  if (!dh || !lens || !dx || B <= 0 || S <= 0 || H <= 0) return VMMT_EINVAL;
  dim3 grid(B, S);
  if (dtype == VMMT_F32) hipLaunchKernelGGL(masked_mean_bwd_kernel<float>, grid, dim3(128), 0, ST, (const float*)dh, (long)lddh, (const long long*)lens, (float*)dx, (long)lddx, B, S, H, batch_major, accumulate);
  else if (dtype == VMMT_BF16) hipLaunchKernelGGL(masked_mean_bwd_kernel<bf16_t>, grid, dim3(128), 0, ST, (const bf16_t*)dh, (long)lddh, (const long long*)lens, (bf16_t*)dx, (long)lddx, B, S, H, batch_major, accumulate);
  else return VMMT_EINVAL;
  return check_launch();
}

// Global (Luong "general") attention, forward and backward, for gfx950.
// One workgroup per sentence: the sentence's source memory Hs[S][H] is staged once in LDS and reused by all T'
// queries (reference: two bmm + masked_fill + softmax + bmm per call, onmt/modules/GlobalAttention.py:113,171-184).
//   score[t][s] = q'[t] . Hs[s]   (q' = W_a r already applied by a GEMM, GlobalAttention.py:106-110)
//   a = softmax over s < len ; c[t] = sum_s a[t][s] Hs[s]
// The context c is written straight into the left half of the [c ; r] concat buffer that feeds linear_out
// (GlobalAttention.py:187), so the torch.cat is never materialised separately.
#include "common.hpp"
#include "vmmt.h"

namespace vmmt {

constexpr int ATT_MAXJ = 16;  // H <= 1024
constexpr int ATT_MAXS = 64;  // S <= 64 (one score per lane)

template <class T>
__device__ __forceinline__ const T* stage_hs(const T* __restrict__ ctx, long ldc, int b, int B, int S, int H,
                                             T* lds, bool use_lds, long* stride) {
  if (!use_lds) { *stride = (long)B * ldc; return ctx + (long)b * ldc; }
  for (int i = threadIdx.x; i < S * H; i += blockDim.x) {
    int s = i / H, h = i - s * H;
    lds[i] = ctx[((long)s * B + b) * ldc + h];
  }
  __syncthreads();
  *stride = H;
  return lds;
}

template <class T>
__global__ void __launch_bounds__(256) attn_fwd_kernel(const T* __restrict__ q, long ldq, const T* __restrict__ ctx,
                                                       long ldc, const long long* __restrict__ lens,
                                                       T* __restrict__ cat, long ldcat, float* __restrict__ probs,
                                                       int Tp, int B, int S, int H, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long hs_stride;
  const T* hs = stage_hs<T>(ctx, ldc, b, B, S, H, reinterpret_cast<T*>(smem_raw), use_lds != 0, &hs_stride);
  int len = (int)lens[b];
  len = len < S ? len : S;
  for (int t = wave; t < Tp; t += 4) {
    const T* qr = q + ((long)t * B + b) * ldq;
    float my = -INFINITY;
    for (int s = 0; s < len; ++s) {
      float part = 0.f;
      for (int h = lane; h < H; h += 64) part += to_f<T>(qr[h]) * to_f<T>(hs[s * hs_stride + h]);
      part = wave_sum(part);
      if (lane == s) my = part;
    }
    float m = wave_max(my);
    float e = lane < len ? __expf(my - m) : 0.f;
    float p = e / wave_sum(e);
    if (lane < S) probs[((long)t * B + b) * S + lane] = p;
    float acc[ATT_MAXJ];
#pragma unroll
    for (int j = 0; j < ATT_MAXJ; ++j) acc[j] = 0.f;
    for (int s = 0; s < len; ++s) {
      float ps = __shfl(p, s, 64);
#pragma unroll
      for (int j = 0; j < ATT_MAXJ; ++j) {
        int h = lane + 64 * j;
        if (h < H) acc[j] += ps * to_f<T>(hs[s * hs_stride + h]);
      }
    }
    T* cr = cat + ((long)t * B + b) * ldcat;
#pragma unroll
    for (int j = 0; j < ATT_MAXJ; ++j) {
      int h = lane + 64 * j;
      if (h < H) cr[h] = from_f<T>(acc[j]);
    }
  }
}

// backward: dc (left half of dcat) -> da -> ds (softmax backward) -> dq' ; dHs[s] = sum_t a[t][s] dc[t] + ds[t][s] q'[t]
template <class T>
__global__ void __launch_bounds__(256) attn_bwd_kernel(const T* __restrict__ dcat, long lddc, const float* __restrict__ probs,
                                                       const T* __restrict__ q, long ldq, const T* __restrict__ ctx,
                                                       long ldc, const long long* __restrict__ lens,
                                                       T* __restrict__ dq, long lddq, T* __restrict__ dctx, long lddx,
                                                       int Tp, int B, int S, int H, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* pbuf = reinterpret_cast<float*>(smem_raw);            // [Tp][64]
  float* dsbuf = pbuf + (long)Tp * ATT_MAXS;                   // [Tp][64]
  T* hs_lds = reinterpret_cast<T*>(dsbuf + (long)Tp * ATT_MAXS);
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long hs_stride;
  const T* hs = stage_hs<T>(ctx, ldc, b, B, S, H, hs_lds, use_lds != 0, &hs_stride);
  int len = (int)lens[b];
  len = len < S ? len : S;
  for (int t = wave; t < Tp; t += 4) {
    const T* dcr = dcat + ((long)t * B + b) * lddc;
    float da = 0.f;
    for (int s = 0; s < len; ++s) {
      float part = 0.f;
      for (int h = lane; h < H; h += 64) part += to_f<T>(dcr[h]) * to_f<T>(hs[s * hs_stride + h]);
      part = wave_sum(part);
      if (lane == s) da = part;
    }
    float p = lane < S ? probs[((long)t * B + b) * S + lane] : 0.f;
    float dot = wave_sum(p * da);
    float ds = p * (da - dot);
    pbuf[t * ATT_MAXS + lane] = p;
    dsbuf[t * ATT_MAXS + lane] = ds;
    float acc[ATT_MAXJ];
#pragma unroll
    for (int j = 0; j < ATT_MAXJ; ++j) acc[j] = 0.f;
    for (int s = 0; s < len; ++s) {
      float dss = __shfl(ds, s, 64);
#pragma unroll
      for (int j = 0; j < ATT_MAXJ; ++j) {
        int h = lane + 64 * j;
        if (h < H) acc[j] += dss * to_f<T>(hs[s * hs_stride + h]);
      }
    }
    T* dqr = dq + ((long)t * B + b) * lddq;
#pragma unroll
    for (int j = 0; j < ATT_MAXJ; ++j) {
      int h = lane + 64 * j;
      if (h < H) dqr[h] = from_f<T>(acc[j]);
    }
  }
  __syncthreads();
  for (int h = threadIdx.x; h < H; h += 256) {
    float acc[ATT_MAXS];
#pragma unroll
    for (int s = 0; s < ATT_MAXS; ++s) acc[s] = 0.f;
    for (int t = 0; t < Tp; ++t) {
      float x = to_f<T>(dcat[((long)t * B + b) * lddc + h]);
      float y = to_f<T>(q[((long)t * B + b) * ldq + h]);
#pragma unroll
      for (int s = 0; s < ATT_MAXS; ++s) acc[s] += pbuf[t * ATT_MAXS + s] * x + dsbuf[t * ATT_MAXS + s] * y;
    }
#pragma unroll
    for (int s = 0; s < ATT_MAXS; ++s)
      if (s < S) dctx[((long)s * B + b) * lddx + h] = from_f<T>(s < len ? acc[s] : 0.f);
  }
}

// masked mean over time of the (detached) encoder memory: hbar[b] = sum_{s<len} ctx[s][b] / len
// (GlobalInferenceNetwork.encode_seq, onmt/modules/NormalVariationalEncoder.py:65-84)
template <class T>
__global__ void masked_mean_kernel(const T* __restrict__ ctx, long ldc, const long long* __restrict__ lens,
                                   T* __restrict__ out, long ldo, int B, int S, int H) {
  int b = blockIdx.x;
  int len = (int)lens[b];
  len = len < S ? len : S;
  for (int h = threadIdx.x; h < H; h += blockDim.x) {
    float a = 0.f;
    for (int s = 0; s < len; ++s) a += to_f<T>(ctx[((long)s * B + b) * ldc + h]);
    out[(long)b * ldo + h] = from_f<T>(a / (float)len);
  }
}

}  // namespace vmmt

extern "C" int vmmt_attn_fwd(int dtype, const void* q, int64_t ldq, const void* ctx, int64_t ldc, const int64_t* lens,
                             void* cat, int64_t ldcat, float* probs, int Tp, int B, int S, int H, void* stream) {
  using namespace vmmt;
  if (!q || !ctx || !lens || !cat || !probs || S > ATT_MAXS || H > 64 * ATT_MAXJ || Tp <= 0 || B <= 0)
    return VMMT_EINVAL;
  size_t esz = dtype == VMMT_F32 ? 4 : 2;
  size_t need = (size_t)S * H * esz;
  int use_lds = need <= 60 * 1024;   // default dynamic-LDS limit without hipFuncSetAttribute
  size_t sm = use_lds ? need : 16;
  if (dtype == VMMT_F32)
    hipLaunchKernelGGL(attn_fwd_kernel<float>, dim3(B), dim3(256), sm, (hipStream_t)stream, (const float*)q, (long)ldq,
                       (const float*)ctx, (long)ldc, (const long long*)lens, (float*)cat, (long)ldcat, probs, Tp, B, S,
                       H, use_lds);
  else if (dtype == VMMT_BF16)
    hipLaunchKernelGGL(attn_fwd_kernel<bf16_t>, dim3(B), dim3(256), sm, (hipStream_t)stream, (const bf16_t*)q,
                       (long)ldq, (const bf16_t*)ctx, (long)ldc, (const long long*)lens, (bf16_t*)cat, (long)ldcat,
                       probs, Tp, B, S, H, use_lds);
  else return VMMT_EINVAL;
  return check_launch();
}

extern "C" int vmmt_attn_bwd(int dtype, const void* dcat, int64_t lddc, const float* probs, const void* q, int64_t ldq,
                             const void* ctx, int64_t ldc, const int64_t* lens, void* dq, int64_t lddq, void* dctx,
                             int64_t lddx, int Tp, int B, int S, int H, void* stream) {
  using namespace vmmt;
  if (!dcat || !probs || !q || !ctx || !lens || !dq || !dctx || S > ATT_MAXS || H > 64 * ATT_MAXJ || Tp <= 0 || B <= 0)
    return VMMT_EINVAL;
  size_t esz = dtype == VMMT_F32 ? 4 : 2;
  size_t base = (size_t)2 * Tp * ATT_MAXS * sizeof(float);
  size_t need = (size_t)S * H * esz;
  int use_lds = base + need <= 60 * 1024;
  size_t sm = base + (use_lds ? need : 0);
  if (base > 60 * 1024) return VMMT_EINVAL;
  if (dtype == VMMT_F32)
    hipLaunchKernelGGL(attn_bwd_kernel<float>, dim3(B), dim3(256), sm, (hipStream_t)stream, (const float*)dcat,
                       (long)lddc, probs, (const float*)q, (long)ldq, (const float*)ctx, (long)ldc,
                       (const long long*)lens, (float*)dq, (long)lddq, (float*)dctx, (long)lddx, Tp, B, S, H, use_lds);
  else if (dtype == VMMT_BF16)
    hipLaunchKernelGGL(attn_bwd_kernel<bf16_t>, dim3(B), dim3(256), sm, (hipStream_t)stream, (const bf16_t*)dcat,
                       (long)lddc, probs, (const bf16_t*)q, (long)ldq, (const bf16_t*)ctx, (long)ldc,
                       (const long long*)lens, (bf16_t*)dq, (long)lddq, (bf16_t*)dctx, (long)lddx, Tp, B, S, H,
                       use_lds);
  else return VMMT_EINVAL;
  return check_launch();
}

extern "C" int vmmt_masked_mean(int dtype, const void* ctx, int64_t ldc, const int64_t* lens, void* out, int64_t ldo,
                                int B, int S, int H, void* stream) {
  using namespace vmmt;
  if (!ctx || !lens || !out || B <= 0) return VMMT_EINVAL;
  if (dtype == VMMT_F32)
    hipLaunchKernelGGL(masked_mean_kernel<float>, dim3(B), dim3(256), 0, (hipStream_t)stream, (const float*)ctx,
                       (long)ldc, (const long long*)lens, (float*)out, (long)ldo, B, S, H);
  else if (dtype == VMMT_BF16)
    hipLaunchKernelGGL(masked_mean_kernel<bf16_t>, dim3(B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)ctx,
                       (long)ldc, (const long long*)lens, (bf16_t*)out, (long)ldo, B, S, H);
  else return VMMT_EINVAL;
  return check_launch();
}

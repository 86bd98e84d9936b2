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

// --------------------------------------------------------------------------------------------------------------
// Long sequences (64 < S <= 256, any T'): the kernels above give every source position a lane (or a 32 / 64-row MFMA tile).  Training
// data is cut at 50 tokens by the reference's preprocess defaults, but a sentence handed to translate_mm_vi.py can be any length: these
// kernels are the fallback -- one WAVE per (sentence, query), lanes hold the positions s = lane + 64 k (k < 4) and the hidden units
// h = lane + 64 j (j < 16); operands straight from global memory / L2.  Same arithmetic as attn_fwd_kernel / attn_bwd_kernel.
constexpr int ATT_LK = 4;     // S <= 64 * ATT_LK

template <class T>
__global__ void __launch_bounds__(256) attn_fwd_long(const T* __restrict__ q, long ldq, const T* __restrict__ ctx, long ldc,
                                                     const long long* __restrict__ lens, T* __restrict__ cat, long ldcat,
                                                     float* __restrict__ probs, int Tp, int B, int S, int H) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + wave;             // (t, b) pair: row = t * B + b
  if (row >= (long)Tp * B) return;
  const int b = (int)(row % B);
  int len = (int)lens[b];
  len = len < S ? len : S;
  const T* qr = q + row * ldq;
  float qv[ATT_MAXJ];
#pragma unroll
  for (int j = 0; j < ATT_MAXJ; ++j) qv[j] = lane + 64 * j < H ? to_f<T>(qr[lane + 64 * j]) : 0.f;
  float my[ATT_LK];
#pragma unroll
  for (int k = 0; k < ATT_LK; ++k) my[k] = -INFINITY;
  for (int s = 0; s < len; ++s) {
    const T* hr = ctx + ((long)s * B + b) * ldc;
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < ATT_MAXJ; ++j)
      if (lane + 64 * j < H) part += qv[j] * to_f<T>(hr[lane + 64 * j]);
    part = wave_sum(part);
#pragma unroll
    for (int k = 0; k < ATT_LK; ++k)
      if ((s >> 6) == k && lane == (s & 63)) my[k] = part;
  }
  float m = my[0];
#pragma unroll
  for (int k = 1; k < ATT_LK; ++k) m = fmaxf(m, my[k]);
  m = wave_max(m);
  float pk[ATT_LK], sum = 0.f;
#pragma unroll
  for (int k = 0; k < ATT_LK; ++k) { pk[k] = lane + 64 * k < len ? __expf(my[k] - m) : 0.f; sum += pk[k]; }
  sum = wave_sum(sum);
#pragma unroll
  for (int k = 0; k < ATT_LK; ++k) {
    pk[k] = pk[k] / sum;
    if (lane + 64 * k < S) probs[row * S + lane + 64 * k] = pk[k];
  }
  float acc[ATT_MAXJ];
#pragma unroll
  for (int j = 0; j < ATT_MAXJ; ++j) acc[j] = 0.f;
  for (int s = 0; s < len; ++s) {
    float ps = 0.f;
#pragma unroll
    for (int k = 0; k < ATT_LK; ++k) { const float v = __shfl(pk[k], s & 63, 64); ps = (s >> 6) == k ? v : ps; }
    const T* hr = ctx + ((long)s * B + b) * ldc;
#pragma unroll
    for (int j = 0; j < ATT_MAXJ; ++j)
      if (lane + 64 * j < H) acc[j] += ps * to_f<T>(hr[lane + 64 * j]);
  }
  T* cr = cat + row * ldcat;
#pragma unroll
  for (int j = 0; j < ATT_MAXJ; ++j)
    if (lane + 64 * j < H) cr[lane + 64 * j] = from_f<T>(acc[j]);
}

// backward, first half: per (sentence, query) dP = dC Hs^T, dot = sum_s P dP (kept in `dots` for the second half), dS = P (dP - dot),
// dQ = dS Hs
template <class T>
__global__ void __launch_bounds__(256) attn_bwd_long_q(const T* __restrict__ dcat, long lddc, const float* __restrict__ probs,
                                                       const T* __restrict__ ctx, long ldc, const long long* __restrict__ lens,
                                                       T* __restrict__ dq, long lddq, float* __restrict__ dots, int Tp, int B, int S,
                                                       int H) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= (long)Tp * B) return;
  const int b = (int)(row % B);
  int len = (int)lens[b];
  len = len < S ? len : S;
  const T* dcr = dcat + row * lddc;
  float dv[ATT_MAXJ];
#pragma unroll
  for (int j = 0; j < ATT_MAXJ; ++j) dv[j] = lane + 64 * j < H ? to_f<T>(dcr[lane + 64 * j]) : 0.f;
  float da[ATT_LK];
#pragma unroll
  for (int k = 0; k < ATT_LK; ++k) da[k] = 0.f;
  for (int s = 0; s < len; ++s) {
    const T* hr = ctx + ((long)s * B + b) * ldc;
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < ATT_MAXJ; ++j)
      if (lane + 64 * j < H) part += dv[j] * to_f<T>(hr[lane + 64 * j]);
    part = wave_sum(part);
#pragma unroll
    for (int k = 0; k < ATT_LK; ++k)
      if ((s >> 6) == k && lane == (s & 63)) da[k] = part;
  }
  float pk[ATT_LK], dot = 0.f;
#pragma unroll
  for (int k = 0; k < ATT_LK; ++k) { pk[k] = lane + 64 * k < len ? probs[row * S + lane + 64 * k] : 0.f; dot += pk[k] * da[k]; }
  dot = wave_sum(dot);
  if (lane == 0) dots[row] = dot;
  float ds[ATT_LK];
#pragma unroll
  for (int k = 0; k < ATT_LK; ++k) ds[k] = pk[k] * (da[k] - dot);
  float acc[ATT_MAXJ];
#pragma unroll
  for (int j = 0; j < ATT_MAXJ; ++j) acc[j] = 0.f;
  for (int s = 0; s < len; ++s) {
    float dss = 0.f;
#pragma unroll
    for (int k = 0; k < ATT_LK; ++k) { const float v = __shfl(ds[k], s & 63, 64); dss = (s >> 6) == k ? v : dss; }
    const T* hr = ctx + ((long)s * B + b) * ldc;
#pragma unroll
    for (int j = 0; j < ATT_MAXJ; ++j)
      if (lane + 64 * j < H) acc[j] += dss * to_f<T>(hr[lane + 64 * j]);
  }
  T* dqr = dq + row * lddq;
#pragma unroll
  for (int j = 0; j < ATT_MAXJ; ++j)
    if (lane + 64 * j < H) dqr[lane + 64 * j] = from_f<T>(acc[j]);
}

// backward, second half: per (source position, sentence)  dHs[s] = sum_t P[t][s] dC[t] + dS[t][s] Q[t],  dS[t][s] = P[t][s] (dC[t] . Hs[s] - dot[t])
template <class T>
__global__ void __launch_bounds__(256) attn_bwd_long_ctx(const T* __restrict__ dcat, long lddc, const float* __restrict__ probs,
                                                         const T* __restrict__ q, long ldq, const T* __restrict__ ctx, long ldc,
                                                         const long long* __restrict__ lens, const float* __restrict__ dots,
                                                         T* __restrict__ dctx, long lddx, int Tp, int B, int S, int H) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long srow = (long)blockIdx.x * 4 + wave;            // row = s * B + b
  if (srow >= (long)S * B) return;
  const int b = (int)(srow % B), s = (int)(srow / B);
  int len = (int)lens[b];
  len = len < S ? len : S;
  T* out = dctx + srow * lddx;
  if (s >= len) {
#pragma unroll
    for (int j = 0; j < ATT_MAXJ; ++j)
      if (lane + 64 * j < H) out[lane + 64 * j] = from_f<T>(0.f);
    return;
  }
  const T* hr = ctx + srow * ldc;
  float hv[ATT_MAXJ], acc[ATT_MAXJ];
#pragma unroll
  for (int j = 0; j < ATT_MAXJ; ++j) { hv[j] = lane + 64 * j < H ? to_f<T>(hr[lane + 64 * j]) : 0.f; acc[j] = 0.f; }
  for (int t = 0; t < Tp; ++t) {
    const long row = (long)t * B + b;
    const T* dcr = dcat + row * lddc;
    const T* qr = q + row * ldq;
    float dv[ATT_MAXJ], part = 0.f;
#pragma unroll
    for (int j = 0; j < ATT_MAXJ; ++j) {
      dv[j] = lane + 64 * j < H ? to_f<T>(dcr[lane + 64 * j]) : 0.f;
      part += dv[j] * hv[j];
    }
    part = wave_sum(part);
    const float p = probs[row * S + s];
    const float ds = p * (part - dots[row]);
#pragma unroll
    for (int j = 0; j < ATT_MAXJ; ++j)
      if (lane + 64 * j < H) acc[j] += p * dv[j] + ds * to_f<T>(qr[lane + 64 * j]);
  }
#pragma unroll
  for (int j = 0; j < ATT_MAXJ; ++j)
    if (lane + 64 * j < H) out[lane + 64 * j] = from_f<T>(acc[j]);
}

// ==============================================================================================================
// MFMA attention for bf16 (T' <= 32, S <= 32, H % 32 == 0): one workgroup per sentence.  Q_b [T' x H] and the
// sentence's source memory Hs_b [S x H] are staged once in LDS (zero-padded to 32 rows);
//   scores = Q Hs^T          32x32 tile, K = H split over the 4 waves, partials folded through LDS
//   softmax over s < len     in registers (32-lane rows)
//   context = P Hs           K = s: Hs is read K-strided with ds_read_b64_tr_b16 straight from the same LDS image
// backward reuses the structure: dP = dC Hs^T, dS = P (dP - sum P dP), dQ = dS Hs, dHs = P^T dC + dS^T Q.
// ==============================================================================================================
typedef short as16x4 __attribute__((ext_vector_type(4)));
typedef short as16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 afragKC(const bf16_t* img, int stride, int row0, int kbase, int lane) {
  return *reinterpret_cast<const bf16x8*>(img + (row0 + (lane & 31)) * stride + kbase + 8 * (lane >> 5));
}
__device__ __forceinline__ bf16x8 afragKS(const bf16_t* img, int stride, int row0, int kbase, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const bf16_t* a0 = img + (kbase + 8 * (g >> 1) + q) * stride + row0 + 16 * (g & 1) + 4 * p;
  typedef __attribute__((address_space(3))) as16x4 lds_v;
  as16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(a0));
  as16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(a0 + 4 * stride));
  as16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
// stage `nrows` rows (row r from src + r*rstride, H elements each) into a [32][pitch] image; rows >= nrows are zeroed
__device__ __forceinline__ void astage(bf16_t* img, int pitch, const bf16_t* src, long rstride, int nrows, int H) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  for (int r = wave; r < 32; r += 4) {
    const bf16_t* g = src + (long)r * rstride;
    const bool vec = (((uintptr_t)g) & 15) == 0;
    for (int c = lane * 8; c < H; c += 512) {
      u32x4 v = zero;
      if (r < nrows) {
        if (vec) v = *reinterpret_cast<const u32x4*>(g + c);
        else { bf16_t t[8]; for (int e = 0; e < 8; ++e) t[e] = g[c + e]; v = *reinterpret_cast<const u32x4*>(t); }
      }
      *reinterpret_cast<u32x4*>(img + r * pitch + c) = v;
    }
  }
}

__global__ void __launch_bounds__(256) attn_fwd_fast(const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ ctx,
                                                     long ldc, const long long* __restrict__ lens, bf16_t* __restrict__ cat,
                                                     long ldcat, float* __restrict__ probs, int Tp, int B, int S, int H) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int P = H + 8;
  bf16_t* Qs = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* Hs = Qs + 32 * P;
  float* Sc = reinterpret_cast<float*>(Hs + 32 * P);          // [4][32][33]
  bf16_t* Pb = reinterpret_cast<bf16_t*>(Sc + 4 * 32 * 33);   // [32][40]
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int len = (int)lens[b];
  len = len < S ? len : S;
  astage(Qs, P, q + (long)b * ldq, (long)B * ldq, Tp, H);
  astage(Hs, P, ctx + (long)b * ldc, (long)B * ldc, S, H);
  __syncthreads();
  {   // scores: this wave's K quarter
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nks = H / 16, per = (nks + 3) / 4;
    const int k0 = wave * per, k1 = min(nks, k0 + per);
    for (int ks = k0; ks < k1; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afragKC(Qs, P, 0, ks * 16, lane), afragKC(Hs, P, 0, ks * 16, lane), acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) Sc[(wave * 32 + acc_row(r, lane)) * 33 + (lane & 31)] = acc[r];
  }
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {   // softmax: wave handles rows 8w .. 8w+7, two rows per pass
    const int t = 8 * wave + 2 * rr + (lane >> 5), s_ = lane & 31;
    float v = Sc[(0 * 32 + t) * 33 + s_] + Sc[(1 * 32 + t) * 33 + s_] + Sc[(2 * 32 + t) * 33 + s_] + Sc[(3 * 32 + t) * 33 + s_];
    v = s_ < len ? v : -INFINITY;
    float m = v;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float e = s_ < len ? __expf(v - m) : 0.f;
    float sum = e;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    float p = e / sum;
    if (t < Tp && s_ < S) probs[((long)t * B + b) * S + s_] = p;
    Pb[t * 40 + s_] = f2bf(t < Tp ? p : 0.f);
  }
  __syncthreads();
  for (int tile = wave; tile < H / 32; tile += 4) {   // context: 32 hidden units per tile
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afragKC(Pb, 40, 0, ks * 16, lane), afragKS(Hs, P, 32 * tile, ks * 16, lane), acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = acc_row(r, lane);
      if (t < Tp) cat[((long)t * B + b) * ldcat + 32 * tile + (lane & 31)] = f2bf(acc[r]);
    }
  }
}

__global__ void __launch_bounds__(256) attn_bwd_fast(const bf16_t* __restrict__ dcat, long lddc, const float* __restrict__ probs,
                                                     const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ ctx,
                                                     long ldc, const long long* __restrict__ lens, bf16_t* __restrict__ dq,
                                                     long lddq, bf16_t* __restrict__ dctx, long lddx, int Tp, int B, int S, int H) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int P = H + 8;
  bf16_t* dCs = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* Qs = dCs + 32 * P;
  bf16_t* Hs = Qs + 32 * P;
  float* Sc = reinterpret_cast<float*>(Hs + 32 * P);          // [4][32][33]
  bf16_t* Pb = reinterpret_cast<bf16_t*>(Sc + 4 * 32 * 33);   // [32][40]
  bf16_t* dSb = Pb + 32 * 40;                                 // [32][40]
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int len = (int)lens[b];
  len = len < S ? len : S;
  astage(dCs, P, dcat + (long)b * lddc, (long)B * lddc, Tp, H);
  astage(Qs, P, q + (long)b * ldq, (long)B * ldq, Tp, H);
  astage(Hs, P, ctx + (long)b * ldc, (long)B * ldc, S, H);
  __syncthreads();
  {   // dP = dC Hs^T, K quarter per wave
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nks = H / 16, per = (nks + 3) / 4;
    const int k0 = wave * per, k1 = min(nks, k0 + per);
    for (int ks = k0; ks < k1; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afragKC(dCs, P, 0, ks * 16, lane), afragKC(Hs, P, 0, ks * 16, lane), acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) Sc[(wave * 32 + acc_row(r, lane)) * 33 + (lane & 31)] = acc[r];
  }
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {   // softmax backward
    const int t = 8 * wave + 2 * rr + (lane >> 5), s_ = lane & 31;
    float dp = Sc[(0 * 32 + t) * 33 + s_] + Sc[(1 * 32 + t) * 33 + s_] + Sc[(2 * 32 + t) * 33 + s_] + Sc[(3 * 32 + t) * 33 + s_];
    float p = (t < Tp && s_ < len) ? probs[((long)t * B + b) * S + s_] : 0.f;
    float dot = p * dp;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    float ds = p * (dp - dot);
    Pb[t * 40 + s_] = f2bf(p);
    dSb[t * 40 + s_] = f2bf(ds);
  }
  __syncthreads();
  for (int tile = wave; tile < H / 32; tile += 4) {
    f32x16 aq, ah;
#pragma unroll
    for (int r = 0; r < 16; ++r) { aq[r] = 0.f; ah[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // dQ[t][h] += dS[t][s] Hs[s][h]        (k = s)
      aq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afragKC(dSb, 40, 0, ks * 16, lane), afragKS(Hs, P, 32 * tile, ks * 16, lane), aq, 0, 0, 0);
      // dHs[s][h] += P[t][s] dC[t][h] + dS[t][s] Q[t][h]      (k = t: both operands K-strided)
      ah = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afragKS(Pb, 40, 0, ks * 16, lane), afragKS(dCs, P, 32 * tile, ks * 16, lane), ah, 0, 0, 0);
      ah = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afragKS(dSb, 40, 0, ks * 16, lane), afragKS(Qs, P, 32 * tile, ks * 16, lane), ah, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, lane), h = 32 * tile + (lane & 31);
      if (row < Tp) dq[((long)row * B + b) * lddq + h] = f2bf(aq[r]);
      if (row < S) dctx[((long)row * B + b) * lddx + h] = f2bf(row < len ? ah[r] : 0.f);
    }
  }
}


// ==============================================================================================================
// MFMA attention for longer sequences / wider memories (bf16, T' <= 64, S <= 64, H % 32 == 0, H <= 1024; BASELINE config 5:
// S = T' = 64, H = 1024).  One workgroup per sentence, four waves.  The source memory Hs_b [S x H] is the ONLY full operand
// image in LDS (129 KB at H = 1024: three images as in the kernels above would need 390 KB); queries / incoming gradients
// are read straight from global memory into MFMA A fragments (16 bytes per lane, rows are K-contiguous), and the products whose
// reduction runs over t (dHs = P^T dC + dS^T Q, both operands K-strided) stage dC and Q column chunk by column chunk into the
// region the memory occupied, after its last use.
//   scores: 2 x 2 tiles of 32 x 32, one per wave, full K = H (no cross-wave fold); softmax row statistics are exchanged through a
//   64 x 2 float table; context / dQ / dHs: 32 x 32 output tiles dealt round-robin to the waves.
// ==============================================================================================================
// stage `rows` rows of `cols` elements (row r from src + r*rstride; rows >= nrows are zero-filled) into a [rows][pitch] image
__device__ __forceinline__ void astage_n(bf16_t* img, int pitch, const bf16_t* src, long rstride, int nrows, int rows, int cols) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  for (int r = wave; r < rows; r += 4) {
    const bf16_t* g = src + (long)r * rstride;
    for (int c = lane * 8; c < cols; c += 512) {
      u32x4 v = zero;
      if (r < nrows) v = *reinterpret_cast<const u32x4*>(g + c);
      *reinterpret_cast<u32x4*>(img + r * pitch + c) = v;
    }
  }
}
// A fragment of a 32 x 16 slab straight from global memory: row = row0 + (lane & 31) (zeros at rows >= nrows), k = kbase + 8 (lane >> 5) ..
__device__ __forceinline__ bf16x8 afragG(const bf16_t* base, long rstride, int row0, int nrows, int kbase, int lane) {
  const int r = row0 + (lane & 31);
  u32x4 v = {0u, 0u, 0u, 0u};
  if (r < nrows) v = *reinterpret_cast<const u32x4*>(base + (long)r * rstride + kbase + 8 * (lane >> 5));
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ float half_max(float v) {       // over the 32 lanes of this lane's half-wave
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int ATT_PB = 72;       // pitch of the 64 x 64 probability / score-gradient images (bf16 elements)

__global__ void __launch_bounds__(256) attn_fwd_big(const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ ctx,
                                                    long ldc, const long long* __restrict__ lens, bf16_t* __restrict__ cat,
                                                    long ldcat, float* __restrict__ probs, int Tp, int B, int S, int H) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int P = H + 8;
  const int rows_s = (S + 15) & ~15, rows_t = (Tp + 15) & ~15;
  bf16_t* Hs = reinterpret_cast<bf16_t*>(smem_raw);                       // [ru32(S)][P]: score tiles read whole 32-row tiles
  bf16_t* Pb = Hs + 64 * P;                                               // [64][ATT_PB]
  float* red = reinterpret_cast<float*>(Pb + 64 * ATT_PB);                // [64][2] row statistics per s-tile
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int len = (int)lens[b];
  len = len < S ? len : S;
  astage_n(Hs, P, ctx + (long)b * ldc, (long)B * ldc, S, (S + 31) & ~31, H);
  __syncthreads();
  const int tt = wave >> 1, st = wave & 1;
  const bool active = tt * 32 < Tp && st * 32 < S;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (active) {
    const bf16_t* qb = q + (long)b * ldq;
    for (int k0 = 0; k0 < H; k0 += 64) {           // four A fragments in flight
      bf16x8 a[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = k0 + 16 * j < H ? afragG(qb, (long)B * ldq, tt * 32, Tp, k0 + 16 * j, lane) : afragG(qb, 0, 0, 0, 0, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k0 + 16 * j < H) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], afragKC(Hs, P, st * 32, k0 + 16 * j, lane), acc, 0, 0, 0);
    }
  }
  const int s_ = st * 32 + (lane & 31);
  float v[16], m[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    v[r] = s_ < len ? acc[r] : -INFINITY;
    m[r] = half_max(v[r]);
    if ((lane & 31) == 0) red[(tt * 32 + acc_row(r, lane)) * 2 + st] = m[r];
  }
  __syncthreads();
  float e[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int t = tt * 32 + acc_row(r, lane);
    const float mm = fmaxf(red[t * 2], red[t * 2 + 1]);          // finite: position 0 is never masked (len >= 1)
    e[r] = s_ < len ? __expf(v[r] - mm) : 0.f;
    m[r] = half_sum(e[r]);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if ((lane & 31) == 0) red[(tt * 32 + acc_row(r, lane)) * 2 + st] = m[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int t = tt * 32 + acc_row(r, lane);
    const float p = t < Tp ? e[r] / (red[t * 2] + red[t * 2 + 1]) : 0.f;
    if (t < Tp && s_ < S) probs[((long)t * B + b) * S + s_] = p;
    Pb[t * ATT_PB + s_] = f2bf(p);
  }
  __syncthreads();
  const int nht = H / 32, nks = rows_s / 16;
  for (int idx = wave; idx < (rows_t + 31) / 32 * nht; idx += 4) {     // context tiles: (t-tile, 32 hidden units)
    const int t0 = (idx / nht) * 32, h0 = (idx % nht) * 32;
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    for (int ks = 0; ks < nks; ++ks)
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afragKC(Pb, ATT_PB, t0, ks * 16, lane), afragKS(Hs, P, h0, ks * 16, lane), c, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = t0 + acc_row(r, lane);
      if (t < Tp) cat[((long)t * B + b) * ldcat + h0 + (lane & 31)] = f2bf(c[r]);
    }
  }
}

__global__ void __launch_bounds__(256) attn_bwd_big(const bf16_t* __restrict__ dcat, long lddc, const float* __restrict__ probs,
                                                    const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ ctx,
                                                    long ldc, const long long* __restrict__ lens, bf16_t* __restrict__ dq,
                                                    long lddq, bf16_t* __restrict__ dctx, long lddx, int Tp, int B, int S, int H,
                                                    int HC, int big_elems) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int P = H + 8, PC = HC + 8;
  const int rows_s = (S + 15) & ~15, rows_t = (Tp + 15) & ~15;
  bf16_t* Hs = reinterpret_cast<bf16_t*>(smem_raw);                       // [rows_s][P]; later dC / Q chunks [rows_t][PC] x 2
  bf16_t* Pb = Hs + big_elems;                                            // [64][ATT_PB]
  bf16_t* dSb = Pb + 64 * ATT_PB;                                         // [64][ATT_PB]
  float* red = reinterpret_cast<float*>(dSb + 64 * ATT_PB);               // [64][2]
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int len = (int)lens[b];
  len = len < S ? len : S;
  astage_n(Hs, P, ctx + (long)b * ldc, (long)B * ldc, S, (S + 31) & ~31, H);
  __syncthreads();
  const int tt = wave >> 1, st = wave & 1;
  const bool active = tt * 32 < Tp && st * 32 < S;
  {   // dP = dC Hs^T (one 32 x 32 tile per wave, full K), softmax backward
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (active) {
      const bf16_t* db = dcat + (long)b * lddc;
      for (int k0 = 0; k0 < H; k0 += 64) {
        bf16x8 a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = k0 + 16 * j < H ? afragG(db, (long)B * lddc, tt * 32, Tp, k0 + 16 * j, lane) : afragG(db, 0, 0, 0, 0, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k0 + 16 * j < H) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], afragKC(Hs, P, st * 32, k0 + 16 * j, lane), acc, 0, 0, 0);
      }
    }
    const int s_ = st * 32 + (lane & 31);
    float p[16], d[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = tt * 32 + acc_row(r, lane);
      const bool on = t < Tp && s_ < len;             // masked entries never enter the arithmetic (0 x anything)
      p[r] = on ? probs[((long)t * B + b) * S + s_] : 0.f;
      acc[r] = on ? acc[r] : 0.f;
      d[r] = half_sum(p[r] * acc[r]);
      if ((lane & 31) == 0) red[t * 2 + st] = d[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = tt * 32 + acc_row(r, lane);
      const float dot = red[t * 2] + red[t * 2 + 1];
      Pb[t * ATT_PB + s_] = f2bf(p[r]);
      dSb[t * ATT_PB + s_] = f2bf(p[r] * (acc[r] - dot));
    }
  }
  __syncthreads();
  const int nht = H / 32;
  {   // dQ[t][h] = sum_s dS[t][s] Hs[s][h]
    const int nks = rows_s / 16;
    for (int idx = wave; idx < (rows_t + 31) / 32 * nht; idx += 4) {
      const int t0 = (idx / nht) * 32, h0 = (idx % nht) * 32;
      f32x16 c;
#pragma unroll
      for (int r = 0; r < 16; ++r) c[r] = 0.f;
      for (int ks = 0; ks < nks; ++ks)
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afragKC(dSb, ATT_PB, t0, ks * 16, lane), afragKS(Hs, P, h0, ks * 16, lane), c, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = t0 + acc_row(r, lane);
        if (t < Tp) dq[((long)t * B + b) * lddq + h0 + (lane & 31)] = f2bf(c[r]);
      }
    }
  }
  // dHs[s][h] = sum_t P[t][s] dC[t][h] + dS[t][s] Q[t][h]: the reduction runs over t, so dC and Q are needed K-strided:
  // they take the memory's place in LDS, HC columns at a time
  bf16_t* dCs = Hs;
  bf16_t* Qs = Hs + 64 * PC;
  const int nkt = rows_t / 16, nhc = HC / 32;
  for (int c0 = 0; c0 < H; c0 += HC) {
    __syncthreads();                                 // everybody is done with the previous occupant of the region
    astage_n(dCs, PC, dcat + (long)b * lddc + c0, (long)B * lddc, Tp, rows_t, HC);
    astage_n(Qs, PC, q + (long)b * ldq + c0, (long)B * ldq, Tp, rows_t, HC);
    __syncthreads();
    for (int idx = wave; idx < (rows_s + 31) / 32 * nhc; idx += 4) {
      const int s0 = (idx / nhc) * 32, h0 = (idx % nhc) * 32;
      f32x16 c;
#pragma unroll
      for (int r = 0; r < 16; ++r) c[r] = 0.f;
      for (int ks = 0; ks < nkt; ++ks) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afragKS(Pb, ATT_PB, s0, ks * 16, lane), afragKS(dCs, PC, h0, ks * 16, lane), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afragKS(dSb, ATT_PB, s0, ks * 16, lane), afragKS(Qs, PC, h0, ks * 16, lane), c, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int srow = s0 + acc_row(r, lane);
        if (srow < S) dctx[((long)srow * B + b) * lddx + c0 + h0 + (lane & 31)] = f2bf(srow < len ? c[r] : 0.f);
      }
    }
  }
}

// The same backward with 15 KiB of LDS instead of 122 (three staged operand images): in the training step this kernel runs while the
// generator's dWg product holds every CU with 144 KiB workgroups, and a workgroup that needs a free CU's worth of LDS waits for one of
// those to retire (137 us in the step against 25 alone).  Here nothing is staged whole:
//   dP = dC Hs^T      both operands are K-contiguous in memory: the MFMA fragments are 16-byte global loads; EVERY wave runs the whole
//                     reduction (H / 16 MFMAs, 1 us) instead of a quarter plus a fold through LDS
//   dS                in the accumulator registers; wave w finishes rows 8w .. 8w+7 and writes P, dS (bf16 [32][40]) for the others
//   dQ, dHs           per 32-column tile (tiles round-robin over the waves): the [32 x 32] tiles of Hs, dC and Q pass one after the other
//                     through ONE wave-private 2.5 KiB buffer (they are read K-strided: ds_read_b64_tr_b16), and the two output tiles
//                     leave through the same buffer as 16-byte rows instead of 2-byte column stores.
// Preconditions (vmmt_attn_bwd falls back to attn_bwd_fast otherwise): T', S <= 32, H % 32 == 0, rows 16-byte aligned.
__global__ void __launch_bounds__(256) attn_bwd_lite(const bf16_t* __restrict__ dcat, long lddc, const float* __restrict__ probs,
                                                     const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ ctx,
                                                     long ldc, const long long* __restrict__ lens, bf16_t* __restrict__ dq,
                                                     long lddq, bf16_t* __restrict__ dctx, long lddx, int Tp, int B, int S, int H) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  bf16_t* Pb = reinterpret_cast<bf16_t*>(smem_raw);          // [32][40]
  bf16_t* dSb = Pb + 32 * 40;                                // [32][40]
  bf16_t* tb = dSb + 32 * 40 + wave * (32 * 40);             // this wave's tile buffer [32][40]
  int len = (int)lens[b];
  len = len < S ? len : S;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int row = lane & 31, kq = 8 * (lane >> 5), nks = H / 16;
    const bf16_t* ar = dcat + ((long)row * B + b) * lddc + kq;
    const bf16_t* br = ctx + ((long)row * B + b) * ldc + kq;
    const bool av = row < Tp, bv = row < S;
    for (int k0 = 0; k0 < nks; k0 += 8) {
      u32x4 fa[8], fb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool in = k0 + j < nks;
        fa[j] = (av && in) ? *reinterpret_cast<const u32x4*>(ar + (k0 + j) * 16) : zero;
        fb[j] = (bv && in) ? *reinterpret_cast<const u32x4*>(br + (k0 + j) * 16) : zero;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k0 + j < nks)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[j]), __builtin_bit_cast(bf16x8, fb[j]), acc, 0, 0, 0);
    }
    // softmax backward on accumulator rows 8 wave .. 8 wave + 7 (registers 4 wave .. 4 wave + 3: static indices per branch)
    const int s_ = lane & 31;
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if (wave == w) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int t = acc_row(4 * w + rr, lane);
          const float dp = acc[4 * w + rr];
          const float p = (t < Tp && s_ < len) ? probs[((long)t * B + b) * S + s_] : 0.f;
          float dot = p * dp;
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
          Pb[t * 40 + s_] = f2bf(p);
          dSb[t * 40 + s_] = f2bf(p * (dp - dot));
        }
      }
  }
  __syncthreads();
  const int r0 = lane >> 2, c8 = 8 * (lane & 3);             // tile staging / output rows r0, r0 + 16; 8 columns from c8
  for (int tile = wave; tile < H / 32; tile += 4) {
    const int h0 = 32 * tile;
    u32x4 vh[2], vc[2], vq[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = r0 + 16 * j;
      vh[j] = r < S ? *reinterpret_cast<const u32x4*>(ctx + ((long)r * B + b) * ldc + h0 + c8) : zero;
      vc[j] = r < Tp ? *reinterpret_cast<const u32x4*>(dcat + ((long)r * B + b) * lddc + h0 + c8) : zero;
      vq[j] = r < Tp ? *reinterpret_cast<const u32x4*>(q + ((long)r * B + b) * ldq + h0 + c8) : zero;
    }
    f32x16 aq, ah;
#pragma unroll
    for (int r = 0; r < 16; ++r) { aq[r] = 0.f; ah[r] = 0.f; }
    // (the buffer is private to this wave and the LDS executes a wave's operations in order: no barrier between a fill and its reads)
#pragma unroll
    for (int j = 0; j < 2; ++j) *reinterpret_cast<u32x4*>(tb + (r0 + 16 * j) * 40 + c8) = vh[j];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)          // dQ[t][h] += dS[t][s] Hs[s][h]        (k = s)
      aq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afragKC(dSb, 40, 0, ks * 16, lane), afragKS(tb, 40, 0, ks * 16, lane), aq, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) *reinterpret_cast<u32x4*>(tb + (r0 + 16 * j) * 40 + c8) = vc[j];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)          // dHs[s][h] += P[t][s] dC[t][h]        (k = t: both operands K-strided)
      ah = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afragKS(Pb, 40, 0, ks * 16, lane), afragKS(tb, 40, 0, ks * 16, lane), ah, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) *reinterpret_cast<u32x4*>(tb + (r0 + 16 * j) * 40 + c8) = vq[j];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)          // dHs[s][h] += dS[t][s] Q[t][h]
      ah = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afragKS(dSb, 40, 0, ks * 16, lane), afragKS(tb, 40, 0, ks * 16, lane), ah, 0, 0, 0);
    // the two output tiles leave as rows of 16 bytes through the same buffer
#pragma unroll
    for (int r = 0; r < 16; ++r) tb[acc_row(r, lane) * 40 + (lane & 31)] = f2bf(aq[r]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = r0 + 16 * j;
      const u32x4 v = *reinterpret_cast<const u32x4*>(tb + r * 40 + c8);
      if (r < Tp) *reinterpret_cast<u32x4*>(dq + ((long)r * B + b) * lddq + h0 + c8) = v;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) tb[acc_row(r, lane) * 40 + (lane & 31)] = f2bf(acc_row(r, lane) < len ? ah[r] : 0.f);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = r0 + 16 * j;
      const u32x4 v = *reinterpret_cast<const u32x4*>(tb + r * 40 + c8);
      if (r < S) *reinterpret_cast<u32x4*>(dctx + ((long)r * B + b) * lddx + h0 + c8) = v;
    }
  }
}

// LDS request of the big kernels; bwd: the large region holds max(memory image, two column-chunk images)
static int attn_big_chunk(int H) { return (H % 64 == 0 && H > 256) ? H / 2 : H; }
static size_t attn_big_lds_fwd(int H) { return (size_t)64 * (H + 8) * 2 + 64 * ATT_PB * 2 + 64 * 2 * 4; }
static size_t attn_big_elems_bwd(int H) {
  size_t a = (size_t)64 * (H + 8), c = (size_t)2 * 64 * (attn_big_chunk(H) + 8);
  return a > c ? a : c;
}
static size_t attn_big_lds_bwd(int H) { return attn_big_elems_bwd(H) * 2 + 2 * 64 * ATT_PB * 2 + 64 * 2 * 4; }

static size_t attn_fast_lds(int H, int nimg) { return (size_t)nimg * 32 * (H + 8) * 2 + 4 * 32 * 33 * 4 + 2 * 32 * 40 * 2; }

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

static bool big_ok(const void* p, long ld) { return (((uintptr_t)p) & 15) == 0 && ld % 8 == 0; }   // 16-byte row loads

extern "C" int vmmt_attn_fwd(int dtype, const void* q, int64_t ldq, const void* ctx, int64_t ldc, const int64_t* lens,
                             void* cat, int64_t ldcat, float* probs, int Tp, int B, int S, int H, void* stream) {
  using namespace vmmt;
  if (!q || !ctx || !lens || !cat || !probs || S <= 0 || S > 64 * ATT_LK || H > 64 * ATT_MAXJ || Tp <= 0 || B <= 0)
    return VMMT_EINVAL;
  if (S > ATT_MAXS) {          // longer than the per-lane / per-tile kernels take: one wave per (sentence, query)
    const unsigned grid = (unsigned)(((long)Tp * B + 3) / 4);
    if (dtype == VMMT_F32)
      hipLaunchKernelGGL(attn_fwd_long<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)q, (long)ldq, (const float*)ctx,
                         (long)ldc, (const long long*)lens, (float*)cat, (long)ldcat, probs, Tp, B, S, H);
    else if (dtype == VMMT_BF16)
      hipLaunchKernelGGL(attn_fwd_long<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, (long)ldq,
                         (const bf16_t*)ctx, (long)ldc, (const long long*)lens, (bf16_t*)cat, (long)ldcat, probs, Tp, B, S, H);
    else return VMMT_EINVAL;
    return check_launch();
  }
  if (dtype == VMMT_BF16 && Tp <= 32 && S <= 32 && H % 32 == 0 && attn_fast_lds(H, 2) <= 150 * 1024) {
    size_t sm = attn_fast_lds(H, 2);
    static size_t attr = 0;
    if (sm > attr) { (void)hipFuncSetAttribute((const void*)attn_fwd_fast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); attr = sm; }
    hipLaunchKernelGGL(attn_fwd_fast, dim3(B), dim3(256), sm, (hipStream_t)stream, (const bf16_t*)q, (long)ldq, (const bf16_t*)ctx,
                       (long)ldc, (const long long*)lens, (bf16_t*)cat, (long)ldcat, probs, Tp, B, S, H);
    return check_launch();
  }
  if (dtype == VMMT_BF16 && Tp <= 64 && S <= 64 && H % 32 == 0 && attn_big_lds_fwd(H) <= 160 * 1024 && big_ok(q, ldq) && big_ok(ctx, ldc)) {
    size_t sm = attn_big_lds_fwd(H);
    static size_t attr = 0;
    if (sm > attr) { (void)hipFuncSetAttribute((const void*)attn_fwd_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); attr = sm; }
    hipLaunchKernelGGL(attn_fwd_big, dim3(B), dim3(256), sm, (hipStream_t)stream, (const bf16_t*)q, (long)ldq, (const bf16_t*)ctx,
                       (long)ldc, (const long long*)lens, (bf16_t*)cat, (long)ldcat, probs, Tp, B, S, H);
    return check_launch();
  }
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
  if (dtype == VMMT_BF16 && Tp <= 32 && S <= 32 && H % 32 == 0 && ((lddc | ldq | ldc | lddq | lddx) & 7) == 0 &&
      ((((uintptr_t)dcat) | ((uintptr_t)q) | ((uintptr_t)ctx) | ((uintptr_t)dq) | ((uintptr_t)dctx)) & 15) == 0) {
    const size_t sm = 6 * 32 * 40 * 2;          // P, dS and one tile buffer per wave: 15 KiB
    hipLaunchKernelGGL(attn_bwd_lite, dim3(B), dim3(256), sm, (hipStream_t)stream, (const bf16_t*)dcat, (long)lddc, probs,
                       (const bf16_t*)q, (long)ldq, (const bf16_t*)ctx, (long)ldc, (const long long*)lens, (bf16_t*)dq, (long)lddq,
                       (bf16_t*)dctx, (long)lddx, Tp, B, S, H);
    return check_launch();
  }
  if (dtype == VMMT_BF16 && Tp <= 32 && S <= 32 && H % 32 == 0 && attn_fast_lds(H, 3) <= 150 * 1024) {
    size_t sm = attn_fast_lds(H, 3);
    static size_t attr = 0;
    if (sm > attr) { (void)hipFuncSetAttribute((const void*)attn_bwd_fast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); attr = sm; }
    hipLaunchKernelGGL(attn_bwd_fast, dim3(B), dim3(256), sm, (hipStream_t)stream, (const bf16_t*)dcat, (long)lddc, probs,
                       (const bf16_t*)q, (long)ldq, (const bf16_t*)ctx, (long)ldc, (const long long*)lens, (bf16_t*)dq, (long)lddq,
                       (bf16_t*)dctx, (long)lddx, Tp, B, S, H);
    return check_launch();
  }
  if (dtype == VMMT_BF16 && Tp <= 64 && S <= 64 && H % 32 == 0 && attn_big_lds_bwd(H) <= 160 * 1024 && big_ok(dcat, lddc) && big_ok(q, ldq) &&
      big_ok(ctx, ldc)) {
    size_t sm = attn_big_lds_bwd(H);
    static size_t attr = 0;
    if (sm > attr) { (void)hipFuncSetAttribute((const void*)attn_bwd_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); attr = sm; }
    hipLaunchKernelGGL(attn_bwd_big, dim3(B), dim3(256), sm, (hipStream_t)stream, (const bf16_t*)dcat, (long)lddc, probs,
                       (const bf16_t*)q, (long)ldq, (const bf16_t*)ctx, (long)ldc, (const long long*)lens, (bf16_t*)dq, (long)lddq,
                       (bf16_t*)dctx, (long)lddx, Tp, B, S, H, attn_big_chunk(H), (int)attn_big_elems_bwd(H));
    return check_launch();
  }
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

extern "C" int vmmt_attn_bwd_long(int dtype, const void* dcat, int64_t lddc, const float* probs, const void* q, int64_t ldq,
                                  const void* ctx, int64_t ldc, const int64_t* lens, void* dq, int64_t lddq, void* dctx, int64_t lddx,
                                  int Tp, int B, int S, int H, float* dots, void* stream) {
  using namespace vmmt;
  if (!dcat || !probs || !q || !ctx || !lens || !dq || !dctx || !dots || S <= 0 || S > 64 * ATT_LK || H > 64 * ATT_MAXJ || Tp <= 0 || B <= 0)
    return VMMT_EINVAL;
  const unsigned gq = (unsigned)(((long)Tp * B + 3) / 4), gc = (unsigned)(((long)S * B + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VMMT_F32) {
    hipLaunchKernelGGL(attn_bwd_long_q<float>, dim3(gq), dim3(256), 0, st, (const float*)dcat, (long)lddc, probs, (const float*)ctx,
                       (long)ldc, (const long long*)lens, (float*)dq, (long)lddq, dots, Tp, B, S, H);
    hipLaunchKernelGGL(attn_bwd_long_ctx<float>, dim3(gc), dim3(256), 0, st, (const float*)dcat, (long)lddc, probs, (const float*)q,
                       (long)ldq, (const float*)ctx, (long)ldc, (const long long*)lens, (const float*)dots, (float*)dctx, (long)lddx, Tp,
                       B, S, H);
  } else if (dtype == VMMT_BF16) {
    hipLaunchKernelGGL(attn_bwd_long_q<bf16_t>, dim3(gq), dim3(256), 0, st, (const bf16_t*)dcat, (long)lddc, probs, (const bf16_t*)ctx,
                       (long)ldc, (const long long*)lens, (bf16_t*)dq, (long)lddq, dots, Tp, B, S, H);
    hipLaunchKernelGGL(attn_bwd_long_ctx<bf16_t>, dim3(gc), dim3(256), 0, st, (const bf16_t*)dcat, (long)lddc, probs, (const bf16_t*)q,
                       (long)ldq, (const bf16_t*)ctx, (long)ldc, (const long long*)lens, (const float*)dots, (bf16_t*)dctx, (long)lddx,
                       Tp, B, S, H);
  } else return VMMT_EINVAL;
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

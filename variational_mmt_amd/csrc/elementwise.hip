// Small HBM-bound kernels of the VI_Model1 step: row gathers (embedding lookup, image-feature rows), column sums
// (bias gradients), dropout masks, activation backward, the fused mu/sigma -> sample -> KL kernel, the image-network
// gate and the (as-executed, H1) image term of the ELBO.
#include "common.hpp"
#include "vmmt.h"

namespace vmmt {

// out[r][:] = table[ids[r]][:]   (one wave per row: 16-byte loads, fully coalesced 8 KB image rows / 2 KB embedding rows)
// Embedding lookup: onmt/modules/Embeddings.py:181; image rows: onmt/TrainerMultimodal.py:632-639 (host fancy-index + H2D
// in the reference; here the table stays resident in HBM).
template <class TO>
__global__ void gather_rows_kernel(const float* __restrict__ table, long ldt, const long long* __restrict__ ids,
                                   TO* __restrict__ out, long ldo, int R, int D) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= R) return;
  const float* src = table + ids[wave] * ldt;
  TO* dst = out + (long)wave * ldo;
  bool vec = ((((uintptr_t)src) & 15) == 0) && (D % 4 == 0);
  if (vec) {
    for (int c = lane * 4; c < D; c += 256) {
      f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[c + e] = from_f<TO>(v[e]);
    }
  } else {
    for (int c = lane; c < D; c += 64) dst[c] = from_f<TO>(src[c]);
  }
}

// out[c] += sum_r X[r][c] (and out2[c] likewise: b_ih and b_hh of an LSTM receive the same gradient).
// A workgroup covers 64 lanes x VEC columns and a chunk of rows split over its 4 waves: 16-byte loads, LDS fold,
// one fp32 atomic per column and chunk.
template <class T>
__global__ void colsum_kernel(const T* __restrict__ X, long ld, int R, int C, float* __restrict__ out, float* __restrict__ out2,
                              int rows_per_block, int blk, int valid) {
  constexpr int VEC = 16 / sizeof(T);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c0 = (blockIdx.x * 64 + lane) * VEC;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
  float a[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) a[e] = 0.f;
  const bool vec = c0 + VEC <= C && ((((uintptr_t)X) & 15) == 0) && ((ld * (long)sizeof(T)) & 15) == 0;
  if (vec) {
    int r = r0 + w;
    for (; r + 12 < r1; r += 16) {                       // four independent 16-byte loads in flight per lane
      u32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const u32x4*>(X + (long)(r + 4 * q) * ld + c0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const T* e_ = reinterpret_cast<const T*>(&v[q]);
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] += to_f<T>(e_[e]);
      }
    }
    for (; r < r1; r += 4) {
      u32x4 v = *reinterpret_cast<const u32x4*>(X + (long)r * ld + c0);
      const T* e_ = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) a[e] += to_f<T>(e_[e]);
    }
  } else if (c0 < C) {
    for (int r = r0 + w; r < r1; r += 4)
#pragma unroll
      for (int e = 0; e < VEC; ++e) if (c0 + e < C) a[e] += to_f<T>(X[(long)r * ld + c0 + e]);
  }
  __shared__ float red[4][64][VEC + 1];
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[w][lane][e] = a[e];
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      if (c0 + e < C) {
        float t = red[0][lane][e] + red[1][lane][e] + red[2][lane][e] + red[3][lane][e];
        int o = c0 + e;
        if (blk) { const int q = o / blk, r = o - q * blk; if (r >= valid) continue; o = q * valid + r; }      // padded blocks -> dense layout
        atomicAdd(out + o, t);
        if (out2) atomicAdd(out2 + o, t);
      }
    }
  }
}

// out[r] += sum_c X[r][c]: one wave per row, 16-byte loads (generator bias gradient = row sums of G^T [V][M])
template <class T>
__global__ void rowsum_kernel(const T* __restrict__ X, long ld, int R, int C, float* __restrict__ out) {
  constexpr int VEC = 16 / sizeof(T);
  int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= R) return;
  const T* x = X + (long)row * ld;
  float a = 0.f;
  if ((((uintptr_t)x) & 15) == 0) {
    int c = lane * VEC;
    for (; c + VEC <= C; c += 64 * VEC) {
      u32x4 v = *reinterpret_cast<const u32x4*>(x + c);
      const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int k = 0; k < VEC; ++k) a += to_f<T>(e[k]);
    }
    if (c < C) for (int k = c; k < C && k < c + VEC; ++k) a += to_f<T>(x[k]);
  } else {
    for (int c = lane; c < C; c += 64) a += to_f<T>(x[c]);
  }
  a = wave_sum(a);
  if (lane == 0) out[row] += a;
}

template <class T>
__global__ void dropout_mask_kernel(T* __restrict__ mask, long n, float p, unsigned long long seed) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float u = u01(rng32(seed, (unsigned long long)i));
  mask[i] = from_f<T>(u >= p ? 1.f / (1.f - p) : 0.f);
}

__global__ void randn_kernel(float* __restrict__ out, long n, unsigned long long seed) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float u1 = u01(rng32(seed, 2ull * i)), u2 = u01(rng32(seed, 2ull * i + 1));
  out[i] = sqrtf(-2.f * logf(u1)) * cospif(2.f * u2);
}

// out = a * b (2-D with leading dimensions; mask multiply for dropout)
template <class T>
__global__ void mul_kernel(const T* __restrict__ a, long lda, const T* __restrict__ b, long ldb, T* __restrict__ out,
                           long ldo, int R, int C) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)R * C) return;
  int r = i / C, c = i - (long)r * C;
  out[(long)r * ldo + c] = from_f<T>(to_f<T>(a[(long)r * lda + c]) * to_f<T>(b[(long)r * ldb + c]));
}

// ... eight bf16 elements per thread (16-byte loads / stores) where rows are 16-byte aligned and C % 8 == 0: the scalar kernel moves
// 1.5 TB/s, and both element-wise passes sit on the step's critical path (dropout between layers / behind the attentional output;
// the output layer's tanh + dropout backward).  Same arithmetic per element: same bits.
__global__ void __launch_bounds__(256) mul8_kernel(const bf16_t* __restrict__ a, long lda, const bf16_t* __restrict__ b, long ldb,
                                                   bf16_t* __restrict__ out, long ldo, int R, int C8) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)R * C8) return;
  const int r = (int)(i / C8), c = (int)(i - (long)r * C8) * 8;
  const u32x4 va = *reinterpret_cast<const u32x4*>(a + (long)r * lda + c), vb = *reinterpret_cast<const u32x4*>(b + (long)r * ldb + c);
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float lo = __uint_as_float(va[e] << 16) * __uint_as_float(vb[e] << 16);
    const float hi = __uint_as_float(va[e] & 0xffff0000u) * __uint_as_float(vb[e] & 0xffff0000u);
    o[e] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
  }
  *reinterpret_cast<u32x4*>(out + (long)r * ldo + c) = o;
}

// TD = float (a split-K accumulated gradient) or bf16_t; W = 8 or 4 elements per thread.  W = 4 is the form for the SMALL tensors of the
// aux stream's chain (q(z|x) / image network backward), which runs beside the vocabulary sweep: gen2p_kernel leaves 24 registers per lane
// free on its 240 CUs, and a kernel WITHOUT LDS that needs more than that was seen to sit in the dispatcher until the sweep had ended
// (tools/probe_under_sweep.py: 80 us per launch against 5; the W = 8 form takes 29 registers, this one stays under 24: test_kernel_resources)
template <class TD, int W>
__global__ void __launch_bounds__(256) act_bwd8_kernel(int act, const TD* __restrict__ dy, long lddy, const bf16_t* __restrict__ y, long ldy,
                                                       const bf16_t* __restrict__ mask, long ldm, bf16_t* __restrict__ out, long ldo, int R, int CW) {
  typedef uint32_t uw __attribute__((ext_vector_type(W / 2)));
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * CW) return;
  const int r = i / CW, c = (i - r * CW) * W;
  float g[W];
  if constexpr (sizeof(TD) == 4) {
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(dy + (long)r * lddy + c + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[4 * q + e] = d[e];
    }
  } else {
    const uw d = *reinterpret_cast<const uw*>(dy + (long)r * lddy + c);
#pragma unroll
    for (int e = 0; e < W / 2; ++e) { g[2 * e] = __uint_as_float(d[e] << 16); g[2 * e + 1] = __uint_as_float(d[e] & 0xffff0000u); }
  }
  if (mask) {
    const uw m = *reinterpret_cast<const uw*>(mask + (long)r * ldm + c);
#pragma unroll
    for (int e = 0; e < W / 2; ++e) { g[2 * e] *= __uint_as_float(m[e] << 16); g[2 * e + 1] *= __uint_as_float(m[e] & 0xffff0000u); }
  }
  float yy[W];
  if (y) {
    const uw v = *reinterpret_cast<const uw*>(y + (long)r * ldy + c);
#pragma unroll
    for (int e = 0; e < W / 2; ++e) { yy[2 * e] = __uint_as_float(v[e] << 16); yy[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u); }
  } else {
#pragma unroll
    for (int e = 0; e < W; ++e) yy[e] = 0.f;
  }
#pragma unroll
  for (int e = 0; e < W; ++e) {
    switch (act) {
      case VMMT_ACT_RELU: g[e] = yy[e] > 0.f ? g[e] : 0.f; break;
      case VMMT_ACT_TANH: g[e] *= 1.f - yy[e] * yy[e]; break;
      case VMMT_ACT_SOFTPLUS: g[e] *= 1.f - __expf(-yy[e]); break;
      case VMMT_ACT_SIGMOID: g[e] *= yy[e] * (1.f - yy[e]); break;
      default: break;
    }
  }
  uw o;
#pragma unroll
  for (int e = 0; e < W / 2; ++e) o[e] = (uint32_t)f2bf(g[2 * e]) | ((uint32_t)f2bf(g[2 * e + 1]) << 16);
  *reinterpret_cast<uw*>(out + (long)r * ldo + c) = o;
}

// out = dy * mask * act'(y), with act' expressed through the activation OUTPUT y
template <class T, class TD>
__global__ void act_bwd_kernel(int act, const TD* __restrict__ dy, long lddy, const T* __restrict__ y, long ldy,
                               const T* __restrict__ mask, long ldm, T* __restrict__ out, long ldo, int R, int C) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)R * C) return;
  int r = i / C, c = i - (long)r * C;
  float g = to_f<TD>(dy[(long)r * lddy + c]);
  if (mask) g *= to_f<T>(mask[(long)r * ldm + c]);
  float yy = y ? to_f<T>(y[(long)r * ldy + c]) : 0.f;
  switch (act) {
    case VMMT_ACT_RELU: g = yy > 0.f ? g : 0.f; break;
    case VMMT_ACT_TANH: g *= 1.f - yy * yy; break;
    case VMMT_ACT_SOFTPLUS: g *= 1.f - __expf(-yy); break;
    case VMMT_ACT_SIGMOID: g *= yy * (1.f - yy); break;
    default: break;
  }
  out[(long)r * ldo + c] = from_f<T>(g);
}

// ---- fused mu/sigma -> sample -> KL -------------------------------------------------------------------------
// z = mu + sigma * eps (training; detached, H2) or mu (eval)          onmt/Models.py:933, modules/Dists.py:21-26
// KL_b = sum_k 0.5 (mu^2 + sigma^2 - 1) - log sigma  (prior N(0,1))     onmt/VILoss.py:446-456
template <class T>
__global__ void latent_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ sigma,
                                  const float* __restrict__ eps, float* __restrict__ z32, T* __restrict__ zT, long ldz,
                                  float* __restrict__ kl_b, float* __restrict__ stats, int B, int Z, int training) {
  int b = blockIdx.x;
  float kl = 0.f;
  for (int k = threadIdx.x; k < Z; k += blockDim.x) {
    float m = mu[(long)b * Z + k], s = sigma[(long)b * Z + k];
    float z = training ? m + s * eps[(long)b * Z + k] : m;
    z32[(long)b * Z + k] = z;
    zT[(long)b * ldz + k] = from_f<T>(z);
    kl += 0.5f * (m * m + s * s - 1.f) - logf(s);
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

// gradient of (kl_after / norm) w.r.t. mu and the pre-softplus scale output; kl_after = max(mult * KL_mean, margin)
// (annealing before free bits: onmt/VILoss.py:463-473).  kl_sum is read from device memory (no host sync).
// dz / eps (both or neither): the reparameterised path z = mu + sigma * eps NOT detached (what the paper describes; the
// reference as executed detaches the sample, H2 -- Dists.py:21-26): d mu += dz, d sigma += dz * eps.
template <class T>
__global__ void latent_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ sigma,
                                  const float* __restrict__ kl_sum, float batch_global, float mult, int use_freebits,
                                  float margin, float inv_norm, const float* __restrict__ dz, const float* __restrict__ eps,
                                  T* __restrict__ dmu, long ld1, T* __restrict__ dpre, long ld2, int B, int Z) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * Z) return;
  int b = i / Z, k = i - (long)b * Z;
  float gs = mult / batch_global * inv_norm;
  if (use_freebits && mult * (*kl_sum) / batch_global < margin) gs = 0.f;
  float m = mu[i], s = sigma[i];
  float gm = gs * m, gsg = gs * (s - 1.f / s);
  if (dz) { gm += dz[i]; gsg += dz[i] * eps[i]; }
  dmu[(long)b * ld1 + k] = from_f<T>(gm);
  dpre[(long)b * ld2 + k] = from_f<T>(gsg * (1.f - __expf(-s)));   // d softplus = 1 - exp(-y)
}

// dL/dz of the reparameterised sample (f32 [B][Z]): through the decoder input (sum over the time steps of dgates_t W_z, given per
// row in dzrow [T*B][ldr]) and through the image network's gate zt = z * sigmoid(w.z + b) (dzt = dL/dzt)
__global__ void reparam_dz_kernel(const float* __restrict__ dzrow, long ldr, int Tsteps, const float* __restrict__ dzt, long ldd,
                                  const float* __restrict__ z, const float* __restrict__ g, const float* __restrict__ w,
                                  float* __restrict__ out, int B, int Z) {
  const int b = blockIdx.x;
  float a = 0.f;
  for (int k = threadIdx.x; k < Z; k += 64) a += dzt[(long)b * ldd + k] * z[(long)b * Z + k];
  a = wave_sum(a);
  const float gg = g[b], dp = a * gg * (1.f - gg);
  for (int k = threadIdx.x; k < Z; k += 64) {
    float d = dzt[(long)b * ldd + k] * gg + dp * w[k];
    for (int t = 0; t < Tsteps; ++t) d += dzrow[((long)t * B + b) * ldr + k];
    out[(long)b * Z + k] = d;
  }
}

// ---- image network gate: g = sigmoid(w.z + b), zt = z * g      onmt/modules/NormalVariationalEncoder.py:286-293
template <class T>
__global__ void gate_fwd_kernel(const float* __restrict__ z, const float* __restrict__ w, const float* __restrict__ bias,
                                float* __restrict__ g, T* __restrict__ zt, long ldzt, int B, int Z) {
  int b = blockIdx.x;
  float a = 0.f;
  for (int k = threadIdx.x; k < Z; k += 64) a += z[(long)b * Z + k] * w[k];
  a = wave_sum(a);
  float gg = sigmoidf_(a + bias[0]);
  if (threadIdx.x == 0) g[b] = gg;
  for (int k = threadIdx.x; k < Z; k += 64) zt[(long)b * ldzt + k] = from_f<T>(z[(long)b * Z + k] * gg);
}
__global__ void gate_bwd_kernel(const float* __restrict__ dzt, long ldd, const float* __restrict__ z,
                                const float* __restrict__ g, float* __restrict__ dw, float* __restrict__ db, int B, int Z) {
  int b = blockIdx.x;
  float a = 0.f;
  for (int k = threadIdx.x; k < Z; k += 64) a += dzt[(long)b * ldd + k] * z[(long)b * Z + k];
  a = wave_sum(a);
  float gg = g[b];
  float dp = a * gg * (1.f - gg);
  for (int k = threadIdx.x; k < Z; k += 64) atomicAdd(dw + k, dp * z[(long)b * Z + k]);
  if (threadIdx.x == 0) atomicAdd(db, dp);
}

// ---- image term of the ELBO, as executed by the reference (hazard H1) ------------------------------------------
// compute_cosine normalises prediction AND observation in place (onmt/VILoss.py:39-44, called at :408-411) before the
// unit-scale Normal log-prob that is summed over the batch and averaged over D (onmt/VILoss.py:321-331):
//   a = mu_v/|mu_v|, vh = v/|v| ; logp = (1/D) sum_b sum_d [ -0.5 (vh - a)^2 - 0.5 log 2 pi ] ; cos = mean_b a.vh
// backward: true derivative of that forward through the normalisation.
template <class T>
__global__ void image_loss_kernel(const float* __restrict__ mu_v, long ldm, const float* __restrict__ img, long ldi,
                                  int D, float inv_norm, T* __restrict__ dmu, long ldd, float* __restrict__ stats) {
  int b = blockIdx.x;
  const float* m = mu_v + (long)b * ldm;
  const float* v = img + (long)b * ldi;
  float nm = 0.f, nv = 0.f, dt = 0.f;
  for (int d = threadIdx.x; d < D; d += blockDim.x) { float x = m[d], y = v[d]; nm += x * x; nv += y * y; dt += x * y; }
  __shared__ float red[3][4];
  nm = wave_sum(nm); nv = wave_sum(nv); dt = wave_sum(dt);
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = nm; red[1][w] = nv; red[2][w] = dt; }
  __syncthreads();
  nm = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  nv = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  dt = red[2][0] + red[2][1] + red[2][2] + red[2][3];
  float rm = rsqrtf(nm), rv = rsqrtf(nv);
  float cosb = dt * rm * rv;                       // a . vh  (both unit vectors)
  // sum_d (vh - a)^2 = 2 - 2 cos
  float logp = (-0.5f * (2.f - 2.f * cosb) - 0.5f * 1.8378770664093453f * (float)D) / (float)D;
  if (threadIdx.x == 0) { atomicAdd(stats + VMMT_STAT_IMG_LOGPROB, logp); atomicAdd(stats + VMMT_STAT_IMG_COS, cosb); }
  if (dmu) {
    // g = d(-logp)/da = (a - vh)/D ; d/dmu = (g - a (a.g)) / |mu| ; a.g = (1 - cos)/D
    float ag = (1.f - cosb) / (float)D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
      float a = m[d] * rm, vh = v[d] * rv;
      float g = (a - vh) / (float)D;
      dmu[(long)b * ldd + d] = from_f<T>((g - a * ag) * rm * inv_norm);
    }
  }
}

// f32 [R][C] (natural layout, ld = lds) -> T shadow (optionally transposed, optionally + second source, e.g. b_ih + b_hh)
template <class T>
__global__ void pack_kernel(const float* __restrict__ src, const float* __restrict__ src2, long lds_, T* __restrict__ dst,
                            long ldd, int R, int C, int transpose) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)R * C) return;
  if (!transpose) {
    int r = i / C, c = i - (long)r * C;
    float v = src[(long)r * lds_ + c] + (src2 ? src2[(long)r * lds_ + c] : 0.f);
    dst[(long)r * ldd + c] = from_f<T>(v);
  } else {
    int c = i / R, r = i - (long)c * R;     // consecutive threads -> consecutive r -> coalesced dst[c][r]
    float v = src[(long)r * lds_ + c] + (src2 ? src2[(long)r * lds_ + c] : 0.f);
    dst[(long)c * ldd + r] = from_f<T>(v);
  }
}

// One launch per step for everything the reference does on the host / in separate tiny ops before the forward:
// token ids into the step workspace, the input/target shift tgt[:-1] / tgt[1:] (Models.py:867, VILoss.py:205), lengths and
// image row indices, statistics reset, and eps ~ N(0, I) for the latent sample.
// The workspace may be larger than the batch (S_ws >= S source positions, T_ws >= T target positions: shapes are bucketed so
// that a bounded set of launch plans serves real data): the extra positions are filled with the pad id, which every consumer
// already masks (encoder / attention by src_len, the loss by y == pad).
__global__ void prepare_batch_kernel(const long long* __restrict__ src, const long long* __restrict__ tgt,
                                     const long long* __restrict__ src_len, const long long* __restrict__ idx, int S, int T, int B,
                                     int S_ws, int T_ws, long long pad,
                                     long long* __restrict__ o_src, long long* __restrict__ o_tin, long long* __restrict__ o_y,
                                     long long* __restrict__ o_len, long long* __restrict__ o_idx, float* __restrict__ stats,
                                     float* __restrict__ eps, long n_eps, unsigned long long seed, int* __restrict__ flags_src, int R_src,
                                     int* __restrict__ flags_tgt, int R_tgt, int gen) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nS = (long)S * B, nT = (long)(T - 1) * B;
  // flags_*: the lazily updated embedding tables' row flags (optim.hip): every row this batch looks up -- the pad row of the bucketed
  // positions included -- is flagged for the update `gen`, in the flag array of that update's parity (the other one may still be read by
  // the half of update gen - 1 that runs on the side stream)
  if (i < (long)S_ws * B) {
    const long long id = i < nS ? src[i] : pad;
    o_src[i] = id;
    if (flags_src && id >= 0 && id < R_src) flags_src[(long)(gen & 1) * R_src + id] = gen;
  }
  if (i < (long)(T_ws - 1) * B) {
    const long long id = i < nT ? tgt[i] : pad;
    o_tin[i] = id; o_y[i] = i < nT ? tgt[i + B] : pad;
    if (flags_tgt && id >= 0 && id < R_tgt) flags_tgt[(long)(gen & 1) * R_tgt + id] = gen;
  }
  if (i < B) { o_len[i] = src_len[i]; o_idx[i] = idx[i]; }
  if (i < VMMT_STAT_COUNT) stats[i] = 0.f;
  if (eps && i < n_eps) {
    float u1 = u01(rng32(seed, 2ull * i)), u2 = u01(rng32(seed, 2ull * i + 1));
    eps[i] = sqrtf(-2.f * logf(u1)) * cospif(2.f * u2);
  }
}

// every buffer a step has to clear (gradient arena, accumulators) in ONE launch: block -> (descriptor, 16-KiB chunk)
__global__ void zero_multi_kernel(const vmmt_zero_desc* __restrict__ descs, int n) {
  const long chunk = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {                       // last descriptor with chunk_start <= chunk
    int mid = (lo + hi + 1) >> 1;
    if (descs[mid].chunk_start <= chunk) lo = mid; else hi = mid - 1;
  }
  const vmmt_zero_desc d = descs[lo];
  char* base = reinterpret_cast<char*>(d.ptr);
  const long off0 = (chunk - d.chunk_start) * 16384;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long off = off0 + ((long)k * 256 + threadIdx.x) * 16;
    if (off + 16 <= d.bytes) __builtin_nontemporal_store(z, reinterpret_cast<f32x4*>(base + off));     // streamed: keep the L2s for the kernels next door
    else
      for (long o = off; o + 4 <= d.bytes; o += 4) *reinterpret_cast<float*>(base + o) = 0.f;     // tail (bytes % 16 != 0)
  }
}

// all shadow refreshes of a step in one launch: block -> (descriptor, 2048-element chunk)
__global__ void pack_multi_kernel(const vmmt_pack_desc* __restrict__ descs, int n) {
  const int chunk = blockIdx.x;
  // last descriptor with chunk_start <= chunk.  The chunk starts are fetched by the whole workgroup in ONE round trip and searched in
  // LDS: a binary search over global memory is five dependent L2 latencies (~2.5 us) in front of 8 KB of work per workgroup
  __shared__ int starts[256];
  __shared__ int found;
  int lo = 0;
  if (n <= 256) {
    if ((int)threadIdx.x < n) starts[threadIdx.x] = (int)descs[threadIdx.x].chunk_start;
    __syncthreads();
    if (threadIdx.x == 0) {
      int l = 0, h = n - 1;
      while (l < h) {
        const int mid = (l + h + 1) >> 1;
        if (starts[mid] <= chunk) l = mid; else h = mid - 1;
      }
      found = l;
    }
    __syncthreads();
    lo = found;
  } else {
    int hi = n - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (descs[mid].chunk_start <= chunk) lo = mid; else hi = mid - 1;
    }
  }
  const vmmt_pack_desc d = descs[lo];
  const long n_el = (long)d.R * d.C;
  const long base = (long)(chunk - d.chunk_start) * 2048;
  // row-major copies whose rows are whole groups of four (every large weight matrix): 16-byte loads, 8- / 16-byte stores,
  // two index divisions per lane instead of eight
  if (!d.transpose && (d.C & 3) == 0 && (d.ld_src & 3) == 0 && (d.ld_dst & 3) == 0 && ((((uintptr_t)d.src) | ((uintptr_t)d.dst)) & 15) == 0 &&
      (!d.src2 || (((uintptr_t)d.src2) & 15) == 0)) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const long i = base + k * 1024 + threadIdx.x * 4;
      if (i >= n_el) break;
      const int r = (int)(i / d.C), c = (int)(i - (long)r * d.C);
      f32x4 v = *reinterpret_cast<const f32x4*>(d.src + (long)r * d.ld_src + c);
      if (d.src2) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(d.src2 + (long)r * d.ld_src + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += w[e];
      }
      const long o = (long)r * d.ld_dst + c;
      if (d.dtype == VMMT_F32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(d.dst) + o) = v;
      else {
        typedef unsigned short us4 __attribute__((ext_vector_type(4)));
        us4 h = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *reinterpret_cast<us4*>(reinterpret_cast<bf16_t*>(d.dst) + o) = h;
      }
    }
    return;
  }
  if (d.transpose) {
    // transposed copies (W_hh^T): a chunk is a [64 rows][32 columns] tile of the source, read by rows and written by columns through
    // LDS, so that both sides move whole 128-byte segments (element-wise, the strided side fetched one line per lane: the two
    // transposes of a step cost more than the 15 M-element generator weight)
    __shared__ float tile[64][33];
    const int tpr = (d.C + 31) / 32, t = chunk - d.chunk_start;
    const int r0 = (t / tpr) * 64, c0 = (t % tpr) * 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rr = (threadIdx.x >> 5) + 8 * i, cc = threadIdx.x & 31;
      const int r = r0 + rr, c = c0 + cc;
      float v = 0.f;
      if (r < d.R && c < d.C) v = d.src[(long)r * d.ld_src + c] + (d.src2 ? d.src2[(long)r * d.ld_src + c] : 0.f);
      tile[rr][cc] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int cc = (threadIdx.x >> 6) + 4 * i, rr = threadIdx.x & 63;
      const int r = r0 + rr, c = c0 + cc;
      if (r < d.R && c < d.C) {
        const long o = (long)c * d.ld_dst + r;
        if (d.dtype == VMMT_F32) reinterpret_cast<float*>(d.dst)[o] = tile[rr][cc];
        else reinterpret_cast<bf16_t*>(d.dst)[o] = f2bf(tile[rr][cc]);
      }
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    long i = base + k * 256 + threadIdx.x;
    if (i >= n_el) break;
    int r, c;
    if (!d.transpose) { r = i / d.C; c = i - (long)r * d.C; } else { c = i / d.R; r = i - (long)c * d.R; }
    float v = d.src[(long)r * d.ld_src + c] + (d.src2 ? d.src2[(long)r * d.ld_src + c] : 0.f);
    long o = d.transpose ? (long)c * d.ld_dst + r : (long)r * d.ld_dst + c;
    if (d.dtype == VMMT_F32) reinterpret_cast<float*>(d.dst)[o] = v;
    else reinterpret_cast<bf16_t*>(d.dst)[o] = f2bf(v);
  }
}

}  // namespace vmmt

using namespace vmmt;
#define ST ((hipStream_t)stream)
#define BLOCKS(n, t) dim3((unsigned)(((n) + (t)-1) / (t)))

extern "C" int vmmt_gather_rows(int out_dtype, const float* table, int64_t ldt, const int64_t* ids, void* out, int64_t ldo,
                                int R, int D, void* stream) {
  if (!table || !ids || !out || R < 0 || D <= 0) return VMMT_EINVAL;
  if (R == 0) return VMMT_OK;
  dim3 grid((R + 3) / 4), block(256);
  if (out_dtype == VMMT_F32)
    hipLaunchKernelGGL(gather_rows_kernel<float>, grid, block, 0, ST, table, (long)ldt, (const long long*)ids, (float*)out, (long)ldo, R, D);
  else if (out_dtype == VMMT_BF16)
    hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, grid, block, 0, ST, table, (long)ldt, (const long long*)ids, (bf16_t*)out, (long)ldo, R, D);
  else return VMMT_EINVAL;
  return check_launch();
}

extern "C" int vmmt_colsum(int dtype, const void* X, int64_t ld, int R, int C, int blk, int valid, float* out, float* out2, void* stream) {
  if (!X || !out || R < 0 || C <= 0 || blk < 0 || (blk > 0 && (valid <= 0 || valid > blk))) return VMMT_EINVAL;
  if (R == 0) return VMMT_OK;
  int vec = dtype == VMMT_F32 ? 4 : 8;
  int gx = (C + 64 * vec - 1) / (64 * vec);
  // the sum is latency-bound: as many workgroups as the matrix allows, but at most 64 row chunks (= atomic adds per column)
  int rpb = 32;
  while (rpb < R && ((R + rpb - 1) / rpb > 64 || (long)gx * ((R + rpb - 1) / rpb) > 2048)) rpb *= 2;
  dim3 grid(gx, (R + rpb - 1) / rpb);
  if (dtype == VMMT_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, ST, (const float*)X, (long)ld, R, C, out, out2, rpb, blk, valid);
  else if (dtype == VMMT_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, ST, (const bf16_t*)X, (long)ld, R, C, out, out2, rpb, blk, valid);
  else return VMMT_EINVAL;
  return check_launch();
}

// out[ids[r]][0:D] += X[r][0:D] (f32, float atomics), rows with ids[r] == pad_id dropped: the embedding-gradient scatter-add with
// padding_idx (modules/Embeddings.py:118).  One wave per row: a wave-instruction adds 256 contiguous bytes, the shape in which the
// memory-side float atomics run at their full rate; as the epilogue of the producing GEMM (a lane per column, rows strided) the same
// 2.5 M adds cost 47 us on top of a 28 us product.
__global__ void __launch_bounds__(256) scatter_add_rows_kernel(const float* __restrict__ X, long ldx, const long long* __restrict__ ids,
                                                               long long pad_id, float* __restrict__ out, long ldo, int R, int D) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const long long id = ids[r];
  if (id == pad_id) return;
  const float* src = X + (long)r * ldx;
  float* dst = out + id * ldo;
  for (int c = lane; c < D; c += 64) atomicAdd(dst + c, src[c]);
}

extern "C" int vmmt_scatter_add_rows(const float* X, int64_t ldx, const int64_t* ids, int64_t pad_id, float* out, int64_t ldo, int R, int D,
                                     void* stream) {
  if (!X || !ids || !out || R < 0 || D <= 0 || ldx < D || ldo < D) return VMMT_EINVAL;
  if (R == 0) return VMMT_OK;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, ST, X, (long)ldx, (const long long*)ids, (long long)pad_id, out,
                     (long)ldo, R, D);
  return check_launch();
}

extern "C" int vmmt_rowsum(int dtype, const void* X, int64_t ld, int R, int C, float* out, void* stream) {
  if (!X || !out || R < 0 || C <= 0) return VMMT_EINVAL;
  if (R == 0) return VMMT_OK;
  dim3 grid((R + 3) / 4);
  if (dtype == VMMT_F32) hipLaunchKernelGGL(rowsum_kernel<float>, grid, dim3(256), 0, ST, (const float*)X, (long)ld, R, C, out);
  else if (dtype == VMMT_BF16) hipLaunchKernelGGL(rowsum_kernel<bf16_t>, grid, dim3(256), 0, ST, (const bf16_t*)X, (long)ld, R, C, out);
  else return VMMT_EINVAL;
  return check_launch();
}

extern "C" int vmmt_dropout_mask(int dtype, void* mask, int64_t n, float p, uint64_t seed, void* stream) {
  if (!mask || n < 0 || p < 0.f || p >= 1.f) return VMMT_EINVAL;
  if (n == 0) return VMMT_OK;
  if (dtype == VMMT_F32) hipLaunchKernelGGL(dropout_mask_kernel<float>, BLOCKS(n, 256), dim3(256), 0, ST, (float*)mask, (long)n, p, (unsigned long long)seed);
  else if (dtype == VMMT_BF16) hipLaunchKernelGGL(dropout_mask_kernel<bf16_t>, BLOCKS(n, 256), dim3(256), 0, ST, (bf16_t*)mask, (long)n, p, (unsigned long long)seed);
  else return VMMT_EINVAL;
  return check_launch();
}

extern "C" int vmmt_randn(float* out, int64_t n, uint64_t seed, void* stream) {
  if (!out || n < 0) return VMMT_EINVAL;
  if (n == 0) return VMMT_OK;
  hipLaunchKernelGGL(randn_kernel, BLOCKS(n, 256), dim3(256), 0, ST, out, (long)n, (unsigned long long)seed);
  return check_launch();
}

extern "C" int vmmt_mul(int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int R,
                        int C, void* stream) {
  if (!a || !b || !out || R < 0 || C < 0) return VMMT_EINVAL;
  long n = (long)R * C;
  if (n == 0) return VMMT_OK;
  const bool vec8 = dtype == VMMT_BF16 && C % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0 &&
                    ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)out)) & 15) == 0;
  if (vec8) hipLaunchKernelGGL(mul8_kernel, BLOCKS(n / 8, 256), dim3(256), 0, ST, (const bf16_t*)a, (long)lda, (const bf16_t*)b, (long)ldb, (bf16_t*)out, (long)ldo, R, C / 8);
  else if (dtype == VMMT_F32) hipLaunchKernelGGL(mul_kernel<float>, BLOCKS(n, 256), dim3(256), 0, ST, (const float*)a, (long)lda, (const float*)b, (long)ldb, (float*)out, (long)ldo, R, C);
  else if (dtype == VMMT_BF16) hipLaunchKernelGGL(mul_kernel<bf16_t>, BLOCKS(n, 256), dim3(256), 0, ST, (const bf16_t*)a, (long)lda, (const bf16_t*)b, (long)ldb, (bf16_t*)out, (long)ldo, R, C);
  else return VMMT_EINVAL;
  return check_launch();
}

extern "C" int vmmt_act_bwd(int dtype, int act, const void* dy, int64_t lddy, int dy_f32, const void* y, int64_t ldy,
                            const void* mask, int64_t ldm, void* out, int64_t ldo, int R, int C, void* stream) {
  if (!dy || !out || R < 0 || C < 0 || (act != VMMT_ACT_NONE && !y)) return VMMT_EINVAL;
  long n = (long)R * C;
  if (n == 0) return VMMT_OK;
  const bool vec8 = dtype == VMMT_BF16 && n < (1l << 33) && C % 8 == 0 && lddy % (dy_f32 ? 4 : 8) == 0 && (!y || ldy % 8 == 0) && (!mask || ldm % 8 == 0) && ldo % 8 == 0 &&
                    ((((uintptr_t)dy) | ((uintptr_t)y) | ((uintptr_t)mask) | ((uintptr_t)out)) & 15) == 0;
  // (small tensors: 4 elements per thread -- the register budget beside the vocabulary sweep, see act_bwd8_kernel)
  const bool small = n <= (1l << 20);
  if (vec8 && dy_f32 && small) hipLaunchKernelGGL((act_bwd8_kernel<float, 4>), BLOCKS(n / 4, 256), dim3(256), 0, ST, act, (const float*)dy, (long)lddy, (const bf16_t*)y, (long)ldy, (const bf16_t*)mask, (long)ldm, (bf16_t*)out, (long)ldo, R, C / 4);
  else if (vec8 && dy_f32) hipLaunchKernelGGL((act_bwd8_kernel<float, 8>), BLOCKS(n / 8, 256), dim3(256), 0, ST, act, (const float*)dy, (long)lddy, (const bf16_t*)y, (long)ldy, (const bf16_t*)mask, (long)ldm, (bf16_t*)out, (long)ldo, R, C / 8);
  else if (vec8 && small) hipLaunchKernelGGL((act_bwd8_kernel<bf16_t, 4>), BLOCKS(n / 4, 256), dim3(256), 0, ST, act, (const bf16_t*)dy, (long)lddy, (const bf16_t*)y, (long)ldy, (const bf16_t*)mask, (long)ldm, (bf16_t*)out, (long)ldo, R, C / 4);
  else if (vec8) hipLaunchKernelGGL((act_bwd8_kernel<bf16_t, 8>), BLOCKS(n / 8, 256), dim3(256), 0, ST, act, (const bf16_t*)dy, (long)lddy, (const bf16_t*)y, (long)ldy, (const bf16_t*)mask, (long)ldm, (bf16_t*)out, (long)ldo, R, C / 8);
  else if (dtype == VMMT_F32) hipLaunchKernelGGL((act_bwd_kernel<float, float>), BLOCKS(n, 256), dim3(256), 0, ST, act, (const float*)dy, (long)lddy, (const float*)y, (long)ldy, (const float*)mask, (long)ldm, (float*)out, (long)ldo, R, C);
  else if (dtype == VMMT_BF16 && dy_f32) hipLaunchKernelGGL((act_bwd_kernel<bf16_t, float>), BLOCKS(n, 256), dim3(256), 0, ST, act, (const float*)dy, (long)lddy, (const bf16_t*)y, (long)ldy, (const bf16_t*)mask, (long)ldm, (bf16_t*)out, (long)ldo, R, C);
  else if (dtype == VMMT_BF16) hipLaunchKernelGGL((act_bwd_kernel<bf16_t, bf16_t>), BLOCKS(n, 256), dim3(256), 0, ST, act, (const bf16_t*)dy, (long)lddy, (const bf16_t*)y, (long)ldy, (const bf16_t*)mask, (long)ldm, (bf16_t*)out, (long)ldo, R, C);
  else return VMMT_EINVAL;
  return check_launch();
}

extern "C" int vmmt_latent_fwd(int dtype, const float* mu, const float* sigma, const float* eps, float* z32, void* zT,
                               int64_t ldz, float* kl_b, float* stats, int B, int Z, int training, void* stream) {
  if (!mu || !sigma || !z32 || !zT || !kl_b || !stats || (training && !eps) || B <= 0 || Z <= 0) return VMMT_EINVAL;
  if (dtype == VMMT_F32) hipLaunchKernelGGL(latent_fwd_kernel<float>, dim3(B), dim3(256), 0, ST, mu, sigma, eps, z32, (float*)zT, (long)ldz, kl_b, stats, B, Z, training);
  else if (dtype == VMMT_BF16) hipLaunchKernelGGL(latent_fwd_kernel<bf16_t>, dim3(B), dim3(256), 0, ST, mu, sigma, eps, z32, (bf16_t*)zT, (long)ldz, kl_b, stats, B, Z, training);
  else return VMMT_EINVAL;
  return check_launch();
}

extern "C" int vmmt_latent_bwd(int dtype, const float* mu, const float* sigma, const float* kl_sum, float batch_global,
                               float mult, int use_freebits, float margin, float inv_norm, const float* dz, const float* eps,
                               void* dmu, int64_t ld1, void* dpre, int64_t ld2, int B, int Z, void* stream) {
  if (!mu || !sigma || !kl_sum || !dmu || !dpre || B <= 0 || Z <= 0 || (dz && !eps)) return VMMT_EINVAL;
  long n = (long)B * Z;
  if (dtype == VMMT_F32) hipLaunchKernelGGL(latent_bwd_kernel<float>, BLOCKS(n, 256), dim3(256), 0, ST, mu, sigma, kl_sum, batch_global, mult, use_freebits, margin, inv_norm, dz, eps, (float*)dmu, (long)ld1, (float*)dpre, (long)ld2, B, Z);
  else if (dtype == VMMT_BF16) hipLaunchKernelGGL(latent_bwd_kernel<bf16_t>, BLOCKS(n, 256), dim3(256), 0, ST, mu, sigma, kl_sum, batch_global, mult, use_freebits, margin, inv_norm, dz, eps, (bf16_t*)dmu, (long)ld1, (bf16_t*)dpre, (long)ld2, B, Z);
  else return VMMT_EINVAL;
  return check_launch();
}

extern "C" int vmmt_reparam_dz(const float* dzrow, int64_t ldr, int T, const float* dzt, int64_t ldd, const float* z, const float* g,
                               const float* w, float* out, int B, int Z, void* stream) {
  if (!dzrow || !dzt || !z || !g || !w || !out || B <= 0 || Z <= 0 || T < 0) return VMMT_EINVAL;
  hipLaunchKernelGGL(reparam_dz_kernel, dim3(B), dim3(64), 0, ST, dzrow, (long)ldr, T, dzt, (long)ldd, z, g, w, out, B, Z);
  return check_launch();
}

extern "C" int vmmt_gate_fwd(int dtype, const float* z, const float* w, const float* bias, float* g, void* zt, int64_t ldzt,
                             int B, int Z, void* stream) {
  if (!z || !w || !bias || !g || !zt || B <= 0 || Z <= 0) return VMMT_EINVAL;
  if (dtype == VMMT_F32) hipLaunchKernelGGL(gate_fwd_kernel<float>, dim3(B), dim3(64), 0, ST, z, w, bias, g, (float*)zt, (long)ldzt, B, Z);
  else if (dtype == VMMT_BF16) hipLaunchKernelGGL(gate_fwd_kernel<bf16_t>, dim3(B), dim3(64), 0, ST, z, w, bias, g, (bf16_t*)zt, (long)ldzt, B, Z);
  else return VMMT_EINVAL;
  return check_launch();
}

extern "C" int vmmt_gate_bwd(const float* dzt, int64_t ldd, const float* z, const float* g, float* dw, float* db, int B, int Z,
                             void* stream) {
  if (!dzt || !z || !g || !dw || !db || B <= 0 || Z <= 0) return VMMT_EINVAL;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(B), dim3(64), 0, ST, dzt, (long)ldd, z, g, dw, db, B, Z);
  return check_launch();
}

extern "C" int vmmt_image_loss(int dtype, const float* mu_v, int64_t ldm, const float* img, int64_t ldi, int B, int D,
                               float inv_norm, void* dmu, int64_t ldd, float* stats, void* stream) {
  if (!mu_v || !img || !stats || B <= 0 || D <= 0) return VMMT_EINVAL;
  if (dtype == VMMT_F32) hipLaunchKernelGGL(image_loss_kernel<float>, dim3(B), dim3(256), 0, ST, mu_v, (long)ldm, img, (long)ldi, D, inv_norm, (float*)dmu, (long)ldd, stats);
  else if (dtype == VMMT_BF16) hipLaunchKernelGGL(image_loss_kernel<bf16_t>, dim3(B), dim3(256), 0, ST, mu_v, (long)ldm, img, (long)ldi, D, inv_norm, (bf16_t*)dmu, (long)ldd, stats);
  else return VMMT_EINVAL;
  return check_launch();
}

extern "C" int vmmt_pack(int dtype, const float* src, const float* src2, int64_t ld_src, void* dst, int64_t ld_dst, int R, int C,
                         int transpose, void* stream) {
  if (!src || !dst || R <= 0 || C <= 0) return VMMT_EINVAL;
  long n = (long)R * C;
  if (dtype == VMMT_F32) hipLaunchKernelGGL(pack_kernel<float>, BLOCKS(n, 256), dim3(256), 0, ST, src, src2, (long)ld_src, (float*)dst, (long)ld_dst, R, C, transpose);
  else if (dtype == VMMT_BF16) hipLaunchKernelGGL(pack_kernel<bf16_t>, BLOCKS(n, 256), dim3(256), 0, ST, src, src2, (long)ld_src, (bf16_t*)dst, (long)ld_dst, R, C, transpose);
  else return VMMT_EINVAL;
  return check_launch();
}

extern "C" int vmmt_zero_multi(const vmmt_zero_desc* descs, int n, int total_chunks, void* stream) {
  if (!descs || n <= 0 || total_chunks <= 0) return VMMT_EINVAL;
  hipLaunchKernelGGL(zero_multi_kernel, dim3(total_chunks), dim3(256), 0, ST, descs, n);
  return check_launch();
}

extern "C" int vmmt_pack_multi(const vmmt_pack_desc* descs, int n, int total_chunks, void* stream) {
  if (!descs || n <= 0 || total_chunks <= 0) return VMMT_EINVAL;
  hipLaunchKernelGGL(pack_multi_kernel, dim3(total_chunks), dim3(256), 0, ST, descs, n);
  return check_launch();
}

extern "C" int vmmt_prepare_batch(const int64_t* src, const int64_t* tgt, const int64_t* src_len, const int64_t* idx, int S, int T,
                                  int B, int S_ws, int T_ws, int pad, int64_t* o_src, int64_t* o_tin, int64_t* o_y, int64_t* o_len,
                                  int64_t* o_idx, float* stats, float* eps, int64_t n_eps, uint64_t seed, int32_t* flags_src, int R_src,
                                  int32_t* flags_tgt, int R_tgt, int gen, void* stream) {
  if (!src || !tgt || !src_len || !idx || !o_src || !o_tin || !o_y || !o_len || !o_idx || !stats || S <= 0 || T < 2 || B <= 0 ||
      S_ws < S || T_ws < T || ((flags_src || flags_tgt) && gen < 1) || (flags_src && R_src <= 0) || (flags_tgt && R_tgt <= 0))
    return VMMT_EINVAL;
  long n = (long)S_ws * B;
  if ((long)(T_ws - 1) * B > n) n = (long)(T_ws - 1) * B;
  if (eps && n_eps > n) n = n_eps;
  if (n < VMMT_STAT_COUNT) n = VMMT_STAT_COUNT;
  hipLaunchKernelGGL(prepare_batch_kernel, BLOCKS(n, 256), dim3(256), 0, ST, (const long long*)src, (const long long*)tgt,
                     (const long long*)src_len, (const long long*)idx, S, T, B, S_ws, T_ws, (long long)pad, (long long*)o_src,
                     (long long*)o_tin, (long long*)o_y,
                     (long long*)o_len, (long long*)o_idx, stats, eps, (long)n_eps, (unsigned long long)seed, flags_src, R_src, flags_tgt, R_tgt, gen);
  return check_launch();
}

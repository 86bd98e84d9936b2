// Fused global-norm clip + Adam over one flat fp32 parameter arena (gfx950, HBM-bound: 28 B/param).
// Reference: onmt/Optim.py:94-96 -> torch.nn.utils.clip_grad_norm(params, 5) then torch.optim.Adam(betas, eps=1e-9).step()
//   clip_coef = max_norm / (||g||_2 + 1e-6), applied only when < 1
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// All parameters that receive gradients live contiguously in one arena, so the norm is one reduction and the update
// one streaming kernel (16-byte loads/stores) instead of ~50 per-tensor launches.
#include "common.hpp"
#include "vmmt.h"

namespace vmmt {

// Deterministic two-stage reduction: every workgroup leaves its partial sum in `partials[blockIdx.x]` (write-through store),
// takes a ticket, and the workgroup whose ticket is the last one adds the partials in index order.  The result depends only on
// (g, n, grid), never on the arrival order -- data-parallel ranks that hold bit-identical all-reduced gradients compute
// bit-identical norms and hence identical clip coefficients (replicas cannot drift apart through the clip).
__global__ void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ partials, unsigned* __restrict__ ticket,
                             float* __restrict__ total) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  long stride = (long)gridDim.x * blockDim.x * 4;
  float a = 0.f;
  // few workgroups (the reduction runs underneath latency-critical kernels of another stream: leave them wave slots),
  // four independent 16-byte loads in flight per lane instead
  for (; i + 3 * stride + 3 < n; i += 4 * stride) {
    f32x4 v0 = *reinterpret_cast<const f32x4*>(g + i), v1 = *reinterpret_cast<const f32x4*>(g + i + stride);
    f32x4 v2 = *reinterpret_cast<const f32x4*>(g + i + 2 * stride), v3 = *reinterpret_cast<const f32x4*>(g + i + 3 * stride);
    a += v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2] + v0[3] * v0[3] + v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2] + v1[3] * v1[3];
    a += v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2] + v2[3] * v2[3] + v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2] + v3[3] * v3[3];
  }
  for (; i + 3 < n; i += stride) {
    f32x4 v = *reinterpret_cast<const f32x4*>(g + i);          // (non-temporal here measured the same)
    a += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  // tail (n % 4 elements) handled by the thread that lands on it
  if (i < n && i + 3 >= n)
    for (long k = i; k < n; ++k) a += g[k] * g[k];
  a = wave_sum(a);
  __shared__ float red[4];
  __shared__ int is_last;
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    // hand-off (MI355X_MICROARCH.md, inter-workgroup visibility, first table row): sc1 store, drained, then ONE agent-scope
    // atomic add per workgroup; the workgroup whose add returned last reads every partial with sc1 loads
    __hip_atomic_store(partials + blockIdx.x, (red[0] + red[1]) + (red[2] + red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == gridDim.x - 1);
  }
  __syncthreads();
  if (!is_last) return;
  float s = 0.f;
  for (unsigned k = threadIdx.x; k < gridDim.x; k += blockDim.x)
    s += __hip_atomic_load(partials + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  s = wave_sum(s);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    *total = (red[0] + red[1]) + (red[2] + red[3]);
    *ticket = 0u;                      // ready for the next step without a memset
  }
}

// ONE Adam element update, shared by the dense kernel and the row-wise kernel below.  Floating-point contraction is switched off: the
// row-wise kernel passes g = 0 for rows without gradient, and its results must be the bits the dense kernel produces for a zero
// gradient whatever the compiler would like to fuse in either place.
#pragma clang fp contract(off)
__device__ __forceinline__ void adam_elem(float& p, float ge, float& m, float& v, float step_size, float b1, float b2, float eps,
                                          float inv_sqrt_bc2) {
  m = b1 * m + (1.f - b1) * ge;
  v = b2 * v + (1.f - b2) * ge * ge;
  p -= step_size * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
}
#pragma clang fp contract(fast)

// `shadow` (optional): bf16 compute copy of this parameter range with the SAME flat layout (a 2-D weight whose shadow rows are not padded):
// written here, so that the shadow refresh does not have to read the 15 M-element generator weight back
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float step_size, float b1, float b2, float eps, float inv_sqrt_bc2, float max_norm,
                            const float* __restrict__ sumsq, float grad_scale, bf16_t* __restrict__ shadow, int* __restrict__ skip) {
  if (skip && __hip_atomic_load(skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {      // vmmt.h: the step is not applied
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skip + 1, 1);
    return;
  }
  float coef = grad_scale;
  if (max_norm > 0.f) {
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < VMMT_SUMSQ_SLOTS; ++k) ss += sumsq[k];       // slot totals in index order (unused slots hold 0)
    float nrm = sqrtf(ss) * grad_scale;
    float c = max_norm / (nrm + 1e-6f);
    if (c < 1.f) coef *= c;
  }
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {
    if (i + 3 < n) {
#ifndef VMMT_EXP_ADAMT
      // streamed once per step: non-temporal, so that 1.7 GB of optimiser traffic does not push the recurrences' exchange lines and
      // the GEMMs' operands out of the L2s (1.799 -> 1.780 ms per step, three same-box pairs)
      f32x4 pp = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p + i)), gg = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + i));
      f32x4 mm = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m + i)), vv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v + i));
#else
      f32x4 pp = *reinterpret_cast<f32x4*>(p + i), gg = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
#endif
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pp[e], me = mm[e], ve = vv[e];
        adam_elem(pe, gg[e] * coef, me, ve, step_size, b1, b2, eps, inv_sqrt_bc2);
        pp[e] = pe; mm[e] = me; vv[e] = ve;
      }
#ifndef VMMT_EXP_ADAMT
      __builtin_nontemporal_store(pp, reinterpret_cast<f32x4*>(p + i));
      __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(m + i));
      __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v + i));
#else
      *reinterpret_cast<f32x4*>(p + i) = pp;
      *reinterpret_cast<f32x4*>(m + i) = mm;
      *reinterpret_cast<f32x4*>(v + i) = vv;
#endif
      if (shadow) {
        typedef unsigned short us4 __attribute__((ext_vector_type(4)));
        const us4 h = {f2bf(pp[0]), f2bf(pp[1]), f2bf(pp[2]), f2bf(pp[3])};
        *reinterpret_cast<us4*>(shadow + i) = h;
      }
    } else {
      for (long k = i; k < n; ++k) {
        float mk = m[k], vk = v[k], pk = p[k];
        adam_elem(pk, g[k] * coef, mk, vk, step_size, b1, b2, eps, inv_sqrt_bc2);
        m[k] = mk; v[k] = vk;
        p[k] = pk;
        if (shadow) shadow[k] = f2bf(pk);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Embedding tables: gradient bookkeeping by ROW.  An embedding table [R][C] receives gradient in the rows of the current batch only
// (<= 5120 of 30 000 at the benchmark shape), but the dense path clears, norms and reads the whole 60 MB gradient of each table every
// step.  With one flag per row (set for the batch's rows by rows_mark_kernel) only the flagged rows' gradient is cleared
// (rows_zero_kernel), summed into the norm (rows_sumsq_kernel) and read by the update (adam_rows_kernel: g = 0 for the other rows
// without touching memory).  EVERY row is still updated at every step with the dense kernel's arithmetic (adam_elem): a row without
// gradient moves under Adam too (its moments decay, the parameter follows them), so results are bit-identical to the dense kernels.
// (A lazy variant -- rows brought up to date only when a batch uses them, replaying the missed zero-gradient steps -- was built and
//  measured: bit-identical as well and 1.5 % faster on the benchmark's eight recurring batches, but on Zipf-distributed word ids most
//  rows come back after hundreds of steps and the replays -- a sqrt and a division per element and missed step, on the critical path in
//  front of the embedding lookup -- cost more than the 28 B/element they save: 2.11 against 1.92 ms per step through the trainer.)
__global__ void rows_mark_kernel(const long long* __restrict__ ids, long n, int* __restrict__ flags, int R) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long id = ids[i];
  if (id >= 0 && id < R) flags[id] = 1;
}

// g[r][:] = 0 for the flagged rows (one wave per row; the rows are about to receive this batch's scatter-add)
__global__ void __launch_bounds__(256) rows_zero_kernel(float* __restrict__ g, int R, int C, const int* __restrict__ flags) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R || !flags[row]) return;
  float* x = g + (long)row * C;
  for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<f32x4*>(x + c) = f32x4{0.f, 0.f, 0.f, 0.f};
}

// Adam over a whole table, one wave per row; the gradient is read for flagged rows only (zero elsewhere); flags are cleared
__global__ void __launch_bounds__(256) adam_rows_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int R, int C, int* __restrict__ flags, float b1, float b2,
                                                        float eps, float step_size, float inv_sqrt_bc2, float max_norm,
                                                        const float* __restrict__ sumsq, float grad_scale, int* __restrict__ skip) {
  if (skip && __hip_atomic_load(skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skip + 1, 1);
    return;
  }
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const bool has_g = flags[row] != 0;
  float coef = grad_scale;
  if (max_norm > 0.f) {
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < VMMT_SUMSQ_SLOTS; ++k) ss += sumsq[k];
    const float c = max_norm / (sqrtf(ss) * grad_scale + 1e-6f);
    if (c < 1.f) coef *= c;
  }
  const long base = (long)row * C;
  for (int c0 = lane * 4; c0 < C; c0 += 256) {
    f32x4 pp = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p + base + c0));
    f32x4 mm = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m + base + c0));
    f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v + base + c0));
    f32x4 gg = {0.f, 0.f, 0.f, 0.f};
    if (has_g) gg = *reinterpret_cast<const f32x4*>(g + base + c0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = pp[e], me = mm[e], ve = vv[e];
      adam_elem(pe, gg[e] * coef, me, ve, step_size, b1, b2, eps, inv_sqrt_bc2);
      pp[e] = pe; mm[e] = me; vv[e] = ve;
    }
    __builtin_nontemporal_store(pp, reinterpret_cast<f32x4*>(p + base + c0));
    __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(m + base + c0));
    __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v + base + c0));
  }
  if (lane == 0 && has_g) flags[row] = 0;
}

// ||g||^2 over the flagged rows (the other rows hold zeros): per-row sums by one wave each, then ONE workgroup adds the R row sums in a
// fixed order into the slot total: deterministic, like sumsq_kernel
__global__ void __launch_bounds__(256) rows_sumsq_kernel(const float* __restrict__ g, int R, int C, const int* __restrict__ flags,
                                                         float* __restrict__ rowsq) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  float a = 0.f;
  if (flags[row]) {
    const float* x = g + (long)row * C;
    for (int c = lane * 4; c < C; c += 256) { const f32x4 q = *reinterpret_cast<const f32x4*>(x + c); a += q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]; }
    a = wave_sum(a);
  }
  if (lane == 0) rowsq[row] = a;
}
__global__ void __launch_bounds__(1024) rows_sumsq_total_kernel(const float* __restrict__ rowsq, int R, float* __restrict__ total) {
  // a fixed order with independent loads in flight (a chain of 30 dependent loads per thread cost 17 us)
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int r = threadIdx.x;
  for (; r + 3 * 1024 < R; r += 4 * 1024) { a0 += rowsq[r]; a1 += rowsq[r + 1024]; a2 += rowsq[r + 2048]; a3 += rowsq[r + 3072]; }
  for (; r < R; r += 1024) a0 += rowsq[r];
  float a = (a0 + a1) + (a2 + a3);
  a = wave_sum(a);
  __shared__ float red[16];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += red[k];
    *total = t;
  }
}

// Data-parallel clip norm with the optimiser state sharded over the ranks: every rank norms ITS shards (sumsq_kernel, one slot per arena
// segment), the ranks exchange ONE row [slot totals | guard word] (an all-gather of 36 bytes), and every rank folds the rows in the same
// fixed order -- slots of a rank, then the ranks -- so that all of them compute the same clip coefficient bit for bit.  The guard word of
// the persistent recurrences (vmmt.h: VMMT_SEQ_GUARD_WORD) rides along as raw bits and is folded with max(): a rank whose recurrence timed
// out must not be the only one that skips the update.
__global__ void dp_norm_pack_kernel(const float* __restrict__ sumsq, const int* __restrict__ guard, float* __restrict__ row) {
  const int t = threadIdx.x;
  if (t < VMMT_SUMSQ_SLOTS) row[t] = sumsq[t];
  else if (t == VMMT_SUMSQ_SLOTS) row[t] = __int_as_float(guard ? guard[0] : 0);
}
__global__ void dp_norm_fold_kernel(const float* __restrict__ rows, int world, float* __restrict__ sumsq, int* __restrict__ guard) {
  if (threadIdx.x != 0) return;
  float tot = 0.f;
  int g = 0;
  for (int r = 0; r < world; ++r) {
    const float* row = rows + (long)r * (VMMT_SUMSQ_SLOTS + 1);
    float s = 0.f;
    for (int k = 0; k < VMMT_SUMSQ_SLOTS; ++k) s += row[k];
    tot += s;
    const int gr = __float_as_int(row[VMMT_SUMSQ_SLOTS]);
    g = gr > g ? gr : g;
  }
  sumsq[0] = tot;
  for (int k = 1; k < VMMT_SUMSQ_SLOTS; ++k) sumsq[k] = 0.f;
  if (guard) guard[0] = g;
}

}  // namespace vmmt

extern "C" int vmmt_dp_norm_pack(const float* sumsq, const int32_t* guard, float* row, void* stream) {
  using namespace vmmt;
  if (!sumsq || !row) return VMMT_EINVAL;
  hipLaunchKernelGGL(dp_norm_pack_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sumsq, (const int*)guard, row);
  return check_launch();
}

extern "C" int vmmt_dp_norm_fold(const float* rows, int world, float* sumsq, int32_t* guard, void* stream) {
  using namespace vmmt;
  if (!rows || !sumsq || world < 1) return VMMT_EINVAL;
  hipLaunchKernelGGL(dp_norm_fold_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rows, world, sumsq, (int*)guard);
  return check_launch();
}

extern "C" int vmmt_rows_mark(const int64_t* ids, int64_t n, int32_t* flags, int R, void* stream) {
  using namespace vmmt;
  if (!ids || !flags || n < 0 || R <= 0) return VMMT_EINVAL;
  if (n == 0) return VMMT_OK;
  hipLaunchKernelGGL(rows_mark_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const long long*)ids, (long)n, flags, R);
  return check_launch();
}

extern "C" int vmmt_rows_zero(float* g, int R, int C, const int32_t* flags, void* stream) {
  using namespace vmmt;
  if (!g || !flags || R <= 0 || C <= 0 || C % 4 != 0 || (((uintptr_t)g) & 15)) return VMMT_EINVAL;
  hipLaunchKernelGGL(rows_zero_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, R, C, flags);
  return check_launch();
}

extern "C" int vmmt_adam_rows_step(float* p, const float* g, float* m, float* v, int R, int C, int32_t* flags, float lr, float beta1,
                                   float beta2, float eps, int step, float max_norm, const float* sumsq, float grad_scale, const int32_t* skip,
                                   void* stream) {
  using namespace vmmt;
  if (!p || !g || !m || !v || !flags || R <= 0 || C <= 0 || C % 4 != 0 || step < 1 || (max_norm > 0.f && !sumsq) ||
      ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15))
    return VMMT_EINVAL;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = (float)(lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));        // as vmmt_adam_step
  hipLaunchKernelGGL(adam_rows_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, R, C, flags, beta1, beta2,
                     eps, step_size, inv_sqrt_bc2, max_norm, sumsq, grad_scale, const_cast<int*>(skip));
  return check_launch();
}

extern "C" int vmmt_sumsq_rows(const float* g, int R, int C, const int32_t* flags, float* rowsq, float* scratch, int slot, void* stream) {
  using namespace vmmt;
  if (!g || !flags || !rowsq || !scratch || R <= 0 || C <= 0 || slot < 0 || slot >= VMMT_SUMSQ_SLOTS || (((uintptr_t)g) & 15) || C % 4 != 0)
    return VMMT_EINVAL;
  hipLaunchKernelGGL(rows_sumsq_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, R, C, flags, rowsq);
  hipLaunchKernelGGL(rows_sumsq_total_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rowsq, R, scratch + slot);
  return check_launch();
}

extern "C" int vmmt_sumsq(const float* g, int64_t n, float* scratch, int slot, void* stream) {
  using namespace vmmt;
  if (!g || !scratch || n < 0 || slot < 0 || slot >= VMMT_SUMSQ_SLOTS || (((uintptr_t)g) & 15)) return VMMT_EINVAL;
  if (n == 0) return VMMT_OK;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > VMMT_SUMSQ_MAXBLOCKS) blocks = VMMT_SUMSQ_MAXBLOCKS;        // 3 workgroups of 4 waves per CU
  if (blocks < 1) blocks = 1;
  float* partials = scratch + 2 * VMMT_SUMSQ_SLOTS + (long)slot * VMMT_SUMSQ_MAXBLOCKS;
  unsigned* ticket = reinterpret_cast<unsigned*>(scratch + VMMT_SUMSQ_SLOTS) + slot;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, (long)n, partials, ticket,
                     scratch + slot);
  return check_launch();
}

extern "C" int vmmt_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                              float eps, int step, float max_norm, const float* sumsq, float grad_scale, int max_blocks,
                              void* shadow_bf16, const int32_t* skip, void* stream) {
  using namespace vmmt;
  if (!p || !g || !m || !v || n < 0 || step < 1 || (max_norm > 0.f && !sumsq)) return VMMT_EINVAL;
  if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return VMMT_EINVAL;
  if (((uintptr_t)shadow_bf16) & 7) return VMMT_EINVAL;
  if (n == 0) return VMMT_OK;
  double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  float step_size = (float)(lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;   // throttle: a background update must not saturate HBM
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, step_size, beta1,
                     beta2, eps, inv_sqrt_bc2, max_norm, sumsq, grad_scale, (bf16_t*)shadow_bf16, const_cast<int*>(skip));
  return check_launch();
}

extern "C" int vmmt_version(void) { return 1; }

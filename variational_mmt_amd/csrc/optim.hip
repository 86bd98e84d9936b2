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
  // v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the correctly rounded sqrtf and division, which expand to ~35 instructions per element:
  // the dense kernel hides them under its HBM streams, the lazy rows' replays (no memory traffic between steps) were bound by them -- 31 us
  // of catch-up in front of the lookups.  The update term carries a relative error of ~2e-7: 5e-10 of a parameter at lr 0.002, below
  // half an ulp of the parameter it is subtracted from.  (denominator >= eps = 1e-9: a normal number; a denormal v flushes to 0 < eps ulp)
  p -= (step_size * m) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) * inv_sqrt_bc2 + eps);
}
#pragma clang fp contract(fast)

// `shadow` (optional): bf16 compute copy of this parameter range with the SAME flat layout (a 2-D weight whose shadow rows are not padded):
// written here, so that the shadow refresh does not have to read the 15 M-element generator weight back
// (sh_lo, sh_n): the shadow covers elements [sh_lo, sh_lo + sh_n) of the range only (shadow[0] = element sh_lo);  (hole_lo, hole_n): the n
// elements are those of [0, n + hole_n) WITHOUT [hole_lo, hole_lo + hole_n) -- a lazily updated embedding table in the middle of the range,
// which vmmt_adam_rows_step updates: one launch for everything around it instead of one per piece (the pieces of the step's last update are
// small: their launches, not their bytes, were the tail of the step).  All four are multiples of 4.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float step_size, float b1, float b2, float eps, float inv_sqrt_bc2, float max_norm,
                            const float* __restrict__ sumsq, float grad_scale, bf16_t* __restrict__ shadow, int* __restrict__ skip,
                            long sh_lo, long sh_n, long hole_lo, long hole_n) {
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
  long j = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  long stride = (long)gridDim.x * blockDim.x * 4;
  for (; j < n; j += stride) {
    const long i = j < hole_lo ? j : j + hole_n;          // (logical -> arena index: the hole is skipped)
    const bool sh = shadow && i >= sh_lo && i < sh_lo + sh_n;
    if (j + 3 < n) {
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
      if (sh) {
        typedef unsigned short us4 __attribute__((ext_vector_type(4)));
        const us4 h = {f2bf(pp[0]), f2bf(pp[1]), f2bf(pp[2]), f2bf(pp[3])};
        *reinterpret_cast<us4*>(shadow + (i - sh_lo)) = h;
      }
    } else {
      for (long k = i; k < i + (n - j); ++k) {              // (the range's last 1-3 elements: behind the hole, if there is one)
        float mk = m[k], vk = v[k], pk = p[k];
        adam_elem(pk, g[k] * coef, mk, vk, step_size, b1, b2, eps, inv_sqrt_bc2);
        m[k] = mk; v[k] = vk;
        p[k] = pk;
        if (sh) shadow[k - sh_lo] = f2bf(pk);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Embedding tables: EXACT LAZY Adam by row.  An embedding table [R][C] receives gradient in the rows of the current batch only
// (<= 5120 of 30 000 at the benchmark shape), yet the dense path clears, norms and streams p, m, v, g of the whole table at every step:
// 36 B per element.  A row without gradient still moves under dense Adam (its moments decay, the parameter follows them) -- but that
// zero-gradient step depends on nothing but the row's own m, v and the step's scalars, so it can be applied LATER, bit for bit:
//   last[r]         the optimiser step row r is current for
//   hist            a ring of the last VMMT_LAZY_HIST steps' scalars (step_size, 1 / sqrt(bias correction 2), applied or skipped)
//   flags[t & 1][r] generation number: row r is "flagged" for the update t when flags[t & 1][r] == t (nothing ever clears a flag; two
//                   arrays by the update's parity: the next batch is flagged while a half of the last update still reads its own)
// rows_mark flags the batch's rows, rows_catchup replays the zero-gradient steps last + 1 .. t - 1 of the flagged rows in registers
// (adam_elem with g = 0: the dense kernel's arithmetic in the dense kernel's order) in front of the lookup and clears their gradient rows,
// the update touches the flagged rows -- and a ROLLING 1 / roll of the table (rows r % roll == t % roll), so that no row is ever more
// than ~roll steps behind: the replay in front of a lookup stays a few microseconds whatever the distribution of the word ids (the
// first lazy variant, rounds 1-3, replayed hundreds of steps for the rare words of a Zipf distribution and lost what it saved).
#define VMMT_LAZY_HDR 4
__device__ __forceinline__ const float* lazy_entry(const int* hist, int s) { return reinterpret_cast<const float*>(hist + VMMT_LAZY_HDR + 4 * (s & (VMMT_LAZY_HIST - 1))); }

__global__ void rows_mark_kernel(const long long* __restrict__ ids, long n, int* __restrict__ flags, int R, const int* __restrict__ hist) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long id = ids[i];
  if (id >= 0 && id < R) { const int gen = hist[0] + 1; flags[(long)(gen & 1) * R + id] = gen; }
}

// The row kernels share one shape: a workgroup owns a block of 64 rows -- every wave reads the block's flags / `last` words with one
// coalesced load each and ballots the rows that have work -- and its waves take those rows in turn, a whole row per wave (C / 4 lanes x
// 16 bytes per array and pass).  One wave per (row, chunk) over the whole table meant 60 000 waves of which 50 000 found nothing to do
// after three dependent memory round trips each: 69 us for an update that moves 90 MB.
#define VMMT_ROWS_WAVES 16

// the scalars of the steps from + 1 .. from + n (n <= 64) out of the ring: lane j takes step from + 1 + j's entry -- ONE load per wave
// instead of one per replayed step (a memory round trip of 1-3 us in every iteration of a loop whose arithmetic takes 0.2 us)
__device__ __forceinline__ f32x4 lazy_entries(const int* __restrict__ hist, int from, int n) {
  const int lane = threadIdx.x & 63;
  f32x4 en = {0.f, 0.f, 0.f, 0.f};
  if (lane < n) en = *reinterpret_cast<const f32x4*>(lazy_entry(hist, from + 1 + lane));
  return en;
}
// ... and the zero-gradient steps themselves on four elements per lane, the scalars handed round by v_readlane (all 64 lanes take part:
// lanes beyond the row's end carry zeros).  An entry that has been overwritten (a row further behind than the ring is long: the caller's
// rolling / flush policy was not kept) sets the error word instead of applying another step's scalars.
__device__ __forceinline__ void lazy_replay(f32x4& pp, f32x4& mm, f32x4& vv, const f32x4& en, int from, int n, int* __restrict__ hist, float b1,
                                            float b2, float eps) {
  for (int j = 0; j < n; ++j) {
    const float ss = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(en[0]), j));
    const float ib = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(en[1]), j));
    const float applied = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(en[2]), j));
    const int sid = __builtin_amdgcn_readlane(__float_as_int(en[3]), j);
    if (sid != from + 1 + j) { if ((threadIdx.x & 63) == 0) hist[1] = from + 1 + j; continue; }
    if (applied == 0.f) continue;             // a step the guard word skipped: nothing moved
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = pp[e], me = mm[e], ve = vv[e];
      adam_elem(pe, 0.f, me, ve, ss, b1, b2, eps, ib);
      pp[e] = pe; mm[e] = me; vv[e] = ve;
    }
  }
}

// one row brought from step `from` up to `upto` by zero-gradient steps and -- STEP -- through the update `upto + 1` with gradient
// g * coef (has_g) or without one; ZERO_G: the row's gradient is cleared.  Two 256-column chunks per pass, their loads issued together.
template <bool STEP, bool ZERO_G>
__device__ __forceinline__ void lazy_row(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long base, int C,
                                         int from, int upto, int* __restrict__ hist, float b1, float b2, float eps, bool has_g, float coef,
                                         float step_size, float inv_sqrt_bc2) {
  const int lane = threadIdx.x & 63;
  const bool behind = from < upto;
  if (!STEP && !behind && !ZERO_G) return;
  for (int cb = 0; cb < C; cb += 512) {
    const int ca = cb + lane * 4, cc = ca + 256;
    const bool la = ca < C, lb = cc < C;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 pa = z4, ma = z4, va = z4, ga = z4, pb = z4, mb = z4, vb = z4, gb = z4;
    if (STEP || behind) {
      if (la) { pa = *reinterpret_cast<f32x4*>(p + base + ca); ma = *reinterpret_cast<f32x4*>(m + base + ca); va = *reinterpret_cast<f32x4*>(v + base + ca); }
      if (lb) { pb = *reinterpret_cast<f32x4*>(p + base + cc); mb = *reinterpret_cast<f32x4*>(m + base + cc); vb = *reinterpret_cast<f32x4*>(v + base + cc); }
      if (STEP && has_g) {
        if (la) ga = *reinterpret_cast<const f32x4*>(g + base + ca);
        if (lb) gb = *reinterpret_cast<const f32x4*>(g + base + cc);
      }
      for (int f0 = from; f0 < upto; f0 += 64) {              // (at most one round under the rolling policy)
        const int n = upto - f0 < 64 ? upto - f0 : 64;
        const f32x4 en = lazy_entries(hist, f0, n);
        lazy_replay(pa, ma, va, en, f0, n, hist, b1, b2, eps);
        if (cc - lane * 4 < C) lazy_replay(pb, mb, vb, en, f0, n, hist, b1, b2, eps);
      }
      if (STEP) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float pe = pa[e], me = ma[e], ve = va[e];
          adam_elem(pe, ga[e] * coef, me, ve, step_size, b1, b2, eps, inv_sqrt_bc2);
          pa[e] = pe; ma[e] = me; va[e] = ve;
          pe = pb[e]; me = mb[e]; ve = vb[e];
          adam_elem(pe, gb[e] * coef, me, ve, step_size, b1, b2, eps, inv_sqrt_bc2);
          pb[e] = pe; mb[e] = me; vb[e] = ve;
        }
      }
      if (la) { *reinterpret_cast<f32x4*>(p + base + ca) = pa; *reinterpret_cast<f32x4*>(m + base + ca) = ma; *reinterpret_cast<f32x4*>(v + base + ca) = va; }
      if (lb) { *reinterpret_cast<f32x4*>(p + base + cc) = pb; *reinterpret_cast<f32x4*>(m + base + cc) = mb; *reinterpret_cast<f32x4*>(v + base + cc) = vb; }
    }
    if (ZERO_G) {
      if (la) *reinterpret_cast<f32x4*>(g + base + ca) = z4;
      if (lb) *reinterpret_cast<f32x4*>(g + base + cc) = z4;
    }
  }
}

// mode 0: the flagged rows (flags[r] == hist[0] + 1) are brought up to step hist[0] and their gradient rows cleared -- call between
// rows_mark and the lookup / the backward's scatter-add; mode 1: every row is brought up to hist[0] (flush: before anything else reads
// the table -- checkpoints, evaluation, the dense kernels)
template <int MODE>
__global__ void __launch_bounds__(64 * VMMT_ROWS_WAVES) rows_catchup_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                                            float* __restrict__ v, int R, int C, const int* __restrict__ flags,
                                                                            int* __restrict__ last, int* __restrict__ hist, float b1, float b2, float eps,
                                                                            bf16_t* __restrict__ shadow, long ld_shadow) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r0 = blockIdx.x * 64;
  const int upto = hist[0];
  int lst = upto;
  bool act = false;
  if (r0 + lane < R) {
    lst = last[r0 + lane];
    // (which wave takes which row must not depend on `last`: the waves of a workgroup read it at different times, and a row another wave
    //  has already finished would drop out of a late wave's ballot and shift every later row to another wave -- or to none)
    act = MODE == 0 ? flags[(long)((upto + 1) & 1) * R + r0 + lane] == upto + 1 : true;
  }
  unsigned long long todo = __ballot(act);
  for (int k = 0; todo; ++k) {
    const int b = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    if (k % VMMT_ROWS_WAVES != wave) continue;
    const int from = __builtin_amdgcn_readlane(lst, b);
    if (MODE == 1 && from >= upto) continue;
    lazy_row<false, MODE == 0>(p, g, m, v, (long)(r0 + b) * C, C, from, upto, hist, b1, b2, eps, false, 0.f, 0.f, 0.f);
    if (lane == 0 && from < upto) last[r0 + b] = upto;
    if (MODE == 0 && shadow) {          // the row's compute copy (its own lanes wrote the row just now: the same addresses, in program order)
      const float* pr = p + (long)(r0 + b) * C;
      bf16_t* sr = shadow + (long)(r0 + b) * ld_shadow;
      for (int c = lane * 4; c < C; c += 256) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(pr + c);
        uint2 o;
        o.x = (uint32_t)f2bf(q[0]) | ((uint32_t)f2bf(q[1]) << 16);
        o.y = (uint32_t)f2bf(q[2]) | ((uint32_t)f2bf(q[3]) << 16);
        *reinterpret_cast<uint2*>(sr + c) = o;
      }
    }
  }
}

// the update `step`: flagged rows take their gradient (after any zero-gradient steps still missing), the rolling rows
// (r % roll == step % roll, not flagged) are brought up to `step` without one; workgroup 0 records the step in the ring.
// A skipped step (guard word set) is recorded as such and changes nothing else.
__global__ void __launch_bounds__(64 * VMMT_ROWS_WAVES) adam_rows_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                                         float* __restrict__ v, int R, int C, const int* __restrict__ flags,
                                                                         int* __restrict__ last, int* __restrict__ hist, int step, int roll, float b1, float b2,
                                                                         float eps, float step_size, float inv_sqrt_bc2, float max_norm,
                                                                         const float* __restrict__ sumsq, float grad_scale, int* __restrict__ skip) {
  const bool skipped = skip && __hip_atomic_load(skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float* en = const_cast<float*>(lazy_entry(hist, step));
    en[0] = step_size; en[1] = inv_sqrt_bc2; en[2] = skipped ? 0.f : 1.f; en[3] = __int_as_float(step);
    hist[0] = step;         // (read by the NEXT forward's mark / catch-up launches only: nothing in this launch looks at it)
    if (skipped) atomicAdd(skip + 1, 1);
  }
  if (skipped) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r0 = blockIdx.x * 64;
  int lst = step;
  bool hg = false, act = false;
  if (r0 + lane < R) {
    lst = last[r0 + lane];
    hg = flags[(long)(step & 1) * R + r0 + lane] == step;
    act = hg || (roll > 0 && (r0 + lane) % roll == step % roll);       // (not a function of `last`: see rows_catchup_kernel)
  }
  unsigned long long todo = __ballot(act);
  const unsigned long long with_g = __ballot(hg);
  if (!todo) return;
  float coef = grad_scale;
  if (max_norm > 0.f) {
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < VMMT_SUMSQ_SLOTS; ++k) ss += sumsq[k];
    const float c = max_norm / (sqrtf(ss) * grad_scale + 1e-6f);
    if (c < 1.f) coef *= c;
  }
  for (int k = 0; todo; ++k) {
    const int b = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    if (k % VMMT_ROWS_WAVES != wave) continue;
    const int from = __builtin_amdgcn_readlane(lst, b);
    if (!((with_g >> b) & 1) && from >= step) continue;          // a rolling row that is up to date already
    lazy_row<true, false>(p, g, m, v, (long)(r0 + b) * C, C, from, step - 1, hist, b1, b2, eps, (with_g >> b) & 1, coef, step_size, inv_sqrt_bc2);
    if (lane == 0) last[r0 + b] = step;
  }
}

// ||g||^2 over the flagged rows (the other rows hold whatever an earlier step left) in ONE launch: a workgroup adds the row sums of its
// 64-row block in row order, leaves the block's sum in `blksq` and takes a ticket; the workgroup whose ticket is the last adds the block
// sums in block order into the slot total: deterministic, like sumsq_kernel (same hand-off).  blksq: f32 [ceil(R / 64) + 1], the last
// word is the ticket (zero before the first launch; it resets itself).
__global__ void __launch_bounds__(64 * VMMT_ROWS_WAVES) rows_sumsq_kernel(const float* __restrict__ g, int R, int C, const int* __restrict__ flags,
                                                                          const int* __restrict__ hist, float* __restrict__ blksq,
                                                                          float* __restrict__ total) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r0 = blockIdx.x * 64;
  const int gen = hist[0] + 1;
  const bool act = r0 + lane < R && flags[(long)(gen & 1) * R + r0 + lane] == gen;
  __shared__ float rowsum[64];
  __shared__ int is_last;
  if (wave == 0) rowsum[lane] = 0.f;
  __syncthreads();
  unsigned long long todo = __ballot(act);
  for (int k = 0; todo; ++k) {
    const int b = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    if (k % VMMT_ROWS_WAVES != wave) continue;
    const float* x = g + (long)(r0 + b) * C;
    float a = 0.f;
    for (int c = lane * 4; c < C; c += 256) { const f32x4 q = *reinterpret_cast<const f32x4*>(x + c); a += q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]; }
    a = wave_sum(a);
    if (lane == 0) rowsum[b] = a;
  }
  __syncthreads();
  const int nblk = (int)gridDim.x;
  unsigned* ticket = reinterpret_cast<unsigned*>(blksq + nblk);
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int r = 0; r < 64; ++r) s += rowsum[r];
    __hip_atomic_store(blksq + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == (unsigned)nblk - 1);
  }
  __syncthreads();
  if (!is_last) return;
  // the block sums in block order: thread i takes blocks i, i + 1024, ... and the partial sums are folded in thread order
  float s = 0.f;
  for (int k = threadIdx.x; k < nblk; k += blockDim.x) s += __hip_atomic_load(blksq + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  s = wave_sum(s);
  __shared__ float red[VMMT_ROWS_WAVES];
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < VMMT_ROWS_WAVES; ++k) t += red[k];
    *total = t;
    *ticket = 0u;
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

extern "C" int vmmt_rows_mark(const int64_t* ids, int64_t n, int32_t* flags, int R, const int32_t* hist, void* stream) {
  using namespace vmmt;
  if (!ids || !flags || !hist || n < 0 || R <= 0) return VMMT_EINVAL;
  if (n == 0) return VMMT_OK;
  hipLaunchKernelGGL(rows_mark_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const long long*)ids, (long)n, flags, R, hist);
  return check_launch();
}

static bool lazy_rows_ok(const void* p, const void* g, const void* m, const void* v, int R, int C) {
  return p && m && v && R > 0 && C > 0 && C % 4 == 0 && ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
}

extern "C" int vmmt_rows_catchup(float* p, float* g, float* m, float* v, int R, int C, const int32_t* flags, int32_t* last, int32_t* hist,
                                 float beta1, float beta2, float eps, int mode, void* stream) {
  using namespace vmmt;
  if (!lazy_rows_ok(p, g, m, v, R, C) || !last || !hist || (mode != 0 && mode != 1) || (mode == 0 && (!flags || !g))) return VMMT_EINVAL;
  const dim3 grid((unsigned)((R + 63) / 64)), block(64 * VMMT_ROWS_WAVES);
  if (mode == 0) hipLaunchKernelGGL(rows_catchup_kernel<0>, grid, block, 0, (hipStream_t)stream, p, g, m, v, R, C, flags, last, hist, beta1, beta2, eps,
                                    (bf16_t*)nullptr, 0L);
  else hipLaunchKernelGGL(rows_catchup_kernel<1>, grid, block, 0, (hipStream_t)stream, p, g, m, v, R, C, flags, last, hist, beta1, beta2, eps,
                          (bf16_t*)nullptr, 0L);
  return check_launch();
}

extern "C" int vmmt_rows_catchup_shadow(float* p, float* g, float* m, float* v, int R, int C, const int32_t* flags, int32_t* last, int32_t* hist,
                                        float beta1, float beta2, float eps, void* shadow_bf16, int64_t ld_shadow, void* stream) {
  using namespace vmmt;
  if (!lazy_rows_ok(p, g, m, v, R, C) || !last || !hist || !flags || !g || !shadow_bf16 || ld_shadow < C || (ld_shadow & 7) || (((uintptr_t)shadow_bf16) & 15))
    return VMMT_EINVAL;
  hipLaunchKernelGGL(rows_catchup_kernel<0>, dim3((unsigned)((R + 63) / 64)), dim3(64 * VMMT_ROWS_WAVES), 0, (hipStream_t)stream, p, g, m, v, R, C, flags,
                     last, hist, beta1, beta2, eps, (bf16_t*)shadow_bf16, (long)ld_shadow);
  return check_launch();
}

extern "C" int vmmt_adam_rows_step(float* p, float* g, float* m, float* v, int R, int C, const int32_t* flags, int32_t* last,
                                   int32_t* hist, float lr, float beta1, float beta2, float eps, int step, int roll, float max_norm,
                                   const float* sumsq, float grad_scale, const int32_t* skip, void* stream) {
  using namespace vmmt;
  if (!lazy_rows_ok(p, g, m, v, R, C) || !g || !flags || !last || !hist || step < 1 || roll < 0 || roll > VMMT_LAZY_HIST / 4 ||
      (max_norm > 0.f && !sumsq))
    return VMMT_EINVAL;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = (float)(lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));        // as vmmt_adam_step
  hipLaunchKernelGGL(adam_rows_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64 * VMMT_ROWS_WAVES), 0, (hipStream_t)stream, p, g, m, v, R, C, flags,
                     last, hist, step, roll, beta1, beta2, eps, step_size, inv_sqrt_bc2, max_norm, sumsq, grad_scale, const_cast<int*>(skip));
  return check_launch();
}

extern "C" int vmmt_sumsq_rows(const float* g, int R, int C, const int32_t* flags, const int32_t* hist, float* rowsq, float* scratch, int slot,
                               void* stream) {
  using namespace vmmt;
  if (!g || !flags || !hist || !rowsq || !scratch || R <= 0 || C <= 0 || slot < 0 || slot >= VMMT_SUMSQ_SLOTS || (((uintptr_t)g) & 15) || C % 4 != 0)
    return VMMT_EINVAL;
  hipLaunchKernelGGL(rows_sumsq_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64 * VMMT_ROWS_WAVES), 0, (hipStream_t)stream, g, R, C, flags, hist, rowsq,
                     scratch + slot);
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

static int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int step,
                       float max_norm, const float* sumsq, float grad_scale, int max_blocks, void* shadow_bf16, int64_t sh_lo, int64_t sh_n,
                       int64_t hole_lo, int64_t hole_n, const int32_t* skip, void* stream);

extern "C" int vmmt_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                              float eps, int step, float max_norm, const float* sumsq, float grad_scale, int max_blocks,
                              void* shadow_bf16, const int32_t* skip, void* stream) {
  return adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, step, max_norm, sumsq, grad_scale, max_blocks, shadow_bf16, 0, shadow_bf16 ? n : 0, n, 0,
                     skip, stream);
}

extern "C" int vmmt_adam_step_ranges(float* p, const float* g, float* m, float* v, int64_t span, int64_t hole_lo, int64_t hole_n, float lr,
                                     float beta1, float beta2, float eps, int step, float max_norm, const float* sumsq, float grad_scale,
                                     int max_blocks, void* shadow_bf16, int64_t shadow_lo, int64_t shadow_n, const int32_t* skip, void* stream) {
  if (span < 0 || hole_lo < 0 || hole_n < 0 || hole_lo + hole_n > span || ((hole_lo | hole_n) & 3)) return VMMT_EINVAL;
  if (shadow_bf16 && (shadow_lo < 0 || shadow_n < 0 || shadow_lo + shadow_n > span || ((shadow_lo | shadow_n) & 3))) return VMMT_EINVAL;
  if (shadow_bf16 && hole_n > 0 && shadow_lo < hole_lo + hole_n && hole_lo < shadow_lo + shadow_n) return VMMT_EINVAL;      // (the shadowed piece lies on one side of the hole)
  return adam_launch(p, g, m, v, span - hole_n, lr, beta1, beta2, eps, step, max_norm, sumsq, grad_scale, max_blocks, shadow_bf16, shadow_lo,
                     shadow_bf16 ? shadow_n : 0, hole_n > 0 ? hole_lo : span, hole_n, skip, stream);
}

static int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int step,
                       float max_norm, const float* sumsq, float grad_scale, int max_blocks, void* shadow_bf16, int64_t sh_lo, int64_t sh_n,
                       int64_t hole_lo, int64_t hole_n, const int32_t* skip, void* stream) {
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
                     beta2, eps, inv_sqrt_bc2, max_norm, sumsq, grad_scale, (bf16_t*)shadow_bf16, const_cast<int*>(skip), (long)sh_lo, (long)sh_n,
                     (long)hole_lo, (long)hole_n);
  return check_launch();
}

extern "C" int vmmt_version(void) { return 1; }

// Persistent LSTM recurrence kernels for gfx950: ONE launch per sequence instead of one per time step.
//
// Replaces cuDNN's single-call nn.LSTM of the reference (encoder: onmt/Models.py:124-129,140-147 with packed sequences;
// decoder: onmt/VI_Model1.py:106,149-152).  The per-step kernels of lstm.hip re-read their W_hh slice from L2 at every step and pay a
// launch boundary (~1.5 us) per step.  Here a workgroup keeps its slice in REGISTERS for the whole sequence and the workgroups of a
// ROW GROUP hand h_t (forward) / dgates_t (backward) to each other in-launch:
//
//   grid  (row groups of 16 sentences) x (H / 32 unit groups) x directions, 4 waves each; every workgroup must be resident at once
//         (<= 256 workgroups: one per CU; the engine cuts larger batches into row chunks, one launch after the other)
//   step  wave w sweeps the exchange buffer for K QUARTER w of the row group's operand (forward: H/4 units of h_{t-1}, backward: gate
//         w's H columns of dgates_{t+1}) -- stored by the producers in the consumers' MFMA fragment order, 1 KiB contiguous per load
//         instruction  ->  v_mfma_f32_16x16x32_bf16 against its register-resident W_hh fragments  ->  every wave leaves its
//         quarter's partial sums in LDS, a cell (two per lane) adds the four in the per-step kernels' order  ->  cell update (c /
//         dL/dc stay in registers)  ->  tile through LDS: into the exchange buffer for the row group, and plain 16-byte stores into
//         the layer's buffers for the kernels that follow.
//   Rounds 2-4 ran this as 32 sentences x 16 units per workgroup with the slice in LDS.  What a workgroup sweeps per step scales with
//   its ROWS, what it keeps resident with its UNITS: 16 x 32 halves the sweep -- the largest piece of a step, out of an L2 that every
//   CU reads at once beside the GEMMs of the other streams -- for the same number of workgroups and MFMAs (round 5, same box: step
//   1.72 -> 1.60 ms at config 2, 17.0 -> 15.0 ms at config 5; idle chip, us per step: forward 4.3 -> 3.4 (H 512), 3.5 -> 3.0 (2 x 256);
//   backward 5.15 -> 4.15, 4.3 -> 3.8).
//
// Hand-off protocol, forward (cdna_hip_programming.md Guideline 16, form R2: the data IS the flag): a granule is ONE naturally aligned
// 8-byte {two bf16 values, 32-bit tag} written by one sc1 (write-through) store; tag = epoch of the launch * 4096 + step + 1, never
// 0.  Consumers re-read their granules with sc1 loads (L1 bypassed) until every tag matches: no flag, no fence, no drain, no
// barrier on the hand-off, one memory latency per step.  A first version with sc1 payload + vmcnt drain + one counter per row
// group + poll + barrier + sc1 loads measured 7 (decoder) to 14 us (encoder) per step -- two more serial memory round trips per
// hop and 256 pollers on one line.  Backward: the payload is 4x the forward's and the sweep is what bounds the step, so the tile
// travels DENSE and validity separately (see lstm_seq_bwd_kernel).  The exchange buffer is double-buffered by step parity (a producer
// can only be one step ahead of the slowest consumer of its row group); the launch epoch lives in device memory and is advanced by
// the last workgroup to finish, so no memset precedes a launch.  Nothing depends on dispatch order or XCD placement.  Spins are
// bounded: a wave that waits longer than ~2 s sets the error word and stops waiting (the launch then finishes with garbage;
// Engine.lstm_seq_errors() reports it).
#include "common.hpp"
#include "vmmt.h"

namespace vmmt {

struct SeqDirF {      // == vmmt_lstm_dir_fwd (checked below)
  const void* h_prev; long ld_hprev;
  const float* c_prev; long ld_cprev;
  const void* w_hh; long ld_w;
  const float* gx; long ld_gx;
  const float* gx2; long ld_gx2;
  void* gates; long ld_gates;
  float* c_out; long ld_c;
  void* h_out; long ld_h;
  void* h_n; long ld_hn;
  float* c_n; long ld_cn;
  int t, capture;
};
static_assert(sizeof(SeqDirF) == sizeof(vmmt_lstm_dir_fwd), "descriptor layouts must match");

struct SeqArgsF {
  const SeqDirF* steps;          // DEVICE array [nsteps][ndir]
  const long long* lens;
  unsigned* sync;                // [0] launch epoch, [1] finished-workgroup count, [2] error word
  unsigned long long* xchg;      // granules [ndir][ngroups][2 slots][32 rows][H / 2]
  int B, nsteps, ndir, ngroups;
};

#ifdef VMMT_SEQ_PROBE        // tools/probe/lstm_seq_probe.hip: in-kernel timestamps (100 MHz), [block x][step][stamp]
__device__ unsigned long long vmmt_seq_ts[8 * 64 * 8];
__device__ unsigned vmmt_seq_xcc[512];          // [block] = XCC id | transport flag << 8
#define SEQ_TS(i) do { if (threadIdx.x == 0 && blockIdx.x < 8 && t < 64) \
    vmmt_seq_ts[(blockIdx.x * 64 + t) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define SEQ_TS(i) do {} while (0)
#endif

typedef __attribute__((address_space(3))) void lds_void_seq;
typedef __attribute__((address_space(1))) const void glb_cvoid_seq;
typedef float f32x4_s __attribute__((ext_vector_type(4)));
typedef float f32x2_s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ u32x4 load16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);          // aux 16 = sc1: served by L2 / memory, never a stale L1 line
}
__device__ __forceinline__ void store16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, 16);             // write-through
}

// The step descriptors are written by the host before the launch and never change: read them through the constant address
// space, i.e. with scalar loads into SGPRs (a plain global pointer makes hipcc use vector loads and then waterfall loops around
// every buffer instruction, because it can no longer prove the descriptor wave-uniform).
typedef const SeqDirF __attribute__((address_space(4))) * SeqDirFConstPtr;
__device__ __forceinline__ SeqDirF load_desc(const SeqDirF* steps, long idx) {
#if defined(__HIP_DEVICE_COMPILE__)
  return reinterpret_cast<SeqDirFConstPtr>(reinterpret_cast<uintptr_t>(steps))[idx];
#else
  return steps[idx];
#endif
}

// What a step needs of the NEXT step's descriptor while it runs (the operands it prefetches), read with scalar loads at the TOP of
// the step -- where their latency (a miss of the scalar cache: 0.3-0.5 us) hides behind the wait for the row group -- together with
// one word of each of the descriptor's other cache lines, so that the full read one step later hits.
struct SeqNextF { const void* h_prev; const float* gx; long ld_gx; const float* gx2; long ld_gx2; void* h_out; int t; };
__device__ __forceinline__ SeqNextF load_next(const SeqDirF* steps, long idx) {
  SeqNextF r;
#if defined(__HIP_DEVICE_COMPILE__)
  const SeqDirFConstPtr p = reinterpret_cast<SeqDirFConstPtr>(reinterpret_cast<uintptr_t>(steps)) + idx;
#else
  const SeqDirF* p = steps + idx;
#endif
  r.h_prev = p->h_prev; r.gx = p->gx; r.ld_gx = p->ld_gx; r.gx2 = p->gx2; r.ld_gx2 = p->ld_gx2; r.h_out = p->h_out; r.t = p->t;
  return r;
}

__device__ __forceinline__ bool timed_out(unsigned long long t0) { return wall_clock64() - t0 > 200000000ull; }   // 2 s at 100 MHz

// A bounded wait ran out: the launch's own error word, and -- when the caller left a pointer in the sync words (vmmt.h:
// VMMT_SEQ_GUARD_WORD) -- the caller's sticky GUARD word, which vmmt_adam_step reads: the optimiser update of a step whose recurrence
// produced garbage is skipped on the device, long before the host learns of it (Engine._seq_timeout_fallback)
__device__ __forceinline__ void seq_fail(unsigned* sync, unsigned code) {
  __hip_atomic_store(sync + 2, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (a GLOBAL store, spelled out: through the generic pointer it is a FLAT instruction, and a flat access that may be pending on some path
  //  into a block makes the compiler wait with vmcnt(0) there instead of counting -- the sweeps' waits sit behind such paths)
  typedef __attribute__((address_space(1))) unsigned glb_u32_seq;
  glb_u32_seq* guard = (glb_u32_seq*)*reinterpret_cast<unsigned* const*>(sync + VMMT_SEQ_GUARD_WORD);
  if (guard) __hip_atomic_store(guard, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ==============================================================================================================
// Forward recurrence: gates_t[16 x (4 gates x 32 units)] = h_{t-1}[16 x H] W_hh slice^T, then the cell.  Wave w multiplies K QUARTER
// w (the k-step blocks of H/128 producers) against all 128 columns: its 8 * H/128 B fragments of W_hh come straight from memory into
// registers, once (128 registers at H = 512, 256 at H = 1024 -- one wave per SIMD has 512).  A producer's h tile (16 x 32) is ONE
// k-step block of the consumers' order: wave 0 stores its "lo" KiB, wave 1 its "hi" KiB.  Fold: every wave leaves its quarter's
// 16 x 128 sums in LDS (32 KiB), a cell adds the four in lstm_step_fwd_fast's order ((q0 + q1) + q2) + q3: the same bits as the
// per-step kernels (H = 1024: those walk the reduction in chunks of 512, another order of the f32 additions: equal within a bf16 ulp).
// H = 64: two k-steps -- quarters 0 and 1 hold one each, 2 and 3 none (as lstm_step_fwd_fast).
// ==============================================================================================================
template <int H> struct SeqCfg {
  static constexpr int NKS = H / 32, KQ = NKS >= 4 ? NKS / 4 : 1, S = H / 32;
  static constexpr int RED_BYTES = 4 * 8 * 4 * 64 * 4;         // [quarter = wave][gate][unit half][lane] x f32x4
  static constexpr int HROW = 40;                              // h tile [16 rows][32 units] bf16, rows padded to 80 bytes
  static constexpr int HT_BYTES = 16 * HROW * 2;
  static constexpr int LDS = RED_BYTES + HT_BYTES;
  static constexpr long SLOT = (long)NKS * 2048;               // one step's h in fragment order: [k-step][lo / hi][64 lanes] x 16 bytes
};

template <int H>
__global__ void __launch_bounds__(256) lstm_seq_fwd_kernel(SeqArgsF a) {
  using Cf = SeqCfg<H>;
  constexpr int KQ = Cf::KQ, S = Cf::S, HROW = Cf::HROW, NKS = Cf::NKS;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* red = reinterpret_cast<float*>(lds);
  bf16_t* htile = reinterpret_cast<bf16_t*>(lds + Cf::RED_BYTES);
  const int B = a.B, ndir = a.ndir;
  const int bid = blockIdx.x;
  // ---- role of this workgroup: (direction k, row group rg, unit slice).  Workgroups b and b + 8 are observed to share an XCD
  //      (round-robin dispatch), so all S = H/32 workgroups of a group are taken from ONE residue class
  //      b % 8.  That is a speed choice only: which transport a group uses is decided below from the XCC ids the hardware reports.
  //      The grid is padded to 8 * S * ceil(groups / 8) workgroups for that (a batch of 40: 3 groups of 16 slices in 128 workgroups instead of
  //      48, a group per XCD; the workgroups without a group leave at once): with 48 in a row the groups lay across all eight XCDs and took the
  //      write-through transport -- 3.7 instead of 3.1 us per backward step, slower than a batch of 256.
  const int grp = (bid % 8) + 8 * ((bid / 8) / S), slice = (bid / 8) % S;
  if (grp >= a.ngroups * ndir) return;
  const int total = a.ngroups * ndir * S;                       // the workgroups that take part
  const int k = grp / a.ngroups, rg = grp % a.ngroups;
  const int m0 = rg * 16, u0 = slice * 32;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // (H = 64 has two k-steps for four waves; from H = 128 on every wave's quarter is whole, and the compiler must know it)
  auto has_k = [&](int q) { return NKS >= 4 || wave * KQ + q < NKS; };
  const int uh = wave & 1, rp = wave >> 1;                      // the two cells of a lane: rows 4*kg + 2*rp + e, unit u0 + 16*uh + n
  const int n = lane & 15, kg = lane >> 4;
  const int u = u0 + uh * 16 + n;
  const unsigned tag0 = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 4096u;
  char* xg = reinterpret_cast<char*>(a.xchg) + (long)grp * (2 * Cf::SLOT);
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(xg);
  // ---- transport of my group.  Every member reports the XCC it runs on; once all S have arrived, a group whose members all
  //      sit on ONE XCD share one L2: its granules are stored with PLAIN stores (they stay in that L2) and the sc1 loads of the
  //      sweep (L1 bypassed) are served from it -- ~200 cycles and L2 bandwidth instead of a round trip through memory at HBM
  //      bandwidth.  A group spread over several XCDs keeps the placement-independent form: sc1 (write-through) stores.
  {
    int* flag = reinterpret_cast<int*>(lds);
    if (threadIdx.x == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned* arrive = a.sync + 4 + 2 * grp;
      __hip_atomic_fetch_or(arrive + 1, 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t0 = wall_clock64();
      int f = 0;
      for (;;) {
        if (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)S) {
          f = __builtin_popcount(__hip_atomic_load(arrive + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 1;
          break;
        }
        if (timed_out(t0)) { seq_fail(a.sync, 0x300u); f = -1; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      *flag = f;
#ifdef VMMT_SEQ_PROBE
      vmmt_seq_xcc[bid & 511] = (xcc & 15u) | ((unsigned)(f & 3) << 8);
#endif
    }
  }
  bool alive = true;                                            // false once a wait timed out: stop waiting, finish the launch
  bool same_xcd = false;
  // ---- this wave's B fragments of W_hh -> registers, once: rows g*H + u0 + 16*h2 + n (gate g, unit), columns (wave*KQ + q)*32 + kg*8 .. +8
  bf16x8 wreg[KQ][4][2];
  {
    const SeqDirF d0 = load_desc(a.steps, k);
    const bf16_t* wp = reinterpret_cast<const bf16_t*>(d0.w_hh);
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int ks = has_k(q) ? wave * KQ + q : 0;   // (H = 64: the waves without a k-step load one they never use)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
          wreg[q][g][h2] = *reinterpret_cast<const bf16x8*>(wp + ((long)g * H + u0 + h2 * 16 + n) * d0.ld_w + ks * 32 + kg * 8);
    }
  }
  int rows[2];
  int len[2];                                                   // (sentence lengths fit 31 bits; two registers less than long long)
  float c_reg[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    rows[e] = m0 + kg * 4 + 2 * rp + e;
    const int rr = rows[e] < B ? rows[e] : B - 1;
    len[e] = a.lens ? (int)a.lens[rr] : 0;
    c_reg[e] = 0.f;
  }
  const int arow = min(m0 + n, B - 1);                          // the h_{t-1} row this lane's A fragments come from (step 0)
  __syncthreads();
  {
    const int f = *reinterpret_cast<const int*>(lds);              // (not through a volatile generic pointer: that is a FLAT load, see seq_fail)
    same_xcd = f == 1;
    alive = f >= 0;
  }
  __syncthreads();                                              // the flag word is part of the fold buffer

  // x W_ih^T + b of the cells' gates (and the decoder's z W_z^T + b beside it), NEXT step's: they do not depend on the recurrence.  Requested
  // behind the sweep; the two are only added where they are used (the add would wait for the loads), and the step descriptor they come
  // from is read at the TOP of the step before -- a scalar load that misses its cache took 0.3-0.5 us between the sweep and the MFMAs
  float gxn[2][4], gx2n[2][4];
  // (buffer loads: a scalar base per operand, one 32-bit offset per cell, the gate as scalar offset; branch-free -- without a second
  //  addend the first is read twice -- so that the compiler can count these loads out of the wait for the sweep's)
  auto fetch_gx = [&](const SeqNextF& dd) {
    const bool two = dd.gx2 != nullptr;
    const __amdgpu_buffer_rsrc_t gr = make_rsrc(dd.gx), g2r = make_rsrc(two ? dd.gx2 : dd.gx);
    const int ld1 = (int)dd.ld_gx, ld2 = two ? (int)dd.ld_gx2 : ld1;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int rr = rows[e] < B ? rows[e] : B - 1;
      const int off = (rr * ld1 + u) * 4, off2 = (rr * ld2 + u) * 4;
#pragma unroll
      for (int g = 0; g < 4; ++g) gxn[e][g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, off, g * H * 4, 0));
#pragma unroll
      for (int g = 0; g < 4; ++g) gx2n[e][g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g2r, off2, g * H * 4, 0));
    }
  };
  SeqNextF dnext = load_next(a.steps, k);
  fetch_gx(dnext);

  for (int t = 0; t < a.nsteps; ++t) {
    const SeqDirF d = load_desc(a.steps, (long)t * ndir + k);
    if (t + 1 < a.nsteps) {
      dnext = load_next(a.steps, (long)(t + 1) * ndir + k);
      asm volatile("" :: "s"(dnext.h_prev), "s"(dnext.h_out), "s"(dnext.t));       // (the words that are only read for their cache lines)
    }
    SEQ_TS(0);
    float gxv[2][4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
      for (int g = 0; g < 4; ++g) gxv[e][g] = d.gx2 ? gxn[e][g] + gx2n[e][g] : gxn[e][g];
      const int rr = rows[e] < B ? rows[e] : B - 1;
      if (t == 0) c_reg[e] = d.c_prev ? d.c_prev[(long)rr * d.ld_cprev + u] : 0.f;
    }
    // ---- gates = h_{t-1} W_hh^T for this wave's K quarter (KQ k-steps of 32 units) against all 128 columns
    f32x4_s acc[4][2];
    auto zero_acc = [&]() {
#pragma unroll
      for (int g = 0; g < 4; ++g) { acc[g][0] = f32x4_s{0.f, 0.f, 0.f, 0.f}; acc[g][1] = f32x4_s{0.f, 0.f, 0.f, 0.f}; }
    };
    auto mfma_q = [&](int q, const u32x4& afq) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
          acc[g][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afq), wreg[q][g][h2], acc[g][h2], 0, 0, 0);
    };
    zero_acc();
    if (t == 0) {                                               // the initial state: an ordinary [B][ld] bf16 buffer of an earlier kernel
      const __amdgpu_buffer_rsrc_t hr = make_rsrc(d.h_prev);
      const unsigned abase = (unsigned)(((long)arow * d.ld_hprev + kg * 8) * 2);
      u32x4 af[KQ];
#pragma unroll
      for (int q = 0; q < KQ; ++q) af[q] = has_k(q) ? load16_sc1(hr, abase + (unsigned)((wave * KQ + q) * 64)) : u32x4{0u, 0u, 0u, 0u};
      fetch_gx(dnext);
      SEQ_TS(2);
#pragma unroll
      for (int q = 0; q < KQ; ++q)
        if (has_k(q)) mfma_q(q, af[q]);
    } else {
      const unsigned want = tag0 + (unsigned)t;                 // tag of step t-1
      const unsigned gbase = (unsigned)(((t - 1) & 1) * Cf::SLOT + (long)(wave * KQ) * 2048 + lane * 16);
      const unsigned long long t_start = wall_clock64();
      // cheap poll first: ONE granule per producer of this wave's quarter (the last of its "hi" block), so that the waiting waves do
      // not sweep: the full sweep repeated by 1024 waves would by itself saturate the memory system the hand-off travels through
      {
        const unsigned pbase = (unsigned)(((t - 1) & 1) * Cf::SLOT + (long)(wave * KQ + lane) * 2048 + 1024 + 63 * 16 + 8);
        while (alive) {
          unsigned long long g = 0;
          const bool mine = lane < KQ && has_k(lane);
          if (mine) g = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(xg + pbase), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__all(!mine || (unsigned)(g >> 32) == want)) break;
          if (timed_out(t_start)) {
            if (lane == 0) seq_fail(a.sync, 0x200u + (unsigned)t);
            alive = false;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      SEQ_TS(1);
      u32x4 lo[KQ], hi[KQ];
      auto sweep = [&]() {
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          if (has_k(q)) {
            lo[q] = load16_sc1(xr, gbase + (unsigned)(q * 2048));
            hi[q] = load16_sc1(xr, gbase + (unsigned)(q * 2048 + 1024));
          } else {
            lo[q] = hi[q] = u32x4{0u, want, 0u, want};
          }
        }
      };
      // the tags are checked k-step by k-step and every k-step's MFMAs go out as soon as ITS granules are in (loads return in order: the
      // waits are counted): the products run under the rest of the sweep instead of behind the last tag.  A sweep that meets a granule of the
      // step before -- behind the cheap poll that is rare -- has multiplied it already: the accumulators start over
      auto multiply = [&]() {
        bool ok = true;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          ok = ok && lo[q][1] == want && lo[q][3] == want && hi[q][1] == want && hi[q][3] == want;
          if (has_k(q)) mfma_q(q, u32x4{lo[q][0], lo[q][2], hi[q][0], hi[q][2]});
        }
        return __all(ok);
      };
      // first attempt, straight-line: the next step's gate operands are requested behind the sweep's loads -- which return first: the compiler
      // counts these 16 out of the waits for the granules -- and in their shadow (issued behind the completed sweep they cost the step 0.4 us
      // of address arithmetic and issue)
      sweep();
      fetch_gx(dnext);                                          // (last step: the last descriptor's once more, unused)
      bool done = multiply();
      while (!done && alive) {
        if (timed_out(t_start)) {
          if (lane == 0) seq_fail(a.sync, 0x100u + (unsigned)t);
          alive = false;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        sweep();
        zero_acc();
        done = multiply();
      }
      SEQ_TS(2);
    }
    // (fold buffer: [quarter = wave][gate][unit half][lane] x f32x4 -- one 16-byte write per accumulator; a lane's two cells are rows
    //  2*rp, 2*rp + 1 of its own lane slot: one 8-byte read per (gate, quarter))
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) reinterpret_cast<f32x4_s*>(red)[((wave * 4 + g) * 2 + h2) * 64 + lane] = acc[g][h2];
    SEQ_TS(3);
    __syncthreads();
    SEQ_TS(4);
    LstmCell cellv[2];
    bool valid[2];
    float pq[2][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x2_s* rq = reinterpret_cast<const f32x2_s*>(red + (((g * 2 + uh) * 64 + lane) * 4 + 2 * rp));
      const f32x2_s q0 = rq[0], q1 = rq[8 * 64 * 2], q2 = rq[16 * 64 * 2], q3 = rq[24 * 64 * 2];
#pragma unroll
      for (int e = 0; e < 2; ++e) pq[e][g] = (((0.f + q0[e]) + q1[e]) + q2[e]) + q3[e];      // lstm_step_fwd_fast's fold order
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float* p = pq[e];
      cellv[e] = lstm_cell_math(p[0] + gxv[e][0], p[1] + gxv[e][1], p[2] + gxv[e][2], p[3] + gxv[e][3], c_reg[e]);
      const float cp = c_reg[e];
      valid[e] = !a.lens || d.t < len[e];
      c_reg[e] = valid[e] ? cellv[e].c : cp;                    // frozen state at pads (packed-sequence semantics)
      htile[(kg * 4 + 2 * rp + e) * HROW + uh * 16 + n] = f2bf(valid[e] ? cellv[e].h : 0.f);
    }
    SEQ_TS(5);
    __syncthreads();
    SEQ_TS(6);
    // ---- publish h_t.  Waves 0-1: the tile as tagged granules = k-step block `slice` of the row group's order, wave 0 its "lo" KiB
    //      (units 8*kg + 0..3 of row n: two granules), wave 1 its "hi" KiB (units 8*kg + 4..7); wave 2: the layer's output buffer
    if (wave < 2) {
      const unsigned* hw = reinterpret_cast<const unsigned*>(htile + n * HROW + kg * 8 + wave * 4);
      const unsigned tg = tag0 + (unsigned)t + 1u;
      const unsigned off = (unsigned)((t & 1) * Cf::SLOT + (long)slice * 2048 + wave * 1024 + lane * 16);
      const u32x4 v = u32x4{hw[0], tg, hw[1], tg};
      if (same_xcd) *reinterpret_cast<u32x4*>(xg + off) = v;   // stays in the group's L2
      else store16_sc1(xr, off, v);                             // write-through: visible to every XCD
    } else if (wave == 2) {
      const int r = lane >> 2, o = lane & 3;
      if (m0 + r < B)
        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(d.h_out) + ((long)(m0 + r)) * d.ld_h + u0 + o * 8) =
            *reinterpret_cast<const u32x4*>(htile + r * HROW + o * 8);
    }
    // ---- what only later kernels read (saved gate activations, cell state, captured final state): behind the publish
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int b = rows[e];
      if (b < B) {
        const LstmCell& cl = cellv[e];
        bf16_t* gs = reinterpret_cast<bf16_t*>(d.gates) + (long)b * d.ld_gates + u;
        gs[0] = f2bf(cl.i); gs[H] = f2bf(cl.f); gs[2 * H] = f2bf(cl.g); gs[3 * (long)H] = f2bf(cl.o);
        d.c_out[(long)b * d.ld_c + u] = c_reg[e];
        const bool cap = d.capture == 3 || (d.capture == 1 && d.t == len[e] - 1) || (d.capture == 2 && d.t == 0);
        if (cap && d.h_n) {
          reinterpret_cast<bf16_t*>(d.h_n)[(long)b * d.ld_hn + u] = f2bf(cl.h);
          d.c_n[(long)b * d.ld_cn + u] = cl.c;
        }
      }
    }
    SEQ_TS(7);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (__hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)total - 1) {
      __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int g2 = 0; g2 < a.ndir * a.ngroups; ++g2) {
        __hip_atomic_store(a.sync + 4 + 2 * g2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.sync + 5 + 2 * g2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int H>
static int launch_seq_fwd(const SeqArgsF& a, hipStream_t st) {
  constexpr int sm = SeqCfg<H>::LDS;
  hipLaunchKernelGGL(lstm_seq_fwd_kernel<H>, dim3(8 * (H / 32) * ((a.ngroups * a.ndir + 7) / 8)), dim3(256), sm, st, a);
  return check_launch();
}

// ==============================================================================================================
// Backward recurrence, same structure: dh_t[16 x 32] = dgates_{t+1}[16 x 4H] W_hh^T slice[32 x 4H]^T, then the cell backward.
// What the workgroups of a row group exchange is dgates (4H values per sentence and step, 4x the forward's payload): every
// workgroup reads the whole 16 x 4H tile of its group at every step, and that traffic is what bounds the step.  So the tile travels
// DENSE (in the consumers' fragment order) and validity travels separately: a producer wave stores its block, waits until it is
// acknowledged, and then stores a flag (the step's tag); consumers poll the flags and sweep without checking anything.  Against the
// forward kernel's self-validating granules this costs one store acknowledgement (~0.7 us) per hand-off and halves the bytes.
// ==============================================================================================================
struct SeqDirB {      // == vmmt_lstm_dir_bwd
  const void* dgates_next; long ld_dgn;
  const void* w_hh_t; long ld_wt;
  const void* dh_above; long ld_dha;
  const void* gates; long ld_gates;
  const float* c_t; long ld_ct;
  const float* c_prev; long ld_cp;
  float* dc_carry; long ld_dcc;
  void* dgates_out; long ld_dgo;
  const float* dh_n; long ld_dhn;
  const float* dc_n; long ld_dcn;
  float* dh0_out; long ld_dh0;
  int t, inject;
};
static_assert(sizeof(SeqDirB) == sizeof(vmmt_lstm_dir_bwd), "descriptor layouts must match");
typedef const SeqDirB __attribute__((address_space(4))) * SeqDirBConstPtr;
__device__ __forceinline__ SeqDirB load_desc_b(const SeqDirB* steps, long idx) {
#if defined(__HIP_DEVICE_COMPILE__)
  return reinterpret_cast<SeqDirBConstPtr>(reinterpret_cast<uintptr_t>(steps))[idx];
#else
  return steps[idx];
#endif
}
// (what a step prefetches of the NEXT cell step, and one word of each of that descriptor's other cache lines: see SeqNextF)
struct SeqNextB {
  const void* dgates_next; const void* dh_above; long ld_dha; const void* gates; long ld_gates; const float* c_t; long ld_ct;
  const float* c_prev; long ld_cp; void* dgates_out; const float* dh_n; long ld_dhn; const float* dc_n; long ld_dcn; int t;
};
__device__ __forceinline__ SeqNextB load_next_b(const SeqDirB* steps, long idx) {
  SeqNextB r;
#if defined(__HIP_DEVICE_COMPILE__)
  const SeqDirBConstPtr p = reinterpret_cast<SeqDirBConstPtr>(reinterpret_cast<uintptr_t>(steps)) + idx;
#else
  const SeqDirB* p = steps + idx;
#endif
  r.dgates_next = p->dgates_next; r.dh_above = p->dh_above; r.ld_dha = p->ld_dha; r.gates = p->gates; r.ld_gates = p->ld_gates;
  r.c_t = p->c_t; r.ld_ct = p->ld_ct; r.c_prev = p->c_prev; r.ld_cp = p->ld_cp; r.dgates_out = p->dgates_out;
  r.dh_n = p->dh_n; r.ld_dhn = p->ld_dhn; r.dc_n = p->dc_n; r.ld_dcn = p->ld_dcn; r.t = p->t;
  return r;
}
struct SeqArgsB {
  const SeqDirB* steps;
  const long long* lens;
  unsigned* sync;
  unsigned long long* xchg;      // granules [ndir][ngroups][2 slots][2 row halves][4H / 32 k-steps][lo / hi][64 lanes] x 16 bytes
  int B, nsteps, ndir, ngroups;
  int with_dh0;                  // steps[nsteps] is a mode-1 descriptor: dh0_out = dgates_{first step in time} W_hh, no cell backward
};

// ==============================================================================================================
// The workgroup: 16 sentences x 32 hidden units.
//   * wave w multiplies K QUARTER w -- which is gate w's H columns -- against both 16-unit halves of the slice: its 2 * H/32 B
//     fragments of W_hh^T live in REGISTERS for the whole sequence (128 at H = 512, 256 at H = 1024, beside 64 / 128 of sweep)
//   * a producer wave g stores gate g's 16 x 32 block = ONE 1-KiB k-step block of the consumers' fragment order (one store
//     instruction, 8 full lines), and a consumer wave w waits only for the S = H/32 flags of the producers' waves w
//   * fold: every wave leaves its quarter's 16 x 32 sums in LDS, the cells (two per lane) add the four in lstm_step_bwd_fast's
//     order ((q0 + q1) + q2) + q3 with q = even + odd k-step accumulators: the same bits as the per-step kernels (H = 1024: those
//     walk the reduction in chunks of 2048: equal within a bf16 ulp or two)
//   * what the cell backward needs besides dL/dh is computed while the wave waits for the sweep (lstm_cell_bwd_pre)
// At H = 512 the kernel takes 304 registers: the GEMMs of the other streams (224-240) do not fit beside it and wait for it -- by
// measurement (LABNOTES round 5): with a quarter / three eighths / half of the fragments in LDS instead (280 / 264 / 248 registers) the
// step was equal / 20 us / 25 us slower.  The encoder's H = 256 (208) does host them; made exclusive by an LDS pad the step lost 65 us.
// ==============================================================================================================
template <int H> struct SeqCfgB {
  static constexpr int K = 4 * H, NKS = K / 32, KQ = NKS / 4, S = H / 32;
  static constexpr int RED_BYTES = 4 * 2 * 4 * 64 * 4;         // [quarter = wave][unit half][lane] x f32x4
  static constexpr int DROW = 4 * 32 + 8;                      // dgates tile [16 rows][4 gates][32 units] bf16, rows padded by 16 bytes
  static constexpr int DT_BYTES = 16 * DROW * 2;
  static constexpr int LDS = RED_BYTES + DT_BYTES;
  static constexpr long DSLOT = (long)NKS * 1024;              // one step's tile in fragment order: [k-step][64 lanes] x 16 bytes
  static constexpr long GROUP_BYTES = 4L * NKS * 1024;         // per (direction, row group of 16): two slots, then the flags
  static constexpr int NFLAG = 4 * S;                          // per slot: [producer wave = gate][producer slice]
  static_assert(2 * DSLOT + 2 * NFLAG * 4 <= GROUP_BYTES, "flags must fit behind the data");
};

template <int H>
__global__ void __launch_bounds__(256) lstm_seq_bwd_kernel(SeqArgsB a) {
  using Cf = SeqCfgB<H>;
  constexpr int KQ = Cf::KQ, S = Cf::S, DROW = Cf::DROW, NFLAG = Cf::NFLAG;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* red = reinterpret_cast<float*>(lds);
  bf16_t* dtile = reinterpret_cast<bf16_t*>(lds + Cf::RED_BYTES);
  const int B = a.B, ndir = a.ndir;
  const int bid = blockIdx.x;
  const int grp = (bid % 8) + 8 * ((bid / 8) / S), slice = (bid / 8) % S;       // role mapping, padded grid, transport choice: see lstm_seq_fwd_kernel
  if (grp >= a.ngroups * ndir) return;
  const int total = a.ngroups * ndir * S;
  const int k = grp / a.ngroups, rg = grp % a.ngroups;
  const int m0 = rg * 16, u0 = slice * 32;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int uh = wave & 1, rp = wave >> 1;                      // the two cells of a lane: rows 4*kg + 2*rp + e, unit u0 + 16*uh + n
  const int n = lane & 15, kg = lane >> 4;
  const int u = u0 + uh * 16 + n;
  const unsigned tag0 = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 4096u;
  char* xg = reinterpret_cast<char*>(a.xchg) + (long)grp * Cf::GROUP_BYTES;
  constexpr unsigned FLAG_OFF = (unsigned)(2 * Cf::DSLOT);
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(xg);
  {
    int* flag = reinterpret_cast<int*>(lds);
    if (threadIdx.x == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned* arrive = a.sync + 4 + 2 * grp;
      __hip_atomic_fetch_or(arrive + 1, 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t0 = wall_clock64();
      int f = 0;
      for (;;) {
        if (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)S) {
          f = __builtin_popcount(__hip_atomic_load(arrive + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 1;
          break;
        }
        if (timed_out(t0)) { seq_fail(a.sync, 0x400u); f = -1; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      *flag = f;
#ifdef VMMT_SEQ_PROBE
      vmmt_seq_xcc[bid & 511] = (xcc & 15u) | ((unsigned)(f & 3) << 8);
#endif
    }
  }
  bool alive = true, same_xcd = false;
  // ---- this wave's B fragments of W_hh^T -> registers, once: units u0 + 16*h2 + n, columns (wave*KQ + q)*32 + kg*8 .. +8
  bf16x8 wreg[KQ][2];
  {
    const SeqDirB d0 = load_desc_b(a.steps, k);
    const bf16_t* wp = reinterpret_cast<const bf16_t*>(d0.w_hh_t);
#pragma unroll
    for (int q = 0; q < KQ; ++q)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
        wreg[q][h2] = *reinterpret_cast<const bf16x8*>(wp + (long)(u0 + h2 * 16 + n) * d0.ld_wt + (wave * KQ + q) * 32 + kg * 8);
  }
  int rows[2];
  int len[2];                                                   // (sentence lengths fit 31 bits; two registers less than long long)
  float dcc[2];                                                 // dL/dc flowing to the previous step: in registers for the whole sequence
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    rows[e] = m0 + kg * 4 + 2 * rp + e;
    const int rr = rows[e] < B ? rows[e] : B - 1;
    len[e] = a.lens ? (int)a.lens[rr] : 0;
    dcc[e] = 0.f;
  }
  __syncthreads();
  {
    const int f = *reinterpret_cast<const int*>(lds);              // (not through a volatile generic pointer: that is a FLAT load, see seq_fail)
    same_xcd = f == 1;
    alive = f >= 0;
  }
  __syncthreads();                                              // the flag word is part of the fold buffer

  // the cell backward's inputs of the NEXT step (they do not depend on the recurrence), requested behind the sweep's loads.  RAW: a
  // conversion here would wait for its load here; branch-free: an operand the step does not have is read from one it has and dropped
  // where it is used -- both so that the compiler can count these 18 loads out of the wait for the sweep; buffer loads: a scalar base per
  // operand, one 32-bit offset per cell
  unsigned short nxg[2][4], nxa[2];
  float nxf[2][4];
  auto fetch_in = [&](const SeqNextB& dd) {
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(dd.gates), rc = make_rsrc(dd.c_t), rp_ = make_rsrc(dd.c_prev ? dd.c_prev : dd.c_t),
                                 ra = make_rsrc(dd.dh_above ? dd.dh_above : dd.gates), rh = make_rsrc(dd.dh_n ? dd.dh_n : dd.c_t),
                                 rn = make_rsrc(dd.dh_n ? dd.dc_n : dd.c_t);
    const int ldp = dd.c_prev ? (int)dd.ld_cp : (int)dd.ld_ct, lda = dd.dh_above ? (int)dd.ld_dha : (int)dd.ld_gates,
              ldh = dd.dh_n ? (int)dd.ld_dhn : (int)dd.ld_ct, ldn = dd.dh_n ? (int)dd.ld_dcn : (int)dd.ld_ct;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int bb = rows[e] < B ? rows[e] : B - 1;
      const int og = (bb * (int)dd.ld_gates + u) * 2;
#pragma unroll
      for (int g = 0; g < 4; ++g) nxg[e][g] = __builtin_amdgcn_raw_buffer_load_b16(rg, og, g * H * 2, 0);
      nxf[e][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, (bb * (int)dd.ld_ct + u) * 4, 0, 0));
      nxf[e][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp_, (bb * ldp + u) * 4, 0, 0));
      nxa[e] = __builtin_amdgcn_raw_buffer_load_b16(ra, (bb * lda + u) * 2, 0, 0);
      nxf[e][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rh, (bb * ldh + u) * 4, 0, 0));
      nxf[e][3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rn, (bb * ldn + u) * 4, 0, 0));
    }
  };
  SeqNextB dnext = load_next_b(a.steps, k);
  fetch_in(dnext);

  for (int t = 0; t < a.nsteps + a.with_dh0; ++t) {
    const SeqDirB d = load_desc_b(a.steps, (long)t * ndir + k);
    // the next CELL step's operands (the last cell step and the dh0 step fetch the last cell step's once more, unused: the prefetch below
    // is issued without a branch); read here, where a miss of the scalar cache hides behind the wait for the row group
    dnext = load_next_b(a.steps, (long)(t + 1 < a.nsteps ? t + 1 : a.nsteps - 1) * ndir + k);
    asm volatile("" :: "s"(dnext.dgates_next), "s"(dnext.dgates_out), "s"(dnext.t));       // (the words that are only read for their cache lines)
    const bool dh0_step = t == a.nsteps;                        // the gradient of the initial hidden state: GEMM only
    SEQ_TS(0);
    // this step's operands out of last step's prefetch, HERE (the top of the step, where nothing younger than them is in flight): left to
    // the compiler the copies sink to just in front of the next prefetch, behind the sweep's loads, and wait for those
    float gi[2], gf[2], gg[2], go[2], cc[2], cpv[2], dha[2], dhn[2], dcn[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      gi[e] = bf2f(nxg[e][0]); gf[e] = bf2f(nxg[e][1]); gg[e] = bf2f(nxg[e][2]); go[e] = bf2f(nxg[e][3]);
      cc[e] = nxf[e][0];
      cpv[e] = d.c_prev ? nxf[e][1] : 0.f;
      dha[e] = d.dh_above ? bf2f(nxa[e]) : 0.f;
      dhn[e] = d.dh_n ? nxf[e][2] : 0.f;
      dcn[e] = d.dh_n ? nxf[e][3] : 0.f;
      asm volatile("" : "+v"(gi[e]), "+v"(gf[e]), "+v"(gg[e]), "+v"(go[e]), "+v"(cc[e]), "+v"(cpv[e]), "+v"(dha[e]), "+v"(dhn[e]), "+v"(dcn[e]));
      const long bb = rows[e] < B ? rows[e] : B - 1;
      if (t == 0) dcc[e] = d.dc_carry[bb * d.ld_dcc + u];
    }
    // everything of the cell backward that does not need dL/dh: computed below while the wave waits for the step's dgates, and held there
    // (the compiler would otherwise sink it behind the fold's barrier, onto the step's critical path)
    LstmCellBwdPre pre[2];
    bool valid[2], inj[2];
    auto cell_pre = [&]() {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        pre[e] = lstm_cell_bwd_pre(gi[e], gf[e], gg[e], go[e], cc[e]);
        asm volatile("" : "+v"(pre[e].tc), "+v"(pre[e].omt2), "+v"(pre[e].omi), "+v"(pre[e].omf), "+v"(pre[e].omo), "+v"(pre[e].omg2));
        valid[e] = !a.lens || d.t < len[e];
        inj[e] = d.inject == 3 || (d.inject == 1 && d.t == len[e] - 1) || (d.inject == 2 && d.t == 0);
      }
    };
    f32x4_s acc[2][2];                                          // [unit half][even / odd K step of the quarter]
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) { acc[h2][0] = f32x4_s{0.f, 0.f, 0.f, 0.f}; acc[h2][1] = f32x4_s{0.f, 0.f, 0.f, 0.f}; }
    if (d.dgates_next) {
      u32x4 af[KQ];
      if (t == 0) {
        // a chain CONTINUED from an earlier launch: the previous step's dgates are in its plain [B][4H] buffer
        const long rr = (m0 + n) < B ? (m0 + n) : B - 1;
        const bf16_t* src = reinterpret_cast<const bf16_t*>(d.dgates_next) + rr * d.ld_dgn + kg * 8;
#pragma unroll
        for (int q = 0; q < KQ; ++q) af[q] = *reinterpret_cast<const u32x4*>(src + (wave * KQ + q) * 32);
      } else {
        const unsigned want = tag0 + (unsigned)t;               // tag of the step processed just before
        const unsigned long long t_start = wall_clock64();
        {   // gate `wave` of that step, from every slice of the row group: S tags, contiguous
          const unsigned* fp = reinterpret_cast<const unsigned*>(xg + FLAG_OFF) + ((t - 1) & 1) * NFLAG + wave * S;
          while (alive) {
            unsigned g = want;
            if (lane < S) g = __hip_atomic_load(fp + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(g == want)) break;
            if (timed_out(t_start)) {
              if (lane == 0) seq_fail(a.sync, 0x500u + (unsigned)t);
              alive = false;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        SEQ_TS(1);
        const unsigned gbase = (unsigned)(((t - 1) & 1) * Cf::DSLOT + (long)(wave * KQ) * 1024 + lane * 16);
#pragma unroll
        for (int q = 0; q < KQ; ++q) af[q] = load16_sc1(xr, gbase + (unsigned)(q * 1024));
      }
      fetch_in(dnext);
      cell_pre();                                               // in the shadow of the sweep's latency
#pragma unroll
      for (int q = 0; q < KQ; ++q)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
          acc[h2][q & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[q]), wreg[q][h2], acc[h2][q & 1], 0, 0, 0);
    } else {
      fetch_in(dnext);
      cell_pre();
    }
    SEQ_TS(2);
    // ---- fold: this wave's quarter sums (even + odd accumulator, as lstm_step_bwd_fast) to LDS; the cells add the four in order
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {        // [quarter = wave][unit half][lane] x f32x4: one 16-byte write each, one 8-byte read per quarter below
      f32x4_s qs;
#pragma unroll
      for (int r = 0; r < 4; ++r) qs[r] = acc[h2][0][r] + acc[h2][1][r];
      reinterpret_cast<f32x4_s*>(red)[(wave * 2 + h2) * 64 + lane] = qs;
    }
    SEQ_TS(3);
    __syncthreads();
    SEQ_TS(4);
    const f32x2_s* rq2 = reinterpret_cast<const f32x2_s*>(red + ((uh * 64 + lane) * 4 + 2 * rp));
    const f32x2_s fq0 = rq2[0], fq1 = rq2[2 * 64 * 2], fq2 = rq2[4 * 64 * 2], fq3 = rq2[6 * 64 * 2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float dh = (((0.f + fq0[e]) + fq1[e]) + fq2[e]) + fq3[e];
      if (dh0_step) {
        if (rows[e] < B) d.dh0_out[(long)rows[e] * d.ld_dh0 + u] = dh;
        continue;
      }
      // branch-free: a padded position computes on whatever it read and stores zeros
      bf16_t* dt = dtile + (kg * 4 + 2 * rp + e) * DROW + uh * 16 + n;
      float dc = dcc[e];
      dh += dha[e];
      const float dh_inj = dh + dhn[e], dc_inj = dc + dcn[e];
      dh = inj[e] ? dh_inj : dh;
      dc = inj[e] ? dc_inj : dc;
      const LstmCellGrad gr = lstm_cell_bwd_post(pre[e], gi[e], gf[e], gg[e], go[e], cpv[e], dh, dc);
      dt[0] = valid[e] ? f2bf(gr.di) : (bf16_t)0; dt[32] = valid[e] ? f2bf(gr.df) : (bf16_t)0;
      dt[64] = valid[e] ? f2bf(gr.dg) : (bf16_t)0; dt[96] = valid[e] ? f2bf(gr.d_o) : (bf16_t)0;
      dcc[e] = valid[e] ? gr.dc_prev : 0.f;
    }
    if (dh0_step) break;                                        // (uniform) nothing to publish
    SEQ_TS(5);
    __syncthreads();
    SEQ_TS(6);
    // ---- publish dgates_t: wave g stores gate g's block (lane = (8-unit piece o, row r): 1 KiB contiguous = k-step g*S + slice of the
    //      consumers' order), flags it once the store is acknowledged; then the plain [B][4H] buffer for the kernels that follow
    {
      const unsigned tg = tag0 + (unsigned)t + 1u;
      {
        const u32x4 v = *reinterpret_cast<const u32x4*>(dtile + (lane & 15) * DROW + wave * 32 + (lane >> 4) * 8);
        const unsigned off = (unsigned)((t & 1) * Cf::DSLOT + (long)(wave * S + slice) * 1024 + lane * 16);
        if (same_xcd) *reinterpret_cast<u32x4*>(xg + off) = v;   // stays in the group's L2
        else store16_sc1(xr, off, v);                             // write-through: visible to every XCD
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's block has reached the L2 (or memory)
      if (lane == 0) {              // (global stores, spelled out: see seq_fail)
        typedef __attribute__((address_space(1))) unsigned glb_u32_seq;
        glb_u32_seq* fl = (glb_u32_seq*)(reinterpret_cast<unsigned*>(xg + FLAG_OFF) + (t & 1) * NFLAG + wave * S + slice);
        if (same_xcd) __hip_atomic_store(fl, tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(fl, tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      // plain copy: 16 rows x 4 gates x 64 bytes = 256 x 16 bytes
      const int r = threadIdx.x >> 4, g = (threadIdx.x >> 2) & 3, o = threadIdx.x & 3;
      if (m0 + r < B)
        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(d.dgates_out) + ((long)(m0 + r)) * d.ld_dgo + (long)g * H + u0 + o * 8) =
            *reinterpret_cast<const u32x4*>(dtile + r * DROW + g * 32 + o * 8);
    }
    if (t == a.nsteps - 1) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
        if (rows[e] < B) d.dc_carry[(long)rows[e] * d.ld_dcc + u] = dcc[e];
    }
    SEQ_TS(7);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (__hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)total - 1) {
      __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int g2 = 0; g2 < a.ndir * a.ngroups; ++g2) {
        __hip_atomic_store(a.sync + 4 + 2 * g2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.sync + 5 + 2 * g2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int H>
static int launch_seq_bwd(const SeqArgsB& a, hipStream_t st) {
  constexpr int sm = SeqCfgB<H>::LDS;
  hipLaunchKernelGGL(lstm_seq_bwd_kernel<H>, dim3(8 * (H / 32) * ((a.ngroups * a.ndir + 7) / 8)), dim3(256), sm, st, a);
  return check_launch();
}

static bool al16s(const void* p, long ld_elems) { return (((uintptr_t)p) & 15) == 0 && (ld_elems * 2) % 16 == 0; }

}  // namespace vmmt

// Whole forward recurrence of one LSTM layer (both directions) in ONE launch.  `dirs` / `dirs_dev`: the same nsteps x ndir step
// descriptors in host and in device memory (step i = [i*ndir, (i+1)*ndir)); `sync` (vmmt_lstm_seq_sync_words() uint32) and `xchg`
// (vmmt_lstm_seq_xchg_bytes(ndir, B, H) bytes): device scratch private to this call site, zeroed ONCE by the caller when it
// allocates them and never touched by it again.  Same results as nsteps vmmt_lstm_step_fwd calls -- which is what this function
// issues when the persistent kernel does not apply (fp32, H not in {64,128,256,512,1024}, more workgroups than CUs, unaligned rows,
// or steps that are not chained h_prev[t] == h_out[t-1]).
extern "C" int vmmt_lstm_seq_sync_words(void) { return VMMT_SEQ_GUARD_WORD + 4; }   // epoch, finish count, error, pad; per group: arrivals, XCC mask; guard pointer
extern "C" int64_t vmmt_lstm_seq_xchg_bytes(int ndir, int B, int H) {
  if (ndir < 1 || B < 1 || H < 1) return 0;
  return (int64_t)ndir * ((B + 15) / 16) * 2 * 16 * H * 4;      // per (direction, row group of 16): two slots of 16 x H/2 granules
}

extern "C" int vmmt_lstm_seq_fwd(int dtype, int ndir, int nsteps, const vmmt_lstm_dir_fwd* dirs, const vmmt_lstm_dir_fwd* dirs_dev,
                                 const int64_t* lens, int B, int H, uint32_t* sync, void* xchg, void* stream) {
  using namespace vmmt;
  if (nsteps < 0 || !dirs || ndir < 1 || ndir > 2 || B <= 0 || H <= 0) return VMMT_EINVAL;
  if (nsteps == 0) return VMMT_OK;
  const int ngroups = (B + 15) / 16;
  bool ok = dtype == VMMT_BF16 && dirs_dev && sync && xchg && (((uintptr_t)xchg) & 15) == 0 && (H == 64 || H == 128 || H == 256 || H == 512 || H == 1024) &&
            (long)ngroups * (H / 32) * ndir <= 256 && nsteps >= 2 && nsteps < 4095;
  for (int i = 0; ok && i < nsteps; ++i)
    for (int k = 0; ok && k < ndir; ++k) {
      const vmmt_lstm_dir_fwd& d = dirs[(long)i * ndir + k];
      ok = d.h_prev && d.w_hh && d.gx && d.gates && d.c_out && d.h_out && al16s(d.h_prev, d.ld_hprev) && al16s(d.h_out, d.ld_h) &&
           al16s(d.w_hh, d.ld_w) && d.w_hh == dirs[k].w_hh && d.ld_w == dirs[k].ld_w;
      if (ok && i > 0) {
        const vmmt_lstm_dir_fwd& p = dirs[(long)(i - 1) * ndir + k];
        ok = d.h_prev == p.h_out && d.ld_hprev == p.ld_h && d.c_prev == p.c_out && d.ld_cprev == p.ld_c;
      }
    }
  if (!ok) return vmmt_lstm_chain_fwd(dtype, ndir, nsteps, dirs, lens, B, H, stream);
  SeqArgsF a;
  a.steps = reinterpret_cast<const SeqDirF*>(dirs_dev); a.lens = (const long long*)lens; a.sync = sync;
  a.xchg = reinterpret_cast<unsigned long long*>(xchg);
  a.B = B; a.nsteps = nsteps; a.ndir = ndir; a.ngroups = ngroups;
  switch (H) {
    case 1024: return launch_seq_fwd<1024>(a, (hipStream_t)stream);
    case 512: return launch_seq_fwd<512>(a, (hipStream_t)stream);
    case 256: return launch_seq_fwd<256>(a, (hipStream_t)stream);
    case 128: return launch_seq_fwd<128>(a, (hipStream_t)stream);
    default: return launch_seq_fwd<64>(a, (hipStream_t)stream);
  }
}

// Whole BACKWARD recurrence (mode 0 steps of vmmt_lstm_step_bwd) in one launch; same contract as vmmt_lstm_seq_fwd.  `xchg`:
// vmmt_lstm_seq_xchg_bytes_bwd(ndir, B, H) bytes.  Falls back to vmmt_lstm_chain_bwd when the persistent kernel does not apply
// (fp32, H not in {64,128,256,512,1024}, more workgroups than CUs, unaligned rows, dgates_next[t] != dgates_out[t-1], a dc_carry
// buffer that changes between steps).  A recurrence may be cut into several calls (the caller then runs the weight gradients of
// the finished part next to the rest): step 0 of a later piece carries dgates_next = the last dgates_out of the piece before,
// which is read from that plain buffer, and dc_carry is read at the first and written at the last step of every call.
extern "C" int64_t vmmt_lstm_seq_xchg_bytes_bwd(int ndir, int B, int H) {
  if (ndir < 1 || B < 1 || H < 1) return 0;
  return (int64_t)ndir * ((B + 15) / 16) * 4 * (4 * H / 32) * 1024;    // == SeqCfgB::GROUP_BYTES per (direction, row group of 16)
}

extern "C" int vmmt_lstm_seq_bwd(int dtype, int ndir, int nsteps, const vmmt_lstm_dir_bwd* dirs, const vmmt_lstm_dir_bwd* dirs_dev,
                                 const int64_t* lens, int B, int H, int with_dh0, uint32_t* sync, void* xchg, void* stream) {
  using namespace vmmt;
  if (nsteps < 0 || !dirs || ndir < 1 || ndir > 2 || B <= 0 || H <= 0) return VMMT_EINVAL;
  if (nsteps == 0) return with_dh0 ? VMMT_EINVAL : VMMT_OK;
  const int ngroups = (B + 15) / 16;
  bool ok = dtype == VMMT_BF16 && dirs_dev && sync && xchg && (((uintptr_t)xchg) & 15) == 0 && (H == 64 || H == 128 || H == 256 || H == 512 || H == 1024) &&
            (long)ngroups * (H / 32) * ndir <= 256 && nsteps >= 2 && nsteps < 4094;
  for (int i = 0; ok && i < nsteps; ++i)
    for (int k = 0; ok && k < ndir; ++k) {
      const vmmt_lstm_dir_bwd& d = dirs[(long)i * ndir + k];
      ok = d.w_hh_t && d.gates && d.c_t && d.dc_carry && d.dgates_out && al16s(d.dgates_out, d.ld_dgo) && al16s(d.w_hh_t, d.ld_wt) &&
           d.w_hh_t == dirs[k].w_hh_t && d.ld_wt == dirs[k].ld_wt && d.dc_carry == dirs[k].dc_carry && d.ld_dcc == dirs[k].ld_dcc &&
           (!d.dh_n || d.dc_n);
      if (ok && i == 0) ok = d.dgates_next == nullptr || al16s(d.dgates_next, d.ld_dgn);   // non-null: a chain continued from an earlier launch
      if (ok && i > 0) {
        const vmmt_lstm_dir_bwd& p = dirs[(long)(i - 1) * ndir + k];
        ok = d.dgates_next == p.dgates_out && d.ld_dgn == p.ld_dgo;
      }
    }
  if (ok && with_dh0)
    for (int k = 0; ok && k < ndir; ++k) {
      const vmmt_lstm_dir_bwd& d = dirs[(long)nsteps * ndir + k], &p = dirs[(long)(nsteps - 1) * ndir + k];
      ok = d.dh0_out && d.dgates_next == p.dgates_out && d.ld_dgn == p.ld_dgo && d.w_hh_t == dirs[k].w_hh_t && d.ld_wt == dirs[k].ld_wt;
    }
  if (!ok) {
    int rc = vmmt_lstm_chain_bwd(dtype, ndir, nsteps, dirs, lens, B, H, 0, stream);
    if (rc == VMMT_OK && with_dh0) rc = vmmt_lstm_step_bwd(dtype, ndir, dirs + (long)nsteps * ndir, lens, B, H, 1, stream);
    return rc;
  }
  SeqArgsB a;
  a.with_dh0 = with_dh0 ? 1 : 0;
  a.steps = reinterpret_cast<const SeqDirB*>(dirs_dev); a.lens = (const long long*)lens; a.sync = sync;
  a.xchg = reinterpret_cast<unsigned long long*>(xchg);
  a.B = B; a.nsteps = nsteps; a.ndir = ndir; a.ngroups = ngroups;
  switch (H) {
    case 1024: return launch_seq_bwd<1024>(a, (hipStream_t)stream);
    case 512: return launch_seq_bwd<512>(a, (hipStream_t)stream);
    case 256: return launch_seq_bwd<256>(a, (hipStream_t)stream);
    case 128: return launch_seq_bwd<128>(a, (hipStream_t)stream);
    default: return launch_seq_bwd<64>(a, (hipStream_t)stream);
  }
}

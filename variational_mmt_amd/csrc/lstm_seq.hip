// Persistent LSTM recurrence kernels for gfx950: ONE launch per sequence instead of one per time step.
//
// Replaces cuDNN's single-call nn.LSTM of the reference (encoder: onmt/Models.py:124-129,140-147 with packed sequences;
// decoder: onmt/VI_Model1.py:106,149-152).  The per-step kernels of lstm.hip re-read their W_hh slice (64 KiB per workgroup)
// from L2 at every step -- 1.6 of the ~5 us a forward step takes -- and pay a launch boundary (~1.5 us) per step.  Here a
// workgroup keeps its slice in LDS for the whole sequence and the workgroups of a ROW GROUP hand h_t to each other in-launch:
//
//   grid  (row groups of 32 sentences) x (H / 16 unit groups) x directions, 4 waves each; every workgroup must be resident at
//         once (<= 256 workgroups, 81 KiB of LDS: one per CU, and a 64-KiB GEMM workgroup of another stream still fits beside it)
//   step  every wave sweeps the exchange buffer for the part of h_{t-1} it multiplies (16 rows x half of K) until every granule
//         carries this step's tag  ->  the granules' data words ARE the MFMA A fragments  ->  v_mfma_f32_16x16x32_bf16 against
//         the LDS-resident W_hh slice  ->  fold the K quarters through LDS  ->  cell update (c stays in registers)  ->  h tile
//         through LDS: plain 16-byte stores into the layer's output buffer (for the kernels that follow) and tagged granules
//         into the exchange buffer (for the other workgroups of the row group).
//
// Hand-off protocol (cdna_hip_programming.md Guideline 16, form R2: the data IS the flag): a granule is ONE naturally aligned
// 8-byte {two bf16 values, 32-bit tag} written by one sc1 (write-through) store; tag = epoch of the launch * 4096 + step + 1, never
// 0.  Consumers re-read their granules with sc1 loads (L1 bypassed) until every tag matches: no flag, no fence, no drain, no
// barrier on the hand-off, one memory latency per step.  A first version with sc1 payload + vmcnt drain + one counter per row
// group + poll + barrier + sc1 loads measured 7 (decoder) to 14 us (encoder) per step -- two more serial memory round trips per
// hop and 256 pollers on one line.  The exchange buffer is double-buffered by step parity (a producer can only be one step
// ahead of the slowest consumer of its row group); the launch epoch lives in device memory and is advanced by the last
// workgroup to finish, so no memset precedes a launch.  Nothing depends on dispatch order or XCD placement.  Spins are
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
  unsigned* guard = *reinterpret_cast<unsigned* const*>(sync + VMMT_SEQ_GUARD_WORD);
  if (guard) __hip_atomic_store(guard, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int H> struct SeqCfg {
  static constexpr int ROWB = H * 2;                           // bytes per staged W_hh row (the whole reduction length)
  static constexpr int NCH = ROWB / 16;
  static constexpr int RPB = ROWB >= 256 ? 1 : 256 / ROWB;     // rows per 256-byte bank row
  static constexpr int KMASK = (NCH < 16 ? NCH : 16) - 1;
  static constexpr int LANES = NCH < 64 ? NCH : 64;
  static constexpr int PIECES = ROWB > 1024 ? ROWB / 1024 : 1; // 1-KiB pieces per row (LDS-DMA moves 64 x 16 B per instruction)
  static constexpr int NKS = H / 32;                           // MFMA K steps
  static constexpr int KQ = NKS >= 4 ? NKS / 4 : 1;            // K steps per QUARTER: the partial sums of lstm_step_fwd_fast's four K
                                                               // quarters are kept apart and added in its order (same bits)
  static constexpr int W_BYTES = 64 * ROWB;
  static constexpr int RED_BYTES = 2 * 2 * 2 * 4 * 2 * 64 * 4; // [row half][writer wave][quarter of the wave][gate][reg pair][lane] f32
  static constexpr int HT_BYTES = 32 * 16 * 2;                 // h tile [32 rows][16 units] bf16
  static constexpr int LDS = W_BYTES + RED_BYTES + HT_BYTES;
  static __device__ __forceinline__ int key(int row) { return (row / RPB) & KMASK; }
};

template <int H>
__global__ void __launch_bounds__(256) lstm_seq_fwd_kernel(SeqArgsF a) {
  using Cf = SeqCfg<H>;
  constexpr int ROWB = Cf::ROWB, KQ = Cf::KQ;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* red = reinterpret_cast<float*>(lds + Cf::W_BYTES);
  bf16_t* htile = reinterpret_cast<bf16_t*>(lds + Cf::W_BYTES + Cf::RED_BYTES);
  const int B = a.B, ndir = a.ndir;
  // ---- role of this workgroup: (direction k, row group rg, unit slice).  Workgroups b and b + 8 are observed to share an XCD
  //      (round-robin dispatch), so when the grid allows it all S = H/16 workgroups of a group are taken from ONE residue class
  //      b % 8.  That is a speed choice only: which transport a group uses is decided below from the XCC ids the hardware reports.
  const int S = H / 16, total = gridDim.x;
  const int bid = blockIdx.x;
  int grp, slice;
  if (total % 8 == 0 && (total / 8) % S == 0) { grp = (bid % 8) + 8 * ((bid / 8) / S); slice = (bid / 8) % S; }
  else { grp = bid / S; slice = bid % S; }
  const int k = grp / a.ngroups, rg = grp % a.ngroups;
  const int m0 = rg * 32, u0 = slice * 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wm = wave & 1, wk = wave >> 1;
  const int n = lane & 15, kg = lane >> 4;
  const int u = u0 + n;
  unsigned* err = a.sync + 2;
  // tags of this launch: epoch * 4096 + step + 1 (the epoch word is only written by the LAST workgroup of a launch to finish)
  const unsigned tag0 = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 4096u;
  // exchange buffer of my (direction, row group): [2 slots][2 row halves][H/32 k-steps][lo / hi][64 lanes] x 16 bytes, i.e. in
  // the ORDER THE CONSUMERS' A FRAGMENTS ARE LOADED: lane (row n of the half, k group kg) of a sweeping wave needs units
  // ks*32 + kg*8 .. +8 of its row = a "lo" and a "hi" piece of two granules each ({units 2j, 2j+1 | tag}); stored this way every
  // load instruction of the sweep reads 1 KiB contiguous (8 full lines) instead of 16 rows x 64 B (fragment-shaped loads ran the
  // sweep at 2.3 us per step: the address path, not the bytes, was the limit)
  char* xg = reinterpret_cast<char*>(a.xchg) + ((long)grp * 2) * (32 * H * 4);
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(xg);
  // ---- transport of my group.  Every member reports the XCC it runs on; once all S have arrived, a group whose members all
  //      sit on ONE XCD share one L2: its granules are stored with PLAIN stores (they stay in that L2) and the sc1 loads of the
  //      sweep (L1 bypassed) are served from it -- ~200 cycles and L2 bandwidth instead of a round trip through memory at HBM
  //      bandwidth (measured: 2.4 of the 5.1 us of a step were the sweep of sc1-stored granules: 16 MB per step chip-wide).
  //      A group spread over several XCDs keeps the placement-independent form: sc1 (write-through) stores.
  {
    int* flag = reinterpret_cast<int*>(lds + Cf::W_BYTES);
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

  // ---- W_hh slice -> LDS, once: rows r = gate * 16 + unit, unpadded, 16-byte chunks XOR-swizzled by the row key (conflict-free
  //      ds_read_b128 fragments); the swizzle is applied to the per-lane SOURCE address of the LDS-DMA
  {
    const SeqDirF d0 = load_desc(a.steps, k);
    const char* wp = reinterpret_cast<const char*>(d0.w_hh);
    if (lane < Cf::LANES) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int r = wave + 4 * j;
        const long row = (long)(r >> 4) * H + u0 + (r & 15);
#pragma unroll
        for (int pc = 0; pc < Cf::PIECES; ++pc)        // rows longer than 1 KiB (H = 1024): one LDS-DMA instruction per 1-KiB piece
          __builtin_amdgcn_global_load_lds((glb_cvoid_seq*)(wp + row * d0.ld_w * 2 + (((pc * 64 + lane) ^ Cf::key(r)) * 16)),
                                           (lds_void_seq*)(lds + r * ROWB + pc * 1024), 16, 0, 0);
      }
    }
  }
  // the two cells this lane finishes every step: rows 4*kg + 2*wk + e of row half wm, unit u
  int rows[2];
  long long len[2];
  float c_reg[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    rows[e] = m0 + wm * 16 + kg * 4 + 2 * wk + e;
    const int rr = rows[e] < B ? rows[e] : B - 1;
    len[e] = a.lens ? a.lens[rr] : 0;
    c_reg[e] = 0.f;
  }
  const int arow = min(m0 + wm * 16 + n, B - 1);                // the h_{t-1} row this lane's A fragments come from (step 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    const int f = *reinterpret_cast<volatile int*>(lds + Cf::W_BYTES);
    same_xcd = f == 1;
    alive = f >= 0;
  }
  __syncthreads();                                              // the flag word is part of the fold buffer
  // ---- this wave's B fragments of W_hh (its two K quarters x four gates) -> REGISTERS, once: they are the same at every step, and
  //      read from LDS per step (32 ds_read_b128 per wave at H = 512) the reads took as long as the MFMAs they feed -- 1.1 us of a
  //      4 us step, 0.85 us with the fragments resident (tools/probe/lstm_seq_probe.hip).  2 * KQ * 4 fragments of 4 registers:
  //      128 registers at H = 512.  (The backward kernel's sweep is bound by its 256 KiB of granules per workgroup and step, not
  //      by its fragment reads: resident fragments and MFMAs pipelined under the sweep measured the same 6.9 us per step.)
  bf16x8 wreg[2][KQ][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int ks = (2 * wk + j) * KQ + q;
      const int c = (ks < Cf::NKS ? ks : 0) * 4 + kg;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = g * 16 + n;
        wreg[j][q][g] = *reinterpret_cast<const bf16x8*>(lds + row * ROWB + ((c ^ Cf::key(row)) * 16));
      }
    }

  // epilogue operands that do not depend on the recurrence (x W_ih^T + b of the cell's four gates): those of step t+1 are
  // requested while step t computes, so that no step waits for them (they come from HBM: ~1 us when fetched at the step's top)
  float gxn[2][4];
  auto fetch_gx = [&](const SeqDirF& dd) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int rr = rows[e] < B ? rows[e] : B - 1;
      const float* gx = dd.gx + (long)rr * dd.ld_gx + u;
      gxn[e][0] = gx[0]; gxn[e][1] = gx[H]; gxn[e][2] = gx[2 * H]; gxn[e][3] = gx[3 * (long)H];
      if (dd.gx2) {
        const float* g2 = dd.gx2 + (long)rr * dd.ld_gx2 + u;
        gxn[e][0] += g2[0]; gxn[e][1] += g2[H]; gxn[e][2] += g2[2 * H]; gxn[e][3] += g2[3 * (long)H];
      }
    }
  };
  fetch_gx(load_desc(a.steps, k));

  for (int t = 0; t < a.nsteps; ++t) {
    const SeqDirF d = load_desc(a.steps, (long)t * ndir + k);
    SEQ_TS(0);
    float gxv[2][4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
      for (int g = 0; g < 4; ++g) gxv[e][g] = gxn[e][g];
      const int rr = rows[e] < B ? rows[e] : B - 1;
      if (t == 0) c_reg[e] = d.c_prev ? d.c_prev[(long)rr * d.ld_cprev + u] : 0.f;
    }
    // ---- A fragments of h_{t-1}: this wave = 16 rows x two of the four K quarters (2 * KQ k-steps of 32 units)
    u32x4 af[2][KQ];
    if (t == 0) {                                               // the initial state: an ordinary [B][ld] bf16 buffer of an earlier kernel
      const __amdgpu_buffer_rsrc_t hr = make_rsrc(d.h_prev);
      const unsigned abase = (unsigned)(((long)arow * d.ld_hprev + kg * 8) * 2);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          const int ks = (2 * wk + j) * KQ + q;
          af[j][q] = ks < Cf::NKS ? load16_sc1(hr, abase + (unsigned)(ks * 64)) : u32x4{0u, 0u, 0u, 0u};
        }
    } else {
      // granule sweep: lane (row n of this half, k group kg) needs units ks*32 + kg*8 .. +8 = four granules = 32 contiguous bytes
      const unsigned want = tag0 + (unsigned)t;                 // tag of step t-1
      const unsigned gbase = (unsigned)(((((t - 1) & 1) * 2 + wm) * (H / 32) * 2) * 1024 + lane * 16);
      const unsigned long long t_start = wall_clock64();
      // cheap poll first: ONE granule per producer workgroup of this wave's K half (row wm*16: every row of this half is stored by
      // one instruction of one wave of the producer), 8 bytes per lane on H/32 lanes -- the full sweep (16 KiB per wave and pass)
      // repeated by 1024 waves while they wait would by itself saturate the memory system the hand-off travels through
      {
        // producer slice sl = wk * S/2 + lane (16 units: k-step sl >> 1, k groups 2 * (sl & 1) + {0, 1}); its last piece (hi of
        // the second k group), row 0 of this half
        const int sl = wk * (S / 2) + lane;
        const unsigned pbase = (unsigned)(((((t - 1) & 1) * 2 + wm) * (H / 32) * 2 + (sl >> 1) * 2 + 1) * 1024 + ((2 * (sl & 1) + 1) * 16) * 16);
        while (alive) {
          unsigned long long g = 0;
          if (lane < S / 2) {
            const unsigned long long* gp = reinterpret_cast<const unsigned long long*>(xg + pbase);
            g = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (__all(lane >= S / 2 || (unsigned)(g >> 32) == want)) break;
          if (timed_out(t_start)) {
            if (lane == 0) seq_fail(a.sync, 0x200u + (unsigned)t);
            alive = false;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      SEQ_TS(1);
      for (;;) {
        u32x4 lo[2][KQ], hi[2][KQ];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < KQ; ++q) {
            const int ks = (2 * wk + j) * KQ + q;
            if (ks < Cf::NKS) {
              lo[j][q] = load16_sc1(xr, gbase + (unsigned)(ks * 2048));
              hi[j][q] = load16_sc1(xr, gbase + (unsigned)(ks * 2048 + 1024));
            } else {
              lo[j][q] = hi[j][q] = u32x4{0u, want, 0u, want};
            }
          }
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < KQ; ++q) {
            ok = ok && lo[j][q][1] == want && lo[j][q][3] == want && hi[j][q][1] == want && hi[j][q][3] == want;
            af[j][q] = u32x4{lo[j][q][0], lo[j][q][2], hi[j][q][0], hi[j][q][2]};
          }
        if (__all(ok) || !alive) break;
        if (timed_out(t_start)) {
          if (lane == 0) seq_fail(a.sync, 0x100u + (unsigned)t);
          alive = false;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    SEQ_TS(2);
    if (t + 1 < a.nsteps) fetch_gx(load_desc(a.steps, (long)(t + 1) * ndir + k));
    f32x4_s acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[j][g] = f32x4_s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        const int ks = (2 * wk + j) * KQ + q;
        if (ks < Cf::NKS) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            acc[j][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[j][q]), wreg[j][q][g], acc[j][g], 0, 0, 0);
        }
      }
    }
    // ---- hand the partner wave (same rows, other two quarters) the accumulator rows it finishes
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 2; ++e) red[((((wm * 2 + wk) * 2 + j) * 4 + g) * 2 + e) * 64 + lane] = wk ? acc[j][g][e] : acc[j][g][2 + e];
    SEQ_TS(3);
    __syncthreads();
    SEQ_TS(4);
    LstmCell cellv[2];
    bool valid[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float p[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // the four quarter sums of this cell's gate g: two of this wave (wk = 0: quarters 0, 1; wk = 1: quarters 2, 3), two of the partner
        const float own0 = wk ? acc[0][g][2 + e] : acc[0][g][e], own1 = wk ? acc[1][g][2 + e] : acc[1][g][e];
        const float oth0 = red[((((wm * 2 + (1 - wk)) * 2 + 0) * 4 + g) * 2 + e) * 64 + lane];
        const float oth1 = red[((((wm * 2 + (1 - wk)) * 2 + 1) * 4 + g) * 2 + e) * 64 + lane];
        const float q0 = wk ? oth0 : own0, q1 = wk ? oth1 : own1, q2 = wk ? own0 : oth0, q3 = wk ? own1 : oth1;
        p[g] = (((0.f + q0) + q1) + q2) + q3;                   // lstm_step_fwd_fast's fold order
      }
      cellv[e] = lstm_cell_math(p[0] + gxv[e][0], p[1] + gxv[e][1], p[2] + gxv[e][2], p[3] + gxv[e][3], c_reg[e]);
      const float cp = c_reg[e];
      valid[e] = !a.lens || d.t < len[e];
      c_reg[e] = valid[e] ? cellv[e].c : cp;                    // frozen state at pads (packed-sequence semantics)
      htile[(wm * 16 + kg * 4 + 2 * wk + e) * 16 + n] = f2bf(valid[e] ? cellv[e].h : 0.f);
    }
    SEQ_TS(5);
    __syncthreads();
    SEQ_TS(6);
    // ---- publish h_t.  Waves 0-1: tagged granules for the row group (one 16-byte store = two granules = four units of a row);
    //      wave 2: the layer's output buffer (plain 16-byte stores, read by later kernels only)
    if (wave < 2) {
      // lane -> (piece, row) with the row fastest: 16 consecutive lanes store 256 contiguous bytes
      const int l = wave * 64 + lane, piece = l >> 5, r = l & 31;
      const unsigned* hw = reinterpret_cast<const unsigned*>(htile + r * 16 + piece * 4);
      const unsigned tg = tag0 + (unsigned)t + 1u;
      const unsigned off = (unsigned)(((((t & 1) * 2 + (r >> 4)) * (H / 32) * 2 + (u0 >> 5) * 2 + (piece & 1)) * 1024) +
                                      (((u0 & 16) ? 2 : 0) + (piece >> 1)) * 256 + (r & 15) * 16);
      const u32x4 v = u32x4{hw[0], tg, hw[1], tg};
      if (same_xcd) *reinterpret_cast<u32x4*>(xg + off) = v;   // stays in the group's L2
      else store16_sc1(xr, off, v);                             // write-through: visible to every XCD
    } else if (wave == 2) {
      const int r = lane >> 1, half = lane & 1;
      if (m0 + r < B)
        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(d.h_out) + ((long)(m0 + r)) * d.ld_h + u0 + half * 8) =
            *reinterpret_cast<const u32x4*>(htile + r * 16 + half * 8);
    }
    // ---- what only later kernels read (saved gate activations, cell state, captured final state): behind the publish, so that
    //      these stores overlap the row group's hand-off instead of delaying it
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
  // ---- the last workgroup to finish advances the launch epoch (kernel boundary = visibility for the next launch)
  __syncthreads();
  if (threadIdx.x == 0) {
    if (__hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)total - 1) {
      __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int g2 = 0; g2 < a.ndir * a.ngroups; ++g2) {          // every workgroup is past its handshake: clear it for the next launch
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
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)lstm_seq_fwd_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, sm); attr_set = true; }
  hipLaunchKernelGGL(lstm_seq_fwd_kernel<H>, dim3(a.ngroups * (H / 16) * a.ndir), dim3(256), sm, st, a);
  return check_launch();
}

// ==============================================================================================================
// Forward recurrence, the shape used for 128 <= H <= 512: a workgroup owns 16 sentences x 32 hidden units (4 gates x 32 = 128 columns
// of the gate product).  Same protocol, same bits; what a workgroup sweeps per step is HALF of the 32-row shape's (16 x H tagged
// granules = 32 KiB at H = 512), and the sweep -- out of an L2 that every CU of the chip reads at once, beside the GEMMs of the other
// streams -- is the largest piece of a step.  Wave w multiplies K QUARTER w (H/4 units of h_{t-1}: the k-step blocks of H/128
// producers) against all 128 columns: its 8 * H/128 B fragments of W_hh come straight from memory into registers, once (128 at
// H = 512; no LDS copy of the slice).  A producer's h tile (16 x 32) is ONE k-step block of the consumers' order: wave 0 stores its
// "lo" KiB, wave 1 its "hi" KiB.  Fold: every wave leaves its quarter's 16 x 128 sums in LDS (32 KiB), a cell adds the four in
// lstm_step_fwd_fast's order ((q0 + q1) + q2) + q3.
// ==============================================================================================================
template <int H> struct SeqCfg16 {
  static constexpr int NKS = H / 32, KQ = NKS / 4, S = H / 32;
  static constexpr int RED_BYTES = 4 * 8 * 4 * 64 * 4;         // [quarter = wave][gate][unit half][reg][lane] f32
  static constexpr int HROW = 40;                              // h tile [16 rows][32 units] bf16, rows padded to 80 bytes
  static constexpr int HT_BYTES = 16 * HROW * 2;
  static constexpr int LDS = RED_BYTES + HT_BYTES;
  static constexpr long SLOT = (long)NKS * 2048;               // one step's h in fragment order: [k-step][lo / hi][64 lanes] x 16 bytes
  static_assert(H % 128 == 0, "a K quarter is a whole number of k-steps");
};

template <int H>
__global__ void __launch_bounds__(256) lstm_seq_fwd16_kernel(SeqArgsF a) {
  using Cf = SeqCfg16<H>;
  constexpr int KQ = Cf::KQ, S = Cf::S, HROW = Cf::HROW;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* red = reinterpret_cast<float*>(lds);
  bf16_t* htile = reinterpret_cast<bf16_t*>(lds + Cf::RED_BYTES);
  const int B = a.B, ndir = a.ndir;
  const int total = gridDim.x, bid = blockIdx.x;
  int grp, slice;                                               // role mapping and transport choice: see lstm_seq_fwd_kernel
  if (total % 8 == 0 && (total / 8) % S == 0) { grp = (bid % 8) + 8 * ((bid / 8) / S); slice = (bid / 8) % S; }
  else { grp = bid / S; slice = bid % S; }
  const int k = grp / a.ngroups, rg = grp % a.ngroups;
  const int m0 = rg * 16, u0 = slice * 32;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int uh = wave & 1, rp = wave >> 1;                      // the two cells of a lane: rows 4*kg + 2*rp + e, unit u0 + 16*uh + n
  const int n = lane & 15, kg = lane >> 4;
  const int u = u0 + uh * 16 + n;
  const unsigned tag0 = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 4096u;
  char* xg = reinterpret_cast<char*>(a.xchg) + (long)grp * (2 * Cf::SLOT);
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
        if (timed_out(t0)) { seq_fail(a.sync, 0x300u); f = -1; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      *flag = f;
#ifdef VMMT_SEQ_PROBE
      vmmt_seq_xcc[bid & 511] = (xcc & 15u) | ((unsigned)(f & 3) << 8);
#endif
    }
  }
  bool alive = true, same_xcd = false;
  // ---- this wave's B fragments of W_hh -> registers, once: rows g*H + u0 + 16*h2 + n (gate g, unit), columns (wave*KQ + q)*32 + kg*8 .. +8
  bf16x8 wreg[KQ][4][2];
  {
    const SeqDirF d0 = load_desc(a.steps, k);
    const bf16_t* wp = reinterpret_cast<const bf16_t*>(d0.w_hh);
#pragma unroll
    for (int q = 0; q < KQ; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
          wreg[q][g][h2] = *reinterpret_cast<const bf16x8*>(wp + ((long)g * H + u0 + h2 * 16 + n) * d0.ld_w + (wave * KQ + q) * 32 + kg * 8);
  }
  int rows[2];
  long long len[2];
  float c_reg[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    rows[e] = m0 + kg * 4 + 2 * rp + e;
    const int rr = rows[e] < B ? rows[e] : B - 1;
    len[e] = a.lens ? a.lens[rr] : 0;
    c_reg[e] = 0.f;
  }
  const int arow = min(m0 + n, B - 1);                          // the h_{t-1} row this lane's A fragments come from (step 0)
  __syncthreads();
  {
    const int f = *reinterpret_cast<volatile int*>(lds);
    same_xcd = f == 1;
    alive = f >= 0;
  }
  __syncthreads();                                              // the flag word is part of the fold buffer

  // x W_ih^T + b of the cells' gates (and the decoder's z W_z^T + b beside it), NEXT step's: they do not depend on the recurrence.  Requested
  // behind the sweep; the two are only added where they are used (the add would wait for the loads), and the step descriptor they come
  // from is read at the TOP of the step before -- a scalar load that misses its cache took 0.3-0.5 us between the sweep and the MFMAs
  float gxn[2][4], gx2n[2][4];
  auto fetch_gx = [&](const SeqNextF& dd) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int rr = rows[e] < B ? rows[e] : B - 1;
      const float* gx = dd.gx + (long)rr * dd.ld_gx + u;
      gxn[e][0] = gx[0]; gxn[e][1] = gx[H]; gxn[e][2] = gx[2 * H]; gxn[e][3] = gx[3 * (long)H];
      if (dd.gx2) {
        const float* g2 = dd.gx2 + (long)rr * dd.ld_gx2 + u;
        gx2n[e][0] = g2[0]; gx2n[e][1] = g2[H]; gx2n[e][2] = g2[2 * H]; gx2n[e][3] = g2[3 * (long)H];
      }
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
    // ---- A fragments of h_{t-1}: 16 rows x this wave's K quarter (KQ k-steps of 32 units)
    u32x4 af[KQ];
    if (t == 0) {                                               // the initial state: an ordinary [B][ld] bf16 buffer of an earlier kernel
      const __amdgpu_buffer_rsrc_t hr = make_rsrc(d.h_prev);
      const unsigned abase = (unsigned)(((long)arow * d.ld_hprev + kg * 8) * 2);
#pragma unroll
      for (int q = 0; q < KQ; ++q) af[q] = load16_sc1(hr, abase + (unsigned)((wave * KQ + q) * 64));
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
          if (lane < KQ) g = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(xg + pbase), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__all(lane >= KQ || (unsigned)(g >> 32) == want)) break;
          if (timed_out(t_start)) {
            if (lane == 0) seq_fail(a.sync, 0x200u + (unsigned)t);
            alive = false;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      SEQ_TS(1);
      for (;;) {
        u32x4 lo[KQ], hi[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          lo[q] = load16_sc1(xr, gbase + (unsigned)(q * 2048));
          hi[q] = load16_sc1(xr, gbase + (unsigned)(q * 2048 + 1024));
        }
        bool ok = true;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          ok = ok && lo[q][1] == want && lo[q][3] == want && hi[q][1] == want && hi[q][3] == want;
          af[q] = u32x4{lo[q][0], lo[q][2], hi[q][0], hi[q][2]};
        }
        if (__all(ok) || !alive) break;
        if (timed_out(t_start)) {
          if (lane == 0) seq_fail(a.sync, 0x100u + (unsigned)t);
          alive = false;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    SEQ_TS(2);
    if (t + 1 < a.nsteps) fetch_gx(dnext);
    f32x4_s acc[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) { acc[g][0] = f32x4_s{0.f, 0.f, 0.f, 0.f}; acc[g][1] = f32x4_s{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int q = 0; q < KQ; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
          acc[g][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[q]), wreg[q][g][h2], acc[g][h2], 0, 0, 0);
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
static int launch_seq_fwd16(const SeqArgsF& a, hipStream_t st) {
  constexpr int sm = SeqCfg16<H>::LDS;
  hipLaunchKernelGGL(lstm_seq_fwd16_kernel<H>, dim3(a.ngroups * (H / 32) * a.ndir), dim3(256), sm, st, a);
  return check_launch();
}

// ==============================================================================================================
// Backward recurrence, same structure: dh_t[32 x 16] = dgates_{t+1}[32 x 4H] W_hh^T slice[16 x 4H]^T, then the cell backward.
// What the workgroups of a row group exchange is dgates (4H values per sentence and step, 4x the forward's payload): every
// workgroup reads the whole 32 x 4H tile of its group at every step, and that traffic is what bounds the step.  So the tile travels
// DENSE (in the consumers' fragment order) and validity travels separately: a producer wave stores its pieces, waits until they
// are acknowledged, and then stores a flag (the step's tag); consumers poll the 4 x S flags of the group and sweep without
// checking anything.  Against the forward kernel's self-validating granules this costs one store acknowledgement (~0.7 us) per
// hand-off and halves the bytes: 6.7 -> 5.1 us per step at H = 512 on an idle chip (tools/probe/lstm_seq_probe.hip), and 365 -> 272
// registers.  The W_hh^T slice (16 rows of 4H) stays in LDS; dL/dc stays in registers.  K quarters, accumulator pairs and fold
// order are those of lstm_step_bwd_fast (same bits).
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
struct SeqArgsB {
  const SeqDirB* steps;
  const long long* lens;
  unsigned* sync;
  unsigned long long* xchg;      // granules [ndir][ngroups][2 slots][2 row halves][4H / 32 k-steps][lo / hi][64 lanes] x 16 bytes
  int B, nsteps, ndir, ngroups;
  int with_dh0;                  // steps[nsteps] is a mode-1 descriptor: dh0_out = dgates_{first step in time} W_hh, no cell backward
};

template <int H> struct SeqCfgB {
  static constexpr int K = 4 * H;
  static constexpr int ROWB = K * 2;
  static constexpr int PPR = ROWB >= 1024 ? ROWB / 1024 : 1;   // 1-KiB pieces per W_hh^T row
  static constexpr int LANES = ROWB >= 1024 ? 64 : ROWB / 16;
  static constexpr int NKS = K / 32, KQ = NKS / 4;             // K steps per quarter (H >= 64: KQ >= 2)
  static constexpr int CH = KQ < 8 ? KQ : 8;                   // K steps per sweep chunk
  static constexpr int W_BYTES = 16 * ROWB;
  static constexpr int RED_BYTES = 4 * 2 * 2 * 64 * 4;         // [wave][quarter of the wave][reg pair][lane] f32
  static constexpr int DT_BYTES = 32 * 64 * 2;                 // dgates tile [32 rows][4 gates][16 units] bf16
  static constexpr int LDS = W_BYTES + RED_BYTES + DT_BYTES;
};

template <int H>
__global__ void __launch_bounds__(256) lstm_seq_bwd_kernel(SeqArgsB a) {
  using Cf = SeqCfgB<H>;
  constexpr int ROWB = Cf::ROWB, KQ = Cf::KQ, NKS = Cf::NKS, CH = Cf::CH;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* red = reinterpret_cast<float*>(lds + Cf::W_BYTES);
  bf16_t* dtile = reinterpret_cast<bf16_t*>(lds + Cf::W_BYTES + Cf::RED_BYTES);
  const int B = a.B, ndir = a.ndir;
  const int S = H / 16, total = gridDim.x;
  const int bid = blockIdx.x;
  int grp, slice;                                               // role mapping and transport choice: see lstm_seq_fwd_kernel
  if (total % 8 == 0 && (total / 8) % S == 0) { grp = (bid % 8) + 8 * ((bid / 8) / S); slice = (bid / 8) % S; }
  else { grp = bid / S; slice = bid % S; }
  const int k = grp / a.ngroups, rg = grp % a.ngroups;
  const int m0 = rg * 32, u0 = slice * 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wm = wave & 1, wk = wave >> 1;
  const int n = lane & 15, kg = lane >> 4;
  const int u = u0 + n;
  unsigned* err = a.sync + 2;
  const unsigned tag0 = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 4096u;
  // exchange area of my (direction, row group): two slots of DENSE dgates in the consumers' fragment order,
  // [slot][row half][NKS k-steps][64 lanes x 16 B] (lane (row n of the half, k group kg) finds its 8 columns of k-step ks in ONE
  // 16-byte load), then the ready flags [slot][4 producer waves x S producers] (a tag each).  The forward kernel's tagged granules
  // carry their own validity but double the bytes, and this kernel is bound by them: every workgroup reads the whole 32 x 4H tile
  // of its group at every step (tagged: 256 KiB per workgroup and step at H = 512, sweep + MFMAs 4.2 us of a 6.1 us step).
  constexpr long DSLOT = 2L * NKS * 1024;
  constexpr long GROUP_BYTES = 4L * NKS * 2048;                 // == vmmt_lstm_seq_xchg_bytes_bwd per (direction, row group)
  constexpr int NFLAG = 4 * (H / 16);                           // per slot
  static_assert(2 * DSLOT + 2 * NFLAG * 4 <= GROUP_BYTES, "flags must fit behind the data");
  char* xg = reinterpret_cast<char*>(a.xchg) + (long)grp * GROUP_BYTES;
  constexpr unsigned FLAG_OFF = (unsigned)(2 * DSLOT);
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(xg);
  {
    int* flag = reinterpret_cast<int*>(lds + Cf::W_BYTES);
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
#ifdef VMMT_EXP_XCCDBG
      if (f == 0) __hip_atomic_fetch_add(a.sync + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
  }
  bool alive = true, same_xcd = false;
  // ---- W_hh^T slice -> LDS, once: 16 rows (units u0 ..) of 4H, 1-KiB pieces, chunk index XOR row (as lstm_step_bwd_fast)
  {
    const SeqDirB d0 = load_desc_b(a.steps, k);
    const char* wp = reinterpret_cast<const char*>(d0.w_hh_t);
    if (lane < Cf::LANES) {
#pragma unroll
      for (int j = 0; j < (16 * Cf::PPR + 3) / 4; ++j) {
        const int p = wave + 4 * j;
        if (p < 16 * Cf::PPR) {
          const int r = p / Cf::PPR, sg = p % Cf::PPR;
          __builtin_amdgcn_global_load_lds((glb_cvoid_seq*)(wp + ((long)(u0 + r) * d0.ld_wt) * 2 + sg * 1024 + ((lane ^ (r & 15)) * 16)),
                                           (lds_void_seq*)(lds + r * ROWB + sg * 1024), 16, 0, 0);
        }
      }
    }
  }
  int rows[2];
  long long len[2];
  float dcc[2];                                                 // dL/dc flowing to the previous step: in registers for the whole sequence
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    rows[e] = m0 + wm * 16 + kg * 4 + 2 * wk + e;
    const int rr = rows[e] < B ? rows[e] : B - 1;
    len[e] = a.lens ? a.lens[rr] : 0;
    dcc[e] = 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    const int f = *reinterpret_cast<volatile int*>(lds + Cf::W_BYTES);
    same_xcd = f == 1;
    alive = f >= 0;
  }
  __syncthreads();

  // everything the cell backward needs besides dh (saved gates, cell states, dh from above, injected final-state gradients) does
  // not depend on the recurrence: step t+1's values are requested while step t computes
  float nx[2][9];
  auto fetch_in = [&](const SeqDirB& dd) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const long bb = rows[e] < B ? rows[e] : B - 1;
      const bf16_t* gs = reinterpret_cast<const bf16_t*>(dd.gates) + bb * dd.ld_gates + u;
      nx[e][0] = bf2f(gs[0]); nx[e][1] = bf2f(gs[H]); nx[e][2] = bf2f(gs[2 * H]); nx[e][3] = bf2f(gs[3 * (long)H]);
      nx[e][4] = dd.c_t[bb * dd.ld_ct + u];
      nx[e][5] = dd.c_prev ? dd.c_prev[bb * dd.ld_cp + u] : 0.f;
      nx[e][6] = dd.dh_above ? bf2f(reinterpret_cast<const bf16_t*>(dd.dh_above)[bb * dd.ld_dha + u]) : 0.f;
      nx[e][7] = dd.dh_n ? dd.dh_n[bb * dd.ld_dhn + u] : 0.f;
      nx[e][8] = dd.dh_n ? dd.dc_n[bb * dd.ld_dcn + u] : 0.f;
    }
  };
  fetch_in(load_desc_b(a.steps, k));

  for (int t = 0; t < a.nsteps + a.with_dh0; ++t) {
    const SeqDirB d = load_desc_b(a.steps, (long)t * ndir + k);
    const bool dh0_step = t == a.nsteps;                        // the gradient of the initial hidden state: GEMM only
    SEQ_TS(0);
    float gi[2], gf[2], gg[2], go[2], cc[2], cpv[2], dha[2], dhn[2], dcn[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      gi[e] = nx[e][0]; gf[e] = nx[e][1]; gg[e] = nx[e][2]; go[e] = nx[e][3]; cc[e] = nx[e][4]; cpv[e] = nx[e][5];
      dha[e] = nx[e][6]; dhn[e] = nx[e][7]; dcn[e] = nx[e][8];
      const long bb = rows[e] < B ? rows[e] : B - 1;
      if (t == 0) dcc[e] = d.dc_carry[bb * d.ld_dcc + u];
    }
    f32x4_s acc[2][2];                                          // [quarter of this wave][even / odd K step]
#pragma unroll
    for (int j = 0; j < 2; ++j) { acc[j][0] = f32x4_s{0.f, 0.f, 0.f, 0.f}; acc[j][1] = f32x4_s{0.f, 0.f, 0.f, 0.f}; }
    if (d.dgates_next) {
      u32x4 af[2][KQ];
      if (t == 0) {
        // a chain CONTINUED from an earlier launch (the caller cut the recurrence into pieces): the previous step's dgates are in
        // its plain [B][4H] buffer, complete since that launch ended.  Same fragments as the sweep below: row n of row half wm,
        // columns ks * 32 + kg * 8 .. + 8
        const long rr = (m0 + wm * 16 + n) < B ? (m0 + wm * 16 + n) : B - 1;
        const bf16_t* src = reinterpret_cast<const bf16_t*>(d.dgates_next) + rr * d.ld_dgn + kg * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < KQ; ++q) af[j][q] = *reinterpret_cast<const u32x4*>(src + ((2 * wk + j) * KQ + q) * 32);
      } else {
        const unsigned want = tag0 + (unsigned)t;               // tag of the step processed just before
        const unsigned long long t_start = wall_clock64();
        {   // wait until every producer wave of the row group has flagged that step: NFLAG tags, 4 bytes each, contiguous
          const unsigned fbase = FLAG_OFF + (unsigned)(((t - 1) & 1) * NFLAG * 4);
          while (alive) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < (NFLAG + 63) / 64; ++i) {
              const int f = i * 64 + lane;
              unsigned g = want;
              if (f < NFLAG) g = __hip_atomic_load(reinterpret_cast<const unsigned*>(xg + fbase) + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ok = ok && g == want;
            }
            if (__all(ok)) break;
            if (timed_out(t_start)) {
              if (lane == 0) seq_fail(a.sync, 0x500u + (unsigned)t);
              alive = false;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        SEQ_TS(1);
        // the flags were stored behind the producers' data (acknowledged by the L2 / by memory): the tile is complete, no tags to check
        const unsigned gbase = (unsigned)((((t - 1) & 1) * 2 + wm) * (NKS * 1024) + (2 * wk * KQ) * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < KQ; ++q) af[j][q] = load16_sc1(xr, gbase + (unsigned)((j * KQ + q) * 1024));
      }
      if (t + 1 < a.nsteps) fetch_in(load_desc_b(a.steps, (long)(t + 1) * ndir + k));
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c0 = 0; c0 < KQ; c0 += CH) {
          bf16x8 bv[CH];
#pragma unroll
          for (int q = 0; q < CH; ++q) {
            const int c = (((2 * wk + j) * KQ + c0 + q) * 4 + kg);
            bv[q] = *reinterpret_cast<const bf16x8*>(lds + n * ROWB + ((c ^ n) * 16));
          }
#pragma unroll
          for (int q = 0; q < CH; ++q)
            acc[j][(c0 + q) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[j][c0 + q]), bv[q], acc[j][(c0 + q) & 1], 0, 0, 0);
        }
    } else if (t + 1 < a.nsteps) {
      fetch_in(load_desc_b(a.steps, (long)(t + 1) * ndir + k));
    }
    SEQ_TS(2);
    // ---- fold: quarter sums (acc0 + acc1, as lstm_step_bwd_fast) to the partner wave, then ((q0 + q1) + q2) + q3
    f32x4_s qs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) qs[j][r] = acc[j][0][r] + acc[j][1][r];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 2; ++e) red[((wave * 2 + j) * 2 + e) * 64 + lane] = wk ? qs[j][e] : qs[j][2 + e];
    SEQ_TS(3);
    __syncthreads();
    SEQ_TS(4);
    const int pw = wm + 2 * (1 - wk);                           // partner wave: same rows, the other two quarters
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float own0 = wk ? qs[0][2 + e] : qs[0][e], own1 = wk ? qs[1][2 + e] : qs[1][e];
      const float oth0 = red[((pw * 2 + 0) * 2 + e) * 64 + lane], oth1 = red[((pw * 2 + 1) * 2 + e) * 64 + lane];
      const float q0 = wk ? oth0 : own0, q1 = wk ? oth1 : own1, q2 = wk ? own0 : oth0, q3 = wk ? own1 : oth1;
      float dh = (((0.f + q0) + q1) + q2) + q3;
      if (dh0_step) {
        if (rows[e] < B) d.dh0_out[(long)rows[e] * d.ld_dh0 + u] = dh;
        continue;
      }
      const bool valid = !a.lens || d.t < len[e];
      bf16_t* dt = dtile + (wm * 16 + kg * 4 + 2 * wk + e) * 64 + n;
      if (!valid) {
        dt[0] = 0; dt[16] = 0; dt[32] = 0; dt[48] = 0;
        dcc[e] = 0.f;
      } else {
        float dc = dcc[e];
        dh += dha[e];
        const bool inj = d.inject == 3 || (d.inject == 1 && d.t == len[e] - 1) || (d.inject == 2 && d.t == 0);
        if (inj) { dh += dhn[e]; dc += dcn[e]; }
        const LstmCellGrad gr = lstm_cell_bwd_math(gi[e], gf[e], gg[e], go[e], cc[e], cpv[e], dh, dc);
        dt[0] = f2bf(gr.di); dt[16] = f2bf(gr.df); dt[32] = f2bf(gr.dg); dt[48] = f2bf(gr.d_o);
        dcc[e] = gr.dc_prev;
      }
    }
    if (dh0_step) break;                                        // (uniform) nothing to publish
    SEQ_TS(5);
    __syncthreads();
    SEQ_TS(6);
    // ---- publish dgates_t: 256 dense 16-byte pieces (gate, half of the 16 units, row), each wave flags its own stores once they
    //      are acknowledged; then the plain [B][4H] buffer for the kernels that follow
    {
      const unsigned tg = tag0 + (unsigned)t + 1u;
      {
        // piece id -> (gate, 8-unit half, row) with the row fastest: 16 consecutive lanes store 256 contiguous bytes
        const int id = threadIdx.x, g = id >> 6, half = (id >> 5) & 1, r = id & 31;
        const u32x4 v = *reinterpret_cast<const u32x4*>(dtile + r * 64 + g * 16 + half * 8);
        const int ks = (g * H + u0) >> 5, kgp = ((u0 & 16) ? 2 : 0) + half;
        const unsigned off = (unsigned)((((t & 1) * 2 + (r >> 4)) * (NKS * 1024)) + ks * 1024 + (kgp * 16 + (r & 15)) * 16);
        if (same_xcd) *reinterpret_cast<u32x4*>(xg + off) = v;   // stays in the group's L2
        else store16_sc1(xr, off, v);                             // write-through: visible to every XCD
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces have reached the L2 (or memory)
      if (lane == 0) {
        unsigned* fl = reinterpret_cast<unsigned*>(xg + FLAG_OFF) + (t & 1) * NFLAG + wave * S + slice;
        if (same_xcd) *reinterpret_cast<volatile unsigned*>(fl) = tg;
        else __hip_atomic_store(fl, tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      // plain copy for the kernels that follow (weight gradients, dx): 32 rows x 4 gates x 32 bytes = 256 x 16 bytes
      const int r = threadIdx.x >> 3, g = (threadIdx.x >> 1) & 3, half = threadIdx.x & 1;
      if (m0 + r < B)
        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(d.dgates_out) + ((long)(m0 + r)) * d.ld_dgo + (long)g * H + u0 + half * 8) =
            *reinterpret_cast<const u32x4*>(dtile + r * 64 + g * 16 + half * 8);
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
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)lstm_seq_bwd_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, sm); attr_set = true; }
  hipLaunchKernelGGL(lstm_seq_bwd_kernel<H>, dim3(a.ngroups * (H / 16) * a.ndir), dim3(256), sm, st, a);
  return check_launch();
}

// ==============================================================================================================
// Backward recurrence, the shape used for H <= 512: a workgroup owns 16 sentences x 32 hidden units.  The kernel above is bound by
// what every workgroup sweeps per step -- its row group's whole dgates tile, 32 x 4H -- out of an L2 that 256 workgroups sweep at
// once (32 MiB per step chip-wide at H = 512: 2.0 of the 4.3-5.1 us of a step on an idle chip, and the part that grows when GEMMs of
// other streams stream through the same L2).  Half the rows per workgroup and twice the units: the same number of workgroups, the
// same MFMAs per wave, HALF the sweep (16 x 4H = 64 KiB per workgroup and step).  What changes with it:
//   * wave w multiplies K QUARTER w -- which is gate w's H columns -- against both 16-unit halves of the slice: its 2 * H/32 B
//     fragments of W_hh^T live in REGISTERS for the whole sequence (128 at H = 512; the sweep holds 64 where the 32-row shape
//     held 128), no LDS copy of the slice at all: 13 KiB of LDS per workgroup instead of 81
//   * a producer wave g stores gate g's 16 x 32 block = ONE 1-KiB k-step block of the consumers' fragment order (one store
//     instruction, 8 full lines), and a consumer wave w waits only for the S = H/32 flags of the producers' waves w
//   * fold: every wave leaves its quarter's 16 x 32 sums in LDS, the cells (two per lane) add the four in lstm_step_bwd_fast's
//     order ((q0 + q1) + q2) + q3 with q = even + odd k-step accumulators: the same bits as the 32-row shape and the per-step kernels.
// ==============================================================================================================
template <int H> struct SeqCfgB16 {
  static constexpr int K = 4 * H, NKS = K / 32, KQ = NKS / 4, S = H / 32;
  static constexpr int RED_BYTES = 4 * 2 * 4 * 64 * 4;         // [quarter = wave][unit half][reg][lane] f32
  static constexpr int DROW = 4 * 32 + 8;                      // dgates tile [16 rows][4 gates][32 units] bf16, rows padded by 16 bytes
  static constexpr int DT_BYTES = 16 * DROW * 2;
  static constexpr int LDS = RED_BYTES + DT_BYTES;
  static constexpr long DSLOT = (long)NKS * 1024;              // one step's tile in fragment order: [k-step][64 lanes] x 16 bytes
  static constexpr long GROUP_BYTES = 4L * NKS * 1024;         // per (direction, row group of 16): two slots, then the flags
  static constexpr int NFLAG = 4 * S;                          // per slot: [producer wave = gate][producer slice]
  static_assert(2 * DSLOT + 2 * NFLAG * 4 <= GROUP_BYTES, "flags must fit behind the data");
};

template <int H>
__global__ void __launch_bounds__(256) lstm_seq_bwd16_kernel(SeqArgsB a) {
  using Cf = SeqCfgB16<H>;
  constexpr int KQ = Cf::KQ, S = Cf::S, DROW = Cf::DROW, NFLAG = Cf::NFLAG;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* red = reinterpret_cast<float*>(lds);
  bf16_t* dtile = reinterpret_cast<bf16_t*>(lds + Cf::RED_BYTES);
  const int B = a.B, ndir = a.ndir;
  const int total = gridDim.x, bid = blockIdx.x;
  int grp, slice;                                               // role mapping and transport choice: see lstm_seq_fwd_kernel
  if (total % 8 == 0 && (total / 8) % S == 0) { grp = (bid % 8) + 8 * ((bid / 8) / S); slice = (bid / 8) % S; }
  else { grp = bid / S; slice = bid % S; }
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
  long long len[2];
  float dcc[2];                                                 // dL/dc flowing to the previous step: in registers for the whole sequence
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    rows[e] = m0 + kg * 4 + 2 * rp + e;
    const int rr = rows[e] < B ? rows[e] : B - 1;
    len[e] = a.lens ? a.lens[rr] : 0;
    dcc[e] = 0.f;
  }
  __syncthreads();
  {
    const int f = *reinterpret_cast<volatile int*>(lds);
    same_xcd = f == 1;
    alive = f >= 0;
  }
  __syncthreads();                                              // the flag word is part of the fold buffer

  float nx[2][9];                                               // the cell backward's inputs of the NEXT step (they do not depend on the recurrence)
  auto fetch_in = [&](const SeqDirB& dd) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const long bb = rows[e] < B ? rows[e] : B - 1;
      const bf16_t* gs = reinterpret_cast<const bf16_t*>(dd.gates) + bb * dd.ld_gates + u;
      nx[e][0] = bf2f(gs[0]); nx[e][1] = bf2f(gs[H]); nx[e][2] = bf2f(gs[2 * H]); nx[e][3] = bf2f(gs[3 * (long)H]);
      nx[e][4] = dd.c_t[bb * dd.ld_ct + u];
      nx[e][5] = dd.c_prev ? dd.c_prev[bb * dd.ld_cp + u] : 0.f;
      nx[e][6] = dd.dh_above ? bf2f(reinterpret_cast<const bf16_t*>(dd.dh_above)[bb * dd.ld_dha + u]) : 0.f;
      nx[e][7] = dd.dh_n ? dd.dh_n[bb * dd.ld_dhn + u] : 0.f;
      nx[e][8] = dd.dh_n ? dd.dc_n[bb * dd.ld_dcn + u] : 0.f;
    }
  };
  SeqDirB dnext = load_desc_b(a.steps, k);                     // (the NEXT step's descriptor is read at the top of a step: see lstm_seq_fwd16_kernel)
  fetch_in(dnext);

  for (int t = 0; t < a.nsteps + a.with_dh0; ++t) {
    const SeqDirB d = dnext;
    if (t + 1 < a.nsteps + a.with_dh0) dnext = load_desc_b(a.steps, (long)(t + 1) * ndir + k);
    const bool dh0_step = t == a.nsteps;                        // the gradient of the initial hidden state: GEMM only
    SEQ_TS(0);
    float gi[2], gf[2], gg[2], go[2], cc[2], cpv[2], dha[2], dhn[2], dcn[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      gi[e] = nx[e][0]; gf[e] = nx[e][1]; gg[e] = nx[e][2]; go[e] = nx[e][3]; cc[e] = nx[e][4]; cpv[e] = nx[e][5];
      dha[e] = nx[e][6]; dhn[e] = nx[e][7]; dcn[e] = nx[e][8];
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
      if (t + 1 < a.nsteps) fetch_in(dnext);
      cell_pre();                                               // in the shadow of the sweep's latency
#pragma unroll
      for (int q = 0; q < KQ; ++q)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
          acc[h2][q & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[q]), wreg[q][h2], acc[h2][q & 1], 0, 0, 0);
    } else {
      if (t + 1 < a.nsteps) fetch_in(dnext);
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
      if (lane == 0) {
        unsigned* fl = reinterpret_cast<unsigned*>(xg + FLAG_OFF) + (t & 1) * NFLAG + wave * S + slice;
        if (same_xcd) *reinterpret_cast<volatile unsigned*>(fl) = tg;
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
static int launch_seq_bwd16(const SeqArgsB& a, hipStream_t st) {
  constexpr int sm = SeqCfgB16<H>::LDS;
  hipLaunchKernelGGL(lstm_seq_bwd16_kernel<H>, dim3(a.ngroups * (H / 32) * a.ndir), dim3(256), sm, st, a);
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
  return (int64_t)ndir * ((B + 31) / 32) * 2 * 32 * H * 4;
}

extern "C" int vmmt_lstm_seq_fwd(int dtype, int ndir, int nsteps, const vmmt_lstm_dir_fwd* dirs, const vmmt_lstm_dir_fwd* dirs_dev,
                                 const int64_t* lens, int B, int H, uint32_t* sync, void* xchg, void* stream) {
  using namespace vmmt;
  if (nsteps < 0 || !dirs || ndir < 1 || ndir > 2 || B <= 0 || H <= 0) return VMMT_EINVAL;
  if (nsteps == 0) return VMMT_OK;
  const int ngroups = (B + 31) / 32;
  bool ok = dtype == VMMT_BF16 && dirs_dev && sync && xchg && (((uintptr_t)xchg) & 15) == 0 && (H == 64 || H == 128 || H == 256 || H == 512 || H == 1024) &&
            (long)ngroups * (H / 16) * ndir <= 256 && nsteps >= 2 && nsteps < 4095;
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
#ifndef VMMT_EXP_FWD32      // (probe build: the 32-row shape at every H, for same-box comparisons)
  if (H >= 128 && H <= 512) {   // 16 sentences x 32 units per workgroup: never more workgroups, never more exchange bytes than the 32-row shape
    a.ngroups = (B + 15) / 16;
    switch (H) {
      case 512: return launch_seq_fwd16<512>(a, (hipStream_t)stream);
      case 256: return launch_seq_fwd16<256>(a, (hipStream_t)stream);
      default: return launch_seq_fwd16<128>(a, (hipStream_t)stream);
    }
  }
#endif
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
  return (int64_t)ndir * ((B + 31) / 32) * 2 * 2 * (4 * H / 32) * 2048;
}

extern "C" int vmmt_lstm_seq_bwd(int dtype, int ndir, int nsteps, const vmmt_lstm_dir_bwd* dirs, const vmmt_lstm_dir_bwd* dirs_dev,
                                 const int64_t* lens, int B, int H, int with_dh0, uint32_t* sync, void* xchg, void* stream) {
  using namespace vmmt;
  if (nsteps < 0 || !dirs || ndir < 1 || ndir > 2 || B <= 0 || H <= 0) return VMMT_EINVAL;
  if (nsteps == 0) return with_dh0 ? VMMT_EINVAL : VMMT_OK;
  const int ngroups = (B + 31) / 32;
  bool ok = dtype == VMMT_BF16 && dirs_dev && sync && xchg && (((uintptr_t)xchg) & 15) == 0 && (H == 64 || H == 128 || H == 256 || H == 512 || H == 1024) &&
            (long)ngroups * (H / 16) * ndir <= 256 && nsteps >= 2 && nsteps < 4094;
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
#ifndef VMMT_EXP_BWD32      // (probe build: the 32-row shape at every H, for same-box comparisons)
  if (H <= 512) {           // 16 sentences x 32 units per workgroup: never more workgroups, never more exchange bytes than the 32-row shape
    a.ngroups = (B + 15) / 16;
    switch (H) {
      case 512: return launch_seq_bwd16<512>(a, (hipStream_t)stream);
      case 256: return launch_seq_bwd16<256>(a, (hipStream_t)stream);
      case 128: return launch_seq_bwd16<128>(a, (hipStream_t)stream);
      default: return launch_seq_bwd16<64>(a, (hipStream_t)stream);
    }
  }
#endif
  switch (H) {
    case 1024: return launch_seq_bwd<1024>(a, (hipStream_t)stream);
    case 512: return launch_seq_bwd<512>(a, (hipStream_t)stream);
    case 256: return launch_seq_bwd<256>(a, (hipStream_t)stream);
    case 128: return launch_seq_bwd<128>(a, (hipStream_t)stream);
    default: return launch_seq_bwd<64>(a, (hipStream_t)stream);
  }
}

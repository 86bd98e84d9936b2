// Fused generator pass of the bf16 training step at H = 512 (256): the vocabulary projection + log-softmax + NLL AND dL/dO in one
// sweep of the generator weight, with the softmax weights stored on the way so that dL/dWg is one plain GEMM afterwards.
//
// Reference semantics: generator = Linear(H, V) + LogSoftmax (onmt/ModelConstructor.py:583-585), NLLLoss(weight[pad] = 0, sum)
// (onmt/Loss.py:163-165), loss.div(normalization).backward() (onmt/Loss.py:129); dL/dlogit[m][v] = (softmax_m[v] - [v == y_m]) * s_m with
// s_m = [y_m != pad] / normalization.
//
// The unfused path (generator.hip) makes four GEMM-sized passes, three of them on the critical path of the step: statistics
// (logits never stored), a recompute that writes G^T = dL/dlogit^T (bf16, 307 MB at M = 5120, V = 30000), dO = G Wg, and dWg = G^T O
// beside it.  Here the critical path is ONE pass:
//
//   vmmt_gen_fwd_dO:  one workgroup = 128 tokens x one slice of the vocabulary.  Per 32-row tile of Wg:
//        S^T = Wg_tile O^T  ->  P = exp(S - ref)  ->  acc^T += Wg_tile^T P^T,   l += rowsum P,   P -> memory (bf16)
//     i.e. the flash-attention forward with K = V = Wg: the un-normalised dO = sum_v P[m][v] Wg[v] accumulates next to the softmax
//     statistics.  `ref` is a LAZY reference: it starts as the maximum of the first tile and only moves (with a rescale of the 256
//     accumulator registers and a rewrite of the slice's stored P) when a later logit exceeds it by more than 60, which keeps exp()
//     inside f32 / bf16 range and costs nothing in the common case.  A small combine kernel folds the vocabulary slices:
//        lse = ref* + log l*,   dO[m] = s_m (acc*/l* - Wg[y_m]),   NLL / accuracy statistics,
//        c_s[m] = s_m exp(ref_s[m] - lse_m)  and  O'_s = diag(c_s) O   for every slice s.
//   dL/dWg[v in slice s] = sum_m P[m][v] O'_s[m] - (one-hot term): one vmmt_gemm (the B operand switches with the slice:
//     vmmt_gemm_args.b_batch_rows) on the side stream, where it shares the chip with the LSTM backward chains, followed by
//     vmmt_gen_dW_finish (bias gradient = weighted column sums of P; one-hot term by atomics).
//   (A second flash-shaped sweep for dWg -- no P in memory -- was built and measured: 976 TFLOP/s in isolation, but its workgroups own
//    whole CUs (132 KB of LDS, 512 registers per lane), so it serialises with whatever stream it is put on; the step came out the
//    same within box-to-box noise (2.06-2.19 ms either way).  Removed: the GEMM shares the chip and is 300 lines less.)
//
// Kernel structure: the "row" operand X = O (128 rows per workgroup, 32 per wave) stays resident for the whole kernel as MFMA
// B-operand fragments (half of the k-steps in registers, half in LDS); the "column" operand Y = Wg streams through a two-deep ring
// of 32-row LDS tiles filled by LDS-DMA, and each tile is used twice: by rows (ds_read_b128) for S^T and by columns
// (ds_read_b64_tr_b16) for the second product.  S^T is computed with the streamed rows on the MFMA A side, so that a lane's
// accumulator registers (one token, 16 vocabulary entries) are -- after conversion to bf16 -- exactly the B-operand fragments of the
// second product; the k-order of that product is the accumulator's row order (c = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)),
// which the transposed reads of Y reproduce.  One wave per SIMD: 256 accumulator registers (acc^T: H x 32 per wave) + 64 for X.
//
// Per byte fetched from L2 this does 4x the MFMA work of the 128 x 128-tile kernels (a 32 KB tile of Y feeds 2 x 4.2 MFLOP),
// which is what bounded them.
#include <cstdlib>
#include <type_traits>
#include "common.hpp"
#include "vmmt.h"

namespace vmmt {

// >>> traffic-key common   (tools/traffic_key.py: profiles/traffic.json names the source text its PMC figures were measured on)
typedef __attribute__((address_space(3))) void f_lds_void_t;
typedef __attribute__((address_space(1))) const void f_glb_cvoid_t;
typedef short fs16x4 __attribute__((ext_vector_type(4)));
typedef short fs16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifndef G2_PD
#define G2_PD 6
#endif
#ifndef G2W_PD
#define G2W_PD 4
#endif
#define G2_HB(i) ((i) >> 1)
#define G2_KK(i) ((i) & 1)
constexpr int G2_COMBINE_GROUPS = 1;          // groups of 8 tokens per workgroup of the combine kernel
constexpr float G2_L2E = 1.4426950408889634f;
constexpr float G2_THR = 60.f;             // lazy-reference slack (natural-log units): exp(60) ~ 1e26 stays far inside bf16 / f32 range

struct Gen2Args {
  const bf16_t* X; long ldx; int nrows;    // resident operand: O [M][ldo]
  const bf16_t* Y; long ldy; int ncols;    // streamed operand: Wg [V][ldw]
  const float* cvec;                       // per-column constants: bias [V]
  const long long* y;                      // targets [M]
  int tiles_per_split, nsplit;             // vocabulary slices
  float* p_acc; float* p_ref; float* p_l; float* p_max; long mpad;      // partials [nsplit][mpad]([D])
  float* tgt_logit;                        // logit of the target [M]
  bf16_t* p_out; long ldp;                 // optional: the un-normalised softmax weights P[m][v] = exp(logit - ref), bf16 [M][ldp]
  const int* rows;                         // optional row map: token m of this launch is row rows[m] of X and y (< 0: no token, a pad)
};

template <int D> struct G2 {
  static constexpr int BC = 32;                    // streamed rows per tile (one 32-row MFMA block)
  static constexpr int NSLOT = 2;                  // ring depth
  static constexpr int CPR = D / 8;                // 16-byte chunks per row
  static constexpr int ROWB = D * 2;               // bytes per row
  static constexpr int TILEB = BC * ROWB;          // 32 KiB at D = 512
  static constexpr int NPIECE = TILEB / 1024;      // 1-KiB LDS-DMA pieces per tile
  static constexpr int PER = NPIECE / 4;           // pieces per wave
  static constexpr int RPP = 1024 / ROWB;          // rows per piece (1 at D = 512, 2 at D = 256)
  static constexpr int KS = D / 16, HB = D / 32;
  // the resident operand: k-steps [0, KR) in registers, [KR, KS) in LDS (the register file holds 256 accumulators + 4 KR operand
  // registers + the working set; the compiler spills beyond ~64 operand registers)
  static constexpr int KR = D >= 512 ? 16 : KS, KL = KS - KR;
  static constexpr int XROWB = KL * 32;            // bytes per row of the LDS part of X (0 or a multiple of 256)
  static constexpr int XB = 128 * XROWB;
  static constexpr int SMALLB = 4 * 256;           // per ring slot and WAVE: 32 f32 column constants + 32 int32 targets
  static constexpr int XOFF = NSLOT * TILEB, SOFF = XOFF + XB;
  static constexpr int POFF = SOFF + NSLOT * SMALLB;   // F with P output: one [32 tokens][32 entries] bf16 patch per wave, 80-byte rows
  static constexpr int PPITCH = 80, PATCHB = 32 * PPITCH;
  static constexpr int LDSB = POFF + 4 * PATCHB;
  static_assert(KS / 4 >= PER, "one DMA piece per four MFMAs of the S^T phase");
  static_assert(D % 128 == 0 && D <= 512, "register budget: 256 accumulators per lane at D = 512");
};

// XOR swizzle of the 16-byte chunk index inside a row of an LDS image whose rows are a multiple of 256 bytes (every row starts on
// bank 0):
//   row reads (ds_read_b128, 16 lanes = 16 rows that differ in row & 15): the low four chunk bits get a permutation of row & 15;
//   transposed reads (ds_read_b64_tr_b16, 16 lanes = 4 consecutive rows x 64 contiguous bytes): chunk bits 2-3 get row & 3, so the
//   four rows land 16 banks apart.
__device__ __forceinline__ int g2_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <int OFF>
__device__ __forceinline__ fs16x4 g2_tr_read(unsigned addr) {
  fs16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
// wait until at most N LDS operations are in flight; the operands tie the wait to the registers the reads fill
template <int N>
__device__ __forceinline__ void g2_wait_lgkm(fs16x4& a, fs16x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

template <int OFF>
__device__ __forceinline__ void g2_lds_write_b64(unsigned addr, fs16x4 v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ u32x4 g2_lds_read_b128(unsigned addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void g2_wait_lgkm_seg(fs16x4& a, fs16x4& b, u32x4& c, u32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

template <int I, int N, class F>
__device__ __forceinline__ void g2_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    g2_static_for<I + 1, N>(f);
  }
}

// acc *= f for one accumulator block that lives in the ACCUMULATOR half of the register file.  Written with explicit
// v_accvgpr moves: a plain `acc[r] *= f` makes the compiler keep all 256 accumulators in the VALU half for the whole tile loop
// (and spill).  The MFMAs that produced `c` finished a whole S^T phase ago and the next reader is a phase away: no hazard.
__device__ __forceinline__ void g2_scale_acc(f32x16& c, float f) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float x = c[r], t;
    asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1" : "+a"(x), "=&v"(t) : "v"(f));
    c[r] = x;
  }
}

// diagnostic build (tools/exp_build.sh G2PROBE): per-phase cycle counts of wave 0 of workgroup 0, summed over its tiles, in a
// buffer that nothing else reads.  No stamp exists in the product build.
#if defined(VMMT_EXP_PROBE)
__device__ unsigned long long g2_probe[16];
#define G2_STAMP(i) do { if (probe) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ps[i] += now_ - last_; last_ = now_; } } while (0)
#else
#define G2_STAMP(i) do { } while (0)
#endif

// <<< traffic-key common
// >>> traffic-key gen2
template <int D, bool HASP>          // HASP: the softmax weights P are stored (a.p_out != NULL)
__global__ void __launch_bounds__(256, 1) gen2_kernel(Gen2Args a) {
  using C = G2<D>;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar: everything derived from it stays in SGPRs
  const int lane = threadIdx.x & 63, half = lane >> 5, r31 = lane & 31;
  // (128-token block rt, vocabulary slice).  Workgroups go to the 8 XCDs round-robin by blockIdx, and every XCD has its own L2:
  // consecutive job numbers j = split * nblocks + rt are dealt to ONE XCD, so that an XCD streams at most two slices of Wg instead
  // of all of them (PMC: 155 MB of reads per launch against 36 MB algorithmic before this mapping).
  const int nwg = (int)gridDim.x, nblk = nwg / a.nsplit;
  const int j = (nwg & 7) == 0 ? ((int)blockIdx.x & 7) * (nwg >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int split = j / nblk, rt = j - split * nblk;
  const int R0 = rt * 128 + wave * 32;                      // this wave's 32 rows
  const int row = R0 + r31;
  const int rowi = row < a.nrows ? row : a.nrows - 1;
  const int rmap = a.rows ? a.rows[rowi] : rowi;            // (compacted tokens: the row of O / y this launch's token comes from)
  const int rowc = rmap < 0 ? 0 : rmap;
  const int ntall = (a.ncols + C::BC - 1) / C::BC;
  const int t0 = split * a.tiles_per_split, t1 = min(ntall, t0 + a.tiles_per_split);

  // ring slot 0 <- first tile (issued before the resident operand is fetched: both are in flight together)
  char* const small = smem + C::SOFF;
  // one LDS-DMA piece of tile t into ring slot `slot`: pieces 0 .. PER-1 are this wave's rows of Y, piece PER the tile's 32 column
  // constants (+ 32 targets; lanes 0..31 constants, 32..63 targets) -- every wave fetches its own copy of those, so that nothing
  // here is conditional: a branch around an issue splits the MFMA phase into basic blocks with full waits at their joins.  Inside
  // the tile loop the pieces are issued one per group of four MFMAs (an issue costs ~100 cycles of the wave's instruction stream:
  // it hides in the shadow of the matrix unit).
  // Per-lane source pointers of this wave's PER pieces, advanced by one tile per issue (no per-piece address arithmetic in the tile
  // loop: with it an issue cost ~87 cycles of the wave's instruction stream).  Rows are NOT clamped: the operand must be readable up
  // to (V rounded up to 32) + 32 rows (vmmt_gen_fwd_dO checks the caller's w_rows); what lies beyond V is masked by the consumer.
  const char* pp[C::PER];
#pragma unroll
  for (int j = 0; j < C::PER; ++j) {
    const int piece = wave * C::PER + j;
    const int prow = piece * C::RPP + (C::RPP > 1 ? lane / C::CPR : 0);
    const int lch = (lane % C::CPR) ^ g2_swz(prow);         // LDS slot (lane % CPR) of the row holds logical chunk lch
    pp[j] = reinterpret_cast<const char*>(a.Y + (long)(t0 * C::BC + prow) * a.ldy) + lch * 16;
  }
  const long tile_step = (long)C::BC * a.ldy * 2;
  auto issue_piece = [&](int t, int slot, int j) {
    if (j < C::PER) {
      __builtin_amdgcn_global_load_lds((f_glb_cvoid_t*)pp[j], (f_lds_void_t*)(smem + slot * C::TILEB + (wave * C::PER + j) * 1024), 16, 0, 0);
      pp[j] += tile_step;
    } else {
      int c = t * C::BC + r31;
      c = c < a.ncols ? c : a.ncols - 1;
      const void* src = (const void*)(a.cvec + c);
      __builtin_amdgcn_global_load_lds((f_glb_cvoid_t*)src, (f_lds_void_t*)(small + slot * C::SMALLB + wave * 256), 4, 0, 0);
    }
  };
  auto issue_tile = [&](int t, int slot) {
#pragma unroll
    for (int j = 0; j <= C::PER; ++j) issue_piece(t, slot, j);
  };
  if (t0 < t1) issue_tile(t0, 0);

  // resident operand: B-operand fragments (lane = row r31, k = 16 ks + 8 half + 0..7)
  bf16x8 xf[C::KR];
  {
    const bf16_t* xr = a.X + (long)rowc * a.ldx + half * 8;
#pragma unroll
    for (int ks = 0; ks < C::KR; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xr + ks * 16);
    if constexpr (C::KL > 0) {
      char* xl = smem + C::XOFF + (wave * 32 + r31) * C::XROWB;
      const int sw = g2_swz(r31);
#pragma unroll
      for (int k = 0; k < C::KL; ++k) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + (C::KR + k) * 16);
        *reinterpret_cast<bf16x8*>(xl + (((2 * k + half) ^ sw) * 16)) = v;       // read back by this wave only
      }
    }
  }
  f32x16 acc[C::HB];
#pragma unroll
  for (int hb = 0; hb < C::HB; ++hb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[hb][r] = 0.f;

  // per-lane fragment addressing: byte offsets into the CURRENT ring slot; everything else is an instruction immediate.
  //   row read of S^T k-step ks:              ua[ks & 7] + (ks >> 3) * 256                       (chunk (2 ks + half) ^ swz(row))
  //   X fragment of k-step KR + k:            ux[k & 7] + (k >> 3) * 256
  //   transposed read of block hb, k-step kk: ul / uh[hb & 3] + (hb >> 2) * 256 + kk * 16 * ROWB (chunk (4 hb + tw) ^ swz(k-row))
  int ua[8], ux[8], ul[4], uh[4];
  {
    const int sw = g2_swz(r31);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ua[e] = r31 * C::ROWB + (((2 * e + half) ^ sw) * 16);
      ux[e] = C::XOFF + (wave * 32 + r31) * C::XROWB + (((2 * e + half) ^ sw) * 16);
    }
    const int i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3, g1 = (lane >> 4) & 1;
    const int tw = 2 * g1 + (p4 >> 1);                      // chunk 4 hb + tw of k-row 16 kk + 4 half + q (+ 8), 8-byte half p4 & 1
    const int x_lo = (q << 2) | half, x_hi = (q << 2) | (half + 2);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ul[e] = (4 * half + q) * C::ROWB + (p4 & 1) * 8 + (((4 * e + tw) ^ x_lo) * 16);
      uh[e] = (4 * half + q + 8) * C::ROWB + (p4 & 1) * 8 + (((4 * e + tw) ^ x_hi) * 16);
    }
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;     // LDS byte address of the ring

  // role state
  float ref = -INFINITY, nrl = 0.f, lsum = 0.f, rmax = -INFINITY;       // F
  int ym = -1;
  ym = (row < a.nrows && rmap >= 0) ? (int)a.y[rowc] : -1;
  // (no instruction: makes the compiler wait for these two loads HERE.  First used inside the tile loop, they get their
  //  s_waitcnt vmcnt(0) there -- in every iteration, in front of the element-wise phase, where it waits out the DMA of the next tile)
  asm volatile("" : "+v"(ym));

#if defined(VMMT_EXP_PROBE)
  const bool probe = blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long ps[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = last_, r_begin = __builtin_amdgcn_s_memrealtime();
#endif
  for (int t = t0; t < t1; ++t) {
    G2_STAMP(5);
    // tile t has landed.  vmcnt counts in issue order: the only operations younger than its DMA pieces are the two stores of P behind
    // the previous tile's element-wise phase, which may stay in flight (waiting for their acknowledgement costs ~0.5 us per tile)
    // (a wave none of whose rows exists issues no store -- its exec mask is empty and the compiler branches around them: it must not
    //  leave two DMA pieces in flight instead)
    if (HASP && R0 < a.nrows) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G2_STAMP(0);
    __syncthreads();                                        // ... for every wave; and every wave is done with tile t-1
    G2_STAMP(1);
    const int cur = (t - t0) & 1;                           // ring slot (its byte offset is folded into ua / ul / uh: see the loop tail)
    constexpr int SB = 0;
    const int tn = t + 1;                                   // (behind the slice's last tile: one tile further, into the slot nobody reads again)
    const char* sb = small + cur * C::SMALLB + wave * 256;
    const int c0 = t * C::BC;

    // ---- S^T[c][r] = sum_h Y[c][h] X[r][h]  (+ column / row constants as the initial accumulator)
    f32x16 sT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(sb + (8 * i + 4 * half) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) sT[4 * i + e] = v[e];
    }
    // Rolling prefetch: the fragment(s) of k-step ks + PD are requested right before the MFMA of k-step ks; the scheduling barriers
    // pin that order (left alone, the compiler hoists every read of the unrolled loop to the top and spills at one wave per SIMD).
    // The next tile's DMA pieces ride along, one per four MFMAs.
    {
      constexpr int PD = G2_PD;
      bf16x8 fa[PD], fx[PD];
      auto rd = [&](int ks) {
        fa[ks % PD] = *reinterpret_cast<const bf16x8*>(smem + ua[ks & 7] + (SB + (ks >> 3) * 256));
        if (ks >= C::KR) fx[ks % PD] = *reinterpret_cast<const bf16x8*>(smem + ux[(ks - C::KR) & 7] + ((ks - C::KR) >> 3) * 256);
      };
#pragma unroll
      for (int ks = 0; ks < PD; ++ks) rd(ks);
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
        if (ks < C::KR) sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks % PD], xf[ks < C::KR ? ks : 0], sT, 0, 0, 0);
        else sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks % PD], fx[ks % PD], sT, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + PD < C::KS) rd(ks + PD);
        if ((ks & 3) == 1 && (ks >> 2) < C::PER) issue_piece(tn, cur ^ 1, ks >> 2);
        if (ks == 3) issue_piece(tn, cur ^ 1, C::PER);
      }
    }

    // The first PD operand pairs of the second product are requested HERE, in front of the element-wise phase (they depend on the tile
    // only): their ~250 cycles of LDS latency pass underneath it instead of in front of the first MFMA.
    constexpr int PD = 4, NM = C::HB * 2;                   // MFMA i: block hb = G2_HB(i), k-step kk = G2_KK(i).  (2 PD early reads + the 6 LDS
                                                            // operations of the P patch must stay below the 15 the lgkmcnt counter can hold: a 16th stalls at issue)
    fs16x4 fl[PD], fh[PD];
    auto rd = [&](auto ic) {
      constexpr int i = decltype(ic)::value, hb = G2_HB(i), kk = G2_KK(i);
      constexpr int off = SB + (hb >> 2) * 256 + kk * 16 * C::ROWB;
      fl[i % PD] = g2_tr_read<off>(lds0 + ul[hb & 3]);
      fh[i % PD] = g2_tr_read<off>(lds0 + uh[hb & 3]);
    };
    g2_static_for<0, PD>([&](auto ic) { rd(ic); });

    G2_STAMP(2);
    // ---- element-wise: S^T -> P^T (bf16 B-operand fragments of the second product: k-step kk = accumulator registers 8 kk .. 8 kk + 7)
    bf16x8 pf[2];
    u32x4 seg[2];                                           // P output: two 16-byte row segments per lane (see below)
    if (c0 + C::BC > a.ncols) {                           // last tile of the vocabulary: rows >= V do not exist
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (c0 + (r & 3) + 8 * (r >> 2) + 4 * half >= a.ncols) sT[r] = -INFINITY;
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; r += 2) tmax = fmaxf(fmaxf(tmax, sT[r]), sT[r + 1]);
    rmax = fmaxf(rmax, tmax);
    if (__any(tmax > ref + G2_THR)) {                     // the reference moves (always in the first tile: ref = -inf; hardly ever later)
      const float nm = fmaxf(tmax, __shfl_xor(tmax, 32, 64));        // both halves of a token's lanes keep the same reference
      const bool mv = nm > ref + G2_THR;
      const float f = mv ? __expf(ref - nm) : 1.f;        // exp(-inf) = 0 in the first tile (accumulators are zero anyway)
      if (mv) { ref = nm; nrl = -nm * G2_L2E; }
      lsum *= f;
#pragma unroll
      for (int hb = 0; hb < C::HB; ++hb) g2_scale_acc(acc[hb], f);
      if (a.p_out && t > t0) {
        // (practically never: a logit more than 60 above everything seen so far in this slice)  The weights already stored for
        // these tokens are in units of the OLD reference: rewrite them.  They were stored by this wave: after its stores have
        // drained, device-scope loads see them.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (mv && row < a.nrows) {
          bf16_t* pr = a.p_out + (long)row * a.ldp;
          for (int v = t0 * C::BC + half * 8; v < c0; v += 16) {
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = __hip_atomic_load(reinterpret_cast<uint32_t*>(pr + v) + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float lo = __uint_as_float(w[e] << 16) * f, hi = __uint_as_float(w[e] & 0xffff0000u) * f;
              w[e] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
            }
            *reinterpret_cast<u32x4*>(pr + v) = u32x4{w[0], w[1], w[2], w[3]};
          }
        }
      }
    }
    if (__any(ym >= c0 && ym < c0 + C::BC)) {             // a target of this wave's tokens lies in this tile: keep its logit
      float tl = 0.f;
      bool hit = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool h = c0 + (r & 3) + 8 * (r >> 2) + 4 * half == ym;
        tl = h ? sT[r] : tl;
        hit = hit || h;
      }
      if (hit) a.tgt_logit[row] = tl;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sT[r], G2_L2E, nrl));
      lsum += pv;
      pf[r >> 3][r & 7] = (__bf16)pv;
    }
    if (HASP) {
      // P[m][c0 .. c0+31] for this wave's 32 tokens: through a [token][entry] patch in LDS (the lane holds 4 x 4 consecutive entries
      // of ONE token: registers 4 i .. 4 i + 3 = entries 8 i + 4 half + 0..3), written out as 16-byte row segments
      // (inline assembly: written as C++ the compiler puts an s_waitcnt vmcnt(0) in front of these LDS accesses -- it cannot tell
      //  the patch from the ring slot the LDS-DMA of the next tile is filling -- and the wave waits out the DMA it just issued.
      //  One wave's LDS operations execute in order: the reads below see the writes above without a wait in between.)
      const unsigned patch = lds0 + C::POFF + wave * C::PATCHB;
      const unsigned wr = patch + r31 * C::PPITCH + 8 * half;
      g2_lds_write_b64<0>(wr, __builtin_shufflevector(__builtin_bit_cast(fs16x8, pf[0]), __builtin_bit_cast(fs16x8, pf[0]), 0, 1, 2, 3));
      g2_lds_write_b64<16>(wr, __builtin_shufflevector(__builtin_bit_cast(fs16x8, pf[0]), __builtin_bit_cast(fs16x8, pf[0]), 4, 5, 6, 7));
      g2_lds_write_b64<32>(wr, __builtin_shufflevector(__builtin_bit_cast(fs16x8, pf[1]), __builtin_bit_cast(fs16x8, pf[1]), 0, 1, 2, 3));
      g2_lds_write_b64<48>(wr, __builtin_shufflevector(__builtin_bit_cast(fs16x8, pf[1]), __builtin_bit_cast(fs16x8, pf[1]), 4, 5, 6, 7));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int q = lane + 64 * j;
        seg[j] = g2_lds_read_b128(patch + (q >> 2) * C::PPITCH + (q & 3) * 16);
      }
      // (the segments are stored behind MFMA PD of the second product: that MFMA's operands were requested after these reads, so its wait covers them)
    }

    G2_STAMP(3);
    // ---- acc^T[h][r] += sum_c Y[c][h] P[r][c]   (A = Y^T by transposed reads, k order = the accumulator row order of S^T)
    // The transposed reads are written as inline assembly: as a builtin the compiler orders them behind the LDS-DMA of the NEXT tile
    // (s_waitcnt vmcnt(0) in front of the first one -- also with the loop unrolled by two and compile-time ring slots), which makes
    // every wave wait out the DMA it has just issued.  The price is that their lgkmcnt bookkeeping is ours (see `inflight` below).
    {
      g2_static_for<0, NM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        // LDS operations younger than pair i when MFMA i is due (LDS returns in order; lgkmcnt has 4 bits: <= 15):
        //   i <  PD: the other early pairs, the P patch's 4 writes + 2 reads (issued in the element-wise phase), the pairs issued behind
        //            MFMAs 0 .. i-1                                  = 2 (PD - 1) + 6 HASP
        //   i >= PD: the pairs behind it                              = 2 min(PD - 1, NM - 1 - i)
        constexpr int inflight = i < PD ? 2 * (PD - 1) + (HASP ? 6 : 0) : 2 * ((PD - 1) < (NM - 1 - i) ? (PD - 1) : (NM - 1 - i));
        static_assert(inflight <= 15, "lgkmcnt is a 4-bit counter");
        if constexpr (i == PD) g2_wait_lgkm_seg<inflight>(fl[i % PD], fh[i % PD], seg[0], seg[1]);     // pair PD is younger than the patch reads
        else g2_wait_lgkm<inflight>(fl[i % PD], fh[i % PD]);
        const fs16x8 v = __builtin_shufflevector(fl[i % PD], fh[i % PD], 0, 1, 2, 3, 4, 5, 6, 7);
        __builtin_amdgcn_sched_barrier(0);
        acc[G2_HB(i)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v), pf[G2_KK(i)], acc[G2_HB(i)], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (i == PD && HASP) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int q = lane + 64 * j, prow = q >> 2, ch = q & 3;
            // (as non-temporal stores: the same step time and kernel time)
            if (R0 + prow < a.nrows) *reinterpret_cast<u32x4*>(a.p_out + (long)(R0 + prow) * a.ldp + c0 + ch * 8) = seg[j];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (i + PD < NM) rd(std::integral_constant<int, i + PD>{});
#if defined(VMMT_EXP_PROBE)
        if constexpr (i == 7 || i == 15 || i == 23) { G2_STAMP(8 + i / 8); }
#endif
      });
    }
    // (no instruction: pins the 256 accumulators to the accumulator half of the register file between tiles; without it the
    //  allocator parks some of them in the VALU half and spills the resident operand instead)
#pragma unroll
    for (int hb = 0; hb < C::HB; ++hb) asm volatile("" : "+a"(acc[hb]));
    G2_STAMP(4);
    // the other ring slot next
#pragma unroll
    for (int e = 0; e < 8; ++e) ua[e] ^= C::TILEB;
#pragma unroll
    for (int e = 0; e < 4; ++e) { ul[e] ^= C::TILEB; uh[e] ^= C::TILEB; }
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no LDS-DMA may outlive the workgroup's LDS allocation
#if defined(VMMT_EXP_PROBE)
  if (probe) {
    for (int i = 0; i < 6; ++i) g2_probe[i] = ps[i];
    for (int i = 8; i < 11; ++i) g2_probe[i] = ps[i];
    g2_probe[6] = (unsigned long long)(t1 - t0);
    // in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, reported in MHz
    g2_probe[7] = (__builtin_amdgcn_s_memtime() - t_begin) * 100ull / (__builtin_amdgcn_s_memrealtime() - r_begin);
  }
#endif
  // ---- write-out: lane (r31, half) owns row `row` and the columns h = 32 hb + 8 i + 4 half + 0..3 of acc^T
  if (row < a.nrows) {
    const long pr = (long)split * a.mpad + row;
    float* dst = a.p_acc + pr * D;
    const float lt = lsum + __shfl_xor(lsum, 32, 64), mt = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
    if (half == 0) { a.p_ref[pr] = ref; a.p_l[pr] = lt; a.p_max[pr] = mt; }
#pragma unroll
    for (int hb = 0; hb < C::HB; ++hb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[hb][4 * i + e];
        *reinterpret_cast<f32x4*>(dst + 32 * hb + 8 * i + 4 * half) = v;
      }
  }
}

// <<< traffic-key gen2
// >>> traffic-key gen2p
// ---- H = 512, software-pipelined across tiles (the product kernel at H = 512) ------------------------------------------------------
// gen2_kernel above runs its three phases -- S^T (32 MFMAs), element-wise (exp, P), second product (32 MFMAs) -- one after the other in
// the ONE instruction stream a SIMD has at 512 registers per lane: the matrix unit idles under the element-wise phase, and the first
// operands of every product are waited for (profiles/r4_gen2_kernel_counters.txt: matrix pipe busy 49 % of the wave's cycles).  Here
// the phases of NEIGHBOURING tiles overlap inside that one stream.  Iteration i:
//     A(i):  the two stores of P(i-1); S^T(i): 32 MFMAs, operands by rows from tile i, the LDS-DMA of tile i + 2 in their shadow; behind
//            the last ones the first operand pairs of B(i) are requested
//     B(i):  acc^T += Y(i-1)^T P(i-1)^T: 32 MFMAs, transposed reads of tile i - 1, with element-wise(i) -- max, exp, row sums, bf16 P(i) --
//            issued BETWEEN those MFMAs
// What a wave's stream can hold next to an MFMA was measured (tools/probe/mfma_mix.hip, mfma_valu.hip, mfma_chain.hip; one wave per
// SIMD, cycles of issue): v_mfma_f32_32x32x16_bf16 4 (the matrix unit then works 32), a VALU instruction ~5, v_exp_f32 ~13,
// ds_read_b128 ~16, ds_read_b64_tr_b16 ~14, each after the other -- LDS reads do NOT issue in parallel with VALU work of the same wave.
// Phase A (MFMA + one row read) leaves ~12 cycles per MFMA free, phase B (MFMA + two transposed reads) none: the sweep is bound by the
// instruction issue of its one wave per SIMD (~2 700 cycles per tile of issue against 2 048 of MFMA), not by LDS latency (deeper
// prefetch changes nothing) nor by dependent MFMAs (a chain on one accumulator runs at 32 cycles per MFMA).  What this kernel removes
// against gen2_kernel: the waits in front of each product's first operands, 16 LDS reads of the resident operand per tile (28 of its
// 32 k-steps live in registers, 4 in LDS), the P patch in LDS (v_permlane32_swap instead), 64 v_accvgpr moves per tile (S^T accumulates
// in VGPRs through an inline-asm MFMA), an XOR in front of every LDS read (eight addresses per tile and phase).  Measured, same box,
// M 5120 x V 30 000: 332 against 368 us (P stored), 299 against 331 us (no P).  The exps moved into phase A's free slots were measured
// too (LABNOTES round 5): slower -- pinned between asm MFMAs they cost hazard nops, and phase A is held up by its DMA and store issues.
// What that costs: tiles i - 1 and i are both alive while i + 1 and i + 2 are landing: a FOUR-deep ring of 32-KiB tiles (128 KiB);
// two generations of P fragments (16 registers); and a lazy reference that moves in the element-wise phase of tile i rescales the
// accumulators only once B(i) has added tile i - 1 in the old units (`fpend`).
// One workgroup barrier per tile, as before.  vmcnt counts in issue order (gfx9: loads and stores on one counter): at the top of
// iteration i the operations younger than tile i's DMA pieces are, per later phase A, two stores of P and nine pieces.
#ifndef G2P_PDA
#define G2P_PDA 5
#endif
#ifndef G2P_KR
#define G2P_KR 28
#endif
#ifndef G2P_PDB
#define G2P_PDB 6
#endif
#ifndef G2P_NSLOT
#define G2P_NSLOT 4
#endif
struct G2P {
  static constexpr int D = 512, BC = 32, NSLOT = G2P_NSLOT, PF = NSLOT - 2;   // ring depth; tiles a DMA runs ahead of its S^T phase
  static constexpr int ROWB = 1024, TILEB = BC * ROWB;        // 32 KiB
  static constexpr int PER = 8;                               // 1-KiB LDS-DMA pieces (= rows) per wave and tile
  static constexpr int KS = 32, HB = 16;
  static constexpr int KR = G2P_KR, KL = KS - KR;             // k-steps of the resident operand in registers / in LDS ([k-step][lane] 16-byte
                                                              // entries per wave: written and read by the same lane, linear, conflict-free)
  static constexpr int SMALLB = 4 * 256;                      // per ring slot: every wave's copy of the tile's 32 column constants
  static constexpr int SOFF = NSLOT * TILEB;
  static constexpr int XOFF = SOFF + NSLOT * SMALLB, XWAVEB = KL * 1024;
  static constexpr int LDSB = XOFF + 4 * XWAVEB;
  static_assert(LDSB <= 160 * 1024, "LDS of one CU");
  // phase B: the element-wise phase of tile i takes the steps EW0 .. EW0 + 15 (one exp each)
  static constexpr int PDB = G2P_PDB, NM = 32, EW0 = 2;
};
// LDS operations younger than operand pair k when MFMA k of phase B is due (LDS returns in order; lgkmcnt has 4 bits)
constexpr int g2p_inflight(int k) {
  using C = G2P;
  return 2 * ((C::PDB - 1) < (C::NM - 1 - k) ? (C::PDB - 1) : (C::NM - 1 - k));
}

// S^T += A B with the accumulator in the VALU half of the register file.  Through the builtin every MFMA of a kernel that uses AGPRs at all
// gets an AGPR accumulator: S^T would take 16 of the 256 that acc^T fills, the allocator would park one block of acc^T in VGPRs during
// phase A and move both back and forth -- 64 v_accvgpr moves per tile in an instruction stream that has no slot to spare.  The compiler does
// not see this instruction: the wait states between the chain's last MFMA and the first VALU read of S^T are written out (g2p_mfma_done).
__device__ __forceinline__ void g2p_mfma_v(f32x16& c, bf16x8 a_, bf16x8 b_) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a_), "v"(b_));
}
__device__ __forceinline__ void g2p_mfma_done(f32x16& c) {
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(c));          // 8 passes: the result is in the registers 18 cycles after the issue at the latest
}

// phase A: LDS operations younger than the fragment(s) of k-step ks when its MFMA is due.  Issue order: 4 constants, the fragments of
// k-steps 0 .. PDA-1, then behind MFMA j: the fragment(s) of k-step j + PDA (one read; two from k-step KR on: the LDS part of the resident
// operand), then [EARLY, j >= KS - PDB] one operand pair of phase B (two reads)
template <bool EARLY>
constexpr int g2p_a_inflight(int ks) {
  using C = G2P;
  constexpr int PD = G2P_PDA;
  int n = 0;
  for (int m = ks + 1; m < ks + PD && m < C::KS; ++m) n += 1 + (m >= C::KR ? 1 : 0);
  if (EARLY)
    for (int jj = (ks - PD > 0 ? ks - PD : 0); jj <= ks - 1; ++jj)
      if (jj >= C::KS - C::PDB) n += 2;
  return n;
}
template <int OFF>
__device__ __forceinline__ u32x4 g2_lds_read_b128o(unsigned addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

template <bool HASP>
__global__ void __launch_bounds__(256, 1) gen2p_kernel(Gen2Args a) {
  using C = G2P;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, half = lane >> 5, r31 = lane & 31;
  const int nwg = (int)gridDim.x, nblk = nwg / a.nsplit;
  const int j = (nwg & 7) == 0 ? ((int)blockIdx.x & 7) * (nwg >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;     // see gen2_kernel
  const int split = j / nblk, rt = j - split * nblk;
  const int R0 = rt * 128 + wave * 32;
  const int row = R0 + r31;
  const int rowi = row < a.nrows ? row : a.nrows - 1;
  const int rmap = a.rows ? a.rows[rowi] : rowi;
  const int rowc = rmap < 0 ? 0 : rmap;
  const int ntall = (a.ncols + C::BC - 1) / C::BC;
  const int t0 = split * a.tiles_per_split, t1 = min(ntall, t0 + a.tiles_per_split);
  const int nt = t1 - t0;

  // LDS-DMA: piece jj (0..7) of this wave = row 4 jj + wave of the tile, whose swizzle is swz(row) = (wave << 2) | (jj & 3): one
  // per-lane offset serves all pieces (gen2w_kernel); the consts piece = the tile's 32 column constants, one copy per wave.  Tiles
  // beyond the vocabulary's last one are fetched from the tile right behind it (the caller's w_rows covers that one: vmmt_gen_fwd_dO).
  char* const small = smem + C::SOFF;
  const unsigned voff0 = (unsigned)(((lane ^ (wave << 2)) & 63) * 16);
  const long row4 = 4 * a.ldy * 2, tile_step = (long)C::BC * a.ldy * 2;
  const char* const ybase = reinterpret_cast<const char*>(a.Y + (long)wave * a.ldy);
  auto issue_piece = [&](const char* yb, int tsrc, int slot, int jj) {
    if (jj < C::PER) {
      const char* src = yb + jj * row4 + (voff0 ^ (unsigned)((jj & 3) << 4));
      __builtin_amdgcn_global_load_lds((f_glb_cvoid_t*)src, (f_lds_void_t*)(smem + slot * C::TILEB + (4 * jj + wave) * C::ROWB), 16, 0, 0);
    } else {
      int c = tsrc * C::BC + r31;
      c = c < a.ncols ? c : a.ncols - 1;
      const void* src = (const void*)(a.cvec + c);
      __builtin_amdgcn_global_load_lds((f_glb_cvoid_t*)src, (f_lds_void_t*)(small + slot * C::SMALLB + wave * 256), 4, 0, 0);
    }
  };
  if (nt > 0) {
#pragma unroll
    for (int d = 0; d < C::PF; ++d) {
      const int ts = min(t0 + d, ntall);
      const char* yb = ybase + (long)ts * tile_step;
#pragma unroll
      for (int jj = 0; jj <= C::PER; ++jj) issue_piece(yb, ts, d, jj);
    }
  }

  // resident operand, all of it in registers: B-operand fragments (lane = row r31, k = 16 ks + 8 half + 0..7)
  bf16x8 xf[C::KR];
  char* const xl = smem + C::XOFF + wave * C::XWAVEB + lane * 16;
  {
    const bf16_t* xr = a.X + (long)rowc * a.ldx + half * 8;
#pragma unroll
    for (int ks = 0; ks < C::KR; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xr + ks * 16);
#pragma unroll
    for (int k = 0; k < C::KL; ++k) *reinterpret_cast<bf16x8*>(xl + k * 1024) = *reinterpret_cast<const bf16x8*>(xr + (C::KR + k) * 16);
  }
  f32x16 acc[C::HB];
#pragma unroll
  for (int hb = 0; hb < C::HB; ++hb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[hb][r] = 0.f;

  // fragment addressing (gen2w_kernel's three constants; the fragment index enters as an XOR, the ring slot as a sum)
  int ua0, ul0, uh0;
  {
    const int sw = g2_swz(r31);
    ua0 = r31 * C::ROWB + ((half ^ sw) * 16);
    const int i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3, g1 = (lane >> 4) & 1;
    const int tw = 2 * g1 + (p4 >> 1);
    const int x_lo = (q << 2) | half, x_hi = (q << 2) | (half + 2);
    ul0 = (4 * half + q) * C::ROWB + (p4 & 1) * 8 + ((tw ^ x_lo) * 16);
    uh0 = (4 * half + q + 8) * C::ROWB + (p4 & 1) * 8 + ((tw ^ x_hi) * 16);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  float ref = -INFINITY, nrl = 0.f, lsum = 0.f, rmax = -INFINITY;
  int ym = (row < a.nrows && rmap >= 0) ? (int)a.y[rowc] : -1;
  asm volatile("" : "+v"(ym));
  const bool stores = HASP && R0 < a.nrows;                 // (a wave without a single row issues no store: its waits must not count one)
  const bool rowok = row < a.nrows;
  const unsigned poff = (unsigned)(((long)rowi * a.ldp + 8 * half) * 2);      // this lane's byte offset into a row-major P (< 4 GB: checked by the host)

  bf16x8 pfo[2];                                            // P(i-1)^T: B-operand fragments of the second product
  pfo[0] = pfo[1] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 sT;

  // ---- A(i): S^T(i)[c][r] = sum_h Y[c][h] X[r][h] + column constants; DMA of tile i + PF; [EARLY] the first PDB operand pairs of B(i)
  // Every LDS read of the phase is inline assembly with a counted wait in front of its MFMA (g2p_a_inflight): the compiler does not see the
  // asm MFMA as the reads' consumer in LDS-return order and waits for lgkmcnt(0) -- all the prefetched fragments -- every few MFMAs.
  fs16x4 fl[C::PDB], fh[C::PDB];                            // operand pairs of the second product (ring of PDB)
  unsigned ulq[4], uhq[4];                                  // their addresses: eight per tile instead of an XOR in front of every read
  auto rdB = [&](auto ic) {
    constexpr int k = decltype(ic)::value, hb = G2_HB(k), kk = G2_KK(k);
    constexpr int off = (hb >> 2) * 256 + kk * 16 * C::ROWB;
    fl[k % C::PDB] = g2_tr_read<off>(ulq[hb & 3]);
    fh[k % C::PDB] = g2_tr_read<off>(uhq[hb & 3]);
  };
  // P[m][c0 .. c0+31] of tile ip, row-major, straight out of the fragment registers.  A lane holds, of ONE token, the entries
  // 8 q + 4 half + 0..3 (q = 0..3): pfo[j] = [q = 2j: two packed registers | q = 2j + 1: two].  v_permlane32_swap (lanes 32..63 of the first
  // operand <-> lanes 0..31 of the second) on the register pairs (q even, q odd) leaves lane t with the 8 consecutive entries 16 j .. 16 j + 7
  // of token t and lane 32 + t with 16 j + 8 .. 16 j + 15, in register order: one 16-byte store per lane, 32 contiguous bytes per token
  // and instruction.  Issued at the head of phase A of the NEXT tile (the fragments live until that tile's phase B is over): the S^T phase
  // has instruction slots to spare, phase B has none.
  auto storeP = [&](int ip) {
    if constexpr (HASP) {
      u32x4 pw[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const u32x4 v = __builtin_bit_cast(u32x4, pfo[jj]);
        const auto s0 = __builtin_amdgcn_permlane32_swap(v[0], v[2], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(v[1], v[3], false, false);
        pw[jj] = u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
      if (rowok) {
        char* pb = reinterpret_cast<char*>(a.p_out + (t0 + ip) * C::BC) + poff;
        *reinterpret_cast<u32x4*>(pb) = pw[0];
        *reinterpret_cast<u32x4*>(pb + 32) = pw[1];
      }
    }
  };
  auto phaseA = [&](auto early, int i) {
    constexpr bool EARLY = decltype(early)::value;
    if constexpr (EARLY) storeP(i - 1);
    constexpr int PD = G2P_PDA;
    const int slot = i % C::NSLOT;
    const unsigned sb = lds0 + C::SOFF + slot * C::SMALLB + wave * 256 + half * 16;
    u32x4 cv[4];
    g2_static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; cv[q] = g2_lds_read_b128o<q * 32>(sb); });
    unsigned uaq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) uaq[e] = (lds0 + ua0 + slot * C::TILEB) ^ (e << 5);
    if constexpr (EARLY) {
      const unsigned sprev = lds0 + (unsigned)(((i + C::NSLOT - 1) % C::NSLOT) * C::TILEB);
#pragma unroll
      for (int q = 0; q < 4; ++q) { ulq[q] = (sprev + ul0) ^ (q << 6); uhq[q] = (sprev + uh0) ^ (q << 6); }
    }
    const unsigned xla = lds0 + C::XOFF + wave * C::XWAVEB + lane * 16;
    const int tn = min(t0 + i + C::PF, ntall), sn = (i + C::PF) % C::NSLOT;
    const char* yn = ybase + (long)tn * tile_step;
    u32x4 fa[PD], fx[PD < C::KL ? C::KL : PD];
    auto rd = [&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      fa[ks % PD] = g2_lds_read_b128o<(ks >> 3) * 256>(uaq[ks & 7]);
      if constexpr (ks >= C::KR) fx[ks % PD] = g2_lds_read_b128o<(ks - C::KR) * 1024>(xla);
    };
    g2_static_for<0, PD>([&](auto kc) { rd(kc); });
    g2_static_for<0, C::KS>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      constexpr int inflight = g2p_a_inflight<EARLY>(ks);
      static_assert(inflight <= 15, "lgkmcnt is a 4-bit counter");
      if constexpr (ks == 0) {
        asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(cv[0]), "+v"(cv[1]), "+v"(cv[2]), "+v"(cv[3]), "+v"(fa[0]) : "n"(inflight));
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) sT[4 * q + e] = __uint_as_float(cv[q][e]);
      } else if constexpr (ks >= C::KR) {
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[ks % PD]), "+v"(fx[ks % PD]) : "n"(inflight));
      } else {
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fa[ks % PD]) : "n"(inflight));
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks < C::KR) g2p_mfma_v(sT, __builtin_bit_cast(bf16x8, fa[ks % PD]), xf[ks < C::KR ? ks : 0]);
      else g2p_mfma_v(sT, __builtin_bit_cast(bf16x8, fa[ks % PD]), __builtin_bit_cast(bf16x8, fx[ks % PD]));
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks + PD < C::KS) rd(std::integral_constant<int, ks + PD>{});
      if constexpr (EARLY && ks >= C::KS - C::PDB) rdB(std::integral_constant<int, ks - (C::KS - C::PDB)>{});
      if constexpr ((ks & 3) == 1) issue_piece(yn, tn, sn, ks >> 2);
      if constexpr (ks == 3) issue_piece(yn, tn, sn, C::PER);
    });
    g2p_mfma_done(sT);
  };

  // ---- B: [WITH2] acc^T += Y(i-1)^T P(i-1)^T with [WITHEW] the element-wise phase of tile i between its MFMAs
  auto phaseB = [&](auto with2, auto withew, int i) {
    constexpr bool WITH2 = decltype(with2)::value, WITHEW = decltype(withew)::value;
    constexpr int PD = C::PDB, NM = C::NM;
    const int c0 = (t0 + i) * C::BC;                        // first vocabulary entry of tile i (element-wise)
    auto rd = rdB;
    if constexpr (WITH2 && !WITHEW) {                       // (the epilogue: no phase A in front of it that has requested the first pairs)
      const unsigned sprev = lds0 + (unsigned)(((i + C::NSLOT - 1) % C::NSLOT) * C::TILEB);
#pragma unroll
      for (int q = 0; q < 4; ++q) { ulq[q] = (sprev + ul0) ^ (q << 6); uhq[q] = (sprev + uh0) ^ (q << 6); }
      g2_static_for<0, PD>([&](auto ic) { rd(ic); });
    }

    bf16x8 pfn[2];
    float tmax = -INFINITY, fpend = 1.f, pe = 0.f;
    bool moved = false;
    if constexpr (WITHEW) {
      if (c0 + C::BC > a.ncols) {                           // last tile of the vocabulary: rows >= V do not exist
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (c0 + (r & 3) + 8 * (r >> 2) + 4 * half >= a.ncols) sT[r] = -INFINITY;
      }
    }
    g2_static_for<0, NM>([&](auto ic) {
      constexpr int k = decltype(ic)::value;
      if constexpr (WITH2) {
        constexpr int inflight = g2p_inflight(k);
        static_assert(inflight <= 15, "lgkmcnt is a 4-bit counter");
        g2_wait_lgkm<inflight>(fl[k % PD], fh[k % PD]);
        const fs16x8 v = __builtin_shufflevector(fl[k % PD], fh[k % PD], 0, 1, 2, 3, 4, 5, 6, 7);
        __builtin_amdgcn_sched_barrier(0);
        acc[G2_HB(k)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v), pfo[G2_KK(k)], acc[G2_HB(k)], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (k + PD < NM) rd(std::integral_constant<int, k + PD>{});
      }
      if constexpr (WITHEW) {
        if constexpr (k == 0) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) tmax = fmaxf(fmaxf(tmax, sT[r]), sT[r + 1]);
          rmax = fmaxf(rmax, tmax);
        }
        if constexpr (k == 1) {
          if (__any(tmax > ref + G2_THR)) {                 // the reference moves (always in the first tile; hardly ever later)
            const float nm = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            moved = nm > ref + G2_THR;
            fpend = moved ? __expf(ref - nm) : 1.f;
            if (moved) { ref = nm; nrl = -nm * G2_L2E; }
            lsum *= fpend;
          }
        }
        if constexpr (k >= C::EW0 && k < C::EW0 + 16) {
          constexpr int r = k - C::EW0;
          const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sT[r], G2_L2E, nrl));
          lsum += pv;
          if constexpr ((r & 1) == 0) pe = pv;
          else {
            pfn[r >> 3][(r & 7) - 1] = (__bf16)pe;
            pfn[r >> 3][r & 7] = (__bf16)pv;
          }
        }
      }
    });
#pragma unroll
    for (int hb = 0; hb < C::HB; ++hb) asm volatile("" : "+a"(acc[hb]));
    if constexpr (WITHEW) {
      if (__any(ym >= c0 && ym < c0 + C::BC)) {             // a target of this wave's tokens lies in tile i: keep its logit
        float tl = 0.f;
        bool hit = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool h = c0 + (r & 3) + 8 * (r >> 2) + 4 * half == ym;
          tl = h ? sT[r] : tl;
          hit = hit || h;
        }
        if (hit) a.tgt_logit[row] = tl;
      }
      if constexpr (WITH2) {
        if (__any(moved)) {
          // (practically never behind the first tile)  acc holds the tiles up to i - 1 in units of the OLD reference, and so do the weights
          // stored for them: rescale, rewrite.  The second product's last MFMAs may still be in flight: the accumulator reads below are
          // inline assembly, which the compiler's hazard recogniser does not see
          asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
          for (int hb = 0; hb < C::HB; ++hb) g2_scale_acc(acc[hb], fpend);
          if constexpr (HASP) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (moved && row < a.nrows) {
              bf16_t* pr = a.p_out + (long)row * a.ldp;
              for (int v = t0 * C::BC + half * 8; v < c0; v += 16) {
                uint32_t w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = __hip_atomic_load(reinterpret_cast<uint32_t*>(pr + v) + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float lo = __uint_as_float(w[e] << 16) * fpend, hi = __uint_as_float(w[e] & 0xffff0000u) * fpend;
                  w[e] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
                }
                *reinterpret_cast<u32x4*>(pr + v) = u32x4{w[0], w[1], w[2], w[3]};
              }
            }
          }
        }
      }
      pfo[0] = pfn[0];
      pfo[1] = pfn[1];
    }
  };

  if (nt > 0) {
    // iteration 0: nothing to add yet
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    phaseA(std::false_type{}, 0);
    phaseB(std::false_type{}, std::true_type{}, 0);
#if defined(VMMT_EXP_PROBE)
    // (scalar stamps of wave 0 of workgroup 0: a per-lane array of counters would cost this kernel the registers it does not have)
    const bool probe = blockIdx.x == 0 && wave == 0;
    unsigned long long ps[6] = {0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = last_, r_begin = __builtin_amdgcn_s_memrealtime();
#endif
    for (int i = 1; i < nt; ++i) {
      G2_STAMP(5);
      // tile i has landed (its pieces went out in phase A(i - PF); younger: per later phase A the two stores of P and 9 pieces)
      if (stores) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((C::PF - 1) * (2 + C::PER + 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((C::PF - 1) * (C::PER + 1)) : "memory");
      G2_STAMP(0);
      // ... for every wave; and every wave is done with tile i - 2.  (The bare barrier: __syncthreads()'s fence makes the compiler wait
      //  for vmcnt(0) here -- the two tiles in flight.  Everything the barrier orders is waited for by hand: the DMA above, phase B's
      //  transposed reads in front of their MFMAs)
      __builtin_amdgcn_s_barrier();
      G2_STAMP(1);
      phaseA(std::true_type{}, i);
      G2_STAMP(2);
      phaseB(std::true_type{}, std::true_type{}, i);
      G2_STAMP(4);
    }
#if defined(VMMT_EXP_PROBE)
    if (probe && lane == 0) {
      for (int q = 0; q < 6; ++q) g2_probe[q] = ps[q];
      g2_probe[3] = 0;
      g2_probe[6] = (unsigned long long)(nt - 1);
      g2_probe[7] = (__builtin_amdgcn_s_memtime() - t_begin) * 100ull / (__builtin_amdgcn_s_memrealtime() - r_begin);
    }
#endif
    storeP(nt - 1);
    __syncthreads();
    phaseB(std::true_type{}, std::false_type{}, nt);
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no LDS-DMA may outlive the workgroup's LDS allocation
  if (row < a.nrows) {
    const long pr = (long)split * a.mpad + row;
    float* dst = a.p_acc + pr * C::D;
    const float lt = lsum + __shfl_xor(lsum, 32, 64), mt = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
    if (half == 0) { a.p_ref[pr] = ref; a.p_l[pr] = lt; a.p_max[pr] = mt; }
#pragma unroll
    for (int hb = 0; hb < C::HB; ++hb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[hb][4 * q + e];
        *reinterpret_cast<f32x4*>(dst + 32 * hb + 8 * q + 4 * half) = v;
      }
  }
}

// <<< traffic-key gen2p
// >>> traffic-key gen2w
// ---- the same sweep at H = 1024 ------------------------------------------------------------------------------------------------
// acc^T for 32 tokens x 1024 columns would be 512 accumulator registers per lane.  Here TWO waves share a group of 32 tokens: wave
// (tg, dh) owns the columns [512 dh, 512 dh + 512) of acc^T (256 accumulator registers, as at H = 512) and -- the k range of S^T being
// those same columns -- computes HALF of the reduction of S^T, from its half of the rows of the streamed tile and its half of O (32
// k-steps, all in registers: 128).  The two partial S^T meet through 4 KiB of LDS per wave (a + b and b + a are the same number, so
// both waves go on with bit-identical logits, statistics and softmax weights) and each wave runs the second product for its columns.
// A workgroup is 2 token groups x 2 column halves = 64 tokens; a tile of Wg is 32 rows x 2 KiB = 64 KiB, the ring two of them.
// Per wave and tile that is the H = 512 kernel's 32 + 32 MFMAs plus the exchange (one more barrier), for twice the L2 -> LDS bytes:
// 64 KiB per tile and CU at the ~13 B per clock an operand byte reaches a CU's LDS is ~5000 cycles, and a tile takes ~5500 (probe build,
// tools/gen_one.py --config5: wait 172, barrier 120, first product 2029, exchange 608, element-wise 989, second product 1377, tail 190)
// -- at H = 1024 the sweep is bound by its operand stream, where the H = 512 kernel (32 KiB per ~4600 cycles) is bound by LDS reads.
struct G2W {
  static constexpr int D = 1024, BC = 32, NSLOT = 2;
  static constexpr int ROWB = 2048, TILEB = BC * ROWB;      // 64 KiB
  static constexpr int PER = TILEB / 1024 / 4;              // 1-KiB LDS-DMA pieces per wave and tile: 16
  static constexpr int KS = 32, HB = 16;                    // per wave: k-steps of its half of S^T, 32-column blocks of its half of acc^T
  static constexpr int SMALLB = 4 * 256;
  static constexpr int SOFF = NSLOT * TILEB;
  static constexpr int EOFF = SOFF + NSLOT * SMALLB;        // exchange of the partial S^T: 4 KiB per wave
  static constexpr int POFF = EOFF + 4 * 4096;
  static constexpr int PPITCH = 80, PATCHB = 32 * PPITCH;
  static constexpr int LDSB = POFF + 4 * PATCHB;
  static_assert(LDSB <= 160 * 1024, "LDS of one CU");
};

__device__ __forceinline__ void g2_lds_write_b128(unsigned addr, f32x4 v) {
  asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 g2_lds_read_f128(unsigned addr) {
  f32x4 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void g2_wait_lgkm_seg1(fs16x4& a, fs16x4& b, u32x4& c) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N));
}

template <bool HASP>
__global__ void __launch_bounds__(256, 1) gen2w_kernel(Gen2Args a) {
  using C = G2W;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tg = wave >> 1, dh = wave & 1;                  // token group, column half
  const int lane = threadIdx.x & 63, half = lane >> 5, r31 = lane & 31;
  const int nwg = (int)gridDim.x, nblk = nwg / a.nsplit;
  const int j = (nwg & 7) == 0 ? ((int)blockIdx.x & 7) * (nwg >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;     // see gen2_kernel
  const int split = j / nblk, rt = j - split * nblk;
  const int R0 = rt * 64 + tg * 32;                         // this wave pair's 32 rows
  const int row = R0 + r31;
  const int rowi = row < a.nrows ? row : a.nrows - 1;
  const int rmap = a.rows ? a.rows[rowi] : rowi;            // (compacted tokens: the row of O / y this launch's token comes from)
  const int rowc = rmap < 0 ? 0 : rmap;
  const int ntall = (a.ncols + C::BC - 1) / C::BC;
  const int t0 = split * a.tiles_per_split, t1 = min(ntall, t0 + a.tiles_per_split);

  // LDS-DMA: piece jj (0..15) of this wave = the 1-KiB half (jj & 1) of row 4 (jj >> 1) + wave of the tile.  The row's swizzle is
  // swz(row) = (wave << 2) | ((jj >> 1) & 3): four per-lane offsets serve all pieces; the rest of the address is uniform.
  char* const small = smem + C::SOFF;
  const unsigned voff0 = (unsigned)(((lane ^ (wave << 2)) & 63) * 16);        // (r & 3 enters as an XOR of bits 4-5)
  const long row4 = 4 * a.ldy * 2, tile_step = (long)C::BC * a.ldy * 2;
  const char* yt = reinterpret_cast<const char*>(a.Y + (long)(t0 * C::BC + wave) * a.ldy);       // row `wave` of the tile being fetched
  auto issue_piece = [&](const char* yb, int t, int slot, int jj) {
    if (jj < C::PER) {
      const int r = jj >> 1, hp = jj & 1;
      const char* src = yb + r * row4 + hp * 1024 + (voff0 ^ (unsigned)((r & 3) << 4));
      __builtin_amdgcn_global_load_lds((f_glb_cvoid_t*)src, (f_lds_void_t*)(smem + slot * C::TILEB + (4 * r + wave) * C::ROWB + hp * 1024), 16, 0, 0);
    } else {
      int c = t * C::BC + r31;
      c = c < a.ncols ? c : a.ncols - 1;
      const void* src = (const void*)(a.cvec + c);
      __builtin_amdgcn_global_load_lds((f_glb_cvoid_t*)src, (f_lds_void_t*)(small + slot * C::SMALLB + wave * 256), 4, 0, 0);
    }
  };
  if (t0 < t1) {
#pragma unroll
    for (int jj = 0; jj <= C::PER; ++jj) issue_piece(yt, t0, 0, jj);
  }

  // resident operand: this wave's half of the k range, B-operand fragments (lane = row r31, k = 512 dh + 16 ks + 8 half + 0..7)
  bf16x8 xf[C::KS];
  {
    const bf16_t* xr = a.X + (long)rowc * a.ldx + dh * 512 + half * 8;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xr + ks * 16);
  }
  f32x16 acc[C::HB];
#pragma unroll
  for (int hb = 0; hb < C::HB; ++hb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[hb][r] = 0.f;

  // fragment addressing as in gen2_kernel, inside this wave's 1-KiB half of every row.  The register file is full (256 accumulators
  // + 128 for X), so instead of eight + four + four per-lane offsets that flip with the ring slot there are three constants, and the
  // fragment index e and the slot enter as ONE uniform XOR per read: ((2 e + half) ^ sw) * 16 = ((half ^ sw) * 16) ^ (e << 5),
  // ((4 e + tw) ^ x) * 16 = ((tw ^ x) * 16) ^ (e << 6), slot = bit 16.
  int ua0, ul0, uh0;
  {
    const int sw = g2_swz(r31);
    ua0 = r31 * C::ROWB + dh * 1024 + ((half ^ sw) * 16);
    const int i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3, g1 = (lane >> 4) & 1;
    const int tw = 2 * g1 + (p4 >> 1);
    const int x_lo = (q << 2) | half, x_hi = (q << 2) | (half + 2);
    ul0 = (4 * half + q) * C::ROWB + dh * 1024 + (p4 & 1) * 8 + ((tw ^ x_lo) * 16);
    uh0 = (4 * half + q + 8) * C::ROWB + dh * 1024 + (p4 & 1) * 8 + ((tw ^ x_hi) * 16);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned ex_mine = lds0 + C::EOFF + wave * 4096 + lane * 16, ex_other = lds0 + C::EOFF + (wave ^ 1) * 4096 + lane * 16;
  const float bsel = dh == 0 ? 1.f : 0.f;                   // the column constants enter through ONE of the two partial sums

  float ref = -INFINITY, nrl = 0.f, lsum = 0.f, rmax = -INFINITY;
  int ym = (row < a.nrows && rmap >= 0) ? (int)a.y[rowc] : -1;
  asm volatile("" : "+v"(ym));

  // (a wave none of whose 16 stored rows exists issues no store: it must not leave a DMA piece in flight instead)
  const bool stores = HASP && R0 + 16 * dh < a.nrows;
#if defined(VMMT_EXP_PROBE)
  const bool probe = blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long ps[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = last_, r_begin = __builtin_amdgcn_s_memrealtime();
#endif
  for (int t = t0; t < t1; ++t) {
    G2_STAMP(5);
    // tile t has landed (the only younger operation is the store of P behind the previous tile's last DMA piece)
    if (stores) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G2_STAMP(0);
    __syncthreads();
    G2_STAMP(1);
    const int cur = (t - t0) & 1, sx = cur << 16;
    const int tn = t + 1;
    const char* yn = yt + tile_step;
    const char* sb = small + cur * C::SMALLB + wave * 256;
    const int c0 = t * C::BC;

    // ---- this wave's half of S^T[c][r] = sum_h Y[c][h] X[r][h]
    f32x16 sT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(sb + (8 * i + 4 * half) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) sT[4 * i + e] = v[e] * bsel;
    }
    {
      constexpr int PD = G2W_PD;
      bf16x8 fa[PD];
      auto rd = [&](int ks) { fa[ks % PD] = *reinterpret_cast<const bf16x8*>(smem + (ua0 ^ (((ks & 7) << 5) | sx)) + (ks >> 3) * 256); };
#pragma unroll
      for (int ks = 0; ks < PD; ++ks) rd(ks);
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
        sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks % PD], xf[ks], sT, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + PD < C::KS) rd(ks + PD);
        // all 16 pieces of the next tile under the FIRST product: they have the exchange, the element-wise phase and the second product to
        // land.  (Half of them under the second product, the last one three MFMAs before the tile's end: 728 instead of 172 cycles of
        // waiting at the top of every tile, 2.34 against 2.27 ms per launch at M 16384 / V 30000 -- tools/gen_one.py --config5, probe build.)
        if ((ks & 1) == 1) issue_piece(yn, tn, cur ^ 1, ks >> 1);
        if (ks == 3) issue_piece(yn, tn, cur ^ 1, C::PER);
      }
    }
    G2_STAMP(2);
    // ---- the partner's half
#pragma unroll
    for (int i = 0; i < 4; ++i) g2_lds_write_b128(ex_mine + i * 1024, f32x4{sT[4 * i], sT[4 * i + 1], sT[4 * i + 2], sT[4 * i + 3]});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
      f32x4 o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = g2_lds_read_f128(ex_other + i * 1024);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) sT[4 * i + e] += o[i][e];
    }

    G2_STAMP(8);
    constexpr int PD = 4, NM = C::HB * 2;
    fs16x4 fl[PD], fh[PD];
    auto rd = [&](auto ic) {
      constexpr int i = decltype(ic)::value, hb = G2_HB(i), kk = G2_KK(i);
      constexpr int off = (hb >> 2) * 256 + kk * 16 * C::ROWB;
      fl[i % PD] = g2_tr_read<off>(lds0 + (ul0 ^ (((hb & 3) << 6) | sx)));
      fh[i % PD] = g2_tr_read<off>(lds0 + (uh0 ^ (((hb & 3) << 6) | sx)));
    };
    g2_static_for<0, PD>([&](auto ic) { rd(ic); });

    // ---- element-wise (both waves of the pair, on identical numbers)
    bf16x8 pf[2];
    u32x4 seg;
    if (c0 + C::BC > a.ncols) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (c0 + (r & 3) + 8 * (r >> 2) + 4 * half >= a.ncols) sT[r] = -INFINITY;
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; r += 2) tmax = fmaxf(fmaxf(tmax, sT[r]), sT[r + 1]);
    rmax = fmaxf(rmax, tmax);
    if (__any(tmax > ref + G2_THR)) {
      const float nm = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const bool mv = nm > ref + G2_THR;
      const float f = mv ? __expf(ref - nm) : 1.f;
      if (mv) { ref = nm; nrl = -nm * G2_L2E; }
      lsum *= f;
#pragma unroll
      for (int hb = 0; hb < C::HB; ++hb) g2_scale_acc(acc[hb], f);
      if (a.p_out && t > t0) {
        // (practically never, see gen2_kernel)  Each wave rewrites the rows IT stored: tokens 16 dh .. 16 dh + 15 of the group
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (mv && row < a.nrows && (r31 >> 4) == dh) {
          bf16_t* pr = a.p_out + (long)row * a.ldp;
          for (int v = t0 * C::BC + half * 8; v < c0; v += 16) {
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = __hip_atomic_load(reinterpret_cast<uint32_t*>(pr + v) + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float lo = __uint_as_float(w[e] << 16) * f, hi = __uint_as_float(w[e] & 0xffff0000u) * f;
              w[e] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
            }
            *reinterpret_cast<u32x4*>(pr + v) = u32x4{w[0], w[1], w[2], w[3]};
          }
        }
      }
    }
    if (__any(ym >= c0 && ym < c0 + C::BC)) {
      float tl = 0.f;
      bool hit = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool h = c0 + (r & 3) + 8 * (r >> 2) + 4 * half == ym;
        tl = h ? sT[r] : tl;
        hit = hit || h;
      }
      if (hit && dh == 0) a.tgt_logit[row] = tl;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sT[r], G2_L2E, nrl));
      lsum += pv;
      pf[r >> 3][r & 7] = (__bf16)pv;
    }
    if (HASP) {
      // P[m][c0 .. c0+31] through the wave's [token][entry] patch (see gen2_kernel); wave dh stores the tokens 16 dh .. 16 dh + 15
      const unsigned patch = lds0 + C::POFF + wave * C::PATCHB;
      const unsigned wr = patch + r31 * C::PPITCH + 8 * half;
      g2_lds_write_b64<0>(wr, __builtin_shufflevector(__builtin_bit_cast(fs16x8, pf[0]), __builtin_bit_cast(fs16x8, pf[0]), 0, 1, 2, 3));
      g2_lds_write_b64<16>(wr, __builtin_shufflevector(__builtin_bit_cast(fs16x8, pf[0]), __builtin_bit_cast(fs16x8, pf[0]), 4, 5, 6, 7));
      g2_lds_write_b64<32>(wr, __builtin_shufflevector(__builtin_bit_cast(fs16x8, pf[1]), __builtin_bit_cast(fs16x8, pf[1]), 0, 1, 2, 3));
      g2_lds_write_b64<48>(wr, __builtin_shufflevector(__builtin_bit_cast(fs16x8, pf[1]), __builtin_bit_cast(fs16x8, pf[1]), 4, 5, 6, 7));
      const int q = lane + 64 * dh;
      seg = g2_lds_read_b128(patch + (q >> 2) * C::PPITCH + (q & 3) * 16);
    }

    G2_STAMP(3);
    // ---- this wave's columns of acc^T[h][r] += sum_c Y[c][h] P[r][c]
    {
      g2_static_for<0, NM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        // younger LDS operations when MFMA i is due: see gen2_kernel (the patch: 4 writes + 1 read)
        constexpr int inflight = i < PD ? 2 * (PD - 1) + (HASP ? 5 : 0) : 2 * ((PD - 1) < (NM - 1 - i) ? (PD - 1) : (NM - 1 - i));
        static_assert(inflight <= 15, "lgkmcnt is a 4-bit counter");
        if constexpr (i == PD && HASP) g2_wait_lgkm_seg1<inflight>(fl[i % PD], fh[i % PD], seg);
        else g2_wait_lgkm<inflight>(fl[i % PD], fh[i % PD]);
        const fs16x8 v = __builtin_shufflevector(fl[i % PD], fh[i % PD], 0, 1, 2, 3, 4, 5, 6, 7);
        __builtin_amdgcn_sched_barrier(0);
        acc[G2_HB(i)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v), pf[G2_KK(i)], acc[G2_HB(i)], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (i + PD < NM) rd(std::integral_constant<int, i + PD>{});
        if constexpr (i == NM - 2 && HASP) {
          // behind the tile's last DMA piece: the wait at the top of the next tile may leave exactly this store in flight
          const int q = lane + 64 * dh, prow = q >> 2, ch = q & 3;
          if (R0 + prow < a.nrows) *reinterpret_cast<u32x4*>(a.p_out + (long)(R0 + prow) * a.ldp + c0 + ch * 8) = seg;
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    }
#pragma unroll
    for (int hb = 0; hb < C::HB; ++hb) asm volatile("" : "+a"(acc[hb]));
    G2_STAMP(4);
    yt = yn;
  }
#if defined(VMMT_EXP_PROBE)
  if (probe) {
    for (int i = 0; i < 6; ++i) g2_probe[i] = ps[i];
    for (int i = 8; i < 11; ++i) g2_probe[i] = ps[i];
    g2_probe[6] = (unsigned long long)(t1 - t0);
    g2_probe[7] = (__builtin_amdgcn_s_memtime() - t_begin) * 100ull / (__builtin_amdgcn_s_memrealtime() - r_begin);
  }
#endif

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (row < a.nrows) {
    const long pr = (long)split * a.mpad + row;
    float* dst = a.p_acc + pr * C::D + dh * 512;
    const float lt = lsum + __shfl_xor(lsum, 32, 64), mt = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
    if (half == 0 && dh == 0) { a.p_ref[pr] = ref; a.p_l[pr] = lt; a.p_max[pr] = mt; }
#pragma unroll
    for (int hb = 0; hb < C::HB; ++hb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[hb][4 * i + e];
        *reinterpret_cast<f32x4*>(dst + 32 * hb + 8 * i + 4 * half) = v;
      }
  }
}

// <<< traffic-key gen2w
// combine of pass F: 8 tokens per workgroup (32 threads per token).  Folds the vocabulary slices' (ref, l, max, acc) and writes
//   lse, tok_nll, the statistics (NLL, words, correct), y32 (targets as int32, -1 at pads; whole 32-token tiles),
//   dO[m][h] = s_m (sum_s w_s acc_s[m][h] / l* - Wg[y_m][h]),  w_s = exp(ref_s - ref*).
// MODE 0: everything.  The training step runs the fold in TWO launches: MODE 1 -- the per-token results (lse, tok_nll, statistics, y32, c_s,
// O'_s: what the dWg product waits for, 10 us) -- and MODE 2 -- dO, the fold of the slices' 63 MB of partial accumulators, which only the
// main stream's backward chain needs: the weight-gradient product starts a launch earlier, beside the fold instead of behind it.
template <int D, int MODE = 0>
__global__ void __launch_bounds__(256) gen2_combine_kernel(const float* __restrict__ p_acc, const float* __restrict__ p_ref,
                                                           const float* __restrict__ p_l, const float* __restrict__ p_max, long mpad, int nsplit,
                                                           const float* __restrict__ tgt_logit, const long long* __restrict__ y, int M, int pad,
                                                           float inv_norm, const bf16_t* __restrict__ W, long ldw,
                                                           float* __restrict__ lse, float* __restrict__ tok_nll,
                                                           int* __restrict__ y32, float* __restrict__ dO, long lddo, float* __restrict__ blk_part,
                                                           const bf16_t* __restrict__ O, long ldo, float* __restrict__ cs,
                                                           bf16_t* __restrict__ Os, long ldos, long os_stride,
                                                           const int* __restrict__ rows, float* __restrict__ stats) {
  // stats != NULL (MODE 1): the launch's LAST workgroup folds the statistics' partial sums itself (ticket: stats[VMMT_STAT_TICKET], zero
  // between steps) -- no gen2_stats_kernel between this launch and the dWg product that waits for it.
  // rows != NULL: the launch's tokens are a compacted list -- token m is row rows[m] of O, y, lse, tok_nll and dO (< 0: no token: treated
  // as a pad); P, y32, c_s, O'_s, the partials and tgt_logit are indexed by m
  constexpr int MAXS = 16;
  __shared__ int s_row[8];
  __shared__ float s_w[8][MAXS];
  __shared__ float s_c[8][MAXS];
  __shared__ float s_invl[8], s_sc[8];
  __shared__ int s_y[8];
  const int tid = threadIdx.x;
  float nll_t = 0.f, nw_t = 0.f, nc_t = 0.f;              // statistics of this wave's tokens: three atomics per wave at the very end
  for (int it = 0; it < G2_COMBINE_GROUPS; ++it) {
  const int m0 = (blockIdx.x * G2_COMBINE_GROUPS + it) * 8;
  if (it) __syncthreads();                                  // the previous group's readers are done with s_w / s_c / ...
  if (tid < 128) {
    // waves 0-1: lane group of 16 = one token, lane s of the group = vocabulary slice s (one round trip for all partials)
    const int tk = tid >> 4, sl = tid & 15, m = m0 + tk;
    const bool live = m < M && sl < nsplit;
    const float r_s = live ? p_ref[(long)sl * mpad + m] : -INFINITY;
    const float l_s = live ? p_l[(long)sl * mpad + m] : 0.f;
    float mx = live ? p_max[(long)sl * mpad + m] : -INFINITY;
    float rstar = r_s;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) { rstar = fmaxf(rstar, __shfl_xor(rstar, o, 16)); mx = fmaxf(mx, __shfl_xor(mx, o, 16)); }
    const float w = live ? __expf(r_s - rstar) : 0.f;
    float l = w * l_s;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) l += __shfl_xor(l, o, 16);
    float nll = 0.f, nw = 0.f, nc = 0.f;
    if (m < M) {
      const float ls = rstar + logf(l);
      const int rmap = rows ? rows[m] : m;
      const long long ym = rmap >= 0 ? y[rmap] : (long long)pad;
      const bool wv = ym != pad;
      const float sc = wv ? inv_norm : 0.f;
      // slice s stored its softmax weights in units of exp(ref_s): dL/dlogit[m][v] = P[m][v] c_s[m] - [v == y_m] s_m
      const float c = live ? sc * __expf(r_s - ls) : 0.f;
      s_w[tk][sl] = w; s_c[tk][sl] = c;
      if (MODE != 2 && cs && sl < nsplit) cs[(long)sl * mpad + m] = c;
      if (sl == 0) {
        const float tl = tgt_logit[m];
        nll = wv ? ls - tl : 0.f;
        if (MODE != 2 && rmap >= 0) { lse[rmap] = ls; tok_nll[rmap] = nll; }
        nw = wv ? 1.f : 0.f;
        nc = (wv && tl >= mx) ? 1.f : 0.f;                  // accuracy: the target's logit is the row maximum (Loss.py:150-160)
        if (MODE != 2) y32[m] = wv ? (int)ym : -1;
        s_invl[tk] = 1.f / l; s_sc[tk] = sc; s_y[tk] = (int)ym; s_row[tk] = rmap;
      }
    } else {
      s_w[tk][sl] = 0.f; s_c[tk][sl] = 0.f;
      if (sl == 0) {
        if (MODE != 2 && m < ((M + 31) / 32) * 32) y32[m] = -1;
        s_sc[tk] = 0.f; s_invl[tk] = 0.f; s_y[tk] = 0; s_row[tk] = -1;
      }
    }
    nll_t += nll; nw_t += nw; nc_t += nc;
  }
  __syncthreads();
  const int tk = tid >> 5, j = tid & 31, m = m0 + tk;
  if (m < M) {
  const float invl = s_invl[tk], sc = s_sc[tk];
  const int srow = s_row[tk];                              // (< 0: no token behind this slot -- nothing to store, O'_s = 0)
  const bf16_t* wr = W + (long)s_y[tk] * ldw;
#pragma unroll
  for (int c = 0; c < D / 128; ++c) {
    const int h = c * 128 + j * 4;
    if constexpr (MODE != 1) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    // the slices' partials eight at a time: independent loads in flight (the kernel is latency-bound: 2.5 workgroups per CU)
    for (int s0 = 0; s0 < nsplit; s0 += 8) {
      f32x4 pa[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int sc_ = s0 + u < nsplit ? s0 + u : nsplit - 1;
        pa[u] = *reinterpret_cast<const f32x4*>(p_acc + ((long)sc_ * mpad + m) * D + h);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float w = s0 + u < nsplit ? s_w[tk][(s0 + u) & (MAXS - 1)] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(w, pa[u][e], v[e]);
      }
    }
    const uint2 wb = *reinterpret_cast<const uint2*>(wr + h);
    const float w4[4] = {__uint_as_float(wb.x << 16), __uint_as_float(wb.x & 0xffff0000u), __uint_as_float(wb.y << 16),
                         __uint_as_float(wb.y & 0xffff0000u)};
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = sc * (v[e] * invl - w4[e]);
    if (srow >= 0) *reinterpret_cast<f32x4*>(dO + (long)srow * lddo + h) = v;
    }
    if (MODE != 2 && Os) {                                  // O'_s[m] = c_s[m] O[m] (bf16): the B operand of dWg = P_s^T O'_s
      const uint2 ob = *reinterpret_cast<const uint2*>(O + (long)(srow < 0 ? 0 : srow) * ldo + h);
      const float o4[4] = {__uint_as_float(ob.x << 16), __uint_as_float(ob.x & 0xffff0000u), __uint_as_float(ob.y << 16),
                           __uint_as_float(ob.y & 0xffff0000u)};
      for (int s = 0; s < nsplit; ++s) {
        const float c = s_c[tk][s];
        uint2 r;
        r.x = (uint32_t)f2bf(c * o4[0]) | ((uint32_t)f2bf(c * o4[1]) << 16);
        r.y = (uint32_t)f2bf(c * o4[2]) | ((uint32_t)f2bf(c * o4[3]) << 16);
        *reinterpret_cast<uint2*>(Os + s * os_stride + (long)m * ldos + h) = r;
      }
    }
  }
  }
  }
  if (MODE != 2 && tid < 128) {
    // statistics: per-wave partial sums into scratch, folded by gen2_stats_kernel (thousands of same-address atomics cost ~10-20 ns
    // each: with them this kernel took 70 us instead of 35)
    nll_t = wave_sum(nll_t); nw_t = wave_sum(nw_t); nc_t = wave_sum(nc_t);
    if ((tid & 63) == 0) {
      float* bp = blk_part + ((long)blockIdx.x * 2 + (tid >> 6)) * 3;
      if (stats) {          // (read by another workgroup of this launch: write-through, at agent scope)
#pragma unroll
        for (int e = 0; e < 3; ++e) __hip_atomic_store(bp + e, e == 0 ? nll_t : e == 1 ? nw_t : nc_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        bp[0] = nll_t; bp[1] = nw_t; bp[2] = nc_t;
      }
    }
  }
  if (MODE == 2 || !stats) return;
  // every workgroup takes a ticket behind its partial sums (the hand-off of optim.hip's sumsq_kernel); the last one adds all of them in
  // index order -- gen2_stats_kernel's order: the same bits
  __shared__ int is_last;
  __shared__ float red[4][3];
  unsigned* ticket = reinterpret_cast<unsigned*>(stats + VMMT_STAT_TICKET);
  __syncthreads();
  if (tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == gridDim.x - 1);
  }
  __syncthreads();
  if (!is_last) return;
  float a[3] = {0.f, 0.f, 0.f};
  const int n = (int)gridDim.x * 2;
  for (int i = tid; i < n; i += 256)
#pragma unroll
    for (int e = 0; e < 3; ++e) a[e] += __hip_atomic_load(blk_part + (long)i * 3 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int e = 0; e < 3; ++e) a[e] = wave_sum(a[e]);
  if ((tid & 63) == 0)
#pragma unroll
    for (int e = 0; e < 3; ++e) red[tid >> 6][e] = a[e];
  __syncthreads();
  if (tid < 3) {
    const int e = tid;
    const int slot = e == 0 ? VMMT_STAT_NLL : e == 1 ? VMMT_STAT_NWORDS : VMMT_STAT_NCORRECT;
    stats[slot] += red[0][e] + red[1][e] + red[2][e] + red[3][e];
  }
  if (tid == 0) *ticket = 0u;
}

// stats[NLL, NWORDS, NCORRECT] += the combine kernel's per-wave partial sums (one workgroup; fixed order: reproducible)
__global__ void __launch_bounds__(256) gen2_stats_kernel(const float* __restrict__ blk_part, int n, float* __restrict__ stats) {
  float a[3] = {0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < n; i += 256)
#pragma unroll
    for (int e = 0; e < 3; ++e) a[e] += blk_part[(long)i * 3 + e];
  __shared__ float red[4][3];
#pragma unroll
  for (int e = 0; e < 3; ++e) a[e] = wave_sum(a[e]);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int e = 0; e < 3; ++e) red[threadIdx.x >> 6][e] = a[e];
  __syncthreads();
  if (threadIdx.x < 3) {
    const int e = threadIdx.x;
    const int slot = e == 0 ? VMMT_STAT_NLL : e == 1 ? VMMT_STAT_NWORDS : VMMT_STAT_NCORRECT;
    stats[slot] += red[0][e] + red[1][e] + red[2][e] + red[3][e];
  }
}

// db[v] += sum_m P[m][v] c_s(v)[m]: 512 vocabulary entries x 128 tokens per workgroup, 16-byte loads of P, one atomic per entry and block
__global__ void __launch_bounds__(256) gen2_db_kernel(const bf16_t* __restrict__ P, long ldp, const float* __restrict__ cs, long mpad,
                                                      int v_per_split, int M, int V, float* __restrict__ db) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int v0 = (blockIdx.x * 64 + lane) * 8;
  const int m0 = blockIdx.y * 128, m1 = min(M, m0 + 128);
  float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (v0 < V) {
    const float* c = cs + (long)(v0 / v_per_split) * mpad;
    for (int mb = m0 + w; mb < m1; mb += 32) {             // eight rows in flight per lane
      u32x4 pw[8];
      float cm[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = mb + 4 * u < m1 ? mb + 4 * u : m1 - 1;
        pw[u] = *reinterpret_cast<const u32x4*>(P + (long)m * ldp + v0);
        cm[u] = mb + 4 * u < m1 ? c[m] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a8[2 * e] = __builtin_fmaf(__uint_as_float(pw[u][e] << 16), cm[u], a8[2 * e]);
          a8[2 * e + 1] = __builtin_fmaf(__uint_as_float(pw[u][e] & 0xffff0000u), cm[u], a8[2 * e + 1]);
        }
    }
  }
  __shared__ float red[4][64][9];
#pragma unroll
  for (int e = 0; e < 8; ++e) red[w][lane][e] = a8[e];
  __syncthreads();
  if (w == 0 && v0 < V) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = red[0][lane][e] + red[1][lane][e] + red[2][lane][e] + red[3][lane][e];
      if (v0 + e < V) atomicAdd(db + v0 + e, t);
    }
  }
}

// the one-hot term of dL/dlogit: dWg[y_m] -= s_m O[m], db[y_m] -= s_m  (one wave per token; after the dWg GEMMs have stored)
template <int D>
__global__ void __launch_bounds__(256) gen2_onehot_kernel(const bf16_t* __restrict__ O, long ldo, const int* __restrict__ y32, float inv_norm,
                                                          int M, float* __restrict__ dW, long lddw, float* __restrict__ db,
                                                          const int* __restrict__ rows) {
  const int mc = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (mc >= M) return;
  const int ym = y32[mc];
  if (ym < 0) return;
  const int m = rows ? rows[mc] : mc;                      // (y32 >= 0 only where a token stands behind the slot)
  float* dst = dW + (long)ym * lddw;
  const int hend = lddw < D ? (int)lddw : D;      // rows of dW narrower than the tiled width (H = 500 in 512): O's padding is not part of the row
  for (int h = lane; h < hend; h += 64) atomicAdd(dst + h, -inv_norm * bf2f(O[(long)m * ldo + h]));
  if (lane == 0) atomicAdd(db + ym, -inv_norm);
}

// Order-preserving list of the decoder rows that carry a target (y != pad): rows[j] = the j-th such row for j < n, -1 for n <= j < Mc
// (Mc = n rounded up by the caller; ONE workgroup: a deterministic scan, so that the token order -- and with it the order of the f32 sums
// over tokens in the dWg product -- is the same in every run).  The generator's kernels then run over Mc compacted tokens instead of all
// T' B rows: pads cost loss weight zero in the reference and FLOPs in a dense sweep (26 % of the rows at lengths U[10, 20]).
__global__ void __launch_bounds__(1024) compact_nonpad_kernel(const long long* __restrict__ y, int M, int pad, int Mc, int* __restrict__ rows,
                                                              int* __restrict__ count) {
  __shared__ int wsum[16];
  __shared__ int total;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (M + 1023) / 1024, lo = tid * per, hi = min(M, lo + per);
  int n = 0;
  for (int i = lo; i < hi; ++i) n += y[i] != pad ? 1 : 0;
  int incl = n;                                             // inclusive scan inside the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  if (tid == 0) { int a = 0; for (int w = 0; w < 16; ++w) { const int t = wsum[w]; wsum[w] = a; a += t; } total = a; }
  __syncthreads();
  int pos = wsum[wave] + incl - n;
  for (int i = lo; i < hi; ++i)
    if (y[i] != pad) { if (pos < Mc) rows[pos] = i; ++pos; }
  const int nt = total;
  for (int j = nt + tid; j < Mc; j += 1024) rows[j] = -1;
  if (tid == 0 && count) {
    count[0] = nt;
    if (nt > Mc) count[1] = 1;          // sticky: the caller promised fewer tokens than the batch holds (its last ones were dropped)
  }
}

// vocabulary slices of the forward sweep: about 256 workgroups in all, whole groups of 8 tiles (256 rows) per slice so that
// a slice boundary is also a tile boundary of the dWg GEMM (vmmt_gemm_args.b_batch_rows)
static int g2_tiles_per_split(int M, int V, int K) {
  const int nmt = (M + 127) / 128, ntiles = (V + 31) / 32;
  int ns = 256 / nmt;
  if (K > 512) {
    // 64-token blocks (gen2w_kernel).  Cost of c slices in sweeps of the vocabulary: ceil(blocks c / 256) / c rounds + ~1 % per slice
    // for its partials.  BASELINE config 5 (T' = 64, B = 256) has exactly 256 blocks: ONE slice, every workgroup sweeps the whole
    // vocabulary and all of them stream the same tile at the same time (measured there, ms per launch: 1 slice 3.96, 2: 4.02, 4: 4.08,
    // 8: 4.31); with 260 blocks (T' = 65): 4 slices 4.13, 8: 4.05, 12: 4.51, 16: 4.45
    const int nb = (M + 63) / 64;
    double best = 1e30;
    for (int c = 1; c <= 16; ++c) {
      const double cost = (double)((nb * c + 255) / 256) / c + 0.012 * c;
      if (cost < best - 1e-9) { best = cost; ns = c; }
    }
  }
  ns = ns < 1 ? 1 : ns > 16 ? 16 : ns;
  ns = ns > ntiles ? ntiles : ns;
  return ((ntiles + ns - 1) / ns + 7) / 8 * 8;
}
static int g2_nsplit(int M, int V, int K) {
  const int ntiles = (V + 31) / 32, tps = g2_tiles_per_split(M, V, K);
  return (ntiles + tps - 1) / tps;
}

static bool g2_applies(int dtype, const void* W, int64_t ldw, const void* O, int64_t ldo, int M, int V, int K) {
  return dtype == VMMT_BF16 && (K == 1024 || K == 512 || K == 256) && M > 0 && V > 0 && ldw % 8 == 0 && ldo % 8 == 0 && ldw >= K && ldo >= K &&
         ((((uintptr_t)W) | ((uintptr_t)O)) & 15) == 0;
}

template <int D, bool HASP>
static int g2_launch(const Gen2Args& a, int grid, hipStream_t st) {
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)gen2_kernel<D, HASP>, hipFuncAttributeMaxDynamicSharedMemorySize, G2<D>::LDSB) != hipSuccess)
      return VMMT_ELAUNCH;
    done = true;
  }
  hipLaunchKernelGGL((gen2_kernel<D, HASP>), dim3(grid), dim3(256), G2<D>::LDSB, st, a);
  return check_launch();
}

template <bool HASP>
static int g2p_launch(const Gen2Args& a, int grid, hipStream_t st) {
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)gen2p_kernel<HASP>, hipFuncAttributeMaxDynamicSharedMemorySize, G2P::LDSB) != hipSuccess) return VMMT_ELAUNCH;
    done = true;
  }
  hipLaunchKernelGGL((gen2p_kernel<HASP>), dim3(grid), dim3(256), G2P::LDSB, st, a);
  return check_launch();
}

template <bool HASP>
static int g2w_launch(const Gen2Args& a, int grid, hipStream_t st) {
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)gen2w_kernel<HASP>, hipFuncAttributeMaxDynamicSharedMemorySize, G2W::LDSB) != hipSuccess) return VMMT_ELAUNCH;
    done = true;
  }
  hipLaunchKernelGGL((gen2w_kernel<HASP>), dim3(grid), dim3(256), G2W::LDSB, st, a);
  return check_launch();
}

}  // namespace vmmt

#if defined(VMMT_EXP_PROBE)
extern "C" int vmmt_g2_probe_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(vmmt::g2_probe), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int vmmt_compact_nonpad(const int64_t* y, int M, int pad, int Mc, int32_t* rows, int32_t* count, void* stream) {
  if (!y || !rows || M <= 0 || Mc <= 0) return VMMT_EINVAL;
  hipLaunchKernelGGL(vmmt::compact_nonpad_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const long long*)y, M, pad, Mc, rows, count);
  return vmmt::check_launch();
}

extern "C" int vmmt_gen_fused_applies(int dtype, int64_t ldw, int64_t ldo, int M, int V, int K) {
  return vmmt::g2_applies(dtype, nullptr, ldw, nullptr, ldo, M, V, K) ? 1 : 0;
}

extern "C" int64_t vmmt_gen_fused_ws_floats(int M, int V, int K) {
  const int64_t mpad = (int64_t)((M + 127) / 128) * 128;
  // partial accumulators + (ref, l, max) per slice, + the combine kernel's per-wave statistics (2 x 3 floats per 8 tokens)
  return (int64_t)vmmt::g2_nsplit(M, V, K) * mpad * (K + 3) + (mpad / 8 + 4) * 6;
}

extern "C" int vmmt_gen_fused_geometry(int M, int V, int K, int* nsplit, int* v_per_split, int64_t* mpad) {
  if (M <= 0 || V <= 0 || K <= 0 || !nsplit || !v_per_split || !mpad) return VMMT_EINVAL;
  *nsplit = vmmt::g2_nsplit(M, V, K);
  *v_per_split = vmmt::g2_tiles_per_split(M, V, K) * 32;
  *mpad = (int64_t)((M + 127) / 128) * 128;
  return VMMT_OK;
}

// the sweep: partial softmax statistics + un-normalised dO per (128-token block, vocabulary slice) into `ws`, target logits, and
// (Pw != NULL) the softmax weights P
extern "C" int vmmt_gen_fwd_dO(int dtype, const void* W, int64_t ldw, int w_rows, const float* bias, const void* O, int64_t ldo,
                               const int64_t* y, int M, int V, int K, float* ws, float* tgt_logit, void* Pw, int64_t ldp,
                               const int32_t* rows, void* stream) {
  using namespace vmmt;
  if (!W || !bias || !O || !y || !ws || !tgt_logit) return VMMT_EINVAL;
  if (w_rows < (V + 31) / 32 * 32 + 32) return VMMT_EINVAL;          // the sweep prefetches one 32-row tile beyond the last one (unclamped)
  if (Pw && (ldp < (V + 31) / 32 * 32 || (ldp & 7) || (((uintptr_t)Pw) & 15))) return VMMT_EINVAL;
  if (!g2_applies(dtype, W, ldw, O, ldo, M, V, K)) return VMMT_EINVAL;
  const int nmt = (M + 127) / 128, ns = g2_nsplit(M, V, K);
  const long mpad = (long)nmt * 128;
  Gen2Args a{};
  a.X = (const bf16_t*)O; a.ldx = ldo; a.nrows = M;
  a.Y = (const bf16_t*)W; a.ldy = ldw; a.ncols = V;
  a.cvec = bias; a.y = (const long long*)y;
  a.nsplit = ns; a.tiles_per_split = g2_tiles_per_split(M, V, K);
  a.mpad = mpad;
  a.p_acc = ws; a.p_ref = ws + (long)ns * mpad * K; a.p_l = a.p_ref + (long)ns * mpad; a.p_max = a.p_l + (long)ns * mpad;
  a.tgt_logit = tgt_logit;
  a.p_out = (bf16_t*)Pw; a.ldp = ldp;
  a.rows = rows;
  hipStream_t st = (hipStream_t)stream;
  if (K == 1024) return Pw ? g2w_launch<true>(a, (M + 63) / 64 * ns, st) : g2w_launch<false>(a, (M + 63) / 64 * ns, st);
#if defined(VMMT_EXP_CLASSIC)      // probe build (tools/exp_build.sh generator_fused.hip CLASSIC): the unpipelined H = 512 sweep, for same-box comparisons
  constexpr bool classic = true;
#else
  constexpr bool classic = false;
#endif
  // (gen2p_kernel addresses P with 32-bit byte offsets from a per-tile base)
  if (K == 512 && !classic && (!Pw || (int64_t)M * ldp * 2 < (int64_t)0xffff0000ll)) return Pw ? g2p_launch<true>(a, nmt * ns, st) : g2p_launch<false>(a, nmt * ns, st);
  if (Pw) return K == 512 ? g2_launch<512, true>(a, nmt * ns, st) : g2_launch<256, true>(a, nmt * ns, st);
  return K == 512 ? g2_launch<512, false>(a, nmt * ns, st) : g2_launch<256, false>(a, nmt * ns, st);
}

// folds the slices of the sweep: lse, tok_nll, statistics, dO, y32 and (cs != NULL) c_s / O'_s for the dWg GEMM
template <int MODE>
static int g2_combine(int dtype, const void* W, int64_t ldw, const void* O, int64_t ldo, const int64_t* y, int M, int V, int K,
                      int pad, float inv_norm, float* ws, const float* tgt_logit, float* lse, float* tok_nll, int* y32,
                      float* dO, int64_t lddo, float* stats, float* cs, void* Os, int64_t ldos, int64_t os_stride,
                      const int32_t* rows, void* stream) {
  using namespace vmmt;
  if (!W || !O || !y || !ws || !tgt_logit) return VMMT_EINVAL;
  if (MODE != 2 && (!lse || !tok_nll || !y32 || !stats)) return VMMT_EINVAL;
  if (MODE != 1 && (!dO || lddo < K || (lddo & 3) || (((uintptr_t)dO) & 15))) return VMMT_EINVAL;
  if ((cs != nullptr) != (Os != nullptr)) return VMMT_EINVAL;
  if (cs && (ldos < K || (ldos & 3) || (((uintptr_t)Os) & 7) || os_stride < (int64_t)M * ldos)) return VMMT_EINVAL;
  if (!g2_applies(dtype, W, ldw, O, ldo, M, V, K)) return VMMT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nmt = (M + 127) / 128, ns = g2_nsplit(M, V, K);
  const long mpad = (long)nmt * 128;
  const float* p_acc = ws;
  const float* p_ref = ws + (long)ns * mpad * K;
  const float* p_l = p_ref + (long)ns * mpad;
  const float* p_max = p_l + (long)ns * mpad;
  float* blk_part = ws + (long)ns * mpad * (K + 3);
  const int mt = (M + 31) / 32 * 32;
  const int ncb = (mt + 8 * G2_COMBINE_GROUPS - 1) / (8 * G2_COMBINE_GROUPS);
#define VMMT_G2_COMBINE(D)                                                                                                              \
  hipLaunchKernelGGL((gen2_combine_kernel<D, MODE>), dim3(ncb), dim3(256), 0, st, p_acc, p_ref, p_l, p_max, mpad, ns, tgt_logit,        \
                     (const long long*)y, M, pad, inv_norm, (const bf16_t*)W, (long)ldw, lse, tok_nll, y32, dO, (long)lddo, blk_part,   \
                     (const bf16_t*)O, (long)ldo, cs, (bf16_t*)Os, (long)ldos, (long)os_stride, rows, MODE == 1 ? stats : nullptr)
  if (K == 1024) VMMT_G2_COMBINE(1024);
  else if (K == 512) VMMT_G2_COMBINE(512);
  else VMMT_G2_COMBINE(256);
#undef VMMT_G2_COMBINE
  if (MODE == 0) hipLaunchKernelGGL(gen2_stats_kernel, dim3(1), dim3(256), 0, st, blk_part, ncb * 2, stats);
  return check_launch();
}

extern "C" int vmmt_gen_fwd_combine(int dtype, const void* W, int64_t ldw, const void* O, int64_t ldo, const int64_t* y, int M, int V, int K,
                                    int pad, float inv_norm, float* ws, const float* tgt_logit, float* lse, float* tok_nll, int* y32,
                                    float* dO, int64_t lddo, float* stats, float* cs, void* Os, int64_t ldos, int64_t os_stride,
                                    const int32_t* rows, void* stream) {
  return g2_combine<0>(dtype, W, ldw, O, ldo, y, M, V, K, pad, inv_norm, ws, tgt_logit, lse, tok_nll, y32, dO, lddo, stats, cs, Os, ldos, os_stride,
                       rows, stream);
}
// the same fold in two launches (same arguments; see gen2_combine_kernel): _stats leaves dO alone, _dO writes nothing but dO
extern "C" int vmmt_gen_fwd_combine_stats(int dtype, const void* W, int64_t ldw, const void* O, int64_t ldo, const int64_t* y, int M, int V, int K,
                                          int pad, float inv_norm, float* ws, const float* tgt_logit, float* lse, float* tok_nll, int* y32,
                                          float* dO, int64_t lddo, float* stats, float* cs, void* Os, int64_t ldos, int64_t os_stride,
                                          const int32_t* rows, void* stream) {
  return g2_combine<1>(dtype, W, ldw, O, ldo, y, M, V, K, pad, inv_norm, ws, tgt_logit, lse, tok_nll, y32, dO, lddo, stats, cs, Os, ldos, os_stride,
                       rows, stream);
}
extern "C" int vmmt_gen_fwd_combine_dO(int dtype, const void* W, int64_t ldw, const void* O, int64_t ldo, const int64_t* y, int M, int V, int K,
                                       int pad, float inv_norm, float* ws, const float* tgt_logit, float* lse, float* tok_nll, int* y32,
                                       float* dO, int64_t lddo, float* stats, float* cs, void* Os, int64_t ldos, int64_t os_stride,
                                       const int32_t* rows, void* stream) {
  return g2_combine<2>(dtype, W, ldw, O, ldo, y, M, V, K, pad, inv_norm, ws, tgt_logit, lse, tok_nll, y32, dO, lddo, stats, nullptr, nullptr, ldos,
                       os_stride, rows, stream);
}

// what the dWg GEMMs (one per vocabulary slice: dWg[slice] = P[:, slice]^T O'_slice, plain vmmt_gemm calls) leave to do:
//   dbias[v] += sum_m P[m][v] c_s(v)[m]   and the one-hot term   dWg[y_m] -= s_m O[m],  dbias[y_m] -= s_m
// colsum_done != 0: the weighted column sums came out of the dWg GEMM itself (vmmt_gemm_args.colsum_w = cs, colsum_w_stride = mpad,
// colsum_out = dbias: no second pass over P); only the one-hot term is left
extern "C" int vmmt_gen_dW_finish(int dtype, const void* Pw, int64_t ldp, const float* cs, const void* O, int64_t ldo, const int* y32,
                                  int M, int V, int K, float inv_norm, float* dW, int64_t lddw, float* dbias, int colsum_done,
                                  const int32_t* rows, void* stream) {
  using namespace vmmt;
  if (dtype != VMMT_BF16 || !Pw || !cs || !O || !y32 || !dW || !dbias || M <= 0 || V <= 0 || (K != 1024 && K != 512 && K != 256) || (ldp & 7) ||
      (((uintptr_t)Pw) & 15) || lddw <= 0)
    return VMMT_EINVAL;          // (lddw < K: rows of dW narrower than the tiled width, e.g. H = 500 in K = 512 -- the one-hot kernel stops at lddw)
  hipStream_t st = (hipStream_t)stream;
  const long mpad = (long)((M + 127) / 128) * 128;
  if (!colsum_done)
    hipLaunchKernelGGL(gen2_db_kernel, dim3((V + 511) / 512, (M + 127) / 128), dim3(256), 0, st, (const bf16_t*)Pw, (long)ldp, cs, mpad,
                       g2_tiles_per_split(M, V, K) * 32, M, V, dbias);
  if (K == 1024)
    hipLaunchKernelGGL((gen2_onehot_kernel<1024>), dim3((M + 3) / 4), dim3(256), 0, st, (const bf16_t*)O, (long)ldo, y32, inv_norm, M, dW,
                       (long)lddw, dbias, rows);
  else if (K == 512)
    hipLaunchKernelGGL((gen2_onehot_kernel<512>), dim3((M + 3) / 4), dim3(256), 0, st, (const bf16_t*)O, (long)ldo, y32, inv_norm, M, dW,
                       (long)lddw, dbias, rows);
  else
    hipLaunchKernelGGL((gen2_onehot_kernel<256>), dim3((M + 3) / 4), dim3(256), 0, st, (const bf16_t*)O, (long)ldo, y32, inv_norm, M, dW,
                       (long)lddw, dbias, rows);
  return check_launch();
}


// Fused generator passes for the bf16 training step at H = 512 (256): the vocabulary projection + log-softmax + NLL and BOTH of
// its gradients without ever writing the [V x M] softmax-gradient matrix G^T to memory.
//
// Reference semantics: generator = Linear(H, V) + LogSoftmax (onmt/ModelConstructor.py:583-585), NLLLoss(weight[pad] = 0, sum)
// (onmt/Loss.py:163-165), loss.div(normalization).backward() (onmt/Loss.py:129); dL/dlogit[m][v] = (softmax_m[v] - [v == y_m]) * s_m with
// s_m = [y_m != pad] / normalization.
//
// The unfused path (generator.hip) makes four GEMM-sized passes: statistics (logits never stored), a recompute that writes G^T
// (bf16, 307 MB at M = 5120, V = 30000), and two library GEMMs that read it back (dO = G Wg, dWg = G^T O).  Here:
//
//   pass F (vmmt_gen_fwd_dO):  one workgroup = 128 tokens x one slice of the vocabulary.  Per 64-row tile of Wg:
//        S^T = Wg_tile O^T  ->  P = exp(S - ref)  ->  acc^T += Wg_tile^T P^T,   l += rowsum P
//     i.e. the flash-attention forward with K = V = Wg: the un-normalised dO = sum_v P[m][v] Wg[v] accumulates next to the
//     softmax statistics.  `ref` is a LAZY reference: it starts as the maximum of the first tile and only moves (with a rescale of
//     the 256 accumulator registers) when a later logit exceeds it by more than 60, which keeps exp() inside f32/bf16 range and
//     costs nothing in the common case.  A small combine kernel folds the vocabulary slices:
//        lse = ref* + log l*,   dO[m] = s_m (acc*/l* - Wg[y_m]),   NLL / accuracy statistics.
//   pass G (vmmt_gen_dW):      one workgroup = 128 vocabulary rows, sweeping all tokens in tiles of 64:
//        S^T = O_tile Wg^T  ->  G = exp(S + b_v - lse_m + ln s_m) - [v == y_m] s_m  ->  dWg^T += O_tile^T G^T,   db_v += rowsum G
//     (lse is final here, so there is no running maximum at all).
//
// Both passes are ONE kernel template: the "row" operand X (128 rows per workgroup, 32 per wave) lives in registers as MFMA
// B-operand fragments for the whole kernel, the "column" operand Y streams through a two-deep ring of 64-row LDS tiles filled by
// LDS-DMA, and each tile of Y is used twice: by rows (ds_read_b128) for S^T and by columns (ds_read_b64_tr_b16) for the second
// product.  S^T is computed with the streamed rows on the MFMA A side, so that a lane's accumulator registers (one row r, 32
// columns c) are -- after conversion to bf16 -- exactly the B-operand fragments of the second product; the k-order of that product
// is the accumulator's row order (c = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)), which the transposed reads of Y reproduce.
// One wave per SIMD, the whole 512-register file: 256 accumulators (acc^T: H x 32 per wave), 128 for X, the rest for S^T / P.
//
// Per byte fetched from L2 this does 4x the MFMA work of the 128 x 128-tile kernels (a 64 KB tile of Y feeds 2 x 8.4 MFLOP),
// which is what bounded them.
#include "common.hpp"
#include "vmmt.h"

namespace vmmt {

typedef __attribute__((address_space(3))) void f_lds_void_t;
typedef __attribute__((address_space(1))) const void f_glb_cvoid_t;
typedef short fs16x4 __attribute__((ext_vector_type(4)));
typedef short fs16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr float G2_L2E = 1.4426950408889634f;
constexpr float G2_THR = 60.f;             // lazy-reference slack (natural-log units): exp(60) ~ 1e26 stays far inside bf16 / f32 range

struct Gen2Args {
  const bf16_t* X; long ldx; int nrows;    // resident operand  (F: O [M][ldo],  G: Wg [V][ldw])
  const bf16_t* Y; long ldy; int ncols;    // streamed operand  (F: Wg,          G: O)
  const float* cvec;                       // per-column constants (F: bias [V];  G: nl [Mpad] = ln s_m - lse_m, -inf beyond M / at pads)
  const int* cy32;                         // G: targets as int32 [Mpad], -1 at pads and beyond M
  const float* rbias;                      // G: bias [V] (per-row constants)
  const long long* y;                      // F: targets [M]
  int tiles_per_split, nsplit;             // F: vocabulary slices
  float* p_acc; float* p_ref; float* p_l; float* p_max; long mpad;      // F: partials [nsplit][mpad]([D])
  float* tgt_logit;                        // F: logit of the target [M]
  float* dW; long lddw; float* db; float inv_norm;                      // G
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
  static constexpr int LDSB = SOFF + NSLOT * SMALLB;
  static_assert(KS / 4 >= PER, "one DMA piece per four MFMAs of the S^T phase");
  static_assert(D % 128 == 0 && D <= 512, "register budget: 256 accumulators per lane at D = 512");
};

// XOR swizzle of the 16-byte chunk index inside a row of an LDS image whose rows are a multiple of 256 bytes (every row starts on
// bank 0):
//   row reads (ds_read_b128, 16 lanes = 16 rows that differ in row & 15): the low four chunk bits get a permutation of row & 15;
//   transposed reads (ds_read_b64_tr_b16, 16 lanes = 4 consecutive rows x 64 contiguous bytes): chunk bits 2-3 get row & 3, so the
//   four rows land 16 banks apart.
__device__ __forceinline__ int g2_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <int D>
__device__ __forceinline__ void g2_issue_one(const bf16_t* __restrict__ Y, long ldy, int c0, int limit, char* buf, int wave, int lane, int j) {
  using C = G2<D>;
  const int piece = wave * C::PER + j;
  if constexpr (C::RPP == 1) {
    // one row per piece: the row address is wave-uniform (scalar registers), the lane supplies a 32-bit byte offset only.
    // Rows beyond the operand are clamped (and masked by the consumer).
    const int rmax = limit - 1 - c0;
    const int rr = __builtin_amdgcn_readfirstlane(piece < rmax ? piece : rmax);
    const char* rowp = reinterpret_cast<const char*>(Y + (long)c0 * ldy) + (unsigned)rr * (unsigned)(ldy * 2);
    const unsigned off = (unsigned)((lane ^ g2_swz(piece)) * 16);      // LDS slot `lane` of the row holds logical chunk lane ^ swz(row)
    __builtin_amdgcn_global_load_lds((f_glb_cvoid_t*)(rowp + off), (f_lds_void_t*)(buf + piece * 1024), 16, 0, 0);
  } else {
    const int row = piece * C::RPP + lane / C::CPR;
    const int lch = (lane % C::CPR) ^ g2_swz(row);
    int g = c0 + row;
    g = g < limit ? g : limit - 1;
    const char* src = reinterpret_cast<const char*>(Y + (long)g * ldy) + lch * 16;
    __builtin_amdgcn_global_load_lds((f_glb_cvoid_t*)src, (f_lds_void_t*)(buf + piece * 1024), 16, 0, 0);
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
#if defined(VMMT_EXP_G2PROBE) || defined(VMMT_EXP_G2PROBE_EW)
__device__ unsigned long long g2_probe[16];
#define G2_STAMP(i) do { if (probe) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ps[i] += now_ - last_; last_ = now_; } } while (0)
#else
#define G2_STAMP(i) do { } while (0)
#endif

// ROLE 0: pass F, ROLE 1: pass G (see the file header)
template <int D, int ROLE>
__global__ void __launch_bounds__(256, 1) gen2_kernel(Gen2Args a) {
  using C = G2<D>;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, r31 = lane & 31;
  const int rt = ROLE == 0 ? (int)blockIdx.x / a.nsplit : (int)blockIdx.x;
  const int split = ROLE == 0 ? (int)blockIdx.x % a.nsplit : 0;
  const int R0 = rt * 128 + wave * 32;                      // this wave's 32 rows
  const int row = R0 + r31;
  const int rowc = row < a.nrows ? row : a.nrows - 1;
  const int ntall = (a.ncols + C::BC - 1) / C::BC;
  const int t0 = ROLE == 0 ? split * a.tiles_per_split : 0;
  const int t1 = ROLE == 0 ? min(ntall, t0 + a.tiles_per_split) : ntall;

  // ring slot 0 <- first tile (issued before the resident operand is fetched: both are in flight together)
  char* const small = smem + C::SOFF;
  // one LDS-DMA piece of tile t into ring slot `slot`: pieces 0 .. PER-1 are this wave's rows of Y, piece PER the tile's 32 column
  // constants (+ 32 targets; lanes 0..31 constants, 32..63 targets) -- every wave fetches its own copy of those, so that nothing
  // here is conditional: a branch around an issue splits the MFMA phase into basic blocks with full waits at their joins.  Inside
  // the tile loop the pieces are issued one per group of four MFMAs (an issue costs ~100 cycles of the wave's instruction stream:
  // it hides in the shadow of the matrix unit).
  auto issue_piece = [&](int t, int slot, int j) {
    if (j < C::PER) {
      g2_issue_one<D>(a.Y, a.ldy, t * C::BC, a.ncols, smem + slot * C::TILEB, wave, lane, j);
    } else {
      int c = t * C::BC + r31;
      if (ROLE == 0) c = c < a.ncols ? c : a.ncols - 1;     // (G: nl / y32 are padded to whole tiles by the combine kernel)
      const void* src = (ROLE == 1 && half) ? (const void*)(a.cy32 + c) : (const void*)(a.cvec + c);
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
  typedef __attribute__((address_space(3))) fs16x4 lds_v4;

  // role state
  float ref = -INFINITY, nrl = 0.f, lsum = 0.f, rmax = -INFINITY;       // F
  int ym = -1;
  float brow = 0.f, rs = 0.f;                                           // G
  if (ROLE == 0) ym = row < a.nrows ? (int)a.y[row] : -1;
  else brow = a.rbias[rowc];

#if defined(VMMT_EXP_G2PROBE) || defined(VMMT_EXP_G2PROBE_EW)
  const bool probe = blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long ps[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
  for (int t = t0; t < t1; ++t) {
    G2_STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // tile t has landed (nothing younger is in flight)
    G2_STAMP(0);
    __syncthreads();                                        // ... for every wave; and every wave is done with tile t-1
    G2_STAMP(1);
    const int cur = (t - t0) & 1;
    const int tn = t + 1 < t1 ? t + 1 : t;                  // (the last tile is fetched once more, into the slot nobody reads again)
    const char* sb = small + cur * C::SMALLB + wave * 256;
    const int c0 = t * C::BC;

    // ---- S^T[c][r] = sum_h Y[c][h] X[r][h]  (+ column / row constants as the initial accumulator)
    f32x16 sT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(sb + (8 * i + 4 * half) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) sT[4 * i + e] = ROLE == 0 ? v[e] : v[e] + brow;
    }
    // Rolling prefetch: the fragment(s) of k-step ks + PD are requested right before the MFMA of k-step ks; the scheduling barriers
    // pin that order (left alone, the compiler hoists every read of the unrolled loop to the top and spills at one wave per SIMD).
    // The next tile's DMA pieces ride along, one per four MFMAs.
    {
      constexpr int PD = 6;
      bf16x8 fa[PD], fx[PD];
      auto rd = [&](int ks) {
        fa[ks % PD] = *reinterpret_cast<const bf16x8*>(smem + ua[ks & 7] + (ks >> 3) * 256);
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
#if !defined(VMMT_EXP_G2EW) && !defined(VMMT_EXP_G2PROBE_EW)
        if ((ks & 3) == 1 && (ks >> 2) < C::PER) issue_piece(tn, cur ^ 1, ks >> 2);
        if (ks == 3) issue_piece(tn, cur ^ 1, C::PER);
#endif
      }
    }

    G2_STAMP(2);
#if defined(VMMT_EXP_G2EW) || defined(VMMT_EXP_G2PROBE_EW)
    issue_tile(tn, cur ^ 1);
#endif
    // ---- element-wise: S^T -> P^T (bf16 B-operand fragments of the second product: k-step kk = accumulator registers 8 kk .. 8 kk + 7)
    bf16x8 pf[2];
    if constexpr (ROLE == 0) {
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
    } else {
      // lanes 0..31 look at one token of the tile each: does any target fall into this wave's 32 vocabulary rows?
      const int yt = *reinterpret_cast<const int*>(sb + 128 + r31 * 4);
      if (__any(yt >= R0 && yt < R0 + 32)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const i32x4 y4 = *reinterpret_cast<const i32x4*>(sb + 128 + (8 * i + 4 * half) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float pv = __builtin_amdgcn_exp2f(sT[4 * i + e] * G2_L2E);
            sT[4 * i + e] = pv - (y4[e] == row ? a.inv_norm : 0.f);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) sT[r] = __builtin_amdgcn_exp2f(sT[r] * G2_L2E);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        rs += sT[r];
        pf[r >> 3][r & 7] = (__bf16)sT[r];
      }
    }

    G2_STAMP(3);
    // ---- acc^T[h][r] += sum_c Y[c][h] P[r][c]   (A = Y^T by transposed reads, k order = the accumulator row order of S^T)
    {
      constexpr int PD = 6, NM = C::HB * 2;                 // MFMA i: block hb = i >> 1, k-step kk = i & 1
      fs16x4 fl[PD], fh[PD];
      auto rd = [&](int i) {
        const int hb = i >> 1, kk = i & 1;
        fl[i % PD] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(smem + ul[hb & 3] + ((hb >> 2) * 256 + kk * 16 * C::ROWB)));
        fh[i % PD] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(smem + uh[hb & 3] + ((hb >> 2) * 256 + kk * 16 * C::ROWB)));
      };
#pragma unroll
      for (int i = 0; i < PD; ++i) rd(i);
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        const fs16x8 v = __builtin_shufflevector(fl[i % PD], fh[i % PD], 0, 1, 2, 3, 4, 5, 6, 7);
        acc[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v), pf[i & 1], acc[i >> 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i + PD < NM) rd(i + PD);
      }
    }
    G2_STAMP(4);
    // the other ring slot next
#pragma unroll
    for (int e = 0; e < 8; ++e) ua[e] ^= C::TILEB;
#pragma unroll
    for (int e = 0; e < 4; ++e) { ul[e] ^= C::TILEB; uh[e] ^= C::TILEB; }
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no LDS-DMA may outlive the workgroup's LDS allocation
#if defined(VMMT_EXP_G2PROBE) || defined(VMMT_EXP_G2PROBE_EW)
  if (probe) {
    for (int i = 0; i < 6; ++i) g2_probe[ROLE * 8 + i] = ps[i];
    g2_probe[ROLE * 8 + 6] = (unsigned long long)(t1 - t0);
  }
#endif
  // ---- write-out: lane (r31, half) owns row `row` and the columns h = 32 hb + 8 i + 4 half + 0..3 of acc^T
  if (row < a.nrows) {
    float* dst;
    if constexpr (ROLE == 0) {
      const long pr = (long)split * a.mpad + row;
      dst = a.p_acc + pr * D;
      const float lt = lsum + __shfl_xor(lsum, 32, 64), mt = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
      if (half == 0) { a.p_ref[pr] = ref; a.p_l[pr] = lt; a.p_max[pr] = mt; }
    } else {
      dst = a.dW + (long)row * a.lddw;
      const float rt_ = rs + __shfl_xor(rs, 32, 64);
      if (half == 0) a.db[row] += rt_;                      // one lane per vocabulary row in the whole grid: a plain accumulate
    }
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

// combine of pass F: 8 tokens per workgroup (32 threads per token).  Folds the vocabulary slices' (ref, l, max, acc) and writes
//   lse, tok_nll, the statistics (NLL, words, correct), nl = ln s_m - lse_m and y32 for pass G (whole 32-token tiles: -inf / -1 beyond M),
//   dO[m][h] = s_m (sum_s w_s acc_s[m][h] / l* - Wg[y_m][h]),  w_s = exp(ref_s - ref*).
template <int D>
__global__ void __launch_bounds__(256) gen2_combine_kernel(const float* __restrict__ p_acc, const float* __restrict__ p_ref,
                                                           const float* __restrict__ p_l, const float* __restrict__ p_max, long mpad, int nsplit,
                                                           const float* __restrict__ tgt_logit, const long long* __restrict__ y, int M, int pad,
                                                           float inv_norm, const bf16_t* __restrict__ W, long ldw,
                                                           float* __restrict__ lse, float* __restrict__ tok_nll, float* __restrict__ nl,
                                                           int* __restrict__ y32, float* __restrict__ dO, long lddo, float* __restrict__ stats) {
  constexpr int MAXS = 16;
  __shared__ float s_w[8][MAXS];
  __shared__ float s_invl[8], s_sc[8];
  __shared__ int s_y[8];
  const int tid = threadIdx.x, m0 = blockIdx.x * 8;
  if (tid < 64) {                                           // wave 0: lanes 0..7 = the block's tokens
    const int m = m0 + tid;
    float nll = 0.f, nw = 0.f, nc = 0.f;
    if (tid < 8) {
      if (m < M) {
        float rstar = -INFINITY, mx = -INFINITY;
        for (int s = 0; s < nsplit; ++s) { rstar = fmaxf(rstar, p_ref[(long)s * mpad + m]); mx = fmaxf(mx, p_max[(long)s * mpad + m]); }
        float l = 0.f;
        for (int s = 0; s < nsplit; ++s) {
          const float w = __expf(p_ref[(long)s * mpad + m] - rstar);
          s_w[tid][s] = w;
          l += w * p_l[(long)s * mpad + m];
        }
        const float ls = rstar + logf(l);
        const long long ym = y[m];
        const bool wv = ym != pad;
        lse[m] = ls;
        nll = wv ? ls - tgt_logit[m] : 0.f;
        tok_nll[m] = nll;
        nw = wv ? 1.f : 0.f;
        nc = (wv && tgt_logit[m] >= mx) ? 1.f : 0.f;         // accuracy: the target's logit is the row maximum (Loss.py:150-160)
        const float sc = wv ? inv_norm : 0.f;
        nl[m] = wv ? logf(inv_norm) - ls : -INFINITY;
        y32[m] = wv ? (int)ym : -1;
        s_invl[tid] = 1.f / l; s_sc[tid] = sc; s_y[tid] = (int)ym;
      } else {
        if (m < ((M + 31) / 32) * 32) { nl[m] = -INFINITY; y32[m] = -1; }
        s_sc[tid] = 0.f; s_invl[tid] = 0.f; s_y[tid] = 0;
        for (int s = 0; s < nsplit; ++s) s_w[tid][s] = 0.f;
      }
    }
    nll = wave_sum(nll); nw = wave_sum(nw); nc = wave_sum(nc);
    if (tid == 0) {
      atomicAdd(stats + VMMT_STAT_NLL, nll);
      atomicAdd(stats + VMMT_STAT_NWORDS, nw);
      atomicAdd(stats + VMMT_STAT_NCORRECT, nc);
    }
  }
  __syncthreads();
  const int tk = tid >> 5, j = tid & 31, m = m0 + tk;
  if (m >= M) return;
  const float invl = s_invl[tk], sc = s_sc[tk];
  const bf16_t* wr = W + (long)s_y[tk] * ldw;
#pragma unroll
  for (int c = 0; c < D / 128; ++c) {
    const int h = c * 128 + j * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsplit; ++s) {
      const f32x4 pa = *reinterpret_cast<const f32x4*>(p_acc + ((long)s * mpad + m) * D + h);
      const float w = s_w[tk][s];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(w, pa[e], v[e]);
    }
    const uint2 wb = *reinterpret_cast<const uint2*>(wr + h);
    const float w4[4] = {__uint_as_float(wb.x << 16), __uint_as_float(wb.x & 0xffff0000u), __uint_as_float(wb.y << 16),
                         __uint_as_float(wb.y & 0xffff0000u)};
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = sc * (v[e] * invl - w4[e]);
    *reinterpret_cast<f32x4*>(dO + (long)m * lddo + h) = v;
  }
}

static int g2_nsplit(int M, int V) {
  const int nmt = (M + 127) / 128, ntiles = (V + 31) / 32;
  int ns = 256 / nmt;
  ns = ns < 1 ? 1 : ns > 16 ? 16 : ns;
  ns = ns > ntiles ? ntiles : ns;
  const int tps = (ntiles + ns - 1) / ns;
  return (ntiles + tps - 1) / tps;
}

static bool g2_applies(int dtype, const void* W, int64_t ldw, const void* O, int64_t ldo, int M, int V, int K) {
  return dtype == VMMT_BF16 && (K == 512 || K == 256) && M > 0 && V > 0 && ldw % 8 == 0 && ldo % 8 == 0 && ldw >= K && ldo >= K &&
         ((((uintptr_t)W) | ((uintptr_t)O)) & 15) == 0;
}

template <int D, int ROLE>
static int g2_launch(const Gen2Args& a, int grid, hipStream_t st) {
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)gen2_kernel<D, ROLE>, hipFuncAttributeMaxDynamicSharedMemorySize, G2<D>::LDSB) != hipSuccess)
      return VMMT_ELAUNCH;
    done = true;
  }
  hipLaunchKernelGGL((gen2_kernel<D, ROLE>), dim3(grid), dim3(256), G2<D>::LDSB, st, a);
  return check_launch();
}

}  // namespace vmmt

#if defined(VMMT_EXP_G2PROBE) || defined(VMMT_EXP_G2PROBE_EW)
extern "C" int vmmt_g2_probe_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(vmmt::g2_probe), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int vmmt_gen_fused_applies(int dtype, int64_t ldw, int64_t ldo, int M, int V, int K) {
  return vmmt::g2_applies(dtype, nullptr, ldw, nullptr, ldo, M, V, K) ? 1 : 0;
}

extern "C" int64_t vmmt_gen_fused_ws_floats(int M, int V, int K) {
  const int64_t mpad = (int64_t)((M + 127) / 128) * 128;
  return (int64_t)vmmt::g2_nsplit(M, V) * mpad * (K + 3);
}

extern "C" int vmmt_gen_fwd_dO(int dtype, const void* W, int64_t ldw, const float* bias, const void* O, int64_t ldo, const int64_t* y,
                               int M, int V, int K, int pad, float inv_norm, float* ws, float* tgt_logit, float* lse, float* tok_nll,
                               float* nl, int* y32, float* dO, int64_t lddo, float* stats, void* stream) {
  using namespace vmmt;
  if (!W || !bias || !O || !y || !ws || !tgt_logit || !lse || !tok_nll || !nl || !y32 || !dO || !stats || lddo < K || (lddo & 3) ||
      (((uintptr_t)dO) & 15))
    return VMMT_EINVAL;
  if (!g2_applies(dtype, W, ldw, O, ldo, M, V, K)) return VMMT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nmt = (M + 127) / 128, ntiles = (V + 31) / 32, ns = g2_nsplit(M, V);
  const long mpad = (long)nmt * 128;
  Gen2Args a{};
  a.X = (const bf16_t*)O; a.ldx = ldo; a.nrows = M;
  a.Y = (const bf16_t*)W; a.ldy = ldw; a.ncols = V;
  a.cvec = bias; a.y = (const long long*)y;
  a.nsplit = ns; a.tiles_per_split = (ntiles + ns - 1) / ns;
  a.mpad = mpad;
  a.p_acc = ws; a.p_ref = ws + (long)ns * mpad * K; a.p_l = a.p_ref + (long)ns * mpad; a.p_max = a.p_l + (long)ns * mpad;
  a.tgt_logit = tgt_logit;
  int rc = K == 512 ? g2_launch<512, 0>(a, nmt * ns, st) : g2_launch<256, 0>(a, nmt * ns, st);
  if (rc) return rc;
  const int mt = (M + 31) / 32 * 32;
  if (K == 512)
    hipLaunchKernelGGL((gen2_combine_kernel<512>), dim3((mt + 7) / 8), dim3(256), 0, st, a.p_acc, a.p_ref, a.p_l, a.p_max, mpad, ns, tgt_logit,
                       (const long long*)y, M, pad, inv_norm, (const bf16_t*)W, (long)ldw, lse, tok_nll, nl, y32, dO, (long)lddo, stats);
  else
    hipLaunchKernelGGL((gen2_combine_kernel<256>), dim3((mt + 7) / 8), dim3(256), 0, st, a.p_acc, a.p_ref, a.p_l, a.p_max, mpad, ns, tgt_logit,
                       (const long long*)y, M, pad, inv_norm, (const bf16_t*)W, (long)ldw, lse, tok_nll, nl, y32, dO, (long)lddo, stats);
  return check_launch();
}

extern "C" int vmmt_gen_dW(int dtype, const void* W, int64_t ldw, const float* bias, const void* O, int64_t ldo, int M, int V, int K,
                           const float* nl, const int* y32, float inv_norm, float* dW, int64_t lddw, float* dbias, void* stream) {
  using namespace vmmt;
  if (!W || !bias || !O || !nl || !y32 || !dW || !dbias || lddw < K || (lddw & 3) || (((uintptr_t)dW) & 15)) return VMMT_EINVAL;
  if (!g2_applies(dtype, W, ldw, O, ldo, M, V, K)) return VMMT_EINVAL;
  Gen2Args a{};
  a.X = (const bf16_t*)W; a.ldx = ldw; a.nrows = V;
  a.Y = (const bf16_t*)O; a.ldy = ldo; a.ncols = M;
  a.cvec = nl; a.cy32 = y32; a.rbias = bias;
  a.nsplit = 1; a.tiles_per_split = 0;
  a.dW = dW; a.lddw = lddw; a.db = dbias; a.inv_norm = inv_norm;
  const int grid = (V + 127) / 128;
  return K == 512 ? g2_launch<512, 1>(a, grid, (hipStream_t)stream) : g2_launch<256, 1>(a, grid, (hipStream_t)stream);
}

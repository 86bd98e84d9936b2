// Generic LDS-tiled MFMA GEMM for gfx950 with a fused, run-time-configured epilogue.
//   C[m][n] (+)= act( alpha * sum_k Aop[m][k] * Bop[n][k] + addend )
// Operand layouts (per operand): K-contiguous ("KC": element (r,k) at P[r*ld + k], e.g. nn.Linear's x and W)
// or K-strided ("KS": element (r,k) at P[k*ld + r], used for the weight-gradient products dW = dY^T X where the
// reduction runs over the token dimension).  Replaces the cuBLAS addmm calls behind every nn.Linear / nn.LSTM
// input projection of the reference hot path.
#include "common.hpp"
#include "glds_gemm.hpp"
#include "vmmt.h"

namespace vmmt {

struct GemmEpi {
  void* C; long ldc;
  const void* addend; long ld_add; int add_rows; int add_is_T;   // add_rows: 0 none, 1 bias row, >1 row modulus, -1 full
  int act; int out_f32; int accumulate; float alpha;
  const long long* scatter_ids; int pad_id;                      // embedding-gradient scatter (atomic add, fp32 C)
  int atomic;                                                    // split-K: fp32 atomicAdd into a pre-zeroed / partial C
  int b_batch_rows; long b_batch_stride;                         // B of output rows [i r, (i + 1) r) starts at B + i * stride (elements)
  const float* colsum_w; long colsum_w_stride; float* colsum_out; // K-strided A only: colsum_out[m] += sum_k A[k][m] w[k] (w per B block)
  float* colsum_out2;                                            // unit weights (colsum_w == NULL): a second destination of the same sums
  int rblk, rvalid, cblk, cvalid;                                // output row / column block map (vmmt_gemm_args.c_row_blk): 0 = identity
};   // (the gathered-A instantiation, EPI = 2, finds vmmt_gemm_args.a_row_ids in `scatter_ids`: one more field here costs every product 16 registers)

// The epilogue runs as a few small, fully unrolled passes over the accumulator registers (static indices only:
// a run-time-indexed accumulator would be demoted to scratch), with the run-time switches hoisted outside.
#define VMMT_FOR_ACC(BODY)                                         \
  _Pragma("unroll") for (int i = 0; i < TI; ++i)                   \
  _Pragma("unroll") for (int j = 0; j < TJ; ++j)                   \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) {                 \
    const int row = m0 + aoff[i] + acc_row(r, lane);               \
    const int col = n0 + boff[j] + acc_col(lane);                  \
    float v = acc[i][j][r];                                        \
    if (row < M && col < N) { BODY }                               \
    acc[i][j][r] = v;                                              \
  }

template <class T, int TI, int TJ, int EPI = 0>
__device__ __forceinline__ void gemm_epilogue(const GemmEpi& e, f32x16 (&acc)[TI][TJ], const int (&aoff)[TI],
                                              const int (&boff)[TJ], int m0, int n0, int M, int N, int lane) {
  if (e.alpha != 1.0f) { VMMT_FOR_ACC(v *= e.alpha;) }
  if (e.add_rows != 0) {
    if (e.add_is_T) {
      const T* ad = reinterpret_cast<const T*>(e.addend);
      VMMT_FOR_ACC(long ar = e.add_rows == -1 ? row : (e.add_rows == 1 ? 0 : row % e.add_rows);
                   v += to_f<T>(ad[ar * e.ld_add + col]);)
    } else {
      const float* ad = reinterpret_cast<const float*>(e.addend);
      VMMT_FOR_ACC(long ar = e.add_rows == -1 ? row : (e.add_rows == 1 ? 0 : row % e.add_rows);
                   v += ad[ar * e.ld_add + col];)
    }
  }
  if (e.act == VMMT_ACT_RELU) { VMMT_FOR_ACC(v = fmaxf(v, 0.f);) }
  else if (e.act == VMMT_ACT_TANH) { VMMT_FOR_ACC(v = tanhf_(v);) }
  else if (e.act == VMMT_ACT_SOFTPLUS) { VMMT_FOR_ACC(v = v > 20.f ? v : log1pf(__expf(v));) }  // nn.Softplus(1, 20)
  else if (e.act == VMMT_ACT_SIGMOID) { VMMT_FOR_ACC(v = sigmoidf_(v);) }
  if (e.rblk | e.cblk) {
    // padded blocks -> the reference's dense layout (out_f32, no scatter: checked by vmmt_gemm): rows / columns in a block's padding are dropped
    float* C = reinterpret_cast<float*>(e.C);
    const int rb = e.rblk ? e.rblk : 0x7fffffff, rv = e.rblk ? e.rvalid : 0x7fffffff;
    const int cb = e.cblk ? e.cblk : 0x7fffffff, cv = e.cblk ? e.cvalid : 0x7fffffff;
#define VMMT_MAPPED(STORE)                                                                  \
    VMMT_FOR_ACC(const int rq = row / rb; const int rr = row - rq * rb; const int cq = col / cb; const int cr = col - cq * cb; \
                 if (rr < rv && cr < cv) { float* p = C + (long)(rq * rv + rr) * e.ldc + (cq * cv + cr); STORE })
    if (e.atomic) { VMMT_MAPPED(atomicAdd(p, v);) }
    else if (e.accumulate) { VMMT_MAPPED(*p += v;) }
    else { VMMT_MAPPED(*p = v;) }
#undef VMMT_MAPPED
  } else if (EPI != 2 && e.scatter_ids) {
    float* C = reinterpret_cast<float*>(e.C);
    VMMT_FOR_ACC(long long id = e.scatter_ids[row]; if (id != e.pad_id) atomicAdd(C + id * e.ldc + col, v);)
  } else if (e.atomic) {
    float* C = reinterpret_cast<float*>(e.C);
    VMMT_FOR_ACC(atomicAdd(C + (long)row * e.ldc + col, v);)
  } else if (e.out_f32) {
    float* C = reinterpret_cast<float*>(e.C);
    if (e.accumulate) { VMMT_FOR_ACC(C[(long)row * e.ldc + col] += v;) }
    else { VMMT_FOR_ACC(C[(long)row * e.ldc + col] = v;) }
  } else {
    T* C = reinterpret_cast<T*>(e.C);
    if (e.accumulate) { VMMT_FOR_ACC(T* p = C + (long)row * e.ldc + col; *p = from_f<T>(to_f<T>(*p) + v);) }
    else { VMMT_FOR_ACC(C[(long)row * e.ldc + col] = from_f<T>(v);) }
  }
}

// one output tile of one product: `tile` = its index among the product's tiles, `split` = which part of the reduction
template <class T, int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, int BK, bool DB, int GL, int EPI = 0>
__device__ __forceinline__ void gemm_tile(const T* __restrict__ A, long lda, const T* __restrict__ B_, long ldb, int M, int N, int K, int a_kmod,
                                          int b_kmod, int tiles_n, int kper, GemmEpi& epi, const int tile, const int split) {
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  constexpr int TI = WM / 32, TJ = WN / 32;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  // per-row-block B (vmmt_gemm_args.b_batch_rows): every row of a tile belongs to one block (checked by vmmt_gemm)
  const T* __restrict__ B = epi.b_batch_rows > 0 ? B_ + (long)(m0 / epi.b_batch_rows) * epi.b_batch_stride : B_;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave / (BN / WN), wn = wave % (BN / WN);
  int aoff[TI], boff[TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i) aoff[i] = wm * WM + i * 32;
#pragma unroll
  for (int j = 0; j < TJ; ++j) boff[j] = wn * WN + j * 32;
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  LinearMap amap{m0, M}, bmap{n0, N};
  const int kbeg = split * kper;                      // split-K: this block reduces over [kbeg, kend)
  const int kend = min(K, kbeg + kper);
  if constexpr (GL == 3) {   // LDS-DMA main loops (glds_gemm.hpp); preconditions checked by launch_layout
    // the weighted column sums of A ride along, shared out over the column tiles of a row slab (one pass over A per launch)
    const float* cw = epi.colsum_w ? epi.colsum_w + (epi.b_batch_rows > 0 ? (long)(m0 / epi.b_batch_rows) * epi.colsum_w_stride : 0L) : nullptr;
    gemm_mainloop_glds3<BM, BN, NT / 64, A_KC, B_KC, TI, TJ>(A, lda, m0, M, B, ldb, n0, N, kbeg, kend, aoff, boff, acc, smem_raw, cw, epi.colsum_out,
                                                             tn, tiles_n);
#if defined(VMMT_EXP_TILE512)
  } else if constexpr (GL == 6) {
    if constexpr (A_KC && B_KC)
      gemm_mainloop_hglds_pipe<BM, BN, NT / 64, TI, TJ>(A, lda, m0, M, B, ldb, n0, N, kbeg, kend, aoff, boff, acc, smem_raw);
    else
      gemm_mainloop_hglds_pipe_t<BM, BN, NT / 64, TI, TJ>(A, lda, m0, M, B, ldb, n0, N, kbeg, kend, aoff, boff, acc, smem_raw);
#endif
  } else if constexpr (GL == 4 || GL == 5) {   // half-depth slabs, both operands K-contiguous (glds_gemm.hpp); 5: four stages
    static_assert((GL != 4 && GL != 5) || (A_KC && B_KC), "the half-depth loop stages K-contiguous operands");
    gemm_mainloop_hglds3<BM, BN, NT / 64, TI, TJ, GL == 5 ? 4 : 3>(A, lda, m0, M, B, ldb, n0, N, kbeg, kend, aoff, boff, acc, smem_raw);
  } else if constexpr (GL == 1) {
    // plain column sums of A (unit weights) in the workgroups of the first column tile, every K split adds its share
    float* co = (epi.colsum_out && !epi.colsum_w && tn == 0) ? epi.colsum_out : nullptr;
    gemm_mainloop_glds<BM, BN, NT / 64, A_KC, B_KC, TI, TJ, true, EPI == 2>(A, lda, m0, M, B, ldb, n0, N, kbeg, kend, aoff, boff, acc, smem_raw, co,
                                                                            epi.colsum_out2, epi.rblk, epi.rvalid, epi.scatter_ids);   // (EPI = 2: the gathered-A instantiation)
  } else {
    gemm_mainloop<T, BM, BN, BK, NT, A_KC, B_KC, TI, TJ, LinearMap, LinearMap, DB>(A, lda, amap, B, ldb, bmap, kend, a_kmod,
                                                                                    b_kmod, aoff, boff, acc, smem, kbeg);
  }
  if (split != 0) epi.add_rows = 0;                  // the addend is added once
#if defined(VMMT_EXP_TILE512)
  if constexpr (GL == 4 || GL == 5 || GL == 6 || (BM == 256 && BN == 256)) {                // probe: plain store only (the general epilogue spills beside 128 accumulators)
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + aoff[i] + acc_row(r, lane), col = n0 + boff[j] + acc_col(lane);
          if (row < M && col < N) {
            if (epi.out_f32) reinterpret_cast<float*>(epi.C)[(long)row * epi.ldc + col] = acc[i][j][r];
            else reinterpret_cast<T*>(epi.C)[(long)row * epi.ldc + col] = from_f<T>(acc[i][j][r]);
          }
        }
    return;
  }
#endif
  gemm_epilogue<T, TI, TJ, EPI>(epi, acc, aoff, boff, m0, n0, M, N, lane);
}

template <class T, int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, int BK = 32, bool DB = true, int GL = 0, int EPI = 0>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64)
gemm_kernel(const T* __restrict__ A, long lda, const T* __restrict__ B_, long ldb, int M, int N, int K, int a_kmod,
            int b_kmod, int tiles_n, int kper, GemmEpi epi) {
  gemm_tile<T, BM, BN, WM, WN, A_KC, B_KC, BK, DB, GL, EPI>(A, lda, B_, ldb, M, N, K, a_kmod, b_kmod, tiles_n, kper, epi,
                                                            xcd_remap(blockIdx.x, gridDim.x), blockIdx.y);
}

// ---- GROUPED launch: the tiles of up to VMMT_GEMM_GROUP_MAX independent products in ONE grid (vmmt_gemm_group).  The weight-gradient
// products of an LSTM layer (dW_hh and dW_ih of each direction, ~20 us each on a quarter of the chip) used to be five launches on two
// streams at the end of the backward pass; as one grid of ~400 workgroups they are a single round on the 256 CUs.  The descriptors
// travel in the kernel-argument segment (uniform: scalar loads); a workgroup finds its product by comparing its index with the
// products' first-workgroup table.
struct GemmGroupItem {
  const void* A; long lda; const void* B; long ldb;
  int M, N, K, tiles_n, ntiles, kper;
  GemmEpi epi;
};
struct GemmGroupArgs {
  int n;
  int start[VMMT_GEMM_GROUP_MAX + 1];          // first workgroup of product i; start[n] = the grid
  GemmGroupItem item[VMMT_GEMM_GROUP_MAX];
};

template <class T, int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, int BK, bool DB, int GL>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64) gemm_group_kernel(const GemmGroupArgs ga) {
  const int bid = blockIdx.x;
  int g = 0;
#pragma unroll
  for (int i = 1; i < VMMT_GEMM_GROUP_MAX; ++i)
    if (i < ga.n && bid >= ga.start[i]) g = i;
  g = __builtin_amdgcn_readfirstlane(g);
  const GemmGroupItem& it = ga.item[g];
  // every (product, K split) owns a range of workgroup indices that starts on a multiple of 8: workgroups are dealt to the 8 XCDs round-robin
  // by index, and xcd_remap() counts on (tile index & 7) being the XCD.  The <= 7 workgroups at the end of a range exit
  const int local = bid - ga.start[g];
  const int stride8 = (it.ntiles + 7) & ~7;
  const int split = local / stride8, tl = local - split * stride8;
  if (tl >= it.ntiles) return;
  GemmEpi epi = it.epi;
  gemm_tile<T, BM, BN, WM, WN, A_KC, B_KC, BK, DB, GL>((const T*)it.A, it.lda, (const T*)it.B, it.ldb, it.M, it.N, it.K, 0, 0, it.tiles_n,
                                                       it.kper, epi, xcd_remap(tl, it.ntiles), split);
}

template <class T, int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, int BK = 32, bool DB = true, int GL = 0, int EPI = 0>
static int launch_cfg(const vmmt_gemm_args* a, const GemmEpi& epi, hipStream_t st, size_t lds_min = 0) {
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  int tm = (a->M + BM - 1) / BM, tn = (a->N + BN - 1) / BN;
  size_t smem = (GL == 5 || GL == 6) ? (size_t)hglds3_smem_bytes<BM, BN, 4>() : GL == 4 ? (size_t)hglds3_smem_bytes<BM, BN>() : GL == 3 ? (size_t)glds3_smem_bytes<BM, BN>() : GL ? (size_t)glds_smem_bytes<BM, BN>() : gemm_smem_elems<T, BM, BN, BK, A_KC, B_KC, DB>() * sizeof(T);
  if (smem < lds_min) smem = lds_min;          // occupancy cap by LDS request (VMMT_TILE_128_ONE_PER_CU)
  if (smem > 64 * 1024) {
    static size_t allowed = 0;
    if (smem > allowed) { (void)hipFuncSetAttribute((const void*)gemm_kernel<T, BM, BN, WM, WN, A_KC, B_KC, BK, DB, GL, EPI>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); allowed = smem; }
  }
  int splits = epi.atomic ? a->split_k : 1;
  if (splits < 1) splits = 1;
  int kper = ((a->K + splits - 1) / splits + BK - 1) / BK * BK;
  if (kper < BK) kper = BK;
  splits = (a->K + kper - 1) / kper;
  if (splits < 1) splits = 1;
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, A_KC, B_KC, BK, DB, GL, EPI>), dim3(tm * tn, splits), dim3(NT), smem, st,
                     (const T*)a->A, (long)a->lda, (const T*)a->B, (long)a->ldb, a->M, a->N, a->K, a->a_kmod,
                     a->b_kmod, tn, kper, epi);
  return check_launch();
}

template <class T, bool A_KC, bool B_KC>
static int launch_layout(const vmmt_gemm_args* a, const GemmEpi& epi, hipStream_t st) {
  long t128 = (long)((a->M + 127) / 128) * ((a->N + 127) / 128) * (epi.atomic && a->split_k > 1 ? a->split_k : 1);
  bool gl_ok = false;
  if constexpr (sizeof(T) == 2)
    gl_ok = a->K % 64 == 0 && a->K > 0 && a->lda % 8 == 0 && a->ldb % 8 == 0 && a->a_kmod == 0 && a->b_kmod == 0 &&
            ((((uintptr_t)a->A) | ((uintptr_t)a->B)) & 15) == 0;
  if (a->a_row_ids) {          // gathered A operand: the two-stage 128 x 128 LDS-DMA loop, K-contiguous operands, table rows from 4-byte aligned addresses
    if constexpr (sizeof(T) == 2 && A_KC && B_KC) {
      if (a->K % 64 == 0 && a->K > 0 && a->lda % 4 == 0 && a->ldb % 8 == 0 && a->a_kmod == 0 && a->b_kmod == 0 && ((((uintptr_t)a->A) & 3) | (((uintptr_t)a->B) & 15)) == 0 &&
          !epi.colsum_out && !epi.colsum_w && (a->tile == 0 || a->tile == 128))
        return launch_cfg<T, 128, 128, 64, 64, A_KC, B_KC, 64, true, 1, 2>(a, epi, st);      // an instantiation of its own: the plain product keeps its registers
    }
    return VMMT_EINVAL;
  }
  // With the LDS-DMA loop a 128 x 128 workgroup retires a 64-deep slab in ~0.55 us against ~0.4 us per 32-deep slab of the
  // 64 x 64 configuration, so for long reductions it is also the lower-LATENCY choice when there are fewer tiles than CUs
  // (M = 256 products of the inference networks at K = 2048: 20 us instead of 26-32 us); below K = 1024 the 64 x 64
  // configuration starts faster (tools/gemm_ab.py).
#if defined(VMMT_EXP_TILE512)        // probe (tools/exp_build.sh gemm.hip TILE512): 256 x 256 tiles, 8 waves of 128 x 64, 32-deep slabs, four / three stages
  if constexpr (sizeof(T) == 2 && !A_KC && !B_KC) {      // 520: probe 518's pipeline for K-strided operands (dWg's layout)
    if (a->tile == 520 && a->K % 32 == 0 && a->lda % 8 == 0 && a->ldb % 8 == 0 && a->a_kmod == 0 && a->b_kmod == 0)
      return launch_cfg<T, 256, 256, 128, 128, A_KC, B_KC, 32, true, 6>(a, epi, st);
  }
  if constexpr (sizeof(T) == 2 && A_KC && B_KC) {
    if (a->tile >= 512 && a->tile <= 518 && a->K % 32 == 0 && a->lda % 8 == 0 && a->ldb % 8 == 0) {
      if (a->tile == 512) return launch_cfg<T, 256, 256, 128, 64, A_KC, B_KC, 32, true, 5>(a, epi, st);
      if (a->tile == 518) return launch_cfg<T, 256, 256, 128, 128, A_KC, B_KC, 32, true, 6>(a, epi, st);      // ... hand-ordered across slabs (gemm_mainloop_hglds_pipe)
      if (a->tile == 516) return launch_cfg<T, 256, 256, 128, 128, A_KC, B_KC, 32, true, 5>(a, epi, st);      // 4 waves of 128 x 128 (one per SIMD), four 32-deep stages
      if (a->tile == 517) return launch_cfg<T, 256, 256, 128, 128, A_KC, B_KC, 32, true, 4>(a, epi, st);      // ... three stages
      if (a->tile == 514 && a->K % 64 == 0) return launch_cfg<T, 256, 256, 128, 128, A_KC, B_KC, 64, true, 1>(a, epi, st);    // 4 waves of 128 x 128, two 64-deep stages
      if (a->tile == 515 && a->K % 64 == 0) return launch_cfg<T, 256, 256, 128, 64, A_KC, B_KC, 64, true, 1>(a, epi, st);     // 8 waves of 128 x 64, two 64-deep stages
      return launch_cfg<T, 256, 256, 128, 64, A_KC, B_KC, 32, true, 4>(a, epi, st);
    }
  }
#endif
  if (a->tile == 128 || a->tile == 256 || a->tile == VMMT_TILE_128_ONE_PER_CU || (a->tile == 0 && (t128 >= 192 || (gl_ok && a->K >= 1024 && t128 >= 8)))) {
    // ONE_PER_CU: an 88-KiB LDS request admits one workgroup per CU (2 x 88 > 160 KiB) and leaves 72 KiB plus half of the
    // registers for a 64-KiB workgroup of another stream (the LSTM step kernels of the critical path)
    const size_t lds_min = a->tile == VMMT_TILE_128_ONE_PER_CU ? 88 * 1024 : 0;
    // bf16: BK = 64 (each row contributes a full 128-byte line per slab) + two LDS buffers: +30 % over BK = 32 on the
    // long-K gradient GEMMs (tools/gemm_ab.py, interleaved in one process)
    if constexpr (sizeof(T) == 2) {
      // large problems: 256 x 128 tiles, 8 waves, three LDS stages with counted waits (two slabs in flight): +7..14 % over
      // the two-stage 128 x 128 loop on the [30000 x 512 x 5120]-class products, equal or worse on small ones (tools/gemm_ab.py)
      // (column sums of A: weighted ones ride in the 256 x 128 loop, plain ones in the two-stage 128 x 128 loop; anything else is
      //  refused -- vmmt_gemm_colsum_applies() says beforehand which it will be)
      const bool cs_w = epi.colsum_w != nullptr, cs_1 = epi.colsum_out != nullptr && !cs_w;
      if (gl_ok && lds_min == 0 && ((a->tile == 0 && t128 >= 768) || a->tile == 256) && a->K >= 512) {   // (tile = 256: forced, tools / large weight gradients)
        if (cs_1) return VMMT_EINVAL;
        return launch_cfg<T, 256, 128, 64, 64, A_KC, B_KC, 64, true, 3>(a, epi, st);
      }
      if (cs_w) return VMMT_EINVAL;
      // LDS-DMA main loop when its preconditions hold (+10..15 % over the register-staged loop, tools/gemm_ab.py)
      if (gl_ok) return launch_cfg<T, 128, 128, 64, 64, A_KC, B_KC, 64, true, 1>(a, epi, st, lds_min);
      if (cs_1) return VMMT_EINVAL;
      return launch_cfg<T, 128, 128, 64, 64, A_KC, B_KC, 64, true>(a, epi, st, lds_min);
    } else {
      if (epi.colsum_out) return VMMT_EINVAL;
      return launch_cfg<T, 128, 128, 64, 64, A_KC, B_KC>(a, epi, st, lds_min);
    }
  }
  if (epi.colsum_out) return VMMT_EINVAL;
  return launch_cfg<T, 64, 64, 32, 32, A_KC, B_KC>(a, epi, st);
}

// The three operand layouts are independent sets of template instantiations; build.py compiles this file three times
// (-DVMMT_GEMM_PART=0/1/2, one layout each, in parallel: the full set takes ~7 minutes in one translation unit).  Without the
// macro everything lands in one object (tools/exp_build.sh).
#ifndef VMMT_GEMM_PART
#define VMMT_GEMM_PART -1
#endif
int gemm_launch_nt(const vmmt_gemm_args* a, const GemmEpi& epi, hipStream_t st);
int gemm_launch_tn(const vmmt_gemm_args* a, const GemmEpi& epi, hipStream_t st);
int gemm_launch_nn(const vmmt_gemm_args* a, const GemmEpi& epi, hipStream_t st);
int gemm_group_launch_tn(const vmmt_gemm_args* args, const GemmEpi* epis, int n, hipStream_t st);
bool gemm_group_ok_tn(const vmmt_gemm_args* a);

template <bool A_KC, bool B_KC>
static int launch_by_dtype(const vmmt_gemm_args* a, const GemmEpi& epi, hipStream_t st) {
  if (a->dtype == VMMT_F32) return launch_layout<float, A_KC, B_KC>(a, epi, st);
  if (a->dtype == VMMT_BF16) return launch_layout<bf16_t, A_KC, B_KC>(a, epi, st);
  return VMMT_EINVAL;
}
#if VMMT_GEMM_PART == -1 || VMMT_GEMM_PART == 0
int gemm_launch_nt(const vmmt_gemm_args* a, const GemmEpi& epi, hipStream_t st) { return launch_by_dtype<true, true>(a, epi, st); }
#endif
#if VMMT_GEMM_PART == -1 || VMMT_GEMM_PART == 1
int gemm_launch_tn(const vmmt_gemm_args* a, const GemmEpi& epi, hipStream_t st) { return launch_by_dtype<false, false>(a, epi, st); }

// would launch_layout run this TN product on the two-stage 128 x 128 LDS-DMA configuration?  (the grouped kernel is that configuration)
static bool group_member_ok(const vmmt_gemm_args* a) {
  if (a->dtype != VMMT_BF16 || a->layout != VMMT_GEMM_TN || a->scatter_ids || !a->out_f32 || a->act != VMMT_ACT_NONE || a->addend) return false;
  if (a->b_batch_rows || a->colsum_w || a->tile != 0 || a->M <= 0 || a->N <= 0) return false;
  const bool gl_ok = a->K % 64 == 0 && a->K > 0 && a->lda % 8 == 0 && a->ldb % 8 == 0 && a->a_kmod == 0 && a->b_kmod == 0 &&
                     ((((uintptr_t)a->A) | ((uintptr_t)a->B)) & 15) == 0;
  if (!gl_ok || a->split_k < 2) return false;                       // members accumulate with atomics: several may share one C
  const long t128 = (long)((a->M + 127) / 128) * ((a->N + 127) / 128) * a->split_k;
  return !(t128 >= 768 && a->K >= 512);                             // (that one takes the 256 x 128 three-stage configuration)
}

int gemm_group_launch_tn(const vmmt_gemm_args* args, const GemmEpi* epis, int n, hipStream_t st) {
  constexpr int BM = 128, BN = 128, BK = 64;
  GemmGroupArgs ga;
  ga.n = n;
  int start = 0;
  for (int i = 0; i < n; ++i) {
    const vmmt_gemm_args* a = &args[i];
    GemmGroupItem& it = ga.item[i];
    const int tm = (a->M + BM - 1) / BM, tn = (a->N + BN - 1) / BN;
    int splits = a->split_k;
    int kper = ((a->K + splits - 1) / splits + BK - 1) / BK * BK;
    if (kper < BK) kper = BK;
    splits = (a->K + kper - 1) / kper;
    it.A = a->A; it.lda = (long)a->lda; it.B = a->B; it.ldb = (long)a->ldb;
    it.M = a->M; it.N = a->N; it.K = a->K; it.tiles_n = tn; it.ntiles = tm * tn; it.kper = kper;
    it.epi = epis[i];
    it.epi.atomic = 1;
    ga.start[i] = start;
    start += ((tm * tn + 7) & ~7) * splits;
  }
  for (int i = n; i <= VMMT_GEMM_GROUP_MAX; ++i) ga.start[i] = start;
  const size_t smem = (size_t)glds_smem_bytes<BM, BN>();
  static bool allowed = false;
  if (!allowed) {
    (void)hipFuncSetAttribute((const void*)gemm_group_kernel<bf16_t, 128, 128, 64, 64, false, false, 64, true, 1>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    allowed = true;
  }
  hipLaunchKernelGGL((gemm_group_kernel<bf16_t, 128, 128, 64, 64, false, false, 64, true, 1>), dim3(start), dim3(256), smem, st, ga);
  return check_launch();
}
bool gemm_group_ok_tn(const vmmt_gemm_args* a) { return group_member_ok(a); }
#endif
#if VMMT_GEMM_PART == -1 || VMMT_GEMM_PART == 2
int gemm_launch_nn(const vmmt_gemm_args* a, const GemmEpi& epi, hipStream_t st) { return launch_by_dtype<true, false>(a, epi, st); }
#endif

}  // namespace vmmt

#if VMMT_GEMM_PART == -1 || VMMT_GEMM_PART == 0
// 1 when vmmt_gemm would compute the column sums of A for these arguments (bf16, layout TN = A K-strided, K a multiple of 64, aligned
// operands, no k-modulus): WEIGHTED sums (colsum_w) in the 256 x 128 three-stage configuration (>= 768 tiles of 128 x 128, K >= 512,
// no split-K), PLAIN sums (colsum_w == NULL, colsum_out and optionally colsum_out2) in the two-stage 128 x 128 configuration, split-K
// included
extern "C" int vmmt_gemm_colsum_applies(const vmmt_gemm_args* a) {
  if (!a || a->dtype != VMMT_BF16 || a->layout != VMMT_GEMM_TN || a->scatter_ids) return 0;
  const long t128 = (long)((a->M + 127) / 128) * ((a->N + 127) / 128) * (a->split_k > 1 ? a->split_k : 1);
  const bool gl_ok = a->K % 64 == 0 && a->K > 0 && a->lda % 8 == 0 && a->ldb % 8 == 0 && a->a_kmod == 0 && a->b_kmod == 0 &&
                     ((((uintptr_t)a->A) | ((uintptr_t)a->B)) & 15) == 0;
  if (!gl_ok) return 0;
  const bool big = a->tile == 0 && t128 >= 768 && a->K >= 512;     // the 256 x 128 three-stage configuration (launch_layout)
  if (a->colsum_w) return (big && a->split_k <= 1 && !a->colsum_out2) ? 1 : 0;
  // plain sums: the two-stage 128 x 128 LDS-DMA configuration
  const bool t128path = a->tile == 128 || (a->tile == 0 && (t128 >= 192 || (a->K >= 1024 && t128 >= 8)));
  return (t128path && !big) ? 1 : 0;
}

// validates the arguments and fills the kernel-side epilogue description; VMMT_OK, or VMMT_EINVAL
static int gemm_make_epi(const vmmt_gemm_args* a, vmmt::GemmEpi& e) {
  using namespace vmmt;
  if (!a || !a->A || !a->B || !a->C || a->M < 0 || a->N < 0 || a->K < 0) return VMMT_EINVAL;
  if (a->scatter_ids && !a->out_f32) return VMMT_EINVAL;
  if (a->split_k > 1 && (!a->out_f32 || a->act != VMMT_ACT_NONE || a->scatter_ids)) return VMMT_EINVAL;
  e.C = a->C; e.ldc = a->ldc; e.addend = a->addend; e.ld_add = a->ld_add; e.add_rows = a->addend ? a->add_rows : 0;
  e.add_is_T = a->add_is_T; e.act = a->act; e.out_f32 = a->out_f32; e.accumulate = a->accumulate;
  e.alpha = a->alpha; e.scatter_ids = (const long long*)a->scatter_ids; e.pad_id = a->pad_id;
  e.atomic = a->split_k > 1 ? 1 : 0;
  e.b_batch_rows = a->b_batch_rows; e.b_batch_stride = (long)a->b_batch_stride;
  e.colsum_w = a->colsum_w; e.colsum_w_stride = (long)a->colsum_w_stride; e.colsum_out = a->colsum_out; e.colsum_out2 = a->colsum_out2;
  e.rblk = a->c_row_blk; e.rvalid = a->c_row_valid; e.cblk = a->c_col_blk; e.cvalid = a->c_col_valid;
  if (a->a_row_ids) {
    if (a->scatter_ids || a->split_k > 1) return VMMT_EINVAL;
    e.scatter_ids = (const long long*)a->a_row_ids;              // (read as the A operand's row ids by the EPI = 2 instantiation, see GemmEpi)
  }
  if (e.rblk < 0 || e.cblk < 0 || (e.rblk > 0 && (e.rvalid <= 0 || e.rvalid > e.rblk)) || (e.cblk > 0 && (e.cvalid <= 0 || e.cvalid > e.cblk)))
    return VMMT_EINVAL;
  if ((e.rblk | e.cblk) && (!a->out_f32 || a->scatter_ids || a->act != VMMT_ACT_NONE)) return VMMT_EINVAL;
  if (e.rblk && a->colsum_w) return VMMT_EINVAL;                 // the weighted sums (generator bias) have no padded blocks
  if ((a->colsum_w || a->colsum_out || a->colsum_out2) && (!a->colsum_out || !vmmt_gemm_colsum_applies(a))) return VMMT_EINVAL;
  if (a->b_batch_rows < 0 || (a->b_batch_rows > 0 && a->b_batch_rows % 256 != 0)) return VMMT_EINVAL;   // whole tiles (<= 256 rows) per block
  return VMMT_OK;
}

extern "C" int vmmt_gemm(const vmmt_gemm_args* a, void* stream) {
  using namespace vmmt;
  GemmEpi e;
  const int rc = gemm_make_epi(a, e);
  if (rc != VMMT_OK) return rc;
  if (a->M == 0 || a->N == 0) return VMMT_OK;
  hipStream_t st = (hipStream_t)stream;
  switch (a->layout) {
    case VMMT_GEMM_NT: return gemm_launch_nt(a, e, st);
    case VMMT_GEMM_TN: return gemm_launch_tn(a, e, st);
    case VMMT_GEMM_NN: return gemm_launch_nn(a, e, st);
    default: return VMMT_EINVAL;
  }
}

// n independent products, results as n vmmt_gemm calls in any order (members ACCUMULATE into C with atomics: split_k >= 2).  When every
// member is a bf16 TN product of the two-stage 128 x 128 configuration (what the weight-gradient products of the LSTM / attention layers
// are) their tiles go out as ONE grid; otherwise one launch per member.
extern "C" int vmmt_gemm_group(const vmmt_gemm_args* args, int n, void* stream) {
  using namespace vmmt;
  if (!args || n < 0) return VMMT_EINVAL;
  if (n == 0) return VMMT_OK;
  GemmEpi epis[VMMT_GEMM_GROUP_MAX];
  bool grouped = n >= 2 && n <= VMMT_GEMM_GROUP_MAX;
  for (int i = 0; i < n && grouped; ++i) {
    if (gemm_make_epi(&args[i], epis[i]) != VMMT_OK) return VMMT_EINVAL;
    grouped = gemm_group_ok_tn(&args[i]);
  }
  if (grouped) return gemm_group_launch_tn(args, epis, n, (hipStream_t)stream);
  for (int i = 0; i < n; ++i) {
    const int rc = vmmt_gemm(&args[i], stream);
    if (rc != VMMT_OK) return rc;
  }
  return VMMT_OK;
}
extern "C" int vmmt_gemm_group_applies(const vmmt_gemm_args* args, int n) {
  if (!args || n < 2 || n > VMMT_GEMM_GROUP_MAX) return 0;
  for (int i = 0; i < n; ++i)
    if (!vmmt::gemm_group_ok_tn(&args[i])) return 0;
  return 1;
}
#endif

// LDS-DMA GEMM main loop for the bf16 fast path (gfx950).
//
// Differences to gemm_mainloop (common.hpp), which remains the general / fp32 / edge-case path:
//   * operand slabs (BK = 64) go HBM/L2 -> LDS directly with global_load_lds_dwordx4: no VGPR staging, no ds_write, and the
//     instruction stream of a K step shrinks to 2 x (ROWS/8)/NW DMA issues + fragment reads + MFMAs;
//   * the LDS images are LINEAR (an LDS-DMA instruction writes base + lane*16 B), so bank conflicts are removed by an XOR
//     swizzle of the 16-byte chunk index that is applied to the per-lane SOURCE address and again to the fragment reads
//     (K-contiguous image: 128-byte rows, chunk ^= (row >> 1) & 7; K-strided image: 256-byte k-rows, chunk ^= 4 * (k & 3),
//     which spreads the four k-rows of a ds_read_b64_tr_b16 block over all 64 banks);
//   * every fragment address is a per-lane constant + immediate: no address arithmetic inside the loop;
//   * two LDS buffers, the loop is unrolled by two so that buffer offsets are immediates; hipcc drains vmcnt(0) at the
//     workgroup barrier, i.e. slab t+1 is in flight while slab t is multiplied.
// Preconditions (checked by the callers, otherwise they use gemm_mainloop): bf16, base pointers 16-byte aligned, leading
// dimensions multiples of 8 elements, K a multiple of 64, every K index < K readable for all tile rows (callers round K up to
// 64 over zero-padded buffers), no k-modulus.  Rows outside [0, limit) are CLAMPED, not masked: they only feed output elements
// that the epilogue never stores.
#pragma once
#include "common.hpp"

namespace vmmt {

typedef __attribute__((address_space(3))) void g_lds_void_t;
typedef __attribute__((address_space(1))) const void g_glb_cvoid_t;

constexpr int GBK = 64;   // K slab

// stages one operand tile per call: ROWS tile rows (KC: ROWS x 64 elements, KS: 64 k-rows x ROWS elements) = ROWS/8 pieces of 1 KiB
template <int ROWS, bool KC, int NW>
struct GldsOperand {
  static constexpr int NP = ROWS / 8;            // 1-KiB pieces per slab
  static constexpr int PER = NP / NW;            // pieces per wave
  static_assert(NP % NW == 0, "pieces must divide over the waves");
  static constexpr int BYTES = ROWS * GBK * 2;
  const char* src[PER];                          // per-lane source address of each of this wave's pieces (advanced every slab)
  long step;                                     // bytes per K slab

  // GATHER (K-contiguous operands): tile row r is row ids[row0 + r] of the table at P -- a gathered operand (vmmt_gemm_args.a_row_ids).  A
  // template parameter, not a null check: the check alone cost the plain product 16 registers
  template <bool GATHER = false>
  __device__ __forceinline__ void init(const bf16_t* P, long ld, int row0, int limit, int k0, int wave, int lane, const long long* ids = nullptr) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int piece = wave * PER + j;
      if constexpr (KC) {
        const int row = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        int g = row0 + row;
        g = g < limit ? g : limit - 1;
        if constexpr (GATHER) src[j] = reinterpret_cast<const char*>(P + (long)ids[g] * ld + k0 + chunk * 8);
        else
        src[j] = reinterpret_cast<const char*>(P + (long)g * ld + k0 + chunk * 8);
      } else {
        // ROWS > 128: side-by-side sub-images of [64 k-rows][128 columns] (16 KiB each), 16 pieces per sub-image
        const int sub = piece >> 4;
        const int krow = (piece & 15) * 4 + (lane >> 4);
        const int chunk = (lane & 15) ^ ((krow & 3) * 4);
        int col = row0 + sub * 128 + chunk * 8;
        const int cmax = ((limit - 1) / 8) * 8;
        col = col < cmax ? col : cmax;           // last chunk that holds a valid column; clamped chunks only feed unstored outputs
        src[j] = reinterpret_cast<const char*>(P + (long)(k0 + krow) * ld + col);
      }
    }
    step = KC ? (long)GBK * 2 : (long)GBK * ld * 2;
  }
  __device__ __forceinline__ void issue(char* lds, int wave) {
#if defined(VMMT_EXP_NODMA)          // experiment: no operand traffic
    return;
#endif
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      char* dst = lds + (wave * PER + j) * 1024;                  // wave-uniform; hardware adds lane * 16
      __builtin_amdgcn_global_load_lds((g_glb_cvoid_t*)src[j], (g_lds_void_t*)dst, 16, 0, 0);
      src[j] += step;
    }
  }
};

// per-lane fragment base offsets (bytes) into an operand image; tile t of the wave adds an immediate
template <bool KC, int NT_>   // NT_: number of 32-row tiles of this operand per wave
struct GldsFrag {
  int off[KC ? 4 : NT_];
  __device__ __forceinline__ void init(const int (&toff)[NT_], int lane) {
    if constexpr (KC) {
      // image [row][8 chunks of 16 B]; lane (r, h) reads chunk c = 2 ks + h of row toff + r at chunk ^ ((row >> 1) & 7)
      const int r = lane & 31, h = lane >> 5, s = (r >> 1) & 7;      // toff multiples of 32 do not change the swizzle key
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) off[ks] = r * 128 + (((2 * ks + h) ^ s) * 16);
    } else {
      // image [k-row][16 chunks of 16 B]; transposed read of a 4(k) x 16(row) block per 16-lane group
      const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
#pragma unroll
      for (int t = 0; t < NT_; ++t) {
        const int chunk = ((toff[t] & 127) >> 3) + 2 * (g & 1) + (p >> 1);
        off[t] = (toff[t] >> 7) * 16384 + (8 * (g >> 1) + q) * 256 + ((chunk ^ (q * 4)) * 16) + (p & 1) * 8;
      }
    }
  }
};

typedef short gs16x4 __attribute__((ext_vector_type(4)));
typedef short gs16x8 __attribute__((ext_vector_type(8)));

template <bool KC, int NT_>
__device__ __forceinline__ bf16x8 glds_frag(const char* img, const GldsFrag<KC, NT_>& f, const int (&toff)[NT_], int t, int ks) {
#if defined(VMMT_EXP_NOREAD)         // experiment: no LDS fragment reads
  gs16x8 c = {(short)ks, (short)t, 1, 2, 3, 4, 5, (short)f.off[0]};
  return __builtin_bit_cast(bf16x8, c);
#endif
  if constexpr (KC) {
    return *reinterpret_cast<const bf16x8*>(img + f.off[ks] + toff[t] * 128);
  } else {
    typedef __attribute__((address_space(3))) gs16x4 lds_v;
    const char* a0 = img + f.off[t] + ks * 16 * 256;
    gs16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(a0));
    gs16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(a0 + 4 * 256));
    gs16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  }
}

template <int BM, int BN, bool A_KC, bool B_KC, int TI, int TJ>
__device__ __forceinline__ void glds_slab(const char* As, const char* Bs, const GldsFrag<A_KC, TI>& fa, const GldsFrag<B_KC, TJ>& fb,
                                          const int (&aoff)[TI], const int (&boff)[TJ], f32x16 (&acc)[TI][TJ]) {
  // fragments of K-step ks+1 are requested before the MFMAs of K-step ks (register double buffer)
  bf16x8 a[2][TI], b[2][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i) a[0][i] = glds_frag<A_KC, TI>(As, fa, aoff, i, 0);
#pragma unroll
  for (int j = 0; j < TJ; ++j) b[0][j] = glds_frag<B_KC, TJ>(Bs, fb, boff, j, 0);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if (ks < 3) {
#pragma unroll
      for (int i = 0; i < TI; ++i) a[(ks + 1) & 1][i] = glds_frag<A_KC, TI>(As, fa, aoff, i, ks + 1);
#pragma unroll
      for (int j = 0; j < TJ; ++j) b[(ks + 1) & 1][j] = glds_frag<B_KC, TJ>(Bs, fb, boff, j, ks + 1);
      __builtin_amdgcn_sched_barrier(0);   // keep the requests ahead of the MFMAs (hipcc otherwise re-serialises them)
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
#if defined(VMMT_EXP_NOMFMA)       // experiment: fragment reads only (results kept alive by a cheap xor into the accumulator)
        acc[i][j][0] += __builtin_bit_cast(f32x4, a[ks & 1][i])[0] + __builtin_bit_cast(f32x4, b[ks & 1][j])[0];
#else
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][i], b[ks & 1][j], acc[i][j], 0, 0, 0);
#endif
      }
  }
}

template <int BM, int BN, bool DBUF = true>
constexpr int glds_smem_bytes() { return (DBUF ? 2 : 1) * (BM + BN) * GBK * 2; }

typedef unsigned gu32x2 __attribute__((ext_vector_type(2)));
// Plain column sums of a K-strided 128-column A slab, 4 waves (the bias gradient db = sum over tokens of dgates next to dW_hh =
// dgates^T h: the dgates would otherwise be read once more by a kernel of their own): wave w owns k-rows 16 w .. 16 w + 15, one
// 8-byte read covers columns 4 (l & 31) .. + 3 of k-row 2 i + (l >> 5).  Inline assembly for the reason given at glds_colsum_slab.
__device__ __forceinline__ void glds_colsum128_slab(const char* As, int wave, int lane, float (&cs)[4]) {
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)As + (unsigned)((lane & 1) * 8);
  const int ch = (lane >> 1) & 15, par = lane >> 5;
  gu32x2 vv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int kr = wave * 16 + 2 * i + par;                       // (kr & 3) = (2 i + par) & 3
    asm volatile("ds_read_b64 %0, %1" : "=v"(vv[i]) : "v"(base + (unsigned)(kr * 256 + ((ch ^ ((kr & 3) * 4)) * 16))) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+v"(vv[4]), "+v"(vv[5]), "+v"(vv[6]), "+v"(vv[7]));
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      cs[2 * e] += __uint_as_float(vv[i][e] << 16);
      cs[2 * e + 1] += __uint_as_float(vv[i][e] & 0xffff0000u);
    }
}

// C_tile += A_tile * B_tile^T over k in [kbeg, kend), kend - kbeg a positive multiple of 64.
template <int BM, int BN, int NW, bool A_KC, bool B_KC, int TI, int TJ, bool DBUF = true, bool GATHER_A = false>
__device__ __forceinline__ void gemm_mainloop_glds(const bf16_t* __restrict__ A, long lda, int m0, int M, const bf16_t* __restrict__ B,
                                                   long ldb, int n0, int N, int kbeg, int kend, const int (&aoff)[TI],
                                                   const int (&boff)[TJ], f32x16 (&acc)[TI][TJ], char* __restrict__ smem,
                                                   float* __restrict__ colsum_out = nullptr, float* __restrict__ colsum_out2 = nullptr,
                                                   int cs_blk = 0, int cs_valid = 0, const long long* __restrict__ a_ids = nullptr) {
  constexpr int ABYTES = BM * GBK * 2, BBYTES = BN * GBK * 2, BUF = ABYTES + BBYTES;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  constexpr bool CS_OK = !A_KC && BM == 128 && NW == 4 && DBUF;   // the shape glds_colsum128_slab is written for
  const bool cs_on = CS_OK && colsum_out != nullptr;
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  GldsOperand<BM, A_KC, NW> ga;
  GldsOperand<BN, B_KC, NW> gb;
  ga.template init<GATHER_A && A_KC>(A, lda, m0, M, kbeg, wave, lane, a_ids);
  gb.init(B, ldb, n0, N, kbeg, wave, lane);
  GldsFrag<A_KC, TI> fa;
  GldsFrag<B_KC, TJ> fb;
  fa.init(aoff, lane);
  fb.init(boff, lane);
  const int nslab = (kend - kbeg) / GBK;
  if (nslab <= 0) return;
  if constexpr (!DBUF) {   // one buffer: short K, several workgroups per CU overlap each other's load and multiply phases
    for (int s = 0; s < nslab; ++s) {
      ga.issue(smem, wave);
      gb.issue(smem + ABYTES, wave);
      __syncthreads();
      glds_slab<BM, BN, A_KC, B_KC, TI, TJ>(smem, smem + ABYTES, fa, fb, aoff, boff, acc);
      __syncthreads();
    }
    return;
  }
  ga.issue(smem, wave);
  gb.issue(smem + ABYTES, wave);
  __syncthreads();
  int s = 0;
  for (; s + 2 <= nslab; s += 2) {
    // slab s in buffer 0, slab s+1 into buffer 1
    ga.issue(smem + BUF, wave);
    gb.issue(smem + BUF + ABYTES, wave);
    if (cs_on) glds_colsum128_slab(smem, wave, lane, csum);
    glds_slab<BM, BN, A_KC, B_KC, TI, TJ>(smem, smem + ABYTES, fa, fb, aoff, boff, acc);
    __syncthreads();
    if (s + 2 < nslab) {
      ga.issue(smem, wave);
      gb.issue(smem + ABYTES, wave);
    }
    if (cs_on) glds_colsum128_slab(smem + BUF, wave, lane, csum);
    glds_slab<BM, BN, A_KC, B_KC, TI, TJ>(smem + BUF, smem + BUF + ABYTES, fa, fb, aoff, boff, acc);
    __syncthreads();
  }
  if (s < nslab) {   // odd count: the last slab sits in buffer 0
    if (cs_on) glds_colsum128_slab(smem, wave, lane, csum);
    glds_slab<BM, BN, A_KC, B_KC, TI, TJ>(smem, smem + ABYTES, fa, fb, aoff, boff, acc);
    __syncthreads();
  }
  if (cs_on) {       // lanes l and l + 32 hold the two k-row parities of the same four columns
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = csum[e] + __shfl_xor(csum[e], 32, 64);
      int col = m0 + (lane & 31) * 4 + e;
      bool ok = lane < 32 && col < M;
      if (cs_blk) { const int q = col / cs_blk, r = col - q * cs_blk; ok = ok && r < cs_valid; col = q * cs_valid + r; }   // padded gate blocks
      if (ok) {
        atomicAdd(colsum_out + col, t);
        if (colsum_out2) atomicAdd(colsum_out2 + col, t);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Three-stage variant: two slabs stay in flight across the workgroup barrier.  `__syncthreads()` would drain them (an
// LDS-DMA is a pending LDS write on the VM counter, so its fence emits vmcnt(0)); here every wave waits with a COUNTED
// s_waitcnt for its own pieces of the slab that is needed next and the workgroup meets at a raw s_barrier.  One barrier per
// slab: it also proves that every wave has finished the multiply of the previous slab, whose buffer is refilled right after.
template <int N>
__device__ __forceinline__ void glds_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, int BN>
constexpr int glds3_smem_bytes() { return 3 * (BM + BN) * GBK * 2; }

// Weighted column sums of a K-strided A slab that sits in LDS anyway: csum[col] += sum_k A[k][col] * w[k] over the 64 k-rows of the
// slab (the generator's bias gradient db[v] = sum_m P[m][v] c[m] next to dWg = P^T O': a second pass over the 307 MB of P
// otherwise).  256 columns, 8 waves: wave w owns k-rows 8 w .. 8 w + 7, whose weights arrive by ONE scalar load (SMEM does not touch
// the VM counter the LDS-DMA pipeline is counted on) and stay wave-uniform: lane l reads 8 bytes = columns 4 l .. 4 l + 3 of one k-row
// per instruction (64 lanes = the whole 512-byte row).
typedef const float __attribute__((address_space(4))) * GldsConstF;
__device__ __forceinline__ void glds_colsum_slab(const char* As, const float* __restrict__ w, int wave, int lane, float (&cs)[4]) {
  float wk[8];
#if defined(__HIP_DEVICE_COMPILE__)
  GldsConstF wp = reinterpret_cast<GldsConstF>(reinterpret_cast<uintptr_t>(w + wave * 8));
#pragma unroll
  for (int i = 0; i < 8; ++i) wk[i] = wp[i];
#else
  for (int i = 0; i < 8; ++i) wk[i] = w[wave * 8 + i];
#endif
  // the reads are inline assembly: as plain loads the compiler orders them behind the LDS-DMA of the slab just requested (s_waitcnt
  // vmcnt(0) in front of their first use), which would drain the three-stage pipeline in every workgroup that carries the sums
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)As + (unsigned)((lane >> 5) * 16384 + (lane & 1) * 8);
  const int ch = (lane >> 1) & 15;
  gu32x2 vv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int kr = wave * 8 + i;                                  // (kr & 3) == (i & 3): the swizzle key is a compile-time constant
    asm volatile("ds_read_b64 %0, %1" : "=v"(vv[i]) : "v"(base + (unsigned)(kr * 256) + (unsigned)((ch ^ ((i & 3) * 4)) * 16)) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+v"(vv[4]), "+v"(vv[5]), "+v"(vv[6]), "+v"(vv[7]));
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      cs[2 * e] = __builtin_fmaf(__uint_as_float(vv[i][e] << 16), wk[i], cs[2 * e]);
      cs[2 * e + 1] = __builtin_fmaf(__uint_as_float(vv[i][e] & 0xffff0000u), wk[i], cs[2 * e + 1]);
    }
  }
}

template <int BM, int BN, int NW, bool A_KC, bool B_KC, int TI, int TJ>
__device__ __forceinline__ void gemm_mainloop_glds3(const bf16_t* __restrict__ A, long lda, int m0, int M, const bf16_t* __restrict__ B,
                                                    long ldb, int n0, int N, int kbeg, int kend, const int (&aoff)[TI],
                                                    const int (&boff)[TJ], f32x16 (&acc)[TI][TJ], char* __restrict__ smem,
                                                    const float* __restrict__ colsum_w = nullptr, float* __restrict__ colsum_out = nullptr,
                                                    int cs_phase = 0, int cs_mod = 1) {
  // (the weighted column sums of the A slab are shared out over the workgroups that read it -- the cs_mod column tiles of one row slab:
  //  the one with column tile t takes the K slabs s % cs_mod == t.  All of them in the first column tile made that workgroup fall behind
  //  the others of its slab, which then miss the slab in L2: 661 MB fetched for 312 MB of operands at the generator's dWg, round 4)
  constexpr int ABYTES = BM * GBK * 2, BBYTES = BN * GBK * 2, BUF = ABYTES + BBYTES;
  constexpr bool CS_OK = !A_KC && BM == 256 && NW == 8;          // the only shape glds_colsum_slab is written for
  const bool cs_on = CS_OK && colsum_w != nullptr;
  int cs_left = cs_phase;                                      // slabs until this workgroup's next turn
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  using GA = GldsOperand<BM, A_KC, NW>;
  using GB = GldsOperand<BN, B_KC, NW>;
  constexpr int PW = GA::PER + GB::PER;           // LDS-DMA instructions per wave and slab
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  GA ga;
  GB gb;
  ga.init(A, lda, m0, M, kbeg, wave, lane);
  gb.init(B, ldb, n0, N, kbeg, wave, lane);
  GldsFrag<A_KC, TI> fa;
  GldsFrag<B_KC, TJ> fb;
  fa.init(aoff, lane);
  fb.init(boff, lane);
  const int nslab = (kend - kbeg) / GBK;
  if (nslab <= 0) return;
  ga.issue(smem, wave);
  gb.issue(smem + ABYTES, wave);
  if (nslab > 1) {
    ga.issue(smem + BUF, wave);
    gb.issue(smem + BUF + ABYTES, wave);
  }
#define VMMT_GLDS3_STEP(CUR, NXT2)                                                                     \
  {                                                                                                    \
    if (s + 1 < nslab) glds_wait_vm<PW>(); else glds_wait_vm<0>();                                     \
    __builtin_amdgcn_s_barrier();                                                                      \
    if (s + 2 < nslab) {                                                                               \
      ga.issue(smem + (NXT2) * BUF, wave);                                                             \
      gb.issue(smem + (NXT2) * BUF + ABYTES, wave);                                                    \
    }                                                                                                  \
    if (cs_on) {                                                                                       \
      if (cs_left == 0) { glds_colsum_slab(smem + (CUR) * BUF, colsum_w + kbeg + s * GBK, wave, lane, csum); cs_left = cs_mod; } \
      --cs_left;                                                                                       \
    }                                                                                                  \
    glds_slab<BM, BN, A_KC, B_KC, TI, TJ>(smem + (CUR) * BUF, smem + (CUR) * BUF + ABYTES, fa, fb, aoff, boff, acc); \
    ++s;                                                                                               \
  }
  int s = 0;
  while (s < nslab) {
    VMMT_GLDS3_STEP(0, 2)
    if (s >= nslab) break;
    VMMT_GLDS3_STEP(1, 0)
    if (s >= nslab) break;
    VMMT_GLDS3_STEP(2, 1)
  }
#undef VMMT_GLDS3_STEP
  __builtin_amdgcn_s_barrier();      // the staging buffers may be reused by the caller's epilogue
  if (cs_on) {                       // every wave holds the sums of its eight k-rows per slab for all 256 columns: one atomic per wave and column
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = m0 + lane * 4 + e;
      if (col < M) atomicAdd(colsum_out + col, csum[e]);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Half-depth variant for K-contiguous operands: K slabs of 32, LDS images with 64-byte rows (4 chunks of 16 B, chunk index
// XOR (row >> 2) & 3 on the source side and in the fragment reads: the 16 lanes of a ds_read_b128 pass cover all 64 banks),
// three stages with counted waits.  A 128 x 128 tile needs 3 x 16 KiB = 48 KiB: THREE workgroups of 4 waves share a CU, so
// that one tile's epilogue and barriers overlap the others' MFMAs (profiles/r1_gen_kernel_investigation.txt).  ONE loop body
// with a scalar stage index: three macro-expanded bodies made hipcc spill the accumulators.
constexpr int HBK = 32;

template <int ROWS, int NW>
struct HalfOperand {
  static constexpr int NP = ROWS / 16, PER = NP / NW;        // 1-KiB pieces (16 rows) per slab / per wave
  static_assert(NP % NW == 0, "pieces must divide over the waves");
  const char* src[PER];
  __device__ __forceinline__ void init(const bf16_t* P, long ld, int row0, int limit, int k0, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int row = (wave * PER + j) * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      int g = row0 + row;
      g = g < limit ? g : limit - 1;                         // clamped rows only feed outputs that are never stored
      src[j] = reinterpret_cast<const char*>(P + (long)g * ld + k0 + chunk * 8);
    }
  }
  __device__ __forceinline__ void issue(char* lds, int wave) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      __builtin_amdgcn_global_load_lds((g_glb_cvoid_t*)src[j], (g_lds_void_t*)(lds + (wave * PER + j) * 1024), 16, 0, 0);
      src[j] += HBK * 2;
    }
  }
  __device__ __forceinline__ void issue_piece(char* lds, int wave, int j) {      // (j a compile-time constant after unrolling)
    __builtin_amdgcn_global_load_lds((g_glb_cvoid_t*)src[j], (g_lds_void_t*)(lds + (wave * PER + j) * 1024), 16, 0, 0);
    src[j] += HBK * 2;
  }
};

template <int BM, int BN, int NS = 3>
constexpr int hglds3_smem_bytes() { return NS * (BM + BN) * HBK * 2; }

// C_tile += A_tile * B_tile^T over k in [kbeg, kend), kend - kbeg a positive multiple of 32; both operands K-contiguous.
template <int BM, int BN, int NW, int TI, int TJ, int NS = 3>
__device__ __forceinline__ void gemm_mainloop_hglds3(const bf16_t* __restrict__ A, long lda, int m0, int M, const bf16_t* __restrict__ B,
                                                     long ldb, int n0, int N, int kbeg, int kend, const int (&aoff)[TI],
                                                     const int (&boff)[TJ], f32x16 (&acc)[TI][TJ], char* __restrict__ smem) {
  constexpr int ABYTES = BM * HBK * 2, BBYTES = BN * HBK * 2, BUF = ABYTES + BBYTES;
  using GA = HalfOperand<BM, NW>;
  using GB = HalfOperand<BN, NW>;
  constexpr int PW = GA::PER + GB::PER;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  GA ga;
  GB gb;
  ga.init(A, lda, m0, M, kbeg, wave, lane);
  gb.init(B, ldb, n0, N, kbeg, wave, lane);
  int foff[2];          // lane (r, h) reads logical chunk 2 ks + h of row toff + r (toff a multiple of 32: same swizzle key)
  {
    const int r = lane & 31, h = lane >> 5, sw = (r >> 2) & 3;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = r * 64 + (((2 * ks + h) ^ sw) * 16);
  }
  const int nslab = (kend - kbeg) / HBK;
  if (nslab <= 0) return;
  // NS stages: NS - 1 slabs in flight (the slab of step s + NS - 1 goes into the buffer step s - 1 has just left)
#pragma unroll
  for (int q = 0; q < NS - 1; ++q)
    if (q < nslab) {
      ga.issue(smem + q * BUF, wave);
      gb.issue(smem + q * BUF + ABYTES, wave);
    }
  int cur = 0;
  for (int s = 0; s < nslab; ++s) {
    const int ahead = nslab - 1 - s;             // slabs requested behind this one
    if constexpr (NS == 4) {
      if (ahead >= 2) glds_wait_vm<2 * PW>(); else if (ahead == 1) glds_wait_vm<PW>(); else glds_wait_vm<0>();
    } else {
      if (ahead >= 1) glds_wait_vm<PW>(); else glds_wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (s + NS - 1 < nslab) {
      const int nb = cur == 0 ? NS - 1 : cur - 1;
      ga.issue(smem + nb * BUF, wave);
      gb.issue(smem + nb * BUF + ABYTES, wave);
    }
    const char* As = smem + cur * BUF;
    const char* Bs = As + ABYTES;
    // fragments of K-step 1 are requested before the MFMAs of K-step 0 (register double buffer, as glds_slab)
    bf16x8 fa_[2][TI], fb_[2][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) fa_[0][i] = *reinterpret_cast<const bf16x8*>(As + foff[0] + aoff[i] * 64);
#pragma unroll
    for (int j = 0; j < TJ; ++j) fb_[0][j] = *reinterpret_cast<const bf16x8*>(Bs + foff[0] + boff[j] * 64);
#pragma unroll
    for (int i = 0; i < TI; ++i) fa_[1][i] = *reinterpret_cast<const bf16x8*>(As + foff[1] + aoff[i] * 64);
#pragma unroll
    for (int j = 0; j < TJ; ++j) fb_[1][j] = *reinterpret_cast<const bf16x8*>(Bs + foff[1] + boff[j] * 64);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[ks][i], fb_[ks][j], acc[i][j], 0, 0, 0);
    cur = cur == NS - 1 ? 0 : cur + 1;
  }
  __builtin_amdgcn_s_barrier();      // the staging buffers may be reused by the caller's epilogue
}

#if defined(VMMT_EXP_TILE512)
// HalfOperand through the BUFFER form of the LDS-DMA (buffer_load_dwordx4 ... offen lds, what hipBLASLt's kernels use): the K position is a
// scalar offset that advances once per slab -- no 64-bit VALU add per request, half the address registers
template <int ROWS, int NW>
struct HalfOperandBuf {
  static constexpr int NP = ROWS / 16, PER = NP / NW;
  static_assert(NP % NW == 0, "pieces must divide over the waves");
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff[PER];
  int soff;
  __device__ __forceinline__ void init(const bf16_t* P, long ld, int row0, int limit, int k0, int wave, int lane) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(P), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int row = (wave * PER + j) * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      int g = row0 + row;
      g = g < limit ? g : limit - 1;
      voff[j] = (unsigned)((long)g * ld * 2 + chunk * 16);
    }
    soff = k0 * 2;
  }
  __device__ __forceinline__ void issue_piece(char* lds, int wave, int j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (g_lds_void_t*)(lds + (wave * PER + j) * 1024), 16, voff[j], soff, 0, 0);
    if (j == PER - 1) soff += HBK * 2;
  }
  __device__ __forceinline__ void issue(char* lds, int wave) {
#pragma unroll
    for (int j = 0; j < PER; ++j) issue_piece(lds, wave, j);
  }
};

// Probe: one wave per SIMD (4 waves of 128 x 128 on a 256 x 256 tile), four 32-deep stages, ONE loop body with a run-time stage index.  The
// fragment reads are inline assembly with counted lgkmcnt waits (hipcc does not see them: no s_waitcnt vmcnt(0) in front of reads it cannot tell
// from the LDS-DMA's destinations, no lgkmcnt(0) per group), requested one K-step ahead ACROSS slabs and issued between the MFMAs.
typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pipe_read(gu32x4& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
template <int N>
__device__ __forceinline__ void pipe_wait_lgkm(gu32x4 (&a)[4], gu32x4 (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
}
template <int BM, int BN, int NW, int TI, int TJ>
__device__ __forceinline__ void gemm_mainloop_hglds_pipe(const bf16_t* __restrict__ A, long lda, int m0, int M, const bf16_t* __restrict__ B,
                                                         long ldb, int n0, int N, int kbeg, int kend, const int (&aoff)[TI],
                                                         const int (&boff)[TJ], f32x16 (&acc)[TI][TJ], char* __restrict__ smem) {
  static_assert(TI == 4 && TJ == 4, "written for 128 x 128 per wave");
  constexpr int NS = 4;
  constexpr int ABYTES = BM * HBK * 2, BBYTES = BN * HBK * 2, BUF = ABYTES + BBYTES;
  using GA = HalfOperandBuf<BM, NW>;
  using GB = HalfOperandBuf<BN, NW>;
  constexpr int PW = GA::PER + GB::PER;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  GA ga;
  GB gb;
  ga.init(A, lda, m0, M, kbeg, wave, lane);
  gb.init(B, ldb, n0, N, kbeg, wave, lane);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  unsigned fo[2];                          // byte offset of this lane's 16 bytes in a 32-row block, K-step 0 / 1 (swizzled as HalfOperand fills it)
  {
    const int r = lane & 31, h = lane >> 5, sw = (r >> 2) & 3;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fo[ks] = (unsigned)(r * 64 + (((2 * ks + h) ^ sw) * 16));
  }
  unsigned ao[4], bo[4];                   // + this wave's four row blocks of each operand
#pragma unroll
  for (int i = 0; i < 4; ++i) { ao[i] = (unsigned)(aoff[i] * 64); bo[i] = (unsigned)(ABYTES + boff[i] * 64); }
  const int nslab = (kend - kbeg) / HBK;
  if (nslab <= 0) return;
#pragma unroll
  for (int q = 0; q < NS - 1; ++q)
    if (q < nslab) {
      ga.issue(smem + q * BUF, wave);
      gb.issue(smem + q * BUF + ABYTES, wave);
    }
  if (nslab >= 3) glds_wait_vm<2 * PW>(); else if (nslab == 2) glds_wait_vm<PW>(); else glds_wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  gu32x4 f0a[4], f0b[4], f1a[4], f1b[4];
  // fragment addresses: one register per operand and K-step + an immediate per 32-row block (2 KiB apart): no VALU add per read
  const unsigned wa = lds0 + ao[0], wb = lds0 + bo[0];
#define PIPE_READ1(DST, ADDR, I) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"((I) * 2048) : "memory")
#pragma unroll
  for (int i = 0; i < 4; ++i) { PIPE_READ1(f0a[i], wa + fo[0], i); PIPE_READ1(f0b[i], wb + fo[0], i); }
  int cur = 0;
  for (int s = 0; s < nslab; ++s) {
    const unsigned off = (unsigned)(cur * BUF);
    const unsigned a1 = wa + off + fo[1], b1 = wb + off + fo[1];
    const int nxt = cur == NS - 1 ? 0 : cur + 1;
    // ---- K-step 0 of slab s: its fragments were requested one K-step ago; the requests of K-step 1 go out between its MFMAs, and so does the
    //      B half of slab s + 2's LDS-DMA (its A half went out between the MFMAs of the previous K-step: one request per four MFMAs --
    //      a burst of eight behind the barrier keeps this wave, the SIMD's only one, from issuing MFMAs while the requests queue up)
    pipe_wait_lgkm<0>(f0a, f0b);
    const bool fillb = s >= 1 && s + NS - 2 < nslab;
    char* const pb = smem + (cur >= 2 ? cur - 2 : cur + 2) * BUF + ABYTES;      // buffer of slab s + 2 = (cur + 2) % 4
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f0a[i]), __builtin_bit_cast(bf16x8, f0b[j]), acc[i][j], 0, 0, 0);
      if (fillb) gb.issue_piece(pb, wave, i);
      PIPE_READ1(f1a[i], a1, i);
      PIPE_READ1(f1b[i], b1, i);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- slab s + 1 must have landed in every wave's share (only slab s + 2 may still be on its way); behind the barrier every wave is past its
    //      last read of slab s - 1, whose buffer takes slab s + 3
    //      (outstanding at this point, in issue order: slab s + 1, then slab s + 2 whole -- its B half went out just now)
    if (s + 2 < nslab) glds_wait_vm<PW>(); else glds_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    // ---- K-step 1 of slab s, the requests of slab s + 1 / K-step 0 and the A half of slab s + 3's LDS-DMA between its MFMAs
    pipe_wait_lgkm<0>(f1a, f1b);
    const unsigned noff = (unsigned)(nxt * BUF);
    const unsigned a0 = wa + noff + fo[0], b0 = wb + noff + fo[0];
    const bool more = s + 1 < nslab;
    const bool filla = s + NS - 1 < nslab;
    char* const pa = smem + (cur == 0 ? NS - 1 : cur - 1) * BUF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f1a[i]), __builtin_bit_cast(bf16x8, f1b[j]), acc[i][j], 0, 0, 0);
      if (filla) ga.issue_piece(pa, wave, i);
      if (more) {
        PIPE_READ1(f0a[i], a0, i);
        PIPE_READ1(f0b[i], b0, i);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    cur = nxt;
  }
#undef PIPE_READ1
  __builtin_amdgcn_s_barrier();
}

// Probe 520: the same pipeline for K-STRIDED operands (TN: A stored [K][M], B [K][N] -- dWg's layout).  Image of a 32-deep slab of one operand:
// two sub-images of [32 k-rows][128 columns] (256 B per k-row, 16 chunks of 16 B, chunk ^= (k-row & 3) * 4 on the source side as in GldsOperand),
// fragments by two ds_read_b64_tr_b16 each (GldsFrag's lane mapping).
template <int COLS, int NW>
struct HalfOperandBufT {
  static constexpr int NP = COLS / 16, PER = NP / NW;           // 1-KiB pieces (4 k-rows x 128 columns) per slab / per wave
  static_assert(NP % NW == 0 && COLS % 128 == 0, "pieces must divide over the waves");
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff[PER];
  int soff, sstep;
  __device__ __forceinline__ void init(const bf16_t* P, long ld, int col0, int limit, int k0, int wave, int lane) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(P), 0, 0x7fffffff, 0x00020000);
    const int cmax = ((limit - 1) / 8) * 8;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int pc = wave * PER + j, sub = pc >> 3, kq = pc & 7;
      const int kr = lane >> 4, ch = lane & 15;
      const int krow = kq * 4 + kr;
      int col = col0 + sub * 128 + ((ch ^ (kr * 4)) * 8);
      col = col < cmax ? col : cmax;
      voff[j] = (unsigned)(((long)krow * ld + col) * 2);
    }
    soff = (int)((long)k0 * ld * 2);
    sstep = (int)(ld * HBK * 2);
  }
  __device__ __forceinline__ void issue_piece(char* lds, int wave, int j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (g_lds_void_t*)(lds + (wave * PER + j) * 1024), 16, voff[j], soff, 0, 0);
    if (j == PER - 1) soff += sstep;
  }
  __device__ __forceinline__ void issue(char* lds, int wave) {
#pragma unroll
    for (int j = 0; j < PER; ++j) issue_piece(lds, wave, j);
  }
};
typedef unsigned gu32x2_ __attribute__((ext_vector_type(2)));
template <int BM, int BN, int NW, int TI, int TJ>
__device__ __forceinline__ void gemm_mainloop_hglds_pipe_t(const bf16_t* __restrict__ A, long lda, int m0, int M, const bf16_t* __restrict__ B,
                                                           long ldb, int n0, int N, int kbeg, int kend, const int (&aoff)[TI],
                                                           const int (&boff)[TJ], f32x16 (&acc)[TI][TJ], char* __restrict__ smem) {
  static_assert(TI == 4 && TJ == 4, "written for 128 x 128 per wave");
  constexpr int NS = 4;
  constexpr int ABYTES = BM * HBK * 2, BBYTES = BN * HBK * 2, BUF = ABYTES + BBYTES;
  using GA = HalfOperandBufT<BM, NW>;
  using GB = HalfOperandBufT<BN, NW>;
  constexpr int PW = GA::PER + GB::PER;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  GA ga;
  GB gb;
  ga.init(A, lda, m0, M, kbeg, wave, lane);
  gb.init(B, ldb, n0, N, kbeg, wave, lane);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  unsigned ta[4], tb[4];                   // this lane's byte offset for 32-column block t of this wave's sub-image (K-step and half: immediates)
  {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int chunk = t * 4 + 2 * (g & 1) + (pp >> 1);
      const unsigned o = (unsigned)((8 * (g >> 1) + q) * 256 + ((chunk ^ (q * 4)) * 16) + (pp & 1) * 8);
      ta[t] = lds0 + (unsigned)((aoff[0] >> 7) * 8192) + o;
      tb[t] = lds0 + (unsigned)(ABYTES + (boff[0] >> 7) * 8192) + o;
    }
  }
  const int nslab = (kend - kbeg) / HBK;
  if (nslab <= 0) return;
#pragma unroll
  for (int q = 0; q < NS - 1; ++q)
    if (q < nslab) {
      ga.issue(smem + q * BUF, wave);
      gb.issue(smem + q * BUF + ABYTES, wave);
    }
  if (nslab >= 3) glds_wait_vm<2 * PW>(); else if (nslab == 2) glds_wait_vm<PW>(); else glds_wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  gu32x2_ f0a[4][2], f0b[4][2], f1a[4][2], f1b[4][2];
#define PIPET_READ(DST, ADDR, KS)                                                                                          \
  {                                                                                                                        \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST[0]) : "v"(ADDR), "n"((KS) * 4096) : "memory");           \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST[1]) : "v"(ADDR), "n"((KS) * 4096 + 1024) : "memory");    \
  }
#define PIPET_WAIT(F, G)                                                                                                   \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[1][0]), "+v"(F[1][1]), "+v"(F[2][0]), "+v"(F[2][1]), "+v"(F[3][0]), \
               "+v"(F[3][1]), "+v"(G[0][0]), "+v"(G[0][1]), "+v"(G[1][0]), "+v"(G[1][1]), "+v"(G[2][0]), "+v"(G[2][1]), "+v"(G[3][0]), "+v"(G[3][1]))
#define PIPET_FRAG(F) __builtin_bit_cast(bf16x8, __builtin_shufflevector(F[0], F[1], 0, 1, 2, 3))
#pragma unroll
  for (int i = 0; i < 4; ++i) { PIPET_READ(f0a[i], ta[i], 0); PIPET_READ(f0b[i], tb[i], 0); }
  int cur = 0;
  for (int s = 0; s < nslab; ++s) {
    const unsigned off = (unsigned)(cur * BUF);
    const int nxt = cur == NS - 1 ? 0 : cur + 1;
    PIPET_WAIT(f0a, f0b);
    const bool fillb = s >= 1 && s + NS - 2 < nslab;
    char* const pb = smem + (cur >= 2 ? cur - 2 : cur + 2) * BUF + ABYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PIPET_FRAG(f0a[i]), PIPET_FRAG(f0b[j]), acc[i][j], 0, 0, 0);
      if (fillb) gb.issue_piece(pb, wave, i);
      PIPET_READ(f1a[i], ta[i] + off, 1);
      PIPET_READ(f1b[i], tb[i] + off, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (s + 2 < nslab) glds_wait_vm<PW>(); else glds_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    PIPET_WAIT(f1a, f1b);
    const unsigned noff = (unsigned)(nxt * BUF);
    const bool more = s + 1 < nslab;
    const bool filla = s + NS - 1 < nslab;
    char* const pa = smem + (cur == 0 ? NS - 1 : cur - 1) * BUF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PIPET_FRAG(f1a[i]), PIPET_FRAG(f1b[j]), acc[i][j], 0, 0, 0);
      if (filla) ga.issue_piece(pa, wave, i);
      if (more) {
        PIPET_READ(f0a[i], ta[i] + noff, 0);
        PIPET_READ(f0b[i], tb[i] + noff, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    cur = nxt;
  }
#undef PIPET_READ
#undef PIPET_WAIT
#undef PIPET_FRAG
  __builtin_amdgcn_s_barrier();
}
#endif

}  // namespace vmmt

// Common device helpers for the gfx950 (MI355X / CDNA4) kernels of the VI_Model1 training step.
// Wave = 64 lanes.  MFMA shapes used: v_mfma_f32_32x32x16_bf16 (bf16 storage) and
// v_mfma_f32_32x32x2_f32 (fp32 storage, exact fp32 "parity" mode).  Both share one C/D layout:
//   col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), reg in [0,16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vmmt {

typedef uint16_t bf16_t;  // raw bf16 storage
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((uint32_t)x) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(uint16_t, b);
}
template <class T> __device__ __forceinline__ float to_f(T x);
template <> __device__ __forceinline__ float to_f<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t x) { return bf2f(x); }
template <class T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float x) { return f2bf(x); }

template <class T> struct Traits;
template <> struct Traits<float> {
  static constexpr int VEC = 4;     // elements per 16-byte vector
  static constexpr int KSTEP = 2;   // K per MFMA
  static constexpr int PAD = 4;     // LDS row padding (elements), keeps rows 16-byte aligned
  static constexpr int PADKS = 4;   // padding of a K-strided image row ([k][rows + PADKS])
};
template <> struct Traits<bf16_t> {
  static constexpr int VEC = 8;
  static constexpr int KSTEP = 16;
  static constexpr int PAD = 8;     // 80-byte rows: ds_read_b128 by 16 consecutive rows is conflict-free
  static constexpr int PADKS = 32;  // [k][rows + 32]: the 4 k-rows of a ds_read_b64_tr_b16 block land 16 banks apart
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  // accurate enough for fp32 parity (1e-6) and cheap: tanh(x) = 1 - 2/(exp(2x)+1)
  float e = __expf(2.0f * x);
  return 1.0f - 2.0f / (e + 1.0f);
}
// The LSTM cells' versions: v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 instructions each) -- five per cell, on the critical
// path of every step of a recurrence (0.68 -> 0.48 us of a 3.4-us forward step).  NOT for the GEMM epilogues: there the other sequence costs the
// 128 x 128 products 12 registers (220 -> 232), and they no longer fit beside the decoder's forward recurrence (step + 60 us, same box)
__device__ __forceinline__ float rcpf_(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}
__device__ __forceinline__ float cell_sigmoid_(float x) { return rcpf_(1.0f + __expf(-x)); }
__device__ __forceinline__ float cell_tanh_(float x) {
  float e = __expf(2.0f * x);
  return 1.0f - 2.0f * rcpf_(e + 1.0f);
}

// the LSTM cell of one (sentence, hidden unit) from the four gate pre-activations: shared by the per-step kernels (lstm.hip) and
// the persistent recurrence (lstm_seq.hip) so that both produce the same bits (the one fused multiply-add is spelled out)
struct LstmCell { float i, f, g, o, c, h; };
__device__ __forceinline__ LstmCell lstm_cell_math(float pi, float pf, float pg, float po, float cp) {
  LstmCell r;
  r.i = cell_sigmoid_(pi); r.f = cell_sigmoid_(pf); r.g = cell_tanh_(pg); r.o = cell_sigmoid_(po);
  r.c = __builtin_fmaf(r.f, cp, r.i * r.g);
  r.h = r.o * cell_tanh_(r.c);
  return r;
}
// ... and its backward: gradients of the four gate pre-activations and of c_{t-1}, given dL/dh and the dL/dc arriving from t+1
struct LstmCellGrad { float di, df, dg, d_o, dc_prev; };
// (in two halves: what depends only on the SAVED forward values -- the persistent backward recurrence computes it while it waits for the
// step's dgates -- and what needs dL/dh; lstm_cell_bwd_math is their composition, so every caller produces the same bits)
struct LstmCellBwdPre { float tc, omt2, omi, omf, omo, omg2; };
__device__ __forceinline__ LstmCellBwdPre lstm_cell_bwd_pre(float i, float f, float g, float o, float c) {
  LstmCellBwdPre p;
  p.tc = cell_tanh_(c);
  p.omt2 = __builtin_fmaf(-p.tc, p.tc, 1.f);
  p.omi = 1.f - i; p.omf = 1.f - f; p.omo = 1.f - o;
  p.omg2 = __builtin_fmaf(-g, g, 1.f);
  return p;
}
__device__ __forceinline__ LstmCellGrad lstm_cell_bwd_post(const LstmCellBwdPre& p, float i, float f, float g, float o, float cp, float dh, float dc) {
  LstmCellGrad r;
  const float d_o = dh * p.tc;
  dc = __builtin_fmaf(dh * o, p.omt2, dc);
  const float d_i = dc * g, d_f = dc * cp, d_g = dc * i;
  r.di = d_i * i * p.omi;
  r.df = d_f * f * p.omf;
  r.dg = d_g * p.omg2;
  r.d_o = d_o * o * p.omo;
  r.dc_prev = dc * f;
  return r;
}
__device__ __forceinline__ LstmCellGrad lstm_cell_bwd_math(float i, float f, float g, float o, float c, float cp, float dh, float dc) {
  return lstm_cell_bwd_post(lstm_cell_bwd_pre(i, f, g, o, c), i, f, g, o, cp, dh, dc);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ----------------------------------------------------------------------------------------------
// Row maps: translate a tile-local row index into a global row index (or -1 = out of range).
// ----------------------------------------------------------------------------------------------
struct LinearMap {   // rows base .. limit-1
  int base, limit;
  __device__ __forceinline__ int operator()(int r) const { int g = base + r; return g < limit ? g : -1; }
  // true when no 16-byte vector of `vec` consecutive local rows straddles the end of the valid range
  __device__ __forceinline__ bool vec_ok(int rows, int vec) const { int n = limit - base; return n >= rows || n % vec == 0; }
  // a 16-byte vector that straddles `limit` may still be LOADED when every stored row is padded to a multiple of vec
  __device__ __forceinline__ bool pad_ok(long ld, int vec) const { return ld >= (long)((limit + vec - 1) / vec) * vec; }
  __device__ __forceinline__ bool full(int rows) const { return limit - base >= rows; }   // every tile row is in range
};
struct GateMap {     // LSTM gate rows: local n = g * BU + u  ->  g * H + u0 + u   (g = i,f,g,o)
  int u0, H, BU;
  __device__ __forceinline__ int operator()(int r) const {
    int g = r / BU, u = u0 + (r - g * BU);
    return u < H ? g * H + u : -1;
  }
  __device__ __forceinline__ bool vec_ok(int rows, int vec) const { int n = H - u0; return (n >= BU || n % vec == 0) && BU % vec == 0; }
  __device__ __forceinline__ bool pad_ok(long, int) const { return false; }
  __device__ __forceinline__ bool full(int) const { return H - u0 >= BU; }
};

// ----------------------------------------------------------------------------------------------
// Tile staging: global -> registers (16-byte vectors when aligned and in range, scalar otherwise)
// -> LDS image: K-contiguous operands [rows][BK + PAD]; K-strided operands keep their global orientation
//    [BK][rows + PADKS] (coalesced 16-byte loads and stores) and are transposed on the way to the MFMA operand
//    registers by ds_read_b64_tr_b16 (bf16) or by plain per-lane reads (fp32).
//   KC = true : operand stored [row][k] (k contiguous), element (row,k) at P[row*ld + k]
//   KC = false: operand stored [k][row] (row contiguous), element (row,k) at P[kmap(k)*ld + row]
//               (kmod > 0: k index taken modulo kmod -- used to broadcast z over time steps)
// ----------------------------------------------------------------------------------------------
template <class T, int ROWS, int BK, int NT, bool KC>
struct Stager {
  static constexpr int VEC = Traits<T>::VEC;
  static constexpr int STRIDE = KC ? BK + Traits<T>::PAD : ROWS + Traits<T>::PADKS;
  static constexpr int ELEMS = KC ? ROWS * STRIDE : BK * STRIDE;
  static constexpr int NVEC = ROWS * BK / VEC;
  static constexpr int PER = (NVEC + NT - 1) / NT;
  static_assert(ROWS % VEC == 0 && BK % VEC == 0, "tile");
  u32x4 regs[PER];
  // per-vector state hoisted out of the K loop.  `ptr` is ALWAYS dereferenceable (out-of-range rows are clamped to
  // the tile's first row and zeroed by a select), so the fast path has no branch around any load: hipcc then issues
  // the whole slab's loads back to back instead of waiting for each one (a load inside a per-vector branch is waited
  // for individually).
  const T* ptr[PER];
  int nvalid[PER];        // leading elements of the vector that are in range (0 = clamped dummy load, VEC = all)
  const T* P_; long ld_; int kmod_;
  bool fast;              // block-uniform: 16-byte aligned operand, no ragged vector
  bool raw;               // regs hold unmasked fast-path loads
  bool edge;              // block-uniform: some tile rows are out of range (only then store() masks anything)

  template <class Map>
  __device__ __forceinline__ void init(const T* __restrict__ P, long ld, const Map& map, int kmod, int tid) {
    P_ = P; ld_ = ld; kmod_ = kmod;
    fast = (((uintptr_t)P) & 15) == 0 && ((ld * (long)sizeof(T)) & 15) == 0 &&
           (KC || map.vec_ok(ROWS, VEC) || map.pad_ok(ld, VEC));
    edge = !map.full(ROWS);
    const int g_first = map(0);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int v = tid + i * NT;
      if (NVEC % NT != 0 && v >= NVEC) v = 0;
      if constexpr (KC) {
        int row = v / (BK / VEC), kq = (v % (BK / VEC)) * VEC;
        int g = map(row);
        nvalid[i] = g >= 0 ? VEC : 0;
        ptr[i] = P + (long)(g >= 0 ? g : g_first) * ld + kq;
      } else {
        int row = (v % (ROWS / VEC)) * VEC;
        int g = map(row);
        int n = 0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) n += map(row + e) >= 0 ? 1 : 0;     // valid rows are a prefix
        nvalid[i] = n;
        ptr[i] = P + (g >= 0 ? g : g_first);
      }
    }
  }

  // whole slab [k0, k0 + BK) inside [0, K) and `fast`: unconditional 16-byte loads
  __device__ __forceinline__ void load_fast(int k0, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      u32x4 v;
      if constexpr (KC) {
        v = *reinterpret_cast<const u32x4*>(ptr[i] + k0);
      } else {
        int kq = (tid + i * NT) / (ROWS / VEC);
        if (NVEC % NT != 0 && tid + i * NT >= NVEC) kq = 0;
        int k = k0 + kq;
        long kr = kmod_ > 0 ? (k % kmod_) : k;
        v = *reinterpret_cast<const u32x4*>(ptr[i] + kr * ld_);
      }
      regs[i] = v;            // NOT consumed here: the zero-select for clamped rows happens in store(), so the loads
    }                         // stay in flight across the MFMA slab
    raw = true;
  }

  // general path: element-wise with full predication (edge tiles, unaligned operands, the K tail)
  template <class Map>
  __device__ __forceinline__ void load_slow(const Map& map, int k0, int K, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int v = tid + i * NT;
      T tmp[VEC];
      if (NVEC % NT != 0 && v >= NVEC) continue;
      if constexpr (KC) {
        int row = v / (BK / VEC), kq = (v % (BK / VEC)) * VEC;
        int g = map(row);
        int k = k0 + kq;
#pragma unroll
        for (int e = 0; e < VEC; ++e) tmp[e] = (g >= 0 && k + e < K) ? P_[(long)g * ld_ + k + e] : T(0);
      } else {
        int kq = v / (ROWS / VEC), row = (v % (ROWS / VEC)) * VEC;
        int k = k0 + kq;
        long kr = kmod_ > 0 ? (k % kmod_) : k;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          int ge = map(row + e);
          tmp[e] = (k < K && ge >= 0) ? P_[kr * ld_ + ge] : T(0);
        }
      }
      regs[i] = *reinterpret_cast<const u32x4*>(tmp);
    }
    raw = false;
  }

  __device__ __forceinline__ void store(T* __restrict__ S, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int v = tid + i * NT;
      if (NVEC % NT != 0 && v >= NVEC) break;
      u32x4 w = regs[i];
      if (edge && raw && nvalid[i] < VEC) {               // clamped / ragged vector: keep the valid prefix only
        if constexpr (sizeof(T) == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = e < nvalid[i] ? w[e] : 0u;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            uint32_t m = (2 * e + 1 < nvalid[i]) ? 0xffffffffu : (2 * e < nvalid[i] ? 0x0000ffffu : 0u);
            w[e] &= m;
          }
        }
      }
      if constexpr (KC) {
        int row = v / (BK / VEC), kq = (v % (BK / VEC)) * VEC;
        *reinterpret_cast<u32x4*>(S + row * STRIDE + kq) = w;
      } else {
        int kq = v / (ROWS / VEC), row = (v % (ROWS / VEC)) * VEC;
        *reinterpret_cast<u32x4*>(S + kq * STRIDE + row) = w;
      }
    }
  }
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// MFMA 32x32x16 bf16 operand fragment for tile rows [off, off+32), k-step ks: lane (r = lane&31, h = lane>>5) needs
// the 8 elements (row off+r, k = 16 ks + 8h .. +7).
template <int ROWS, int BK, bool KC>
__device__ __forceinline__ bf16x8 frag_bf16(const bf16_t* __restrict__ img, int off, int ks, int lane) {
  if constexpr (KC) {
    constexpr int STRIDE = BK + Traits<bf16_t>::PAD;
    return *reinterpret_cast<const bf16x8*>(img + (off + (lane & 31)) * STRIDE + ks * 16 + 8 * (lane >> 5));
  } else {
    // image [k][row]: each 16-lane group g transposes a 4(k) x 16(row) block per read; lane 4q+p of the group supplies
    // the address of k-row q, rows 4p..4p+3 and receives row (lane&15) of the block with k = q' in element q'.
    constexpr int STRIDE = ROWS + Traits<bf16_t>::PADKS;
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const bf16_t* a0 = img + (ks * 16 + 8 * (g >> 1) + q) * STRIDE + off + 16 * (g & 1) + 4 * p;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * STRIDE));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  }
}
template <int ROWS, int BK, bool KC>
__device__ __forceinline__ float frag_f32(const float* __restrict__ img, int off, int ks, int lane) {
  if constexpr (KC) return img[(off + (lane & 31)) * (BK + Traits<float>::PAD) + ks * 2 + (lane >> 5)];
  else return img[(ks * 2 + (lane >> 5)) * (ROWS + Traits<float>::PADKS) + off + (lane & 31)];
}

// One BK-deep slab of MFMAs for a wave: TI x TJ tiles of 32x32.
template <class T, int BM, int BN, int BK, bool A_KC, bool B_KC, int TI, int TJ>
__device__ __forceinline__ void mfma_slab(const T* __restrict__ As, const T* __restrict__ Bs, const int (&aoff)[TI],
                                          const int (&boff)[TJ], f32x16 (&acc)[TI][TJ], int lane) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 a[TI], b[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) a[i] = frag_bf16<BM, BK, A_KC>(As, aoff[i], ks, lane);
#pragma unroll
      for (int j = 0; j < TJ; ++j) b[j] = frag_bf16<BN, BK, B_KC>(Bs, boff[j], ks, lane);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  } else {
#pragma unroll 4
    for (int ks = 0; ks < BK / 2; ++ks) {
      float a[TI], b[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) a[i] = frag_f32<BM, BK, A_KC>(As, aoff[i], ks, lane);
#pragma unroll
      for (int j = 0; j < TJ; ++j) b[j] = frag_f32<BN, BK, B_KC>(Bs, boff[j], ks, lane);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
}

// Block-level main loop: C_tile += A_tile(BM x K) * B_tile(BN x K)^T.  Two LDS buffers: while slab k is multiplied
// out of buffer `cur`, the global loads of slab k+1 are in flight into registers and are written to the other buffer
// after the MFMAs -- one barrier per slab, and the loads have a whole MFMA phase to land.
template <class T, int BM, int BN, int BK, int NT, bool A_KC, bool B_KC, int TI, int TJ, class AMap, class BMap, bool DB = true>
__device__ __forceinline__ void gemm_mainloop(const T* __restrict__ A, long lda, const AMap& amap,
                                              const T* __restrict__ B, long ldb, const BMap& bmap, int K, int a_kmod,
                                              int b_kmod, const int (&aoff)[TI], const int (&boff)[TJ],
                                              f32x16 (&acc)[TI][TJ], T* __restrict__ smem, int kbeg = 0) {
  Stager<T, BM, BK, NT, A_KC> sa;
  Stager<T, BN, BK, NT, B_KC> sb;
  constexpr int AE = (Stager<T, BM, BK, NT, A_KC>::ELEMS + 7) & ~7;
  constexpr int BE = (Stager<T, BN, BK, NT, B_KC>::ELEMS + 7) & ~7;
  const int tid = threadIdx.x, lane = tid & 63;
  if (K <= kbeg) return;              // K is the END of this block's reduction range, kbeg its start (split-K)
  sa.init(A, lda, amap, a_kmod, tid);
  sb.init(B, ldb, bmap, b_kmod, tid);
  const int nslab = (K - kbeg + BK - 1) / BK;
  // slabs [0, nfast) are full and both operands take the branch-free vector path; the rest (K tail, ragged or
  // unaligned tiles) use the predicated element-wise path.  The two phases are SEPARATE loops: a fast/slow choice
  // inside one loop body makes hipcc wait for operand A's loads before it issues operand B's.
  const int nfast = (sa.fast && sb.fast) ? (K - kbeg) / BK : 0;
  if (nfast > 0) { sa.load_fast(kbeg, tid); sb.load_fast(kbeg, tid); }
  else { sa.load_slow(amap, kbeg, K, tid); sb.load_slow(bmap, kbeg, K, tid); }
  sa.store(smem, tid);
  sb.store(smem + AE, tid);
  __syncthreads();
  int cur = 0, sidx = 0;
  if constexpr (DB) {
    // fast phase, two slabs per trip so that both LDS buffer addresses are compile-time constants
    for (; sidx + 2 < nfast; sidx += 2) {
      sa.load_fast(kbeg + (sidx + 1) * BK, tid);
      sb.load_fast(kbeg + (sidx + 1) * BK, tid);
      mfma_slab<T, BM, BN, BK, A_KC, B_KC, TI, TJ>(smem, smem + AE, aoff, boff, acc, lane);
      sa.store(smem + (AE + BE), tid);
      sb.store(smem + (AE + BE) + AE, tid);
      __syncthreads();
      sa.load_fast(kbeg + (sidx + 2) * BK, tid);
      sb.load_fast(kbeg + (sidx + 2) * BK, tid);
      mfma_slab<T, BM, BN, BK, A_KC, B_KC, TI, TJ>(smem + (AE + BE), smem + (AE + BE) + AE, aoff, boff, acc, lane);
      sa.store(smem, tid);
      sb.store(smem + AE, tid);
      __syncthreads();
    }
  }
  for (; sidx + 1 < nfast; ++sidx) {                    // fast phase: next slab is a fast one
    sa.load_fast(kbeg + (sidx + 1) * BK, tid);
    sb.load_fast(kbeg + (sidx + 1) * BK, tid);
    const T* As = smem + cur * (AE + BE);
    mfma_slab<T, BM, BN, BK, A_KC, B_KC, TI, TJ>(As, As + AE, aoff, boff, acc, lane);
    if constexpr (!DB) __syncthreads();                  // single buffer: everyone done reading before it is refilled
    T* Ns = smem + (DB ? (cur ^ 1) : 0) * (AE + BE);
    sa.store(Ns, tid);
    sb.store(Ns + AE, tid);
    __syncthreads();
    if constexpr (DB) cur ^= 1;
  }
  for (; sidx < nslab; ++sidx) {                        // remaining slabs: next slab (if any) via the slow path
    const bool more = sidx + 1 < nslab;
    if (more) {
      sa.load_slow(amap, kbeg + (sidx + 1) * BK, K, tid);
      sb.load_slow(bmap, kbeg + (sidx + 1) * BK, K, tid);
    }
    const T* As = smem + cur * (AE + BE);
    mfma_slab<T, BM, BN, BK, A_KC, B_KC, TI, TJ>(As, As + AE, aoff, boff, acc, lane);
    if constexpr (!DB) __syncthreads();
    if (more) {
      T* Ns = smem + (DB ? (cur ^ 1) : 0) * (AE + BE);
      sa.store(Ns, tid);
      sb.store(Ns + AE, tid);
    }
    __syncthreads();
    if constexpr (DB) cur ^= 1;
  }
}

template <class T, int BM, int BN, int BK, bool A_KC = true, bool B_KC = true, bool DB = true>
constexpr int gemm_smem_elems() {
  return (DB ? 2 : 1) * (((Stager<T, BM, BK, 64, A_KC>::ELEMS + 7) & ~7) + ((Stager<T, BN, BK, 64, B_KC>::ELEMS + 7) & ~7));
}

// XCD-aware workgroup -> tile remap (MI355X: 8 XCDs with private L2s; workgroups are dealt round-robin over the XCDs, so
// ids b and b+8 share an L2).  Logical tiles [x*cpx, (x+1)*cpx) go to the workgroups of XCD-group x: tiles that are
// neighbours in the logical order (and share an operand panel) then hit the same L2 instead of 8 different ones.
// Bijective for any grid size; affects speed only, never results.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// accumulator element -> (row, col) inside a 32x32 tile
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ int acc_col(int lane) { return lane & 31; }

// counter-based RNG (for dropout masks and eps): 32-bit mix of (seed, index)
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t rng32(uint64_t seed, uint64_t idx) {
  uint32_t a = hash32((uint32_t)idx ^ (uint32_t)seed);
  uint32_t b = hash32((uint32_t)(idx >> 32) + 0x9e3779b9U + (uint32_t)(seed >> 32) + a);
  return hash32(a ^ (b * 0x85ebca6bU));
}
__device__ __forceinline__ float u01(uint32_t r) { return ((r >> 8) + 0.5f) * (1.0f / 16777216.0f); }

#define VMMT_OK 0
#define VMMT_EINVAL 1
#define VMMT_ELAUNCH 2

inline int check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VMMT_OK : VMMT_ELAUNCH;
}

}  // namespace vmmt

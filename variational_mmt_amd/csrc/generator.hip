// Fused vocabulary projection + log-softmax + NLL (forward statistics and backward seed) for gfx950.
//
// Reference: generator = Sequential(Linear(H, V), LogSoftmax) (onmt/ModelConstructor.py:583-585), criterion =
// NLLLoss(weight[pad] = 0, sum) (onmt/Loss.py:163-165), argmax accuracy (onmt/VILoss.py:515-531); the reference
// materialises the [T'B, V] fp32 log-probabilities and re-reads them >= 4 times.  Here the logits never reach HBM
// in the forward pass: the GEMM is computed TRANSPOSED (rows = vocabulary entries, columns = tokens), so that a
// lane owns one token and its vocabulary entries sit in that lane's accumulator registers -- the running
// max / sum-exp / argmax over the vocabulary are lane-local (no cross-lane reduction).  Pass 1 emits per-token
// partial statistics per 64-row vocabulary block, a tiny combine kernel produces logsumexp / NLL / accuracy, and
// pass 2 recomputes the logits tile and writes the loss gradient G^T[v][m] = (softmax - onehot) * w_m / norm
// directly in the storage type, coalesced along tokens.  dO = G Wg and dWg = G^T O are then plain GEMMs.
#include "common.hpp"
#include "glds_gemm.hpp"
#include "vmmt.h"

namespace vmmt {

struct GenArgs {
  const void* W; long ldw;            // T [V][ldw] generator weight (k contiguous)
  const float* bias;                  // f32 [V]
  const void* O; long ldo;            // T [M][ldo] decoder outputs (after dropout)
  const long long* y;                 // [M] target ids (tgt[1:], flattened t*B+b)
  int M, V, K, pad, npart;
  // pass 1
  float* part_max; float* part_sum; int* part_idx; float* tgt_logit;   // [NPART][M], [M]
  // pass 2
  const float* lse; float inv_norm; void* GT; long ldgt;               // T [V][ldgt]
  float* dbias;                                                        // optional f32 [V] += row sums of G^T (bias gradient)
};

// Tile: 128 vocabulary rows x 256 tokens, 8 waves (2 along V x 4 along tokens, 64x64 each), BK = 64, one LDS buffer
// (K = H is short: 8 slabs at H = 512) -- the fastest of the shapes tried on [30000 x 5120 x 512] (tools/gemm_ab.py).
// BMV_ = 256 (bf16 LDS-DMA path only): 256 x 256 tiles, waves of 128 x 64 -- a third fewer operand bytes per FLOP through the
// L2 -> LDS path, which is what bounds this kernel (8 slabs of 48 KiB per 128 x 256 tile at ~45 GB/s per CU).
template <class T, int BMV_ = 128> struct GenCfg {
  static constexpr int BMV = BMV_, BNM = 256, BK = sizeof(T) == 2 ? 64 : 32, NT = 512, TI = BMV_ / 64, TJ = 2;
  static constexpr bool DB = false;
  static constexpr int PP = 64 + 16 / sizeof(T);     // pitch (elements) of a wave's 64 x 64 output patch in LDS
};

// GL: 0 = register-staged main loop (any dtype / alignment); LDS-DMA main loops: 1 = one buffer, 2 = two buffers,
// 3 = three buffers with counted waits (two slabs in flight across the barrier)
template <class T, int MODE, int GL, int BMV_ = 128>
__global__ void __launch_bounds__(512) gen_kernel(GenArgs a, int tiles_m) {
  using Cf = GenCfg<T, BMV_>;
  constexpr int BK = Cf::BK, NT = Cf::NT, BMV = Cf::BMV, BNM = Cf::BNM, TI = Cf::TI, TJ = Cf::TJ;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  // L2-friendly order: an XCD walks groups of 8 vocabulary tiles x all token tiles with the vocabulary tile index
  // fastest, so the ~64 workgroups resident on an XCD share 8 Wg tiles (1 MB) and 8 O tiles (2 MB) inside its 4 MB L2
  // instead of streaming the whole O matrix from the Infinity Cache for every vocabulary tile.
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles_v = (a.V + BMV - 1) / BMV;
  constexpr int GV = 8;
  const int group = tile / (GV * tiles_m), in_g = tile - group * (GV * tiles_m);
  const int gv = min(GV, tiles_v - group * GV);
  const int tv = group * GV + in_g % gv, tm = in_g / gv;
  const int v0 = tv * BMV, m0 = tm * BNM;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wv = wave >> 2, wm = wave & 3;
  int aoff[TI], boff[TJ] = {wm * 64, wm * 64 + 32};
#pragma unroll
  for (int i = 0; i < TI; ++i) aoff[i] = wv * (32 * TI) + 32 * i;
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  LinearMap amap{v0, a.V}, bmap{m0, a.M};
  if constexpr (GL == 3) {
    gemm_mainloop_glds3<BMV, BNM, NT / 64, true, true, TI, TJ>((const bf16_t*)a.W, a.ldw, v0, a.V, (const bf16_t*)a.O, a.ldo, m0, a.M, 0,
                                                              a.K, aoff, boff, acc, smem_raw);
  } else if constexpr (GL != 0) {
    gemm_mainloop_glds<BMV, BNM, NT / 64, true, true, TI, TJ, GL == 2>((const bf16_t*)a.W, a.ldw, v0, a.V, (const bf16_t*)a.O, a.ldo,
                                                                     m0, a.M, 0, a.K, aoff, boff, acc, smem_raw);
  } else {
    gemm_mainloop<T, BMV, BNM, BK, NT, true, true, TI, TJ, LinearMap, LinearMap, Cf::DB>(
        (const T*)a.W, a.ldw, amap, (const T*)a.O, a.ldo, bmap, a.K, 0, 0, aoff, boff, acc, smem);
  }
  if constexpr (MODE == 1) __syncthreads();     // the staging buffers are reused as per-wave output patches
  // A lane owns token columns m_j and, per 32-row tile i, the vocabulary rows vb_i + 8q + s (q, s = 0..3) where
  // vb_i = v0 + aoff[i] + 4*(lane>>5): four consecutive rows per register group q -> one 16-byte bias load per group.
  const int hi4 = 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int vb = v0 + aoff[i] + hi4 + 8 * q;
      f32x4 bv;
      if (vb + 3 < a.V && ((((uintptr_t)(a.bias + vb)) & 15) == 0)) bv = *reinterpret_cast<const f32x4*>(a.bias + vb);
      else {
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) bv[s_] = vb + s_ < a.V ? a.bias[vb + s_] : 0.f;
      }
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        const bool ok = vb + s_ < a.V;
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j][4 * q + s_] = ok ? acc[i][j][4 * q + s_] + bv[s_] : -INFINITY;
      }
    }
  // The epilogue works in units of 64 vocabulary rows (two 32-row MFMA tiles of a wave): one partial-statistics row (MODE 0)
  // or one pass through the wave's 64 x 64 output patch (MODE 1) per unit; a wave owns TI / 2 of them.
  constexpr int NH = TI / 2;
#pragma unroll
  for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int m = m0 + boff[j] + (lane & 31);
      const bool mv = m < a.M;
      const int ym = mv ? (int)a.y[m] : -1;
      if constexpr (MODE == 0) {
        float mx = -INFINITY, tl = 0.f;
        int mi = 0x7fffffff;
        bool hit = false;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = 2 * hf + ii;
            const int v = v0 + aoff[i] + hi4 + (r & 3) + 8 * (r >> 2);
            const float x = acc[i][j][r];
            const bool gt = x > mx;                    // rows visited in increasing v per lane: first max wins
            mi = gt ? v : mi;
            mx = gt ? x : mx;
            const bool h = v == ym;
            tl = h ? x : tl;
            hit = hit || h;
          }
        float sm = 0.f;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int r = 0; r < 16; ++r) sm += __expf(acc[2 * hf + ii][j][r] - mx);      // exp(-inf - finite) = 0; all -inf handled below
        if (mx == -INFINITY) sm = 0.f;
        if (hit) a.tgt_logit[m] = tl;
        // combine with the other half-wave (same token column, interleaved rows)
        float omx = __shfl_xor(mx, 32, 64), osm = __shfl_xor(sm, 32, 64);
        int omi = __shfl_xor(mi, 32, 64);
        float nm = fmaxf(mx, omx);
        float ns = (nm == -INFINITY) ? 0.f : sm * __expf(mx - nm) + osm * __expf(omx - nm);
        int ni = (omx > mx || (omx == mx && omi < mi)) ? omi : mi;
        const int prow = (tv * 2 + wv) * NH + hf;                       // partial row = 64-row vocabulary block index
        if (mv && lane < 32 && prow < a.npart) {                        // (a 256-row tile can reach past vmmt_gen_npart(V) rows)
          long p = (long)prow * a.M + m;
          a.part_max[p] = nm; a.part_sum[p] = ns; a.part_idx[p] = ni;
        }
      } else {
        // gradient values -> this wave's 64(v) x 64(m) patch of an LDS image [v][m] (bf16/f32), written back below as
        // whole 16-byte row segments (a lane-per-token 2-byte store per element is store-issue bound)
        const float l = mv ? a.lse[m] : 0.f;
        const float sc = (mv && ym != a.pad) ? a.inv_norm : 0.f;
        T* patch = smem + wave * (64 * Cf::PP);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int i = 2 * hf + ii;
          const int vb = v0 + aoff[i] + hi4;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dv = (r & 3) + 8 * (r >> 2);
            float g = (__expf(acc[i][j][r] - l) - ((vb + dv) == ym ? 1.f : 0.f)) * sc;
            patch[(ii * 32 + hi4 + dv) * Cf::PP + j * 32 + (lane & 31)] = from_f<T>(g);
          }
        }
      }
    }
    if constexpr (MODE == 1) {
      // each wave owns its patch (LDS operations of a wave complete in order): no workgroup barrier between its writes, its
      // reads and the next unit's writes; the staging buffers of the main loop are being reused -> one barrier in front
      constexpr int VEC = 16 / sizeof(T), PP = Cf::PP;
      const T* patch = smem + wave * (64 * PP);
      const int vbase = v0 + wv * (64 * NH) + hf * 64, mbase = m0 + wm * 64;
      constexpr int CH = 64 / VEC;                       // 16-byte chunks per 64-token row
#pragma unroll
      for (int it = 0; it < (64 * CH) / 64; ++it) {
        const int idx = it * 64 + lane, row = idx / CH, ch = idx % CH;
        const int v = vbase + row, mm = mbase + ch * VEC;
        if (v < a.V && mm < a.M) {
          T* dst = reinterpret_cast<T*>(a.GT) + (long)v * a.ldgt + mm;
          const T* srcp = patch + row * PP + ch * VEC;
          if (mm + VEC <= a.M && ((((uintptr_t)dst) & 15) == 0)) *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(srcp);
          else for (int e = 0; e < VEC && mm + e < a.M; ++e) dst[e] = srcp[e];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent variant (bf16, LDS-DMA, three stages): 128 x 256 tiles need 144 KiB of LDS, so only ONE workgroup fits a CU and
// nothing hides a tile's prologue (first slabs' L2 latency) and epilogue (exp / max / stores) -- about 40 % of a tile's time at
// K = 512 (8 slabs).  Here a workgroup walks its share of the tiles and issues the first two slabs of the NEXT tile (buffers 0
// and 1, free after the last multiply) before it runs the epilogue of the current one; the gradient patches of MODE 1 live in
// buffer 2 (32-row units: 36 KiB).  Tile order = the L2-friendly order of gen_kernel: XCD x owns a contiguous range of the
// grouped tile sequence and its workgroups take consecutive tiles of it.
template <int MODE>
__global__ void __launch_bounds__(512) gen_kernel_p(GenArgs a, int tiles_m, int ntiles) {
  using T = bf16_t;
  using Cf = GenCfg<T, 128>;
  constexpr int NT = Cf::NT, BMV = Cf::BMV, BNM = Cf::BNM, TI = Cf::TI, TJ = Cf::TJ, NW = NT / 64;
  constexpr int ABYTES = BMV * GBK * 2, BBYTES = BNM * GBK * 2, BUF = ABYTES + BBYTES;
  using GA = GldsOperand<BMV, true, NW>;
  using GB = GldsOperand<BNM, true, NW>;
  constexpr int PW = GA::PER + GB::PER;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  char* smem = smem_raw;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wv = wave >> 2, wm = wave & 3;
  int aoff[TI], boff[TJ] = {wm * 64, wm * 64 + 32};
#pragma unroll
  for (int i = 0; i < TI; ++i) aoff[i] = wv * (32 * TI) + 32 * i;
  GldsFrag<true, TI> fa;
  GldsFrag<true, TJ> fb;
  fa.init(aoff, lane);
  fb.init(boff, lane);
  // this workgroup's tiles: XCD x = blockIdx % 8 owns [lo, lo + cnt) of the grouped order, slot = blockIdx / 8 strides by per
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int lo = x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8, cnt = q8 + (x < r8 ? 1 : 0);
  const int tiles_v = (a.V + BMV - 1) / BMV;
  constexpr int GV = 8;
  auto coords = [&](int tile, int& tv_, int& v0_, int& m0_) {
    const int group = tile / (GV * tiles_m), in_g = tile - group * (GV * tiles_m);
    const int gv = min(GV, tiles_v - group * GV);
    tv_ = group * GV + in_g % gv;
    v0_ = tv_ * BMV;
    m0_ = (in_g / gv) * BNM;
  };
  int idx = slot;
  if (idx >= cnt) return;
  const int nslab = a.K / GBK;
  int tv, v0, m0;
  coords(lo + idx, tv, v0, m0);
  GA ga;
  GB gb;
  ga.init((const T*)a.W, a.ldw, v0, a.V, 0, wave, lane);
  gb.init((const T*)a.O, a.ldo, m0, a.M, 0, wave, lane);
  ga.issue(smem, wave);
  gb.issue(smem + ABYTES, wave);
  if (nslab > 1) {
    ga.issue(smem + BUF, wave);
    gb.issue(smem + BUF + ABYTES, wave);
  }
  const int hi4 = 4 * (lane >> 5);
  while (true) {
    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#define VMMT_GENP_STEP(CUR, NXT2)                                                                      \
  {                                                                                                    \
    if (s + 1 < nslab) glds_wait_vm<PW>(); else glds_wait_vm<0>();                                     \
    __builtin_amdgcn_s_barrier();                                                                      \
    if (s + 2 < nslab) {                                                                               \
      ga.issue(smem + (NXT2) * BUF, wave);                                                             \
      gb.issue(smem + (NXT2) * BUF + ABYTES, wave);                                                    \
    }                                                                                                  \
    glds_slab<BMV, BNM, true, true, TI, TJ>(smem + (CUR) * BUF, smem + (CUR) * BUF + ABYTES, fa, fb, aoff, boff, acc); \
    ++s;                                                                                               \
  }
    int s = 0;
    while (s < nslab) {
      VMMT_GENP_STEP(0, 2)
      if (s >= nslab) break;
      VMMT_GENP_STEP(1, 0)
      if (s >= nslab) break;
      VMMT_GENP_STEP(2, 1)
    }
#undef VMMT_GENP_STEP
    __builtin_amdgcn_s_barrier();        // every wave has finished its last multiply: buffers 0 and 1 are free
    // ---- epilogue inputs first (a dependent load issued AFTER the prefetch would wait for the prefetch: vmcnt is in order)
    f32x4 bv[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int vb = v0 + aoff[i] + hi4 + 8 * q;
        if (vb + 3 < a.V && ((((uintptr_t)(a.bias + vb)) & 15) == 0)) bv[i][q] = *reinterpret_cast<const f32x4*>(a.bias + vb);
        else {
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_) bv[i][q][s_] = vb + s_ < a.V ? a.bias[vb + s_] : 0.f;
        }
      }
    int ymv[TJ];
    float lsev[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int m = m0 + boff[j] + (lane & 31);
      ymv[j] = m < a.M ? (int)a.y[m] : -1;
      lsev[j] = 0.f;
      if constexpr (MODE == 1) lsev[j] = m < a.M ? a.lse[m] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- prefetch: first two slabs of the next tile
    const int nidx = idx + per;
    const bool has_next = nidx < cnt;
    int ntv = 0, nv0 = 0, nm0 = 0;
    if (has_next) {
      coords(lo + nidx, ntv, nv0, nm0);
      ga.init((const T*)a.W, a.ldw, nv0, a.V, 0, wave, lane);
      gb.init((const T*)a.O, a.ldo, nm0, a.M, 0, wave, lane);
      ga.issue(smem, wave);
      gb.issue(smem + ABYTES, wave);
      if (nslab > 1) {
        ga.issue(smem + BUF, wave);
        gb.issue(smem + BUF + ABYTES, wave);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue of the current tile
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int vb = v0 + aoff[i] + hi4 + 8 * q;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          const bool ok = vb + s_ < a.V;
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j][4 * q + s_] = ok ? acc[i][j][4 * q + s_] + bv[i][q][s_] : -INFINITY;
        }
      }
    if constexpr (MODE == 0) {
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int m = m0 + boff[j] + (lane & 31);
        const bool mv = m < a.M;
        const int ym = ymv[j];
        float mx = -INFINITY, tl = 0.f;
        int mi = 0x7fffffff;
        bool hit = false;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int v = v0 + aoff[i] + hi4 + (r & 3) + 8 * (r >> 2);
            const float xx = acc[i][j][r];
            const bool gt = xx > mx;
            mi = gt ? v : mi;
            mx = gt ? xx : mx;
            const bool h = v == ym;
            tl = h ? xx : tl;
            hit = hit || h;
          }
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) sm += __expf(acc[i][j][r] - mx);
        if (mx == -INFINITY) sm = 0.f;
        if (hit) a.tgt_logit[m] = tl;
        float omx = __shfl_xor(mx, 32, 64), osm = __shfl_xor(sm, 32, 64);
        int omi = __shfl_xor(mi, 32, 64);
        float nm = fmaxf(mx, omx);
        float ns = (nm == -INFINITY) ? 0.f : sm * __expf(mx - nm) + osm * __expf(omx - nm);
        int ni = (omx > mx || (omx == mx && omi < mi)) ? omi : mi;
        const int prow = tv * 2 + wv;
        if (mv && lane < 32 && prow < a.npart) {
          long p = (long)prow * a.M + m;
          a.part_max[p] = nm; a.part_sum[p] = ns; a.part_idx[p] = ni;
        }
      }
    } else {
      // 32-row units through this wave's patch in buffer 2 (the only staging buffer the prefetch leaves alone)
      constexpr int PP = Cf::PP, VEC = 8, CH = 64 / VEC;
      T* patch = reinterpret_cast<T*>(smem + 2 * BUF) + wave * (32 * PP);
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const int vb = v0 + aoff[i] + hi4;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const int m = m0 + boff[j] + (lane & 31);
          const bool mv = m < a.M;
          const float l = lsev[j];
          const float sc = (mv && ymv[j] != a.pad) ? a.inv_norm : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dv = (r & 3) + 8 * (r >> 2);
            float g = (__expf(acc[i][j][r] - l) - ((vb + dv) == ymv[j] ? 1.f : 0.f)) * sc;
            patch[(hi4 + dv) * PP + j * 32 + (lane & 31)] = from_f<T>(g);
          }
        }
        const int vbase = v0 + aoff[i], mbase = m0 + wm * 64;
#pragma unroll
        for (int it = 0; it < (32 * CH) / 64; ++it) {
          const int id2 = it * 64 + lane, row = id2 / CH, ch = id2 % CH;
          const int v = vbase + row, mm = mbase + ch * VEC;
          if (v < a.V && mm < a.M) {
            T* dst = reinterpret_cast<T*>(a.GT) + (long)v * a.ldgt + mm;
            const T* srcp = patch + row * PP + ch * VEC;
            if (mm + VEC <= a.M && ((((uintptr_t)dst) & 15) == 0)) *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(srcp);
            else for (int e = 0; e < VEC && mm + e < a.M; ++e) dst[e] = srcp[e];
          }
        }
      }
    }
    if (!has_next) break;
    idx = nidx; tv = ntv; v0 = nv0; m0 = nm0;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Half-depth variant (bf16): K slabs of 32, three stages of 24 KiB = 72 KiB of LDS and at most 128 VGPRs, so that TWO
// workgroups share a CU and one tile's epilogue / barriers overlap the other's MFMAs (profiles/r1_gen_kernel_investigation.txt).
// LDS images have 64-byte rows (4 chunks of 16 B); a 1-KiB LDS-DMA piece is 16 rows; the chunk index is XOR-swizzled with
// (row >> 2) & 3 on the source side and in the fragment reads (16 lanes of a ds_read_b128 pass then cover all 64 banks).
template <int MODE>
__global__ void __launch_bounds__(512, 4) gen_kernel_h(GenArgs a, int tiles_m) {
  using T = bf16_t;
  using Cf = GenCfg<T, 128>;
  constexpr int NT = Cf::NT, BMV = Cf::BMV, BNM = Cf::BNM, TI = Cf::TI, TJ = Cf::TJ, NW = NT / 64;
  constexpr int ABYTES = BMV * HBK * 2, BBYTES = BNM * HBK * 2, BUF = ABYTES + BBYTES;       // 8 + 16 = 24 KiB
  using GA = HalfOperand<BMV, NW>;
  using GB = HalfOperand<BNM, NW>;
  constexpr int PW = GA::PER + GB::PER;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  char* smem = smem_raw;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles_v = (a.V + BMV - 1) / BMV;
  constexpr int GV = 8;
  const int group = tile / (GV * tiles_m), in_g = tile - group * (GV * tiles_m);
  const int gv = min(GV, tiles_v - group * GV);
  const int tv = group * GV + in_g % gv, tm = in_g / gv;
  const int v0 = tv * BMV, m0 = tm * BNM;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wv = wave >> 2, wm = wave & 3;
  int aoff[TI], boff[TJ] = {wm * 64, wm * 64 + 32};
#pragma unroll
  for (int i = 0; i < TI; ++i) aoff[i] = wv * (32 * TI) + 32 * i;
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  GA ga;
  GB gb;
  ga.init((const T*)a.W, a.ldw, v0, a.V, 0, wave, lane);
  gb.init((const T*)a.O, a.ldo, m0, a.M, 0, wave, lane);
  // fragment offsets: lane (r, h) reads logical chunk 2 ks + h of row toff + r
  int foff[2];
  {
    const int r = lane & 31, h = lane >> 5, sw = (r >> 2) & 3;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = r * 64 + (((2 * ks + h) ^ sw) * 16);
  }
  const int nslab = a.K / HBK;
  ga.issue(smem, wave);
  gb.issue(smem + ABYTES, wave);
  if (nslab > 1) {
    ga.issue(smem + BUF, wave);
    gb.issue(smem + BUF + ABYTES, wave);
  }
#define VMMT_GENH_STEP(CUR, NXT2)                                                                      \
  {                                                                                                    \
    if (s + 1 < nslab) glds_wait_vm<PW>(); else glds_wait_vm<0>();                                     \
    __builtin_amdgcn_s_barrier();                                                                      \
    if (s + 2 < nslab) {                                                                               \
      ga.issue(smem + (NXT2) * BUF, wave);                                                             \
      gb.issue(smem + (NXT2) * BUF + ABYTES, wave);                                                    \
    }                                                                                                  \
    const char* As = smem + (CUR) * BUF;                                                               \
    const char* Bs = As + ABYTES;                                                                      \
    bf16x8 fa_[2][TI], fb_[2][TJ];                                                                     \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                 \
      _Pragma("unroll") for (int i = 0; i < TI; ++i) fa_[ks][i] = *reinterpret_cast<const bf16x8*>(As + foff[ks] + aoff[i] * 64); \
      _Pragma("unroll") for (int j = 0; j < TJ; ++j) fb_[ks][j] = *reinterpret_cast<const bf16x8*>(Bs + foff[ks] + boff[j] * 64); \
    }                                                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                   \
      _Pragma("unroll") for (int i = 0; i < TI; ++i)                                                   \
        _Pragma("unroll") for (int j = 0; j < TJ; ++j)                                                 \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[ks][i], fb_[ks][j], acc[i][j], 0, 0, 0); \
    ++s;                                                                                               \
  }
  int s = 0;
  while (s < nslab) {
    VMMT_GENH_STEP(0, 2)
    if (s >= nslab) break;
    VMMT_GENH_STEP(1, 0)
    if (s >= nslab) break;
    VMMT_GENH_STEP(2, 1)
  }
#undef VMMT_GENH_STEP
  __builtin_amdgcn_s_barrier();        // staging buffers are reused as gradient patches (MODE 1)
  const int hi4 = 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int vb = v0 + aoff[i] + hi4 + 8 * q;
      f32x4 bv;
      if (vb + 3 < a.V && ((((uintptr_t)(a.bias + vb)) & 15) == 0)) bv = *reinterpret_cast<const f32x4*>(a.bias + vb);
      else {
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) bv[s_] = vb + s_ < a.V ? a.bias[vb + s_] : 0.f;
      }
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        const bool ok = vb + s_ < a.V;
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j][4 * q + s_] = ok ? acc[i][j][4 * q + s_] + bv[s_] : -INFINITY;
      }
    }
  if constexpr (MODE == 0) {
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int m = m0 + boff[j] + (lane & 31);
      const bool mv = m < a.M;
      const int ym = mv ? (int)a.y[m] : -1;
      float mx = -INFINITY, tl = 0.f;
      int mi = 0x7fffffff;
      bool hit = false;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int v = v0 + aoff[i] + hi4 + (r & 3) + 8 * (r >> 2);
          const float xx = acc[i][j][r];
          const bool gt = xx > mx;
          mi = gt ? v : mi;
          mx = gt ? xx : mx;
          const bool h = v == ym;
          tl = h ? xx : tl;
          hit = hit || h;
        }
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sm += __expf(acc[i][j][r] - mx);
      if (mx == -INFINITY) sm = 0.f;
      if (hit) a.tgt_logit[m] = tl;
      float omx = __shfl_xor(mx, 32, 64), osm = __shfl_xor(sm, 32, 64);
      int omi = __shfl_xor(mi, 32, 64);
      float nm = fmaxf(mx, omx);
      float ns = (nm == -INFINITY) ? 0.f : sm * __expf(mx - nm) + osm * __expf(omx - nm);
      int ni = (omx > mx || (omx == mx && omi < mi)) ? omi : mi;
      const int prow = tv * 2 + wv;
      if (mv && lane < 32 && prow < a.npart) {
        long p = (long)prow * a.M + m;
        a.part_max[p] = nm; a.part_sum[p] = ns; a.part_idx[p] = ni;
      }
    }
  } else {
    constexpr int PP = Cf::PP, VEC = 8, CH = 64 / VEC;
    T* patch = reinterpret_cast<T*>(smem) + wave * (32 * PP);
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int vb = v0 + aoff[i] + hi4;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int m = m0 + boff[j] + (lane & 31);
        const bool mv = m < a.M;
        const int ym = mv ? (int)a.y[m] : -1;
        const float l = mv ? a.lse[m] : 0.f;
        const float sc = (mv && ym != a.pad) ? a.inv_norm : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dv = (r & 3) + 8 * (r >> 2);
          float g = (__expf(acc[i][j][r] - l) - ((vb + dv) == ym ? 1.f : 0.f)) * sc;
          patch[(hi4 + dv) * PP + j * 32 + (lane & 31)] = from_f<T>(g);
        }
      }
      const int vbase = v0 + aoff[i], mbase = m0 + wm * 64;
#pragma unroll
      for (int it = 0; it < (32 * CH) / 64; ++it) {
        const int id2 = it * 64 + lane, row = id2 / CH, ch = id2 % CH;
        const int v = vbase + row, mm = mbase + ch * VEC;
        if (v < a.V && mm < a.M) {
          T* dst = reinterpret_cast<T*>(a.GT) + (long)v * a.ldgt + mm;
          const T* srcp = patch + row * PP + ch * VEC;
          *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(srcp);      // M % 8 == 0, G^T rows 16-byte aligned (dispatch)
        }
      }
    }
  }
}

// 128 x 128 tiles, 4 waves (2 x 2 of 64 x 64), 48 KiB: THREE workgroups per CU at 3 waves per SIMD (170 VGPRs: no spills)
template <int MODE>
__global__ void __launch_bounds__(256, 3) gen_kernel_q(GenArgs a, int tiles_m) {
  using T = bf16_t;
  using Cf = GenCfg<T, 128>;
  constexpr int NT = 256, BMV = Cf::BMV, BNM = 128, TI = Cf::TI, TJ = Cf::TJ, NW = NT / 64;
  constexpr int ABYTES = BMV * HBK * 2, BBYTES = BNM * HBK * 2, BUF = ABYTES + BBYTES;       // 8 + 16 = 24 KiB
  using GA = HalfOperand<BMV, NW>;
  using GB = HalfOperand<BNM, NW>;
  constexpr int PW = GA::PER + GB::PER;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  char* smem = smem_raw;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles_v = (a.V + BMV - 1) / BMV;
  constexpr int GV = 8;
  const int group = tile / (GV * tiles_m), in_g = tile - group * (GV * tiles_m);
  const int gv = min(GV, tiles_v - group * GV);
  const int tv = group * GV + in_g % gv, tm = in_g / gv;
  const int v0 = tv * BMV, m0 = tm * BNM;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wv = wave >> 1, wm = wave & 1;
  int aoff[TI], boff[TJ] = {wm * 64, wm * 64 + 32};
#pragma unroll
  for (int i = 0; i < TI; ++i) aoff[i] = wv * (32 * TI) + 32 * i;
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  GA ga;
  GB gb;
  ga.init((const T*)a.W, a.ldw, v0, a.V, 0, wave, lane);
  gb.init((const T*)a.O, a.ldo, m0, a.M, 0, wave, lane);
  // fragment offsets: lane (r, h) reads logical chunk 2 ks + h of row toff + r
  int foff[2];
  {
    const int r = lane & 31, h = lane >> 5, sw = (r >> 2) & 3;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = r * 64 + (((2 * ks + h) ^ sw) * 16);
  }
  const int nslab = a.K / HBK;
  ga.issue(smem, wave);
  gb.issue(smem + ABYTES, wave);
  if (nslab > 1) {
    ga.issue(smem + BUF, wave);
    gb.issue(smem + BUF + ABYTES, wave);
  }
  // one loop body, the stage index is a scalar (three macro-expanded bodies made hipcc spill the accumulators)
  int cur = 0;
  for (int s = 0; s < nslab; ++s) {
    if (s + 1 < nslab) glds_wait_vm<PW>(); else glds_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (s + 2 < nslab) {
      const int nb = cur == 0 ? 2 : cur - 1;
      ga.issue(smem + nb * BUF, wave);
      gb.issue(smem + nb * BUF + ABYTES, wave);
    }
    const char* As = smem + cur * BUF;
    const char* Bs = As + ABYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa_[TI], fb_[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) fa_[i] = *reinterpret_cast<const bf16x8*>(As + foff[ks] + aoff[i] * 64);
#pragma unroll
      for (int j = 0; j < TJ; ++j) fb_[j] = *reinterpret_cast<const bf16x8*>(Bs + foff[ks] + boff[j] * 64);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[i], fb_[j], acc[i][j], 0, 0, 0);
    }
    cur = cur == 2 ? 0 : cur + 1;
  }
  __builtin_amdgcn_s_barrier();        // staging buffers are reused as gradient patches (MODE 1)
  const int hi4 = 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int vb = v0 + aoff[i] + hi4 + 8 * q;
      f32x4 bv;
      if (vb + 3 < a.V && ((((uintptr_t)(a.bias + vb)) & 15) == 0)) bv = *reinterpret_cast<const f32x4*>(a.bias + vb);
      else {
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) bv[s_] = vb + s_ < a.V ? a.bias[vb + s_] : 0.f;
      }
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        const bool ok = vb + s_ < a.V;
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j][4 * q + s_] = ok ? acc[i][j][4 * q + s_] + bv[s_] : -INFINITY;
      }
    }
  if constexpr (MODE == 0) {
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int m = m0 + boff[j] + (lane & 31);
      const bool mv = m < a.M;
      const int ym = mv ? (int)a.y[m] : -1;
      float mx = -INFINITY, tl = 0.f;
      int mi = 0x7fffffff;
      bool hit = false;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int v = v0 + aoff[i] + hi4 + (r & 3) + 8 * (r >> 2);
          const float xx = acc[i][j][r];
          const bool gt = xx > mx;
          mi = gt ? v : mi;
          mx = gt ? xx : mx;
          const bool h = v == ym;
          tl = h ? xx : tl;
          hit = hit || h;
        }
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sm += __expf(acc[i][j][r] - mx);
      if (mx == -INFINITY) sm = 0.f;
      if (hit) a.tgt_logit[m] = tl;
      float omx = __shfl_xor(mx, 32, 64), osm = __shfl_xor(sm, 32, 64);
      int omi = __shfl_xor(mi, 32, 64);
      float nm = fmaxf(mx, omx);
      float ns = (nm == -INFINITY) ? 0.f : sm * __expf(mx - nm) + osm * __expf(omx - nm);
      int ni = (omx > mx || (omx == mx && omi < mi)) ? omi : mi;
      const int prow = tv * 2 + wv;
      if (mv && lane < 32 && prow < a.npart) {
        long p = (long)prow * a.M + m;
        a.part_max[p] = nm; a.part_sum[p] = ns; a.part_idx[p] = ni;
      }
    }
  } else {
    // one 64(v) x 64(m) patch per wave (4 x 9 KiB inside the 48 KiB of staging), written back as 16-byte row segments
    constexpr int PP = Cf::PP, VEC = 8, CH = 64 / VEC;
    T* patch = reinterpret_cast<T*>(smem) + wave * (64 * PP);
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int m = m0 + boff[j] + (lane & 31);
      const bool mv = m < a.M;
      const int ym = mv ? (int)a.y[m] : -1;
      const float l = mv ? a.lse[m] : 0.f;
      const float sc = (mv && ym != a.pad) ? a.inv_norm : 0.f;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int vb = v0 + aoff[ii] + hi4;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dv = (r & 3) + 8 * (r >> 2);
          float g = (__expf(acc[ii][j][r] - l) - ((vb + dv) == ym ? 1.f : 0.f)) * sc;
          patch[(ii * 32 + hi4 + dv) * PP + j * 32 + (lane & 31)] = from_f<T>(g);
        }
      }
    }
    const int vbase = v0 + wv * 64, mbase = m0 + wm * 64;
    float* rsum = reinterpret_cast<float*>(smem + 4 * 64 * PP * (int)sizeof(T)) + wave * 64;     // behind the four patches
#pragma unroll
    for (int it = 0; it < (64 * CH) / 64; ++it) {
      const int id2 = it * 64 + lane, row = id2 / CH, ch = id2 % CH;
      const int v = vbase + row, mm = mbase + ch * VEC;
      const T* srcp = patch + row * PP + ch * VEC;
      const u32x4 seg = *reinterpret_cast<const u32x4*>(srcp);
      if (v < a.V && mm < a.M) {
        T* dst = reinterpret_cast<T*>(a.GT) + (long)v * a.ldgt + mm;
        *reinterpret_cast<u32x4*>(dst) = seg;      // M % 8 == 0, G^T rows 16-byte aligned (dispatch)
      }
      if (a.dbias) {
        // bias gradient = row sums of G^T (of the stored bf16 values, as the stand-alone row-sum kernel would read them); columns
        // beyond M hold exact zeros.  The eight lanes of a row fold by shuffles, one atomic per row and wave.
        float rs = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) rs += bf2f((bf16_t)(seg[e] & 0xffffu)) + bf2f((bf16_t)(seg[e] >> 16));
        // (v_dot2_f32_bf16 with a vector of ones would halve these instructions, but its sums did not match: measured, rejected)
        rs += __shfl_xor(rs, 1, 64);
        rs += __shfl_xor(rs, 2, 64);
        rs += __shfl_xor(rs, 4, 64);
        if (ch == 0) rsum[row] = rs;          // collected in LDS: ONE 64-lane atomic per wave below (atomics cost ~50 ns per instruction)
      }
    }
    if (a.dbias) {                            // same-wave LDS traffic is ordered: no barrier needed
      const int v = vbase + lane;
      const float rs = rsum[lane];
      if (v < a.V) atomicAdd(a.dbias + v, rs);
    }
  }
}

template <int MODE>
static int launch_gen_q(const GenArgs& a, hipStream_t st) {
  using Cf = GenCfg<bf16_t, 128>;
  const int tv = (a.V + Cf::BMV - 1) / Cf::BMV, tm = (a.M + 127) / 128;
  const size_t sm = (size_t)3 * (Cf::BMV + 128) * HBK * 2;          // 48 KiB
  hipLaunchKernelGGL((gen_kernel_q<MODE>), dim3(tv * tm), dim3(256), sm, st, a, tm);
  return check_launch();
}

template <int MODE>
static int launch_gen_h(const GenArgs& a, hipStream_t st) {
  using Cf = GenCfg<bf16_t, 128>;
  const int tv = (a.V + Cf::BMV - 1) / Cf::BMV, tm = (a.M + Cf::BNM - 1) / Cf::BNM;
  const size_t sm = (size_t)3 * (Cf::BMV + Cf::BNM) * HBK * 2;          // 72 KiB
  static bool done[2] = {false, false};
  if (!done[MODE]) { (void)hipFuncSetAttribute((const void*)gen_kernel_h<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); done[MODE] = true; }
  hipLaunchKernelGGL((gen_kernel_h<MODE>), dim3(tv * tm), dim3(Cf::NT), sm, st, a, tm);
  return check_launch();
}

template <int MODE>
static int launch_gen_p(const GenArgs& a, hipStream_t st) {
  using Cf = GenCfg<bf16_t, 128>;
  const int tv = (a.V + Cf::BMV - 1) / Cf::BMV, tm = (a.M + Cf::BNM - 1) / Cf::BNM, ntiles = tv * tm;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return VMMT_ELAUNCH;
    ncu = pr.multiProcessorCount > 8 ? (pr.multiProcessorCount / 8) * 8 : 8;
  }
  const size_t sm = (size_t)glds3_smem_bytes<Cf::BMV, Cf::BNM>();
  static bool done[2] = {false, false};
  if (!done[MODE]) { (void)hipFuncSetAttribute((const void*)gen_kernel_p<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); done[MODE] = true; }
  int grid = ncu < ((ntiles + 7) / 8) * 8 ? ncu : ((ntiles + 7) / 8) * 8;      // a multiple of 8: one slot set per XCD
  hipLaunchKernelGGL((gen_kernel_p<MODE>), dim3(grid), dim3(Cf::NT), sm, st, a, tm, ntiles);
  return check_launch();
}

// per token: logsumexp, NLL, argmax-correct; block-reduced sums are added to stats[0..2].
// Block = 16 waves x 64 tokens: wave w folds partials w, w+16, ... (coalesced along tokens) with an online
// (max, sum-exp, argmax) merge, then the 16 wave results are merged through LDS.
__global__ void __launch_bounds__(1024) gen_combine_kernel(const float* __restrict__ part_max, const float* __restrict__ part_sum,
                                   const int* __restrict__ part_idx, const float* __restrict__ tgt_logit,
                                   const long long* __restrict__ y, int M, int npart, int pad, float* __restrict__ lse,
                                   float* __restrict__ tok_nll, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int m = blockIdx.x * 64 + lane;
  float mx = -INFINITY, s = 0.f;
  int mi = 0x7fffffff;
  if (m < M) {
    // four partials per round trip (the merge is a dependent chain, the loads are not)
    for (int p0 = w; p0 < npart; p0 += 64) {
      float xv[4], sv[4];
      int iv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p = p0 + 16 * q;
        const bool ok = p < npart;
        const long o = (long)(ok ? p : p0) * M + m;
        xv[q] = ok ? part_max[o] : -INFINITY; sv[q] = ok ? part_sum[o] : 0.f; iv[q] = ok ? part_idx[o] : 0x7fffffff;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float x = xv[q], xs = sv[q];
        const int xi = iv[q];
        if (x > mx || (x == mx && xi < mi)) mi = xi;
        float nm = fmaxf(mx, x);
        if (nm != -INFINITY) s = s * __expf(mx - nm) + xs * __expf(x - nm);
        mx = nm;
      }
    }
  }
  __shared__ float smx[16][64], ssm[16][64];
  __shared__ int smi[16][64];
  smx[w][lane] = mx; ssm[w][lane] = s; smi[w][lane] = mi;
  __syncthreads();
  if (w != 0) return;
  float nll = 0.f, nw = 0.f, nc = 0.f;
  if (m < M) {
    for (int k = 1; k < 16; ++k) {
      float x = smx[k][lane], xs = ssm[k][lane];
      int xi = smi[k][lane];
      if (x > mx || (x == mx && xi < mi)) mi = xi;
      float nm = fmaxf(mx, x);
      if (nm != -INFINITY) s = s * __expf(mx - nm) + xs * __expf(x - nm);
      mx = nm;
    }
    float l = mx + logf(s);
    lse[m] = l;
    long long ym = y[m];
    bool wv = ym != pad;
    nll = wv ? l - tgt_logit[m] : 0.f;
    tok_nll[m] = nll;
    nw = wv ? 1.f : 0.f;
    nc = (wv && mi == (int)ym) ? 1.f : 0.f;
  }
  nll = wave_sum(nll); nw = wave_sum(nw); nc = wave_sum(nc);
  if (lane == 0) {
    atomicAdd(stats + VMMT_STAT_NLL, nll);
    atomicAdd(stats + VMMT_STAT_NWORDS, nw);
    atomicAdd(stats + VMMT_STAT_NCORRECT, nc);
  }
}

int g_gen_variant = -1;   // -1: automatic; 0/1/2 force a main loop (tools/gen_ab.py via vmmt_gen_set_variant)

template <class T, int MODE, int GL, int BMV_ = 128>
static int launch_gen_v(const GenArgs& a, hipStream_t st) {
  using Cf = GenCfg<T, BMV_>;
  int tv = (a.V + Cf::BMV - 1) / Cf::BMV, tm = (a.M + Cf::BNM - 1) / Cf::BNM;
  size_t sm = GL == 3 ? (size_t)glds3_smem_bytes<Cf::BMV, Cf::BNM>() : GL ? (size_t)glds_smem_bytes<Cf::BMV, Cf::BNM, GL == 2>()
                 : gemm_smem_elems<T, Cf::BMV, Cf::BNM, Cf::BK, true, true, Cf::DB>() * sizeof(T);
  if (MODE == 1 && sm < (size_t)8 * 64 * Cf::PP * sizeof(T)) sm = (size_t)8 * 64 * Cf::PP * sizeof(T);
  if (sm > 64 * 1024) {
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)gen_kernel<T, MODE, GL, BMV_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); done = true; }
  }
  hipLaunchKernelGGL((gen_kernel<T, MODE, GL, BMV_>), dim3(tv * tm), dim3(Cf::NT), sm, st, a, tm);
  return check_launch();
}

// arg-max over the vocabulary from the partial statistics of pass 0 (step-wise decoding: the next input token)
__global__ void gen_argmax_kernel(const float* __restrict__ part_max, const int* __restrict__ part_idx, int M, int npart,
                                  long long* __restrict__ out_idx, float* __restrict__ out_max) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float mx = -INFINITY;
  int mi = 0x7fffffff;
  for (int p = 0; p < npart; ++p) {
    const float x = part_max[(long)p * M + m];
    const int xi = part_idx[(long)p * M + m];
    if (x > mx || (x == mx && xi < mi)) { mx = x; mi = xi; }      // ties: lowest vocabulary index
  }
  out_idx[m] = mi;
  if (out_max) out_max[m] = mx;
}

// preconditions of the default bf16 kernel (gen_kernel_q) for this call, as launch_gen checks them
template <int MODE>
static bool gen_q_applies(const GenArgs& a) {
  const bool ok = a.K % 64 == 0 && a.ldw % 8 == 0 && a.ldo % 8 == 0 && ((((uintptr_t)a.W) | ((uintptr_t)a.O)) & 15) == 0;
  const bool gt_ok = MODE == 0 || (a.M % 8 == 0 && a.ldgt % 8 == 0 && (((uintptr_t)a.GT) & 15) == 0);
  return ok && gt_ok && (g_gen_variant < 0 || g_gen_variant == 8);
}

template <class T, int MODE>
static int launch_gen(const GenArgs& a, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    // LDS-DMA main loop: 16-byte aligned operands, K in whole 64-slabs (callers round K up over zero-padded rows)
    const bool ok = a.K % 64 == 0 && a.ldw % 8 == 0 && a.ldo % 8 == 0 && ((((uintptr_t)a.W) | ((uintptr_t)a.O)) & 15) == 0;
    // variants (tools/gen_ab.py; results are bit-identical across them):
    //   8 (default) 128 x 128 tiles, 32-deep slabs, 48 KiB, three workgroups per CU        fwd 219 us, bwd 221 us
    //   3           128 x 256 tiles, 64-deep slabs, three stages (144 KiB), one per CU      fwd 259 us, bwd 256 us
    //   7 / 6       128 x 256 tiles, 32-deep slabs, 72 KiB, two per CU (fwd only / both; the backward instantiation spills)
    //   5           persistent variant of 3 with the next tile prefetched under the epilogue (no gain)
    //   1, 2, 4     one / two LDS buffers, 256 x 256 tiles
    int v = g_gen_variant < 0 ? 8 : g_gen_variant;
    // the half-depth kernels store G^T as unconditional 16-byte segments
    const bool gt_ok = MODE == 0 || (a.M % 8 == 0 && a.ldgt % 8 == 0 && (((uintptr_t)a.GT) & 15) == 0);
    if (v == 8 && !(a.K % 32 == 0 && gt_ok)) v = 3;
    if (v == 6 && !(a.K % 32 == 0 && gt_ok)) v = 3;
    if (v == 7 && MODE == 1) v = 3;
    if (ok && v == 8) return launch_gen_q<MODE>(a, st);
    if (ok && (v == 6 || v == 7)) return launch_gen_h<MODE>(a, st);
    if (ok && v == 5 && a.K >= 128) return launch_gen_p<MODE>(a, st);
    if (ok && v == 4) return launch_gen_v<T, MODE, 2, 256>(a, st);
    if (ok && v == 3) return launch_gen_v<T, MODE, 3>(a, st);
    if (ok && v == 2) return launch_gen_v<T, MODE, 2>(a, st);
    if (ok && v == 1) return launch_gen_v<T, MODE, 1>(a, st);
  }
  return launch_gen_v<T, MODE, 0>(a, st);
}

}  // namespace vmmt

extern "C" int vmmt_gen_npart(int V) { return ((V + 127) / 128) * 2; }

extern "C" int vmmt_gen_argmax(const float* part_max, const int* part_idx, int M, int npart, int64_t* out_idx, float* out_max,
                               void* stream) {
  using namespace vmmt;
  if (!part_max || !part_idx || !out_idx || M <= 0 || npart <= 0) return VMMT_EINVAL;
  hipLaunchKernelGGL(gen_argmax_kernel, dim3((M + 127) / 128), dim3(128), 0, (hipStream_t)stream, part_max, part_idx, M, npart,
                     (long long*)out_idx, out_max);
  return check_launch();
}

extern "C" int vmmt_gen_set_variant(int v) { vmmt::g_gen_variant = v; return VMMT_OK; }

extern "C" int vmmt_gen_loss_fwd(int dtype, const void* W, int64_t ldw, const float* bias, const void* O, int64_t ldo,
                                 const int64_t* y, int M, int V, int K, int pad, float* part_max, float* part_sum,
                                 int* part_idx, float* tgt_logit, float* lse, float* tok_nll, float* stats,
                                 void* stream) {
  using namespace vmmt;
  if (!W || !bias || !O || !y || !part_max || !part_sum || !part_idx || !tgt_logit || !lse || !tok_nll || !stats ||
      M <= 0 || V <= 0 || K <= 0)
    return VMMT_EINVAL;
  GenArgs a{};
  a.W = W; a.ldw = ldw; a.bias = bias; a.O = O; a.ldo = ldo; a.y = (const long long*)y; a.M = M; a.V = V; a.K = K;
  a.pad = pad; a.npart = vmmt_gen_npart(V); a.part_max = part_max; a.part_sum = part_sum; a.part_idx = part_idx; a.tgt_logit = tgt_logit;
  hipStream_t st = (hipStream_t)stream;
  int rc = dtype == VMMT_F32 ? launch_gen<float, 0>(a, st) : dtype == VMMT_BF16 ? launch_gen<bf16_t, 0>(a, st)
                                                                                 : VMMT_EINVAL;
  if (rc) return rc;
  hipLaunchKernelGGL(gen_combine_kernel, dim3((M + 63) / 64), dim3(1024), 0, st, part_max, part_sum, part_idx,
                     tgt_logit, (const long long*)y, M, vmmt_gen_npart(V), pad, lse, tok_nll, stats);
  return check_launch();
}

extern "C" int vmmt_rowsum(int dtype, const void* X, int64_t ld, int R, int C, float* out, void* stream);

// vmmt_gen_loss_bwd + the generator bias gradient dbias[v] += sum_m G^T[v][m]: fused into the bf16 default kernel's write-out
// (saves re-reading G^T), otherwise one vmmt_rowsum launch behind the gradient pass.
extern "C" int vmmt_gen_loss_bwd_db(int dtype, const void* W, int64_t ldw, const float* bias, const void* O, int64_t ldo,
                                    const int64_t* y, int M, int V, int K, int pad, const float* lse, float inv_norm,
                                    void* GT, int64_t ldgt, float* dbias, void* stream) {
  using namespace vmmt;
  if (!W || !bias || !O || !y || !lse || !GT || !dbias || M <= 0 || V <= 0 || K <= 0 || ldgt < M) return VMMT_EINVAL;
  GenArgs a{};
  a.W = W; a.ldw = ldw; a.bias = bias; a.O = O; a.ldo = ldo; a.y = (const long long*)y; a.M = M; a.V = V; a.K = K;
  a.pad = pad; a.lse = lse; a.inv_norm = inv_norm; a.GT = GT; a.ldgt = ldgt;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VMMT_BF16 && gen_q_applies<1>(a)) {
    a.dbias = dbias;
    return launch_gen_q<1>(a, st);
  }
  int rc = dtype == VMMT_F32 ? launch_gen<float, 1>(a, st) : dtype == VMMT_BF16 ? launch_gen<bf16_t, 1>(a, st) : VMMT_EINVAL;
  if (rc) return rc;
  return vmmt_rowsum(dtype, GT, ldgt, V, M, dbias, stream);
}

extern "C" int vmmt_gen_loss_bwd(int dtype, const void* W, int64_t ldw, const float* bias, const void* O, int64_t ldo,
                                 const int64_t* y, int M, int V, int K, int pad, const float* lse, float inv_norm,
                                 void* GT, int64_t ldgt, void* stream) {
  using namespace vmmt;
  if (!W || !bias || !O || !y || !lse || !GT || M <= 0 || V <= 0 || K <= 0 || ldgt < M) return VMMT_EINVAL;
  GenArgs a{};
  a.W = W; a.ldw = ldw; a.bias = bias; a.O = O; a.ldo = ldo; a.y = (const long long*)y; a.M = M; a.V = V; a.K = K;
  a.pad = pad; a.lse = lse; a.inv_norm = inv_norm; a.GT = GT; a.ldgt = ldgt;
  hipStream_t st = (hipStream_t)stream;
  return dtype == VMMT_F32 ? launch_gen<float, 1>(a, st) : dtype == VMMT_BF16 ? launch_gen<bf16_t, 1>(a, st)
                                                                               : VMMT_EINVAL;
}

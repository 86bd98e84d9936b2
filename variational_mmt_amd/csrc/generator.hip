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
  int v_off;                                                           // pass 2 over a vocabulary chunk: W/bias/GT/dbias start at this row
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
          a.part_max[p] = nm; a.part_sum[p] = ns;
          if (a.part_idx) a.part_idx[p] = ni;
        }
      } else {
        // gradient values -> this wave's 64(v) x 64(m) patch of an LDS image [v][m] (bf16/f32), written back below as
        // whole 16-byte row segments (a lane-per-token 2-byte store per element is store-issue bound)
        const float l = mv ? a.lse[m] : 0.f;
        const float sc = (mv && ym != a.pad) ? a.inv_norm : 0.f;
        const int yl = ym - a.v_off;                              // target's row inside this vocabulary chunk
        T* patch = smem + wave * (64 * Cf::PP);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int i = 2 * hf + ii;
          const int vb = v0 + aoff[i] + hi4;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dv = (r & 3) + 8 * (r >> 2);
            float g = (__expf(acc[i][j][r] - l) - ((vb + dv) == yl ? 1.f : 0.f)) * sc;
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
  // ---- epilogue.  A lane owns one token per j and 32 vocabulary rows of it (2 x 16 accumulator registers); ~900 vector
  //      instructions per wave in the first version of this epilogue held the MFMA pipe at 35 % busy (PMC, profiles/r1_gen_kernel_
  //      investigation.txt), so everything that is not needed per element has been taken out of the per-element path:
  //      * the -inf masking of rows >= V only in the last vocabulary tile (uniform branch);
  //      * no arg-max index in the training pass (accuracy = "the target's logit is the maximum", decided by the combine kernel);
  //        the decoding pass (part_idx != NULL) still tracks it;
  //      * the target logit / the one-hot term only in waves whose 64 vocabulary rows contain one of their 64 tokens' targets
  //        (a wave-uniform branch, taken by ~13 % of the waves at V = 30 000).
  const int hi4 = 4 * (lane >> 5);
  const bool full_v = v0 + BMV <= a.V;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int vb = v0 + aoff[i] + hi4 + 8 * q;
      f32x4 bv;
      if (full_v) bv = *reinterpret_cast<const f32x4*>(a.bias + vb);        // bias + 4 * k: 16-byte aligned (hipMalloc'ed arena, offsets % 64)
      else {
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) bv[s_] = vb + s_ < a.V ? a.bias[vb + s_] : -INFINITY;
      }
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j][4 * q + s_] += bv[s_];      // rows >= V: -inf (their accumulators are finite)
    }
  const int vlo = v0 + wv * 64;                                   // this wave's 64 vocabulary rows
  if constexpr (MODE == 0) {
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int m = m0 + boff[j] + (lane & 31);
      const bool mv = m < a.M;
      const int ym = mv ? (int)a.y[m] : -1;
      float mx = -INFINITY;
      int mi = 0x7fffffff;
      if (a.part_idx) {                                           // decoding: arg-max index, ties -> lowest vocabulary index
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int v = v0 + aoff[i] + hi4 + (r & 3) + 8 * (r >> 2);
            const float xx = acc[i][j][r];
            const bool gt = xx > mx;
            mi = gt ? v : mi;
            mx = gt ? xx : mx;
          }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, acc[i][j][r]), acc[i][j][r + 1]);      // v_max3_f32
      }
      if (__any(ym >= vlo && ym < vlo + 64)) {                    // some token of this wave has its target in these rows
        float tl = 0.f;
        bool hit = false;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool h = (v0 + aoff[i] + hi4 + (r & 3) + 8 * (r >> 2)) == ym;
            tl = h ? acc[i][j][r] : tl;
            hit = hit || h;
          }
        if (hit) a.tgt_logit[m] = tl;
      }
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sm += __expf(acc[i][j][r] - mx);
      if (mx == -INFINITY) sm = 0.f;
      float omx = __shfl_xor(mx, 32, 64), osm = __shfl_xor(sm, 32, 64);
      int omi = __shfl_xor(mi, 32, 64);
      float nm = fmaxf(mx, omx);
      float ns = (nm == -INFINITY) ? 0.f : sm * __expf(mx - nm) + osm * __expf(omx - nm);
      int ni = (omx > mx || (omx == mx && omi < mi)) ? omi : mi;
      const int prow = tv * 2 + wv;
      if (mv && lane < 32 && prow < a.npart) {
        long p = (long)prow * a.M + m;
        a.part_max[p] = nm; a.part_sum[p] = ns;
        if (a.part_idx) a.part_idx[p] = ni;
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
      const int yraw = mv ? (int)a.y[m] : a.pad;
      const int ym = mv ? yraw - a.v_off : -1;                    // target's row inside this vocabulary chunk (or outside it)
      const float l = mv ? a.lse[m] : 0.f;
      const float sc = (mv && yraw != a.pad) ? a.inv_norm : 0.f;
      if (__any(ym >= vlo && ym < vlo + 64)) {                    // with the one-hot term (rare)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int vb = v0 + aoff[ii] + hi4;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dv = (r & 3) + 8 * (r >> 2);
            const float g = (__expf(acc[ii][j][r] - l) - ((vb + dv) == ym ? 1.f : 0.f)) * sc;
            patch[(ii * 32 + hi4 + dv) * PP + j * 32 + (lane & 31)] = from_f<T>(g);
          }
        }
      } else {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dv = (r & 3) + 8 * (r >> 2);
            patch[(ii * 32 + hi4 + dv) * PP + j * 32 + (lane & 31)] = from_f<T>(__expf(acc[ii][j][r] - l) * sc);
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

// per token: logsumexp, NLL, argmax-correct; block-reduced sums are added to stats[0..2].  part_idx may be NULL.
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
        xv[q] = ok ? part_max[o] : -INFINITY; sv[q] = ok ? part_sum[o] : 0.f; iv[q] = (ok && part_idx) ? part_idx[o] : 0x7fffffff;
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
    // accuracy (Loss.py:150-160: pred = scores.max(1)[1]; pred.eq(target)).  With the arg-max partials: their index; without
    // (training pass, part_idx == NULL): the target's logit is the maximum (an exact tie counts as correct)
    nc = (wv && (part_idx ? mi == (int)ym : tgt_logit[m] >= mx)) ? 1.f : 0.f;
  }
  nll = wave_sum(nll); nw = wave_sum(nw); nc = wave_sum(nc);
  if (lane == 0) {
    atomicAdd(stats + VMMT_STAT_NLL, nll);
    atomicAdd(stats + VMMT_STAT_NWORDS, nw);
    atomicAdd(stats + VMMT_STAT_NCORRECT, nc);
  }
}

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
// (one wave per token: the ~2 V / 128 partials of a token are scanned by its 64 lanes -- as one lane per token this kernel took
//  74 us per decoded position at V = 30 000, 38 % of it)
__global__ void __launch_bounds__(256) gen_argmax_kernel(const float* __restrict__ part_max, const int* __restrict__ part_idx, int M,
                                                         int npart, long long* __restrict__ out_idx, float* __restrict__ out_max) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= M) return;
  float mx = -INFINITY;
  int mi = 0x7fffffff;
  for (int p = lane; p < npart; p += 64) {
    const float x = part_max[(long)p * M + m];
    const int xi = part_idx[(long)p * M + m];
    if (x > mx || (x == mx && xi < mi)) { mx = x; mi = xi; }      // ties: lowest vocabulary index
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ox = __shfl_xor(mx, off, 64);
    const int oi = __shfl_xor(mi, off, 64);
    if (ox > mx || (ox == mx && oi < mi)) { mx = ox; mi = oi; }
  }
  if (lane == 0) {
    out_idx[m] = mi == 0x7fffffff ? 0 : mi;                       // (a row of NaNs has no maximum: a valid id all the same -- it is fed to a table lookup)
    if (out_max) out_max[m] = mx;
  }
}

// preconditions of the bf16 throughput kernel (gen_kernel_q): LDS-DMA operands (16-byte aligned rows, K in whole 32-slabs) and,
// for the gradient pass, G^T rows that take unconditional 16-byte segments; everything else (fp32 parity mode, ragged shapes)
// runs the register-staged general kernel
template <int MODE>
static bool gen_q_applies(const GenArgs& a) {
  const bool ok = a.K % 64 == 0 && a.ldw % 8 == 0 && a.ldo % 8 == 0 && ((((uintptr_t)a.W) | ((uintptr_t)a.O)) & 15) == 0;
  const bool gt_ok = MODE == 0 || (a.M % 8 == 0 && a.ldgt % 8 == 0 && (((uintptr_t)a.GT) & 15) == 0);
  return ok && gt_ok;
}

template <class T, int MODE>
static int launch_gen(const GenArgs& a, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    if (gen_q_applies<MODE>(a)) return launch_gen_q<MODE>(a, st);
  }
  return launch_gen_v<T, MODE, 0>(a, st);
}

}  // namespace vmmt

extern "C" int vmmt_gen_npart(int V) { return ((V + 127) / 128) * 2; }

extern "C" int vmmt_gen_argmax(const float* part_max, const int* part_idx, int M, int npart, int64_t* out_idx, float* out_max,
                               void* stream) {
  using namespace vmmt;
  if (!part_max || !part_idx || !out_idx || M <= 0 || npart <= 0) return VMMT_EINVAL;
  hipLaunchKernelGGL(gen_argmax_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, part_max, part_idx, M, npart,
                     (long long*)out_idx, out_max);
  return check_launch();
}

extern "C" int vmmt_gen_loss_fwd(int dtype, const void* W, int64_t ldw, const float* bias, const void* O, int64_t ldo,
                                 const int64_t* y, int M, int V, int K, int pad, float* part_max, float* part_sum,
                                 int* part_idx, float* tgt_logit, float* lse, float* tok_nll, float* stats,
                                 void* stream) {
  using namespace vmmt;
  if (!W || !bias || !O || !y || !part_max || !part_sum || !tgt_logit || !lse || !tok_nll || !stats ||
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
                                    void* GT, int64_t ldgt, float* dbias, int v_off, void* stream) {
  using namespace vmmt;
  if (!W || !bias || !O || !y || !lse || !GT || !dbias || M <= 0 || V <= 0 || K <= 0 || ldgt < M || v_off < 0) return VMMT_EINVAL;
  GenArgs a{};
  a.v_off = v_off;
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

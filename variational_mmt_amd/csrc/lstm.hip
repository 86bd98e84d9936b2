// LSTM time-step kernels (forward and backward) for gfx950.
//
// The reference runs nn.LSTM through cuDNN (encoder: onmt/Models.py:124-129,140-147 with packed sequences;
// decoder: onmt/VI_Model1.py:106,149-152).  Here the input projections x W_ih^T + b are one big GEMM per layer
// (gemm.hip) and only the sequential part runs per time step: one launch per step, cut at the all-to-all seam
// (every hidden unit of step t needs all of h_{t-1}); a launch boundary (~1.5 us) is cheaper on this chip than an
// in-kernel grid barrier (~4-5 us).  Each step is a fused [B x H] x [H x 4H] MFMA GEMM + gate non-linearities +
// state update; the four gate tiles (i,f,g,o) of a hidden unit land in the same lane, so the cell update is
// lane-local.  Both directions of a bidirectional layer run in one launch (blockIdx.z).
//
// Packed-sequence semantics (pack_padded_sequence, Models.py:140-147) are reproduced by masking: a sentence's
// state is not updated and its output is zero at positions >= its length; the reverse direction starts from the
// zero state at the sentence's last token.
#include "common.hpp"
#include "vmmt.h"

namespace vmmt {

struct StepDirF {
  const void* h_prev; long ld_hprev;     // T [B][ld]   (out[t-1] / out[t+1] / h0)
  const float* c_prev; long ld_cprev;    // f32 [B][ld]
  const void* w_hh; long ld_w;           // T [4H][ld] (k contiguous)
  const float* gx; long ld_gx;           // f32 [B][ld]: x_t W_ih^T + b_ih + b_hh (gate-major columns g*H+u)
  const float* gx2; long ld_gx2;         // f32 [B][ld] or null: per-sentence addend (z W_z^T + b, constant over time)
  void* gates; long ld_gates;            // T [B][ld]: saved post-activation i,f,g,o
  float* c_out; long ld_c;               // f32 [B][ld]
  void* h_out; long ld_h;                // T [B][ld] (masked output)
  void* h_n; long ld_hn;                 // T [B][ld] or null: final state capture
  float* c_n; long ld_cn;
  int t;                                 // time index of this step (for masking)
  int capture;                           // 0 none, 1 when t == len-1, 2 when t == 0, 3 always
};
struct StepArgsF {
  StepDirF d[2];
  const long long* lens;                 // [B] or null (decoder: no masking)
  int B, H;
};

// cell forward for one (batch row b, hidden unit u): lane-local
template <class T>
__device__ __forceinline__ void cell_fwd(const StepDirF& d, const long long* lens, int b, int u, int H, float pi, float pf,
                                         float pg, float po) {
  const float* gx = d.gx + (long)b * d.ld_gx + u;
  pi += gx[0]; pf += gx[H]; pg += gx[2 * H]; po += gx[3 * (long)H];
  if (d.gx2) {
    const float* g2 = d.gx2 + (long)b * d.ld_gx2 + u;
    pi += g2[0]; pf += g2[H]; pg += g2[2 * H]; po += g2[3 * (long)H];
  }
  float i = cell_sigmoid_(pi), f = cell_sigmoid_(pf), g = cell_tanh_(pg), o = cell_sigmoid_(po);
  float cp = d.c_prev ? d.c_prev[(long)b * d.ld_cprev + u] : 0.f;
  float c = f * cp + i * g;
  float h = o * cell_tanh_(c);
  bool valid = true;
  long long len = 0;
  if (lens) { len = lens[b]; valid = d.t < len; }
  T* gs = reinterpret_cast<T*>(d.gates) + (long)b * d.ld_gates + u;
  gs[0] = from_f<T>(i); gs[H] = from_f<T>(f); gs[2 * H] = from_f<T>(g); gs[3 * (long)H] = from_f<T>(o);
  d.c_out[(long)b * d.ld_c + u] = valid ? c : cp;          // frozen state at pads
  reinterpret_cast<T*>(d.h_out)[(long)b * d.ld_h + u] = from_f<T>(valid ? h : 0.f);
  bool cap = d.capture == 3 || (d.capture == 1 && d.t == len - 1) || (d.capture == 2 && d.t == 0);
  if (cap && d.h_n) {
    reinterpret_cast<T*>(d.h_n)[(long)b * d.ld_hn + u] = from_f<T>(h);
    d.c_n[(long)b * d.ld_cn + u] = c;
  }
}

template <class T>
__global__ void __launch_bounds__(128) lstm_step_fwd_kernel(StepArgsF a) {
  constexpr int BK = 32, NT = 128, BM = 64, BU = 32, BN = 4 * BU;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const StepDirF& d = a.d[blockIdx.z];
  const int B = a.B, H = a.H;
  const int m0 = blockIdx.x * BM, u0 = blockIdx.y * BU;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int aoff[1] = {wave * 32};
  int boff[4] = {0, BU, 2 * BU, 3 * BU};
  f32x16 acc[1][4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][g][r] = 0.f;
  LinearMap amap{m0, B};
  GateMap bmap{u0, H, BU};
  gemm_mainloop<T, BM, BN, BK, NT, true, true, 1, 4>((const T*)d.h_prev, d.ld_hprev, amap, (const T*)d.w_hh, d.ld_w,
                                                      bmap, H, 0, 0, aoff, boff, acc, smem);
  const int u = u0 + (lane & 31);
  if (u >= H) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int b = m0 + aoff[0] + acc_row(r, lane);
    if (b >= B) continue;
    cell_fwd<T>(d, a.lens, b, u, H, acc[0][0][r], acc[0][1][r], acc[0][2][r], acc[0][3][r]);
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward step: dh_rec = dgates_next W_hh (GEMM over 4H), then the cell backward for step t, lane-local.
// ------------------------------------------------------------------------------------------------------------
struct StepDirB {
  const void* dgates_next; long ld_dgn;  // T [B][ld] or null (first processed step: no recurrent term)
  const void* w_hh_t; long ld_wt;        // T [H][ld]: W_hh^T shadow (k = gate index contiguous)
  const void* dh_above; long ld_dha;     // T [B][ld] or null: gradient arriving at this step's output
  const void* gates; long ld_gates;      // T saved activations of step t
  const float* c_t; long ld_ct;          // f32 c after step t
  const float* c_prev; long ld_cp;       // f32 c before step t (null = zeros)
  float* dc_carry; long ld_dcc;          // f32 [B][ld] in/out: dL/dc flowing to the previous step
  void* dgates_out; long ld_dgo;         // T [B][ld]: dL/d(gate pre-activations) of step t
  const float* dh_n; long ld_dhn;        // f32 [B][ld] or null: gradient of the captured final state
  const float* dc_n; long ld_dcn;
  float* dh0_out; long ld_dh0;           // mode 1 only: f32 [B][ld] receives dgates_next W_hh
  int t, inject;                         // inject: 0 none, 1 when t == len-1, 2 when t == 0, 3 always
};
struct StepArgsB {
  StepDirB d[2];
  const long long* lens;
  int B, H, mode;                        // mode 0: GEMM + cell backward; 1: GEMM only -> dh0_out
};

// cell backward for one (b, u) given dh = dgates_next W_hh (recurrent part): lane-local
template <class T>
__device__ __forceinline__ void cell_bwd(const StepDirB& d, const long long* lens, int mode, int b, int u, int H, float dh) {
  if (mode == 1) { d.dh0_out[(long)b * d.ld_dh0 + u] = dh; return; }
  long long len = 0;
  bool valid = true;
  if (lens) { len = lens[b]; valid = d.t < len; }
  T* dgo = reinterpret_cast<T*>(d.dgates_out) + (long)b * d.ld_dgo + u;
  float* dcc = d.dc_carry + (long)b * d.ld_dcc + u;
  if (!valid) {
    dgo[0] = T(0); dgo[H] = T(0); dgo[2 * H] = T(0); dgo[3 * (long)H] = T(0);
    *dcc = 0.f;
    return;
  }
  float dc = *dcc;
  if (d.dh_above) dh += to_f<T>(reinterpret_cast<const T*>(d.dh_above)[(long)b * d.ld_dha + u]);
  bool inj = d.inject == 3 || (d.inject == 1 && d.t == len - 1) || (d.inject == 2 && d.t == 0);
  if (inj && d.dh_n) { dh += d.dh_n[(long)b * d.ld_dhn + u]; dc += d.dc_n[(long)b * d.ld_dcn + u]; }
  const T* gs = reinterpret_cast<const T*>(d.gates) + (long)b * d.ld_gates + u;
  float i = to_f<T>(gs[0]), f = to_f<T>(gs[H]), g = to_f<T>(gs[2 * H]), o = to_f<T>(gs[3 * (long)H]);
  float c = d.c_t[(long)b * d.ld_ct + u];
  float cp = d.c_prev ? d.c_prev[(long)b * d.ld_cp + u] : 0.f;
  float tc = cell_tanh_(c);
  float d_o = dh * tc;
  dc += dh * o * (1.f - tc * tc);
  float d_i = dc * g, d_f = dc * cp, d_g = dc * i;
  dgo[0] = from_f<T>(d_i * i * (1.f - i));
  dgo[H] = from_f<T>(d_f * f * (1.f - f));
  dgo[2 * H] = from_f<T>(d_g * (1.f - g * g));
  dgo[3 * (long)H] = from_f<T>(d_o * o * (1.f - o));
  *dcc = dc * f;
}

template <class T>
__global__ void __launch_bounds__(256) lstm_step_bwd_kernel(StepArgsB a) {
  constexpr int BK = 32, NT = 256, BM = 64, BN = 64;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const StepDirB& d = a.d[blockIdx.z];
  const int B = a.B, H = a.H;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int aoff[1] = {(wave >> 1) * 32};
  int boff[1] = {(wave & 1) * 32};
  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  if (d.dgates_next) {
    LinearMap amap{m0, B}, bmap{n0, H};
    gemm_mainloop<T, BM, BN, BK, NT, true, true, 1, 1>((const T*)d.dgates_next, d.ld_dgn, amap, (const T*)d.w_hh_t,
                                                        d.ld_wt, bmap, 4 * H, 0, 0, aoff, boff, acc, smem);
  }
  const int u = n0 + boff[0] + (lane & 31);
  if (u >= H) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int b = m0 + aoff[0] + acc_row(r, lane);
    if (b >= B) continue;
    cell_bwd<T>(d, a.lens, a.mode, b, u, H, acc[0][0][r]);
  }
}

// ==============================================================================================================
// Latency-optimised bf16 step kernels.  A step is a tiny GEMM ([B x K] x [K x N]) on the critical path of the
// recurrence, so what matters is the number of dependent memory round trips and the issue time of the loads:
//   * tiles of 32 batch rows x 16 hidden units: 256 workgroups at B = 256, H = 512 (one per CU);
//   * the operands of a whole K chunk are staged with direct-to-LDS loads (global_load_lds_dwordx4, one row piece per
//     wave instruction, no VGPR round trip), ALL issued before a single wait.  In-kernel timestamps (tools/probe/
//     lstm_probe.hip) showed that the first version spent 110 ns PER PIECE in address arithmetic and branches (2.7 us of
//     a 5 us forward step, 7 us of a 14 us backward step): here every piece address is a wave-uniform base (scalar
//     registers, rows clamped instead of skipped) plus one per-lane offset, fully unrolled;
//   * v_mfma_f32_16x16x32_bf16 straight from LDS (row pitch = chunk bytes + 16: conflict-free ds_read_b128);
//     its C layout (col = lane&15, row = 4*(lane>>4)+reg) puts the four gates of a hidden unit in one lane;
//   * the four waves split the tile as 2 row halves x 2 K halves; the K halves are folded through LDS and every wave
//     finishes two of its four accumulator rows, so the cell epilogue's loads and stores are spread over all lanes.
// W_hh (2 MB at H = 512) stays L2-resident across the steps; h_{t-1} / dgates are re-read from L2 by the blocks that
// share them.
// ==============================================================================================================
#ifdef VMMT_PROBE
__device__ unsigned long long vmmt_probe_ts[64 * 16];
#define VMMT_TS(i) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) \
    vmmt_probe_ts[(blockIdx.x & 63) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define VMMT_TS(i) do {} while (0)
#endif

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_cvoid_t;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const char* src, char* dst_uniform) {
  __builtin_amdgcn_global_load_lds((glb_cvoid_t*)src, (lds_void_t*)dst_uniform, 16, 0, 0);
}

// epilogue operands of the forward cell, fetched right behind the staging loads so that their latency hides behind them
struct CellInF { float gx[4]; float cp; long long len; };
__device__ __forceinline__ CellInF cell_fwd_load(const StepDirF& d, const long long* lens, int b, int u, int H, int B) {
  CellInF in;
  const bool ok = b < B;
  const float* gx = d.gx + (long)(ok ? b : 0) * d.ld_gx + u;
  in.gx[0] = gx[0]; in.gx[1] = gx[H]; in.gx[2] = gx[2 * H]; in.gx[3] = gx[3 * (long)H];
  if (d.gx2) {
    const float* g2 = d.gx2 + (long)(ok ? b : 0) * d.ld_gx2 + u;
    in.gx[0] += g2[0]; in.gx[1] += g2[H]; in.gx[2] += g2[2 * H]; in.gx[3] += g2[3 * (long)H];
  }
  in.cp = d.c_prev ? d.c_prev[(long)(ok ? b : 0) * d.ld_cprev + u] : 0.f;
  in.len = lens ? lens[ok ? b : 0] : 0;
  return in;
}
__device__ __forceinline__ void cell_fwd_apply(const StepDirF& d, bool has_lens, const CellInF& in, int b, int u, int H,
                                               float pi, float pf, float pg, float po) {
  const LstmCell cell = lstm_cell_math(pi + in.gx[0], pf + in.gx[1], pg + in.gx[2], po + in.gx[3], in.cp);
  const float i = cell.i, f = cell.f, g = cell.g, o = cell.o, c = cell.c, h = cell.h;
  const bool valid = !has_lens || d.t < in.len;
  bf16_t* gs = reinterpret_cast<bf16_t*>(d.gates) + (long)b * d.ld_gates + u;
  gs[0] = f2bf(i); gs[H] = f2bf(f); gs[2 * H] = f2bf(g); gs[3 * (long)H] = f2bf(o);
  d.c_out[(long)b * d.ld_c + u] = valid ? c : in.cp;
  reinterpret_cast<bf16_t*>(d.h_out)[(long)b * d.ld_h + u] = f2bf(valid ? h : 0.f);
  const bool cap = d.capture == 3 || (d.capture == 1 && d.t == in.len - 1) || (d.capture == 2 && d.t == 0);
  if (cap && d.h_n) {
    reinterpret_cast<bf16_t*>(d.h_n)[(long)b * d.ld_hn + u] = f2bf(h);
    d.c_n[(long)b * d.ld_cn + u] = c;
  }
}

// W_hh rows are staged UNPADDED (a 64-row x 1-KiB image is exactly 64 KiB, so a 64-KiB GEMM workgroup of the side stream
// still fits beside a step workgroup on the same CU); bank conflicts of the fragment reads are removed by XOR-ing the
// 16-byte chunk index with a row key, applied to the per-lane SOURCE address of the LDS-DMA and again to the reads.
template <int KC> struct FwdCfg {
  static constexpr int ROWB = KC * 2;                          // bytes per staged row
  static constexpr int NCH = ROWB / 16;                        // 16-byte chunks per row
  static constexpr int RPB = ROWB >= 256 ? 1 : 256 / ROWB;     // rows per 256-byte bank row
  static constexpr int KMASK = (NCH < 16 ? NCH : 16) - 1;
  static constexpr int LANES = NCH < 64 ? NCH : 64;            // active lanes of a row piece
  static constexpr int NKS = KC / 32;                          // MFMA K steps per chunk
  static constexpr int KQ = NKS >= 4 ? NKS / 4 : 1;            // K steps per wave (the 8 waves are 2 row halves x 4 K quarters)
  static constexpr int LDS = 64 * ROWB > 32768 ? 64 * ROWB : 32768;   // >= the 32-KiB fold buffer
  static __device__ __forceinline__ int key(int row) { return (row / RPB) & KMASK; }
};

// Forward: gates[32 x (4 x 16)] += h_prev[32 x H] W_hh[(4 x 16) x H]^T, K in chunks of KC (H % KC == 0).
// W_hh chunk: LDS-DMA (shared by the two row-half waves); h_prev: each wave needs a private 16-row x K-quarter slice, loaded
// straight into MFMA A fragments (no LDS round trip, no sharing to exploit).
template <int KC>
__global__ void __launch_bounds__(512) lstm_step_fwd_fast(StepArgsF a) {
  using Cf = FwdCfg<KC>;
  constexpr int ROWB = Cf::ROWB, KQ = Cf::KQ;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const StepDirF& d = a.d[blockIdx.z];
  const int B = a.B, H = a.H;
  const int m0 = blockIdx.x * 32, u0 = blockIdx.y * 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wm = wave & 1, wk = wave >> 1;
  const int n = lane & 15, kg = lane >> 4;
  const int u = u0 + n;
  const int brow = m0 + wm * 16 + kg * 4 + wk;                     // the accumulator row (register wk) this lane finishes
  const bool kact = wk * KQ < Cf::NKS;
  VMMT_TS(0);
  f32x4_t acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const char* wp = reinterpret_cast<const char*>(d.w_hh);
  const char* ap = reinterpret_cast<const char*>(d.h_prev) + ((long)min(m0 + wm * 16 + n, B - 1) * d.ld_hprev + kg * 8) * 2;
  CellInF in;
  for (int k0 = 0; k0 < H; k0 += KC) {
    if (k0 > 0) __syncthreads();
    if (lane < Cf::LANES) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = wave + 8 * j;                                 // staged row: gate r >> 4, unit u0 + (r & 15)
        const long row = (long)(r >> 4) * H + u0 + (r & 15);
        dma16(wp + (row * d.ld_w + k0) * 2 + ((lane ^ Cf::key(r)) * 16), lds + r * ROWB);
      }
    }
    u32x4 af[KQ];
    if (kact) {
#pragma unroll
      for (int q = 0; q < KQ; ++q) af[q] = *reinterpret_cast<const u32x4*>(ap + (long)(k0 + (wk * KQ + q) * 32) * 2);
    }
    if (k0 == 0) in = cell_fwd_load(d, a.lens, brow, u, H, B);
    VMMT_TS(1);
    __syncthreads();                                                // hipcc drains vmcnt(0) here: all pieces landed
    VMMT_TS(2);
    if (kact) {
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        const int c = (wk * KQ + q) * 4 + kg;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = g * 16 + n;
          const bf16x8 bv = *reinterpret_cast<const bf16x8*>(lds + row * ROWB + ((c ^ Cf::key(row)) * 16));
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[q]), bv, acc[g], 0, 0, 0);
        }
      }
    }
  }
  // fold the four K quarters through LDS; every lane then finishes ONE (row, unit) cell
  VMMT_TS(3);
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);                      // [wk][wm][g][reg][lane]
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(((wk * 2 + wm) * 4 + g) * 4 + r) * 64 + lane] = acc[g][r];
  __syncthreads();
  if (brow < B) {
    float p[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      p[g] = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) p[g] += red[(((w * 2 + wm) * 4 + g) * 4 + wk) * 64 + lane];
    }
    cell_fwd_apply(d, a.lens != nullptr, in, brow, u, H, p[0], p[1], p[2], p[3]);
  }
  VMMT_TS(7);
}

struct CellInB { float i, f, g, o, c, cp, dha, dcc, dhn, dcn; long long len; };
__device__ __forceinline__ CellInB cell_bwd_load(const StepDirB& d, const long long* lens, int b, int u, int H, int B) {
  CellInB in;
  const long bb = b < B ? b : 0;
  const bf16_t* gs = reinterpret_cast<const bf16_t*>(d.gates) + bb * d.ld_gates + u;
  in.i = bf2f(gs[0]); in.f = bf2f(gs[H]); in.g = bf2f(gs[2 * H]); in.o = bf2f(gs[3 * (long)H]);
  in.c = d.c_t[bb * d.ld_ct + u];
  in.cp = d.c_prev ? d.c_prev[bb * d.ld_cp + u] : 0.f;
  in.dha = d.dh_above ? bf2f(reinterpret_cast<const bf16_t*>(d.dh_above)[bb * d.ld_dha + u]) : 0.f;
  in.dcc = d.dc_carry[bb * d.ld_dcc + u];
  in.dhn = d.dh_n ? d.dh_n[bb * d.ld_dhn + u] : 0.f;
  in.dcn = d.dh_n ? d.dc_n[bb * d.ld_dcn + u] : 0.f;
  in.len = lens ? lens[bb] : 0;
  return in;
}
__device__ __forceinline__ void cell_bwd_apply(const StepDirB& d, bool has_lens, const CellInB& in, int b, int u, int H, float dh) {
  const bool valid = !has_lens || d.t < in.len;
  bf16_t* dgo = reinterpret_cast<bf16_t*>(d.dgates_out) + (long)b * d.ld_dgo + u;
  float* dcc = d.dc_carry + (long)b * d.ld_dcc + u;
  if (!valid) {
    dgo[0] = 0; dgo[H] = 0; dgo[2 * H] = 0; dgo[3 * (long)H] = 0;
    *dcc = 0.f;
    return;
  }
  float dc = in.dcc;
  dh += in.dha;
  const bool inj = d.inject == 3 || (d.inject == 1 && d.t == in.len - 1) || (d.inject == 2 && d.t == 0);
  if (inj) { dh += in.dhn; dc += in.dcn; }
  const LstmCellGrad gr = lstm_cell_bwd_math(in.i, in.f, in.g, in.o, in.c, in.cp, dh, dc);
  dgo[0] = f2bf(gr.di);
  dgo[H] = f2bf(gr.df);
  dgo[2 * H] = f2bf(gr.dg);
  dgo[3 * (long)H] = f2bf(gr.d_o);
  *dcc = gr.dc_prev;
}

// Backward: dh[32 x 16] = dgates_next[32 x 4H] W_hh^T[16 x 4H]^T over K = 4H in rounds of KCB (one round up to H = 512), then
// the cell backward of step t.  W_hh^T rows (16 x up to 4 KiB, unpadded + swizzled as above) by LDS-DMA; the dgates_next slice of a
// wave (16 rows x a K quarter) straight into MFMA A fragments.
template <int KCB> struct BwdCfg {
  static constexpr int ROWB = KCB * 2;
  static constexpr int PPR = ROWB >= 1024 ? ROWB / 1024 : 1;   // 1-KiB pieces per row
  static constexpr int LANES = ROWB >= 1024 ? 64 : ROWB / 16;
  static constexpr int PER = (16 * PPR + 7) / 8;               // pieces per wave
  static constexpr int NKS = KCB / 32, KQ = NKS / 4;           // KCB >= 128
  static constexpr int LDS = 16 * ROWB > 8192 ? 16 * ROWB : 8192;
};

template <int KCB>
__global__ void __launch_bounds__(512) lstm_step_bwd_fast(StepArgsB a) {
  using Cf = BwdCfg<KCB>;
  constexpr int ROWB = Cf::ROWB, KQ = Cf::KQ, PPR = Cf::PPR;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const StepDirB& d = a.d[blockIdx.z];
  const int B = a.B, H = a.H, K = 4 * a.H;
  const int m0 = blockIdx.x * 32, u0 = blockIdx.y * 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wm = wave & 1, wk = wave >> 1;
  const int n = lane & 15, kg = lane >> 4;
  const int u = u0 + n;
  const int brow = m0 + wm * 16 + kg * 4 + wk;
  VMMT_TS(0);
  CellInB in;
  f32x4_t acc0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (d.dgates_next) {
    const char* wp = reinterpret_cast<const char*>(d.w_hh_t);
    const char* ap = reinterpret_cast<const char*>(d.dgates_next) + ((long)min(m0 + wm * 16 + n, B - 1) * d.ld_dgn + kg * 8) * 2;
    for (int k0 = 0; k0 < K; k0 += KCB) {
      if (k0 > 0) __syncthreads();
      if (lane < Cf::LANES) {
#pragma unroll
        for (int j = 0; j < Cf::PER; ++j) {
          const int p = wave + 8 * j;
          if (p < 16 * PPR) {
            const int r = p / PPR, sg = p % PPR;
            dma16(wp + ((long)(u0 + r) * d.ld_wt + k0) * 2 + sg * 1024 + ((lane ^ (r & 15)) * 16), lds + r * ROWB + sg * 1024);
          }
        }
      }
      u32x4 af[KQ];
#pragma unroll
      for (int q = 0; q < KQ; ++q) af[q] = *reinterpret_cast<const u32x4*>(ap + (long)(k0 + (wk * KQ + q) * 32) * 2);
      if (k0 == 0 && a.mode == 0) in = cell_bwd_load(d, a.lens, brow, u, H, B);
      VMMT_TS(1);
      __syncthreads();
      VMMT_TS(2);
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        const int c = (wk * KQ + q) * 4 + kg;
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(lds + n * ROWB + ((c ^ n) * 16));
        f32x4_t& acc = (q & 1) ? acc1 : acc0;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[q]), bv, acc, 0, 0, 0);
      }
      VMMT_TS(3);
    }
  } else if (a.mode == 0) {
    in = cell_bwd_load(d, a.lens, brow, u, H, B);
  }
  // fold the four K quarters through LDS; every lane then finishes ONE (row, unit) cell
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);                      // [wk][wm][reg][lane]
#pragma unroll
  for (int r = 0; r < 4; ++r) red[((wk * 2 + wm) * 4 + r) * 64 + lane] = acc0[r] + acc1[r];
  __syncthreads();
  if (brow < B) {
    float dh = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) dh += red[((w * 2 + wm) * 4 + wk) * 64 + lane];
    if (a.mode == 1) d.dh0_out[(long)brow * d.ld_dh0 + u] = dh;
    else cell_bwd_apply(d, a.lens != nullptr, in, brow, u, H, dh);
  }
  VMMT_TS(7);
}

// largest chunk in {kmax, kmax/2, ..., kmin} that divides K
static int fast_chunk(int K, int kmax, int kmin) { for (int kc = kmax; kc >= kmin; kc >>= 1) if (K % kc == 0) return kc; return 0; }

template <int KC>
static int launch_fwd_fast(const StepArgsF& a, int ndir, hipStream_t st) {
  constexpr int sm = FwdCfg<KC>::LDS;
  static bool attr_set = false;
  if (!attr_set && sm >= 64 * 1024) { (void)hipFuncSetAttribute((const void*)lstm_step_fwd_fast<KC>, hipFuncAttributeMaxDynamicSharedMemorySize, sm); attr_set = true; }
  hipLaunchKernelGGL(lstm_step_fwd_fast<KC>, dim3((a.B + 31) / 32, a.H / 16, ndir), dim3(512), sm, st, a);
  return check_launch();
}
template <int KCB>
static int launch_bwd_fast(const StepArgsB& a, int ndir, hipStream_t st) {
  constexpr int sm = BwdCfg<KCB>::LDS;
  static bool attr_set = false;
  if (!attr_set && sm >= 64 * 1024) { (void)hipFuncSetAttribute((const void*)lstm_step_bwd_fast<KCB>, hipFuncAttributeMaxDynamicSharedMemorySize, sm); attr_set = true; }
  hipLaunchKernelGGL(lstm_step_bwd_fast<KCB>, dim3((a.B + 31) / 32, a.H / 16, ndir), dim3(512), sm, st, a);
  return check_launch();
}

static bool al16(const void* p, long ld_elems) { return (((uintptr_t)p) & 15) == 0 && (ld_elems * 2) % 16 == 0; }

static void fill_f(StepDirF& o, const vmmt_lstm_dir_fwd& i) {
  o.h_prev = i.h_prev; o.ld_hprev = i.ld_hprev; o.c_prev = (const float*)i.c_prev; o.ld_cprev = i.ld_cprev;
  o.w_hh = i.w_hh; o.ld_w = i.ld_w; o.gx = (const float*)i.gx; o.ld_gx = i.ld_gx; o.gx2 = (const float*)i.gx2;
  o.ld_gx2 = i.ld_gx2; o.gates = i.gates;
  o.ld_gates = i.ld_gates; o.c_out = (float*)i.c_out; o.ld_c = i.ld_c; o.h_out = i.h_out; o.ld_h = i.ld_h;
  o.h_n = i.h_n; o.ld_hn = i.ld_hn; o.c_n = (float*)i.c_n; o.ld_cn = i.ld_cn; o.t = i.t; o.capture = i.capture;
}
static void fill_b(StepDirB& o, const vmmt_lstm_dir_bwd& i) {
  o.dgates_next = i.dgates_next; o.ld_dgn = i.ld_dgn; o.w_hh_t = i.w_hh_t; o.ld_wt = i.ld_wt;
  o.dh_above = i.dh_above; o.ld_dha = i.ld_dha; o.gates = i.gates; o.ld_gates = i.ld_gates;
  o.c_t = (const float*)i.c_t; o.ld_ct = i.ld_ct; o.c_prev = (const float*)i.c_prev; o.ld_cp = i.ld_cp;
  o.dc_carry = (float*)i.dc_carry; o.ld_dcc = i.ld_dcc; o.dgates_out = i.dgates_out; o.ld_dgo = i.ld_dgo;
  o.dh_n = (const float*)i.dh_n; o.ld_dhn = i.ld_dhn; o.dc_n = (const float*)i.dc_n; o.ld_dcn = i.ld_dcn;
  o.dh0_out = (float*)i.dh0_out; o.ld_dh0 = i.ld_dh0; o.t = i.t; o.inject = i.inject;
}

}  // namespace vmmt


extern "C" int vmmt_lstm_step_fwd(int dtype, int ndir, const vmmt_lstm_dir_fwd* dirs, const int64_t* lens, int B,
                                  int H, void* stream) {
  using namespace vmmt;
  if (ndir < 1 || ndir > 2 || !dirs || B <= 0 || H <= 0) return VMMT_EINVAL;
  StepArgsF a;
  for (int k = 0; k < ndir; ++k) {
    if (!dirs[k].h_prev || !dirs[k].w_hh || !dirs[k].gx || !dirs[k].gates || !dirs[k].c_out || !dirs[k].h_out)
      return VMMT_EINVAL;
    fill_f(a.d[k], dirs[k]);
  }
  if (ndir == 1) a.d[1] = a.d[0];
  a.lens = (const long long*)lens; a.B = B; a.H = H;
  if (dtype == VMMT_BF16 && H % 32 == 0) {
    bool ok = true;
    for (int k = 0; k < ndir; ++k) ok = ok && al16(dirs[k].h_prev, dirs[k].ld_hprev) && al16(dirs[k].w_hh, dirs[k].ld_w);
    if (ok) {
      switch (fast_chunk(H, 512, 32)) {
        case 512: return launch_fwd_fast<512>(a, ndir, (hipStream_t)stream);
        case 256: return launch_fwd_fast<256>(a, ndir, (hipStream_t)stream);
        case 128: return launch_fwd_fast<128>(a, ndir, (hipStream_t)stream);
        case 64: return launch_fwd_fast<64>(a, ndir, (hipStream_t)stream);
        default: return launch_fwd_fast<32>(a, ndir, (hipStream_t)stream);
      }
    }
  }
  dim3 grid((B + 63) / 64, (H + 31) / 32, ndir);
  if (dtype == VMMT_F32) {
    size_t sm = gemm_smem_elems<float, 64, 128, 32>() * sizeof(float);
    hipLaunchKernelGGL(lstm_step_fwd_kernel<float>, grid, dim3(128), sm, (hipStream_t)stream, a);
  } else if (dtype == VMMT_BF16) {
    size_t sm = gemm_smem_elems<bf16_t, 64, 128, 32>() * sizeof(bf16_t);
    hipLaunchKernelGGL(lstm_step_fwd_kernel<bf16_t>, grid, dim3(128), sm, (hipStream_t)stream, a);
  } else return VMMT_EINVAL;
  return check_launch();
}

extern "C" int vmmt_lstm_step_bwd(int dtype, int ndir, const vmmt_lstm_dir_bwd* dirs, const int64_t* lens, int B,
                                  int H, int mode, void* stream) {
  using namespace vmmt;
  if (ndir < 1 || ndir > 2 || !dirs || B <= 0 || H <= 0 || mode < 0 || mode > 1) return VMMT_EINVAL;
  StepArgsB a;
  for (int k = 0; k < ndir; ++k) {
    const vmmt_lstm_dir_bwd& i = dirs[k];
    if (mode == 1 && (!i.dgates_next || !i.dh0_out)) return VMMT_EINVAL;
    if (mode == 0 && (!i.gates || !i.c_t || !i.dc_carry || !i.dgates_out)) return VMMT_EINVAL;
    if (i.dgates_next && !i.w_hh_t) return VMMT_EINVAL;
    fill_b(a.d[k], i);
  }
  if (ndir == 1) a.d[1] = a.d[0];
  a.lens = (const long long*)lens; a.B = B; a.H = H; a.mode = mode;
  if (dtype == VMMT_BF16 && H % 32 == 0) {
    bool ok = true;
    for (int k = 0; k < ndir; ++k)
      ok = ok && (!dirs[k].dgates_next || (al16(dirs[k].dgates_next, dirs[k].ld_dgn) && al16(dirs[k].w_hh_t, dirs[k].ld_wt)));
    if (ok) {
      switch (fast_chunk(4 * H, 2048, 128)) {
        case 2048: return launch_bwd_fast<2048>(a, ndir, (hipStream_t)stream);
        case 1024: return launch_bwd_fast<1024>(a, ndir, (hipStream_t)stream);
        case 512: return launch_bwd_fast<512>(a, ndir, (hipStream_t)stream);
        case 256: return launch_bwd_fast<256>(a, ndir, (hipStream_t)stream);
        default: return launch_bwd_fast<128>(a, ndir, (hipStream_t)stream);
      }
    }
  }
  dim3 grid((B + 63) / 64, (H + 63) / 64, ndir);
  if (dtype == VMMT_F32) {
    size_t sm = gemm_smem_elems<float, 64, 64, 32>() * sizeof(float);
    hipLaunchKernelGGL(lstm_step_bwd_kernel<float>, grid, dim3(256), sm, (hipStream_t)stream, a);
  } else if (dtype == VMMT_BF16) {
    size_t sm = gemm_smem_elems<bf16_t, 64, 64, 32>() * sizeof(bf16_t);
    hipLaunchKernelGGL(lstm_step_bwd_kernel<bf16_t>, grid, dim3(256), sm, (hipStream_t)stream, a);
  } else return VMMT_EINVAL;
  return check_launch();
}

// A whole recurrence in one host call: `nsteps` consecutive launches of the step kernel, step i described by
// dirs[i*ndir .. i*ndir+ndir).  Same kernels and semantics as nsteps calls of vmmt_lstm_step_*; it only removes the
// per-launch cost of the host language (the conditional model's encoder_tgt walks the batch axis: 2 x 256 steps per
// training step at B = 256, which made the Python/ctypes host side the bottleneck).
extern "C" int vmmt_lstm_chain_fwd(int dtype, int ndir, int nsteps, const vmmt_lstm_dir_fwd* dirs, const int64_t* lens, int B,
                                   int H, void* stream) {
  if (nsteps < 0 || !dirs) return VMMT_EINVAL;
  for (int i = 0; i < nsteps; ++i) {
    int rc = vmmt_lstm_step_fwd(dtype, ndir, dirs + (long)i * ndir, lens, B, H, stream);
    if (rc != VMMT_OK) return rc;
  }
  return VMMT_OK;
}

extern "C" int vmmt_lstm_chain_bwd(int dtype, int ndir, int nsteps, const vmmt_lstm_dir_bwd* dirs, const int64_t* lens, int B,
                                   int H, int mode, void* stream) {
  if (nsteps < 0 || !dirs) return VMMT_EINVAL;
  for (int i = 0; i < nsteps; ++i) {
    int rc = vmmt_lstm_step_bwd(dtype, ndir, dirs + (long)i * ndir, lens, B, H, mode, stream);
    if (rc != VMMT_OK) return rc;
  }
  return VMMT_OK;
}

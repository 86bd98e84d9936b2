// Beam search, device side: one position of Beam.advance (onmt/translate/Beam.py:63-121) for every sentence of a batch, and
// the beam re-ordering of the decoder state (RNNDecoderState.beam_update, onmt/Models.py:589-594).
//   word_probs = log_softmax(logits)                       [K rows of one sentence]   TranslatorMultimodalVI.py:186-188
//   cur_len < min_length : word_probs[:, eos] = -1e20                                  Beam.py:77-80
//   first position       : candidates = word_probs[0]                                  Beam.py:91-92
//   later positions      : candidates = word_probs + scores[:, None]; rows whose last token is </s> = -1e20   Beam.py:83-90
//   top-K of the flattened K x V candidates -> scores, parent beam = id / V, token = id % V                 Beam.py:93-103
// Row layout of the decoder batch: row = k * B + b (TranslatorMultimodalVI.py:105-108, `view(beam_size, batch_size, -1)`).
// The logits of the K*B rows are materialised by the MFMA GEMM (K*B <= a few hundred rows, <= 40 MB at V = 30 000); this
// kernel streams them twice out of L2 (log-sum-exp, then selection): one workgroup of 1024 lanes per sentence.
#include "common.hpp"
#include "vmmt.h"

namespace vmmt {

struct Cand {
  float v;
  int id;
};
__device__ __forceinline__ bool better(float av, int ai, float bv, int bi) { return av > bv || (av == bv && ai < bi); }

__device__ __forceinline__ Cand wave_best(Cand c) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    float ov = __shfl_xor(c.v, off, 64);
    int oi = __shfl_xor(c.id, off, 64);
    if (better(ov, oi, c.v, c.id)) { c.v = ov; c.id = oi; }
  }
  return c;
}

constexpr int BEAM_NT = 1024;
constexpr int BEAM_NW = BEAM_NT / 64;

template <int KMAX>
__global__ void __launch_bounds__(BEAM_NT)
beam_advance_kernel(const float* __restrict__ logits, long ld, int B, int K, int V, const long long* __restrict__ cur_tok,
                    float* __restrict__ scores, int first, int mask_eos, int eos, long long* __restrict__ next_tok,
                    long long* __restrict__ sel_rows, float* __restrict__ h_score, int* __restrict__ h_prev,
                    long long* __restrict__ h_next) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float s_red[BEAM_NW];
  __shared__ float s_red2[BEAM_NW];
  __shared__ float s_lse[KMAX];
  __shared__ float s_sc[KMAX];
  __shared__ int s_dead[KMAX];
  __shared__ Cand s_c[BEAM_NW];
  __shared__ Cand s_win;
  const int rows = first ? 1 : K;
  // ---- log-sum-exp of each row (max, then sum of exp(x - max): the two passes of log_softmax)
  for (int k = 0; k < rows; ++k) {
    const float* x = logits + ((long)k * B + b) * ld;
    float m = -INFINITY;
    for (int v = tid; v < V; v += BEAM_NT) m = fmaxf(m, x[v]);
    m = wave_max(m);
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = s_red[0];
#pragma unroll
    for (int w = 1; w < BEAM_NW; ++w) m = fmaxf(m, s_red[w]);
    float s = 0.f;
    for (int v = tid; v < V; v += BEAM_NT) s += expf(x[v] - m);
    s = wave_sum(s);
    if (lane == 0) s_red2[wave] = s;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int w = 0; w < BEAM_NW; ++w) t += s_red2[w];
      s_lse[k] = m + logf(t);
      s_sc[k] = first ? 0.f : scores[(long)b * K + k];
      s_dead[k] = (!first && cur_tok[(long)k * B + b] == eos) ? 1 : 0;
    }
    __syncthreads();
  }
  // ---- lane-local top-K over the candidates this lane scans (sorted, best first)
  float tv[KMAX];
  int ti[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
  for (int k = 0; k < rows; ++k) {
    const float* x = logits + ((long)k * B + b) * ld;
    const float lse = s_lse[k], sc = s_sc[k];
    const bool dead = s_dead[k] != 0;
    for (int v = tid; v < V; v += BEAM_NT) {
      float lp = x[v] - lse;
      if (mask_eos && v == eos) lp = -1e20f;
      float val = first ? lp : lp + sc;
      if (dead) val = -1e20f;
      const int id = k * V + v;
      if (better(val, id, tv[KMAX - 1], ti[KMAX - 1])) {
        tv[KMAX - 1] = val;
        ti[KMAX - 1] = id;
#pragma unroll
        for (int j = KMAX - 1; j >= 1; --j) {
          if (better(tv[j], ti[j], tv[j - 1], ti[j - 1])) {
            float fv = tv[j]; tv[j] = tv[j - 1]; tv[j - 1] = fv;
            int fi = ti[j]; ti[j] = ti[j - 1]; ti[j - 1] = fi;
          }
        }
      }
    }
  }
  // ---- K rounds of workgroup-wide arg-max over the heads of the lane-local lists
  for (int r = 0; r < K; ++r) {
    Cand c{tv[0], ti[0]};
    c = wave_best(c);
    if (lane == 0) s_c[wave] = c;
    __syncthreads();
    if (tid == 0) {
      Cand w = s_c[0];
      for (int i = 1; i < BEAM_NW; ++i)
        if (better(s_c[i].v, s_c[i].id, w.v, w.id)) w = s_c[i];
      s_win = w;
      const int pk = w.id / V, tok = w.id - pk * V;
      scores[(long)b * K + r] = w.v;
      next_tok[(long)r * B + b] = tok;
      sel_rows[(long)r * B + b] = (long)pk * B + b;
      h_score[(long)b * K + r] = w.v;
      h_prev[(long)b * K + r] = pk;
      h_next[(long)b * K + r] = tok;
    }
    __syncthreads();
    if (ti[0] == s_win.id) {            // the owner of the winner pops it
#pragma unroll
      for (int j = 0; j < KMAX - 1; ++j) { tv[j] = tv[j + 1]; ti[j] = ti[j + 1]; }
      tv[KMAX - 1] = -INFINITY;
      ti[KMAX - 1] = 0x7fffffff;
    }
    __syncthreads();
  }
}

// dst[r][:] = src[rows[r]][:], rows of `units` 4-byte (or 2-byte) words
template <class U>
__global__ void rows_select_kernel(const U* __restrict__ src, long ld_src, const long long* __restrict__ rows, U* __restrict__ dst,
                                   long ld_dst, int R, int units) {
  const int r = blockIdx.x;
  const U* s = src + rows[r] * ld_src;
  U* d = dst + (long)r * ld_dst;
  for (int c = threadIdx.x; c < units; c += blockDim.x) d[c] = s[c];
}

}  // namespace vmmt
using namespace vmmt;
#define ST ((hipStream_t)stream)

extern "C" int vmmt_beam_advance(const float* logits, int64_t ld, int B, int K, int V, const int64_t* cur_tok, float* scores,
                                 int first, int mask_eos, int eos, int64_t* next_tok, int64_t* sel_rows, float* hist_score,
                                 int* hist_prev, int64_t* hist_next, void* stream) {
  if (!logits || !cur_tok || !scores || !next_tok || !sel_rows || !hist_score || !hist_prev || !hist_next) return VMMT_EINVAL;
  if (B <= 0 || K <= 0 || K > 16 || V <= 0 || ld < V || eos < 0 || eos >= V || (long)K * V > 0x7fffffffL) return VMMT_EINVAL;
  if ((long)V < K) return VMMT_EINVAL;                  // the first position draws K candidates from one row
  dim3 grid(B), block(BEAM_NT);
  if (K <= 8)
    hipLaunchKernelGGL(beam_advance_kernel<8>, grid, block, 0, ST, logits, (long)ld, B, K, V, (const long long*)cur_tok, scores, first,
                       mask_eos, eos, (long long*)next_tok, (long long*)sel_rows, hist_score, hist_prev, (long long*)hist_next);
  else
    hipLaunchKernelGGL(beam_advance_kernel<16>, grid, block, 0, ST, logits, (long)ld, B, K, V, (const long long*)cur_tok, scores, first,
                       mask_eos, eos, (long long*)next_tok, (long long*)sel_rows, hist_score, hist_prev, (long long*)hist_next);
  return check_launch();
}

extern "C" int vmmt_rows_select(const void* src, int64_t ld_src_bytes, const int64_t* rows, void* dst, int64_t ld_dst_bytes, int R,
                                int row_bytes, void* stream) {
  if (!src || !rows || !dst || R < 0 || row_bytes <= 0 || (row_bytes & 1)) return VMMT_EINVAL;
  if (R == 0) return VMMT_OK;
  const bool w4 = ((row_bytes | ld_src_bytes | ld_dst_bytes) & 3) == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 3) == 0;
  if (w4)
    hipLaunchKernelGGL(rows_select_kernel<uint32_t>, dim3(R), dim3(256), 0, ST, (const uint32_t*)src, (long)(ld_src_bytes / 4),
                       (const long long*)rows, (uint32_t*)dst, (long)(ld_dst_bytes / 4), R, row_bytes / 4);
  else {
    if ((ld_src_bytes | ld_dst_bytes) & 1) return VMMT_EINVAL;
    hipLaunchKernelGGL(rows_select_kernel<uint16_t>, dim3(R), dim3(256), 0, ST, (const uint16_t*)src, (long)(ld_src_bytes / 2),
                       (const long long*)rows, (uint16_t*)dst, (long)(ld_dst_bytes / 2), R, row_bytes / 2);
  }
  return check_launch();
}

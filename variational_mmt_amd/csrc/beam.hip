// Beam search, device side: one position of Beam.advance (onmt/translate/Beam.py:63-121) for every sentence of a batch, and
// the beam re-ordering of the decoder state (RNNDecoderState.beam_update, onmt/Models.py:589-594).
//   word_probs = log_softmax(logits)                       [K rows of one sentence]   TranslatorMultimodalVI.py:186-188
//   cur_len < min_length : word_probs[:, eos] = -1e20                                  Beam.py:77-80
//   first position       : candidates = word_probs[0]                                  Beam.py:91-92
//   later positions      : candidates = word_probs + scores[:, None]; rows whose last token is </s> = -1e20   Beam.py:83-90
//   top-K of the flattened K x V candidates -> scores, parent beam = id / V, token = id % V                 Beam.py:93-103
// Row layout of the decoder batch: row = k * B + b (TranslatorMultimodalVI.py:105-108, `view(beam_size, batch_size, -1)`).
// The logits of the K*B rows are materialised by the MFMA GEMM (K*B <= a few hundred rows, <= 40 MB at V = 30 000) and streamed
// twice out of L2 (log-sum-exp, then selection) by one workgroup per (row, 2048-entry vocabulary chunk).
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

// Three launches per position, so that the K*B rows x V logits are scanned by (rows x vocabulary chunks) workgroups instead of one
// workgroup per sentence (which took 216 us per position at V = 30 000, K = 5 -- 63 % of a decoded position):
//   beam_part_kernel   (chunk, row): max and sum exp(x - max) of the chunk                      -> the row's log-sum-exp
//   beam_cand_kernel   (chunk, row): the chunk's K best candidates under the FINAL order (value = Beam.advance's candidate score,
//                                    ties: lowest flat index k V + v), i.e. every candidate that can be among the sentence's K best
//   beam_merge_kernel  (sentence):   K best of its rows x chunks x K candidates, outputs as Beam.advance leaves them
constexpr int BEAM_CH = 2048;            // vocabulary entries per chunk: 8 per lane of a 256-lane workgroup
constexpr int BEAM_EPL = BEAM_CH / 256;
constexpr int BEAM_MAXC = 1024;          // chunks per row the merge kernel's loops are sized for (V <= 2 M)

__device__ __forceinline__ float beam_row_lse(const float* __restrict__ pm, const float* __restrict__ ps, int C) {
  float m = -INFINITY;
  for (int c = 0; c < C; ++c) m = fmaxf(m, pm[c]);
  float t = 0.f;
  for (int c = 0; c < C; ++c) t += ps[c] * expf(pm[c] - m);
  return m + logf(t);
}

__global__ void __launch_bounds__(256) beam_part_kernel(const float* __restrict__ logits, long ld, int V, int C,
                                                        float* __restrict__ part_m, float* __restrict__ part_s) {
  const int c = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = logits + (long)r * ld;
  __shared__ float s_m[4], s_s[4];
  float xv[BEAM_EPL];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < BEAM_EPL; ++j) {
    const int v = c * BEAM_CH + tid + 256 * j;
    xv[j] = v < V ? x[v] : -INFINITY;
    m = fmaxf(m, xv[j]);
  }
  m = wave_max(m);
  if (lane == 0) s_m[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
  float t = 0.f;
#pragma unroll
  for (int j = 0; j < BEAM_EPL; ++j) t += xv[j] > -INFINITY ? expf(xv[j] - m) : 0.f;
  t = wave_sum(t);
  if (lane == 0) s_s[wave] = t;
  __syncthreads();
  if (tid == 0) {
    part_m[(long)r * C + c] = m;
    part_s[(long)r * C + c] = (s_s[0] + s_s[1]) + (s_s[2] + s_s[3]);
  }
}

__global__ void __launch_bounds__(256) beam_cand_kernel(const float* __restrict__ logits, long ld, int B, int K, int V, int C,
                                                        const long long* __restrict__ cur_tok, const float* __restrict__ scores,
                                                        int first, int mask_eos, int eos, const float* __restrict__ part_m,
                                                        const float* __restrict__ part_s, float* __restrict__ cand_v,
                                                        int* __restrict__ cand_i) {
  const int c = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = r / B, b = r - k * B;
  const float* x = logits + (long)r * ld;
  const float lse = beam_row_lse(part_m + (long)r * C, part_s + (long)r * C, C);      // the same arithmetic in every chunk of the row
  const float sc = first ? 0.f : scores[(long)b * K + k];
  const bool dead = !first && cur_tok[(long)k * B + b] == eos;
  float val[BEAM_EPL];
  int id[BEAM_EPL];
#pragma unroll
  for (int j = 0; j < BEAM_EPL; ++j) {
    const int v = c * BEAM_CH + tid + 256 * j;
    if (v < V) {
      float lp = x[v] - lse;
      if (mask_eos && v == eos) lp = -1e20f;
      float t = first ? lp : lp + sc;
      if (dead) t = -1e20f;
      val[j] = t;
      id[j] = k * V + v;
    } else { val[j] = -INFINITY; id[j] = 0x7fffffff; }
  }
  __shared__ Cand s_c[4];
  __shared__ Cand s_win;
  for (int round = 0; round < K; ++round) {
    Cand best{-INFINITY, 0x7fffffff};
#pragma unroll
    for (int j = 0; j < BEAM_EPL; ++j)
      if (better(val[j], id[j], best.v, best.id)) { best.v = val[j]; best.id = id[j]; }
    best = wave_best(best);
    if (lane == 0) s_c[wave] = best;
    __syncthreads();
    if (tid == 0) {
      Cand w = s_c[0];
      for (int i = 1; i < 4; ++i)
        if (better(s_c[i].v, s_c[i].id, w.v, w.id)) w = s_c[i];
      s_win = w;
      cand_v[((long)r * C + c) * K + round] = w.v;
      cand_i[((long)r * C + c) * K + round] = w.id;
    }
    __syncthreads();
    const int wid = s_win.id;
#pragma unroll
    for (int j = 0; j < BEAM_EPL; ++j)
      if (id[j] == wid && wid != 0x7fffffff) { val[j] = -INFINITY; id[j] = 0x7fffffff; }      // the owner retires the winner
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) beam_merge_kernel(const float* __restrict__ cand_v, const int* __restrict__ cand_i, int B, int K,
                                                         int V, int C, int first, float* __restrict__ scores,
                                                         long long* __restrict__ next_tok, long long* __restrict__ sel_rows,
                                                         float* __restrict__ h_score, int* __restrict__ h_prev,
                                                         long long* __restrict__ h_next) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rows = first ? 1 : K, per = C * K, n = rows * per;
  __shared__ Cand s_c[4];
  __shared__ Cand s_win;
  for (int round = 0; round < K; ++round) {
    // this lane's best candidate not yet taken (candidates are few -- rows x chunks x K -- and sit in L2: re-read per round; a
    // candidate is "taken" when it is better than or equal to the previous round's winner in the total order)
    Cand best{-INFINITY, 0x7fffffff};
    const Cand last = round ? s_win : Cand{INFINITY, -1};
    for (int i = tid; i < n; i += 256) {
      const int kk = i / per, j = i - kk * per;
      const long at = ((long)(kk * B + b) * C) * K + j;
      const float v = cand_v[at];
      const int id = cand_i[at];
      const bool open = round == 0 || better(last.v, last.id, v, id);          // strictly after the last winner
      if (open && better(v, id, best.v, best.id)) { best.v = v; best.id = id; }
    }
    __syncthreads();                                        // everybody has read s_win
    best = wave_best(best);
    if (lane == 0) s_c[wave] = best;
    __syncthreads();
    if (tid == 0) {
      Cand w = s_c[0];
      for (int i = 1; i < 4; ++i)
        if (better(s_c[i].v, s_c[i].id, w.v, w.id)) w = s_c[i];
      s_win = w;
      const int wid = w.id == 0x7fffffff ? 0 : w.id;             // (NaN logits leave the lists empty: valid parent / token all the same -- they index rows)
      const int pk = wid / V, tok = wid - pk * V;
      scores[(long)b * K + round] = w.v;
      next_tok[(long)round * B + b] = tok;
      sel_rows[(long)round * B + b] = (long)pk * B + b;
      h_score[(long)b * K + round] = w.v;
      h_prev[(long)b * K + round] = pk;
      h_next[(long)b * K + round] = tok;
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

// history[counter][segment] <- staging buffers of one decoded position, then counter += bump.  The position index lives on the
// device so that every position of a decoding loop is the SAME sequence of launches with the SAME arguments: one hipGraph, replayed.
struct HistSegs {
  const uint32_t* src[VMMT_HIST_MAX_SEGS];
  uint32_t* dst[VMMT_HIST_MAX_SEGS];
  long words[VMMT_HIST_MAX_SEGS], stride_words[VMMT_HIST_MAX_SEGS];
};
__global__ void __launch_bounds__(256) history_append_kernel(HistSegs h, const int* __restrict__ counter, int limit) {
  const long t = *counter;
  if (t < 0 || t >= limit) return;                          // a replay beyond the history's capacity records nothing
  const int seg = blockIdx.y;
  const uint32_t* s = h.src[seg];
  uint32_t* d = h.dst[seg] + t * h.stride_words[seg];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < h.words[seg]; i += (long)gridDim.x * 256) d[i] = s[i];
}
__global__ void counter_add_kernel(int* counter, int v) { *counter += v; }

}  // namespace vmmt
using namespace vmmt;
#define ST ((hipStream_t)stream)

extern "C" int vmmt_history_append(const vmmt_hist_seg* segs, int n, int* counter, int limit, int bump, void* stream) {
  if (!segs || n < 1 || n > VMMT_HIST_MAX_SEGS || !counter || limit < 0) return VMMT_EINVAL;
  HistSegs h{};
  long most = 0;
  for (int i = 0; i < n; ++i) {
    if (!segs[i].src || !segs[i].dst || segs[i].bytes <= 0 || ((segs[i].bytes | segs[i].stride_bytes) & 3) ||
        ((((uintptr_t)segs[i].src) | ((uintptr_t)segs[i].dst)) & 3))
      return VMMT_EINVAL;
    h.src[i] = (const uint32_t*)segs[i].src; h.dst[i] = (uint32_t*)segs[i].dst;
    h.words[i] = segs[i].bytes / 4; h.stride_words[i] = segs[i].stride_bytes / 4;
    most = most > h.words[i] ? most : h.words[i];
  }
  const int gx = (int)((most + 1023) / 1024 < 1 ? 1 : (most + 1023) / 1024 > 64 ? 64 : (most + 1023) / 1024);
  hipLaunchKernelGGL(history_append_kernel, dim3(gx, n), dim3(256), 0, ST, h, (const int*)counter, limit);
  if (bump) hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, ST, counter, bump);
  return check_launch();
}

extern "C" int64_t vmmt_beam_advance_ws_bytes(int B, int K, int V) {
  if (B <= 0 || K <= 0 || V <= 0) return 0;
  const int64_t C = (V + BEAM_CH - 1) / BEAM_CH, R = (int64_t)K * B;
  return 4 * (2 * R * C + 2 * R * C * K);
}

extern "C" int vmmt_beam_advance(const float* logits, int64_t ld, int B, int K, int V, const int64_t* cur_tok, float* scores,
                                 int first, int mask_eos, int eos, int64_t* next_tok, int64_t* sel_rows, float* hist_score,
                                 int* hist_prev, int64_t* hist_next, void* ws, int64_t ws_bytes, void* stream) {
  if (!logits || !cur_tok || !scores || !next_tok || !sel_rows || !hist_score || !hist_prev || !hist_next || !ws) return VMMT_EINVAL;
  if (B <= 0 || K <= 0 || K > 16 || V <= 0 || ld < V || eos < 0 || eos >= V || (long)K * V > 0x7fffffffL) return VMMT_EINVAL;
  if ((long)V < K) return VMMT_EINVAL;                  // the first position draws K candidates from one row
  if (ws_bytes < vmmt_beam_advance_ws_bytes(B, K, V) || (((uintptr_t)ws) & 3)) return VMMT_EINVAL;
  const int C = (V + BEAM_CH - 1) / BEAM_CH;
  if (C > BEAM_MAXC) return VMMT_EINVAL;
  const long R = (long)K * B;
  float* part_m = (float*)ws;
  float* part_s = part_m + R * C;
  float* cand_v = part_s + R * C;
  int* cand_i = (int*)(cand_v + R * C * K);
  const int rows = first ? B : (int)R;                  // the first position scores beam 0 only: rows 0 .. B-1
  if (rows > 65535) return VMMT_EINVAL;
  hipLaunchKernelGGL(beam_part_kernel, dim3(C, rows), dim3(256), 0, ST, logits, (long)ld, V, C, part_m, part_s);
  hipLaunchKernelGGL(beam_cand_kernel, dim3(C, rows), dim3(256), 0, ST, logits, (long)ld, B, K, V, C, (const long long*)cur_tok,
                     (const float*)scores, first, mask_eos, eos, (const float*)part_m, (const float*)part_s, cand_v, cand_i);
  hipLaunchKernelGGL(beam_merge_kernel, dim3(B), dim3(256), 0, ST, (const float*)cand_v, (const int*)cand_i, B, K, V, C, first, scores,
                     (long long*)next_tok, (long long*)sel_rows, hist_score, hist_prev, (long long*)hist_next);
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

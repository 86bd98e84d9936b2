/* vmmt.h -- C-ABI of libvmmt.so: hand-written HIP kernels (gfx950 / MI355X) for ONE hot path of
 * iacercalixto/variational_mmt: the VI_Model1 training step
 *   (onmt/TrainerMultimodal.py:625-718 -> onmt/Models.py:850-1011 -> onmt/VILoss.py:217-513 -> onmt/Optim.py:78-96).
 *
 * The reference is pure Python on torch 0.3.1 and has no FFI of its own; the device work it gets from torch
 * (cuDNN LSTM, cuBLAS addmm/bmm, embedding gather/scatter, softmax, NLL, Adam) is what
 * these entry points replace.  Each entry point cites the reference lines whose arithmetic it performs.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every buffer is CALLER-OWNED device memory (e.g. a PyTorch-ROCm tensor's
 *     data_ptr()); nothing is allocated, freed or synchronised inside; launches are asynchronous on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream) and are hipGraph-capturable.
 *   - return value: 0 = VMMT_OK, 1 = invalid argument, 2 = launch failure (hipGetLastError).
 *   - `dtype` selects the storage/compute type "T" of activations and weight shadows:
 *       VMMT_F32  : fp32 storage, v_mfma_f32_32x32x2_f32 (exact fp32; parity mode)
 *       VMMT_BF16 : bf16 storage, v_mfma_f32_32x32x16_bf16, fp32 accumulate (throughput mode)
 *     Buffers documented as "T" hold that type; "f32" buffers are always float.
 *   - leading dimensions (ld*) are in ELEMENTS.  Any value is accepted; rows that start 16-byte aligned take the
 *     vector path, others a scalar path.
 *   - thread-compatible: no global state.
 */
#ifndef VMMT_H
#define VMMT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VMMT_F32 0
#define VMMT_BF16 1

#define VMMT_ACT_NONE 0
#define VMMT_ACT_RELU 1
#define VMMT_ACT_TANH 2
#define VMMT_ACT_SOFTPLUS 3 /* nn.Softplus(beta=1, threshold=20): NormalVariationalEncoder.py:42 */
#define VMMT_ACT_SIGMOID 4

/* 128 x 128 tiles whose LDS request admits ONE workgroup per CU: for bulk GEMMs of a side stream that must leave room for
 * the latency-critical kernels of another stream on every CU */
#define VMMT_TILE_128_ONE_PER_CU 129

#define VMMT_GEMM_NT 0 /* A[M][K] k-contiguous, B[N][K] k-contiguous   : y = x W^T (nn.Linear forward)        */
#define VMMT_GEMM_TN 1 /* A stored [K][M], B stored [K][N]             : dW = dY^T X (reduction over tokens)   */
#define VMMT_GEMM_NN 2 /* A[M][K] k-contiguous, B stored [K][N]        : dx = dY W                              */

int vmmt_version(void);

/* ---- streams ---------------------------------------------------------------------------------------------------
 * A HIP stream whose kernels may only occupy the compute units set in `mask` (`words` x 32 bits, bit i = CU i as
 * hipExtStreamCreateWithCUMask numbers them).  Used for the step's bulk side stream so that a share of the CUs stays
 * free for the latency-critical kernels of the main stream (the reference serialises everything on one CUDA stream:
 * onmt/TrainerMultimodal.py:624-705). */
int vmmt_stream_create_masked(const uint32_t* mask, int words, int priority, void** stream);
int vmmt_stream_destroy(void* stream);
/* diagnostic (tools/probe_cu_mask.py): workgroup i of a grid of n (each `threads` wide, resident for hold_us) writes XCC id | HW_ID << 8 to out[i] */
int vmmt_probe_where(uint32_t* out, int n_workgroups, int threads, int hold_us, void* stream);

/* ---- generic fused GEMM --------------------------------------------------------------------------------------
 * C[m][n] (+)= act(alpha * sum_k A(m,k) B(n,k) + addend(m,n)).
 * Replaces every nn.Linear / addmm on the path: LSTM input projections (Models.py:124-129, VI_Model1.py:149-152),
 * attention linear_in / linear_out (GlobalAttention.py:106-110,187-190), LocationLayer/ScaleLayer fc1/fc2
 * (NormalVariationalEncoder.py:12-43) and all their backward products.
 *   addend: add_rows = 1 -> bias row; add_rows = R > 1 -> addend[(m % R)][n] (z W_z^T broadcast over time steps,
 *           VI_Model1.py:99-100 without the materialised repeat); add_rows = -1 -> addend[m][n].
 *   a_kmod / b_kmod (K-strided operands only): reduction index taken modulo this (0 = off).
 *   split_k: see the struct field.
 *   scatter_ids != NULL: out_f32 must be 1; C row = scatter_ids[m] (int64), atomic add, rows with id == pad_id are
 *           dropped: the embedding-gradient scatter-add with padding_idx (modules/Embeddings.py:118). */
typedef struct vmmt_gemm_args {
  int dtype, layout;
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  void* C; int64_t ldc;
  int M, N, K;
  int a_kmod, b_kmod;
  const void* addend; int64_t ld_add; int add_rows; int add_is_T;
  int act; int out_f32; int accumulate; float alpha;
  const int64_t* scatter_ids; int pad_id;
  int tile; /* 0 = auto, 64, 128, VMMT_TILE_128_ONE_PER_CU */
  int split_k; /* > 1: the reduction is split over this many workgroups per tile which atomicAdd fp32 partial sums into C
                  (out_f32 = 1, act = NONE; C must hold zeros or a partial sum) -- for weight gradients dW = dY^T X whose
                  reduction runs over thousands of tokens while the output is small */
  int b_batch_rows; int64_t b_batch_stride; /* b_batch_rows > 0 (a multiple of 256): output rows [i r, (i + 1) r) use the B operand
                  at B + i * b_batch_stride elements -- one launch for row blocks that share A's layout but not B (the generator's
                  weight gradient: one scaled copy of the decoder outputs per vocabulary slice) */
  const float* colsum_w; int64_t colsum_w_stride; float* colsum_out; /* colsum_w != NULL (layout TN only): colsum_out[m] += sum_k
                  A[k][m] * w[k] with w = colsum_w + (m / b_batch_rows) * colsum_w_stride (w = colsum_w without B blocks), f32, K
                  entries readable -- the weighted column sums of the K-strided A operand out of the pass that multiplies it (the
                  generator's bias gradient next to its weight gradient).  colsum_w == NULL, colsum_out != NULL: plain sums
                  colsum_out[m] += sum_k A[k][m] (an LSTM bias gradient next to dW_hh).  Only where vmmt_gemm_colsum_applies()
                  returns 1; otherwise VMMT_EINVAL */
  float* colsum_out2; /* plain sums only: a second destination of the same sums (nn.LSTM's bias_ih and bias_hh), or NULL */
  int c_row_blk, c_row_valid; /* c_row_blk > 0 (out_f32 = 1, no scatter): the M output rows are blocks of c_row_blk rows of which the first
                  c_row_valid are stored, block after block without gaps: row m goes to C row (m / blk) * valid + m % blk, rows with
                  m % blk >= valid are dropped.  The caller computes with hidden sizes padded to what the MFMA kernels tile
                  (opts.py:54 -rnn_size 500 -> 4 gate blocks of 512) while C keeps nn.LSTM's [4H][I] gradient layout.  The plain
                  column sums (colsum_out / colsum_out2) follow the same map */
  int c_col_blk, c_col_valid; /* the same for the N output columns (e.g. dW_out [H][2H] of GlobalAttention.py:187 from a [c ; r]
                  buffer whose halves are padded) */
  const int64_t* a_row_ids; /* != NULL (layout NT, bf16, the 128 x 128 LDS-DMA configuration: K % 64 == 0, lda % 4 == 0; VMMT_EINVAL otherwise):
                  row m of the A operand is row a_row_ids[m] of the TABLE at A (lda = the table's row stride) -- the embedding lookup
                  (modules/Embeddings.py:169-188) as the A-operand fetch of the LSTM's input projection (Models.py:124-129): the product's
                  LDS staging reads the table rows by token id, no [tokens x E] copy is written or read.  Every id must be a row of the
                  table; a row stride that is not a multiple of 8 elements is fine (the LDS-DMA takes any 4-byte aligned source), the
                  K - lda columns read beyond a row's end must be finite and meet zero columns of B */
} vmmt_gemm_args;
int vmmt_gemm(const vmmt_gemm_args* args, void* stream);
int vmmt_gemm_colsum_applies(const vmmt_gemm_args* args);
/* n independent products in one call: the same results as n vmmt_gemm calls (in any order: members must ACCUMULATE into C with
 * atomics, split_k >= 2; several may share one C).  When every member is a bf16 TN product that vmmt_gemm would run on its two-stage
 * 128 x 128 LDS-DMA tiles -- the weight-gradient products dW = dY^T X of the LSTM and attention layers, the backward of nn.LSTM /
 * nn.Linear in onmt/Models.py:124-129, onmt/VI_Model1.py:149-152, modules/GlobalAttention.py:106-118 -- and 2 <= n <= VMMT_GEMM_GROUP_MAX,
 * the tiles of all of them go out as ONE grid (vmmt_gemm_group_applies() == 1); otherwise one launch per member. */
#define VMMT_GEMM_GROUP_MAX 8
int vmmt_gemm_group(const vmmt_gemm_args* args, int n, void* stream);
int vmmt_gemm_group_applies(const vmmt_gemm_args* args, int n);

/* ---- LSTM time steps ------------------------------------------------------------------------------------------
 * One launch per time step (both directions of a bidirectional layer in the same launch).  The input projection
 * gx = x W_ih^T + b_ih + b_hh of the whole sequence is a vmmt_gemm; a step computes
 *   gates = gx[t] + h_prev W_hh^T ; i,f,o = sigmoid, g = tanh ; c = f c_prev + i g ; h = o tanh(c)
 * (torch nn.LSTM, gate order i,f,g,o -- encoder onmt/Models.py:124-129,140-147 incl. pack_padded_sequence
 * semantics via `lens`; decoder onmt/VI_Model1.py:106,149-152 with lens = NULL).
 * Column layout of gx / gates / dgates: gate-major, g*H + u (== torch's weight row order). */
typedef struct vmmt_lstm_dir_fwd {
  const void* h_prev; int64_t ld_hprev;  /* T   [B][ld]  state entering the step                               */
  const void* c_prev; int64_t ld_cprev;  /* f32 [B][ld]  (NULL = zeros)                                         */
  const void* w_hh;   int64_t ld_w;      /* T   [4H][ld] recurrent weight, k contiguous                         */
  const void* gx;     int64_t ld_gx;     /* f32 [B][ld]  row block of step t                                    */
  const void* gx2;    int64_t ld_gx2;    /* f32 [B][ld]  optional per-sentence addend (z W_z^T + b; VI_Model1.py:99-100) */
  void* gates;        int64_t ld_gates;  /* T   [B][ld]  out: post-activation i,f,g,o (saved for backward)      */
  void* c_out;        int64_t ld_c;      /* f32 [B][ld]  out: cell state (frozen at padded positions)           */
  void* h_out;        int64_t ld_h;      /* T   [B][ld]  out: h (zero at padded positions)                      */
  void* h_n;          int64_t ld_hn;     /* T   [B][ld]  out (optional): final state, see `capture`             */
  void* c_n;          int64_t ld_cn;     /* f32 [B][ld]                                                         */
  int t;                                 /* time index (compared with lens[b])                                  */
  int capture;                           /* 0 none; 1 if t == lens[b]-1; 2 if t == 0; 3 always                  */
} vmmt_lstm_dir_fwd;
int vmmt_lstm_step_fwd(int dtype, int ndir, const vmmt_lstm_dir_fwd* dirs, const int64_t* lens, int B, int H,
                       void* stream);

/* backward of one step (autograd of the above).  mode 0: dh = dgates_next W_hh + dh_above (+ dh_n when injected);
 * writes dgates_out (gate pre-activation gradients, T) and updates dc_carry in place.  mode 1: only
 * dh0_out = dgates_next W_hh (gradient of the initial hidden state). */
typedef struct vmmt_lstm_dir_bwd {
  const void* dgates_next; int64_t ld_dgn; /* T [B][ld] of the step processed just before (NULL = none)         */
  const void* w_hh_t;      int64_t ld_wt;  /* T [H][ld]: W_hh transposed (k = gate row index contiguous)         */
  const void* dh_above;    int64_t ld_dha; /* T [B][ld] or NULL                                                  */
  const void* gates;       int64_t ld_gates;
  const void* c_t;         int64_t ld_ct;  /* f32 */
  const void* c_prev;      int64_t ld_cp;  /* f32, NULL = zeros */
  void* dc_carry;          int64_t ld_dcc; /* f32 in/out */
  void* dgates_out;        int64_t ld_dgo; /* T */
  const void* dh_n;        int64_t ld_dhn; /* f32 or NULL: gradient of the captured final h */
  const void* dc_n;        int64_t ld_dcn; /* f32 */
  void* dh0_out;           int64_t ld_dh0; /* f32, mode 1 */
  int t;
  int inject;                              /* 0 none; 1 if t == lens[b]-1; 2 if t == 0; 3 always */
} vmmt_lstm_dir_bwd;
int vmmt_lstm_step_bwd(int dtype, int ndir, const vmmt_lstm_dir_bwd* dirs, const int64_t* lens, int B, int H,
                       int mode, void* stream);

/* a whole recurrence in ONE host call: nsteps launches of the step kernel, step i described by dirs[i*ndir .. (i+1)*ndir)
 * (nn.LSTM's time loop, onmt/Models.py:131-149; same semantics as nsteps calls of vmmt_lstm_step_fwd / _bwd) */
int vmmt_lstm_chain_fwd(int dtype, int ndir, int nsteps, const vmmt_lstm_dir_fwd* dirs, const int64_t* lens, int B, int H,
                        void* stream);
int vmmt_lstm_chain_bwd(int dtype, int ndir, int nsteps, const vmmt_lstm_dir_bwd* dirs, const int64_t* lens, int B, int H,
                        int mode, void* stream);

/* a whole FORWARD recurrence in ONE LAUNCH (persistent kernel): every workgroup (16 sentences x 32 hidden units) keeps its W_hh slice in
 * registers for the whole sequence and the workgroups of a 16-sentence row group hand h_t to each other in-launch as tagged 8-byte granules
 * ({two bf16 | tag}, one sc1 write-through store each; consumers re-read with sc1 loads until the tags match; bounded spins).
 * `dirs` = the nsteps x ndir step descriptors in HOST memory (validated here), `dirs_dev` = the same array in DEVICE memory
 * (read by the kernel); `sync` (vmmt_lstm_seq_sync_words() uint32: launch epoch, finish count, error code -- 0 = every wait
 * completed) and `xchg` (vmmt_lstm_seq_xchg_bytes() bytes, 16-byte aligned): device scratch private to the call site, zeroed
 * ONCE when allocated and then left alone.  Same semantics and BITS as vmmt_lstm_chain_fwd (H = 1024: the same numbers up to the
 * order of the f32 partial sums, <= 1 bf16 ulp), which this call falls back to when the persistent kernel does not apply (fp32, H not
 * in {64,128,256,512,1024}, more workgroups than CUs -- (B / 16) (H / 32) ndir > 256: cut the batch into row chunks --, unaligned rows, steps not
 * chained h_prev[t] == h_out[t-1] / c_prev[t] == c_out[t-1]). */
/* words VMMT_SEQ_GUARD_WORD, +1 of `sync` (8-byte aligned): an optional device pointer to a uint32 GUARD word of the caller, written
 * by the host once after the scratch is zeroed (0 = none).  A launch whose bounded wait runs out stores its error code there as well:
 * vmmt_adam_step(skip = guard) then leaves the parameters of that step alone (the recurrence's outputs are garbage) until the host has
 * seen the word, switched to the per-step kernels and cleared it. */
#define VMMT_SEQ_GUARD_WORD (4 + 2 * 256)
int vmmt_lstm_seq_sync_words(void);
int64_t vmmt_lstm_seq_xchg_bytes(int ndir, int B, int H);
int vmmt_lstm_seq_fwd(int dtype, int ndir, int nsteps, const vmmt_lstm_dir_fwd* dirs, const vmmt_lstm_dir_fwd* dirs_dev,
                      const int64_t* lens, int B, int H, uint32_t* sync, void* xchg, void* stream);
/* ... and the BACKWARD recurrence (in-launch hand-off of dgates: dense pieces + one ready flag per producer wave, stored behind
 * the acknowledged pieces).  The mode-0 steps of vmmt_lstm_step_bwd: step t consumes dgates_out of step t-1; dc_carry is
 * read at the first and written at the last step; step 0 has dgates_next == NULL, or -- a recurrence cut into several calls --
 * the plain dgates_out buffer of the last step of the call before, which is then read from memory).  What the row group exchanges is dgates
 * (4H values per sentence and step).  `xchg`: vmmt_lstm_seq_xchg_bytes_bwd() bytes; otherwise the contract of vmmt_lstm_seq_fwd;
 * falls back to vmmt_lstm_chain_bwd.  with_dh0 != 0: `dirs` holds nsteps + 1 steps, the last one a mode-1 descriptor (dgates_next,
 * w_hh_t, dh0_out): the gradient of the initial hidden state comes out of the same launch. */
int64_t vmmt_lstm_seq_xchg_bytes_bwd(int ndir, int B, int H);
int vmmt_lstm_seq_bwd(int dtype, int ndir, int nsteps, const vmmt_lstm_dir_bwd* dirs, const vmmt_lstm_dir_bwd* dirs_dev,
                      const int64_t* lens, int B, int H, int with_dh0, uint32_t* sync, void* xchg, void* stream);


/* ---- statistics vector (f32[VMMT_STAT_COUNT], device memory, zeroed by the caller before each step) ------------
 * sums that VIStatistics needs (onmt/TrainerMultimodal.py:32-228, filled at onmt/VILoss.py:483-497). */
#define VMMT_STAT_NLL 0          /* sum of token NLL over non-pad targets                                       */
#define VMMT_STAT_NWORDS 1       /* number of non-pad targets                                                    */
#define VMMT_STAT_NCORRECT 2     /* argmax == target, non-pad                                                    */
#define VMMT_STAT_KL_SUM 3       /* sum_b KL(q(z|x_b) || p(z))  (batch mean = / B)                               */
#define VMMT_STAT_IMG_LOGPROB 4  /* image log-prob as executed (H1): sum over batch of the per-sentence mean over D */
#define VMMT_STAT_IMG_COS 5      /* sum_b cosine(mu_v_b, v_b)   (reported value = / B)                           */
#define VMMT_STAT_GRAD_SUMSQ 6   /* ||g||^2 (written by vmmt_sumsq when pointed here)                            */
#define VMMT_STAT_TICKET 7       /* a uint32 work word of vmmt_gen_fwd_combine_stats (zero between steps)              */
#define VMMT_STAT_COUNT 8

/* ---- attention ("general" Luong), one workgroup per sentence, source memory staged in LDS --------------------
 * forward: onmt/modules/GlobalAttention.py:113 (bmm), :171-176 (mask), :179-180 (softmax), :184 (bmm).
 *   q    T [Tp*B][ldq]   rows t*B+b: W_a r (linear_in already applied by vmmt_gemm)
 *   ctx  T [S*B][ldc]    rows s*B+b: encoder memory ; lens int64 [B]
 *   cat  T [Tp*B][ldcat] out: context vectors, columns [0,H) of the [c ; r] concat buffer (GlobalAttention.py:187)
 *   probs f32 [Tp][B][S] out: attention distributions (attns["std"])
 * limits: S <= 256, H <= 1024 (S <= 64: a lane / an MFMA tile row per source position; longer sources -- a sentence handed to
 * translate_mm_vi.py can be any length -- take a one-wave-per-query kernel). */
int vmmt_attn_fwd(int dtype, const void* q, int64_t ldq, const void* ctx, int64_t ldc, const int64_t* lens, void* cat,
                  int64_t ldcat, float* probs, int Tp, int B, int S, int H, void* stream);
/* backward: dcat columns [0,H) hold dL/dc; writes dq (T [Tp*B][lddq]) and dctx (T [S*B][lddx], zero at pads).
 * vmmt_attn_bwd: S <= 64 and T' <= 64 (VMMT_EINVAL beyond).  vmmt_attn_bwd_long: any T', S <= 256 -- two launches (per query:
 * dS and dQ; per source position: dHs) with `dots` f32 [Tp*B] as scratch between them. */
int vmmt_attn_bwd(int dtype, const void* dcat, int64_t lddc, const float* probs, const void* q, int64_t ldq,
                  const void* ctx, int64_t ldc, const int64_t* lens, void* dq, int64_t lddq, void* dctx, int64_t lddx,
                  int Tp, int B, int S, int H, void* stream);
int vmmt_attn_bwd_long(int dtype, const void* dcat, int64_t lddc, const float* probs, const void* q, int64_t ldq,
                       const void* ctx, int64_t ldc, const int64_t* lens, void* dq, int64_t lddq, void* dctx, int64_t lddx,
                       int Tp, int B, int S, int H, float* dots, void* stream);
/* hbar[b] = mean_{s<len_b} ctx[s][b]  (GlobalInferenceNetwork.encode_seq, modules/NormalVariationalEncoder.py:65-84) */
int vmmt_masked_mean(int dtype, const void* ctx, int64_t ldc, const int64_t* lens, void* out, int64_t ldo, int B, int S,
                     int H, void* stream);

/* ---- fused vocabulary projection + log-softmax + NLL ------------------------------------------------------------
 * generator Linear(H,V)+LogSoftmax (onmt/ModelConstructor.py:583-585), NLLLoss(weight[pad]=0, sum)
 * (onmt/Loss.py:163-165), accuracy (onmt/VILoss.py:515-531).  Logits are never written to memory in the forward.
 *   W T [V][ldw], bias f32 [V], O T [M][ldo] (M = Tp*B rows t*B+b), y int64 [M] = tgt[1:]
 *   workspaces: part_max/part_sum f32 [vmmt_gen_npart(V)][M], part_idx int32 same, tgt_logit f32 [M]
 *   out: lse f32 [M], tok_nll f32 [M]; stats[NLL,NWORDS,NCORRECT] += sums.
 *   part_idx may be NULL (training): no arg-max index is kept, and a token counts as correct when its target's logit
 *   equals the row maximum; decoding passes part_idx and reads the index back with vmmt_gen_argmax. */
int vmmt_gen_npart(int V);
/* arg-max over the vocabulary (and its logit) from the partials written by vmmt_gen_loss_fwd: the next input token of
 * step-wise decoding with beam size 1 (onmt/translate/TranslatorMultimodalVI.py:185-200); log-prob = out_max - lse */
int vmmt_gen_argmax(const float* part_max, const int* part_idx, int M, int npart, int64_t* out_idx, float* out_max, void* stream);
int vmmt_gen_loss_fwd(int dtype, const void* W, int64_t ldw, const float* bias, const void* O, int64_t ldo,
                      const int64_t* y, int M, int V, int K, int pad, float* part_max, float* part_sum, int* part_idx,
                      float* tgt_logit, float* lse, float* tok_nll, float* stats, void* stream);
/* backward seed: GT[v][m] = (exp(logit - lse[m]) - [v == y_m]) * (y_m != pad) * inv_norm   (T [V][ldgt], ldgt >= M).
 * The weight / input gradients are then vmmt_gemm calls: dWg = GT O (NN), dO = GT^T Wg (TN). */
int vmmt_gen_loss_bwd(int dtype, const void* W, int64_t ldw, const float* bias, const void* O, int64_t ldo,
                      const int64_t* y, int M, int V, int K, int pad, const float* lse, float inv_norm, void* GT,
                      int64_t ldgt, void* stream);

/* vmmt_gen_loss_bwd plus the generator bias gradient dbias[v] += sum_m G^T[v][m] (f32 [V], accumulated): fused into the bf16
 * kernel's write-out where that kernel applies, otherwise a row-sum pass over G^T behind it.
 * v_off > 0: the pass covers the vocabulary CHUNK [v_off, v_off + V) only -- W, bias, GT and dbias point at the chunk's first
 * row, y keeps whole-vocabulary ids (a caller short of memory for G^T [V][M] can walk the vocabulary in chunks of whole
 * 128-row tiles with a chunk-sized G^T; the results are bit-identical to one pass). */
int vmmt_gen_loss_bwd_db(int dtype, const void* W, int64_t ldw, const float* bias, const void* O, int64_t ldo,
                         const int64_t* y, int M, int V, int K, int pad, const float* lse, float inv_norm, void* GT,
                         int64_t ldgt, float* dbias, int v_off, void* stream);

/* ---- the same loss with dL/dO in ONE sweep of Wg (csrc/generator_fused.hip; bf16, K = 1024, 512 or 256) --------------
 * vmmt_gen_fwd_dO + vmmt_gen_fwd_combine replace vmmt_gen_loss_fwd + vmmt_gen_loss_bwd_db + the dO GEMM of the training step.
 *   vmmt_gen_fwd_dO: per 128-token block (64 at K = 1024) and vocabulary slice (vmmt-chosen: vmmt_gen_fused_geometry) a flash-attention-shaped sweep
 *     accumulates the softmax statistics and the un-normalised dO = sum_v P[m][v] Wg[v] into `ws`
 *     (vmmt_gen_fused_ws_floats(M, V, K) floats) and writes tgt_logit f32 [M].  W must be READABLE for w_rows >= (V rounded up to
 *     32) + 32 rows (the sweep prefetches whole 32-row tiles, one beyond the last; contents beyond V are ignored).  P != NULL (the training step): it also stores
 *     P T [M][ldp] (ldp >= V rounded up to 32), the un-normalised softmax weights exp(logit - ref_s[m]) of slice s = v / v_per_split.
 *   vmmt_gen_fwd_combine: folds the slices.  out: lse, tok_nll f32 [M]; stats[NLL,NWORDS,NCORRECT] += sums;
 *     dO f32 [M][lddo] = s_m (softmax_m Wg - Wg[y_m]), s_m = [y_m != pad] inv_norm; y32 int32 [(M+31)/32*32]: targets, -1 at pads
 *     and beyond M.  cs / Os != NULL (both or neither):
 *       cs  f32 [nsplit][mpad]: c_s[m] = s_m exp(ref_s[m] - lse_m), so that dL/dlogit[m][v] = P[m][v] c_s[m] - [v == y_m] s_m;
 *       Os  T [nsplit][os_stride]: O'_s[m][:] = c_s[m] O[m][:] (rows ld ldos; rows >= M are never written: keep them zero).
 * Then dWg[v in slice s][:] = sum_m P[m][v] O'_s[m][:] is ONE vmmt_gemm (GEMM_TN, K = M, b_batch_rows = v_per_split,
 * b_batch_stride = os_stride, plain store) and vmmt_gen_dW_finish adds the bias gradient and the one-hot term; where
 * vmmt_gemm_colsum_applies() holds the bias gradient's weighted column sums of P come out of that GEMM (colsum_w = cs, stride mpad,
 * colsum_out = dbias; entries of cs beyond M must be zero) and the finish call is told so (colsum_done = 1).
 * vmmt_gen_fused_applies() tells whether the shape is served (otherwise VMMT_EINVAL: the caller uses the G^T path above).
 * Reference: the same lines as above (ModelConstructor.py:583-585, Loss.py:129,163-165). */
int vmmt_gen_fused_applies(int dtype, int64_t ldw, int64_t ldo, int M, int V, int K);
int64_t vmmt_gen_fused_ws_floats(int M, int V, int K);
int vmmt_gen_fused_geometry(int M, int V, int K, int* nsplit, int* v_per_split, int64_t* mpad);
int vmmt_gen_fwd_dO(int dtype, const void* W, int64_t ldw, int w_rows, const float* bias, const void* O, int64_t ldo, const int64_t* y,
                    int M, int V, int K, float* ws, float* tgt_logit, void* P, int64_t ldp, const int32_t* rows, void* stream);
int vmmt_gen_fwd_combine(int dtype, const void* W, int64_t ldw, const void* O, int64_t ldo, const int64_t* y, int M, int V, int K,
                         int pad, float inv_norm, float* ws, const float* tgt_logit, float* lse, float* tok_nll, int* y32,
                         float* dO, int64_t lddo, float* stats, float* cs, void* Os, int64_t ldos, int64_t os_stride, const int32_t* rows,
                         void* stream);
/* vmmt_gen_fwd_combine in two launches, same arguments: _stats writes everything but dO (what the dWg product waits for: c_s, O'_s; and
 * lse, tok_nll, y32, the statistics -- folded by the launch's last workgroup: stats must hold VMMT_STAT_COUNT floats with
 * stats[VMMT_STAT_TICKET] zero), _dO writes dO alone (the fold of the slices' partial accumulators; cs / Os are ignored) -- the training
 * step issues the second one behind the point where the weight-gradient product starts. */
int vmmt_gen_fwd_combine_stats(int dtype, const void* W, int64_t ldw, const void* O, int64_t ldo, const int64_t* y, int M, int V, int K,
                               int pad, float inv_norm, float* ws, const float* tgt_logit, float* lse, float* tok_nll, int* y32,
                               float* dO, int64_t lddo, float* stats, float* cs, void* Os, int64_t ldos, int64_t os_stride, const int32_t* rows,
                               void* stream);
int vmmt_gen_fwd_combine_dO(int dtype, const void* W, int64_t ldw, const void* O, int64_t ldo, const int64_t* y, int M, int V, int K,
                            int pad, float inv_norm, float* ws, const float* tgt_logit, float* lse, float* tok_nll, int* y32,
                            float* dO, int64_t lddo, float* stats, float* cs, void* Os, int64_t ldos, int64_t os_stride, const int32_t* rows,
                            void* stream);
int vmmt_gen_dW_finish(int dtype, const void* P, int64_t ldp, const float* cs, const void* O, int64_t ldo, const int* y32, int M, int V,
                       int K, float inv_norm, float* dW, int64_t lddw, float* dbias, int colsum_done, const int32_t* rows, void* stream);
/* COMPACTED tokens (rows != NULL in the three calls above): pads carry loss weight zero (Loss.py:163-165) but a dense sweep spends FLOPs on
 * them -- 26 % of the decoder rows at target lengths U[10, 20].  vmmt_compact_nonpad lists the rows with y != pad in order (rows[j], j < n;
 * -1 for n <= j < Mc; count, optional int32[2]: [0] receives n, [1] is set to 1 -- and never cleared -- when n > Mc, i.e. when tokens were
 * left out; one workgroup, deterministic).  The generator calls then take M = Mc (n rounded up, e.g.
 * to 128) and `rows`: token m of the launch is row rows[m] of O, y, lse, tok_nll and dO (rows not listed are NOT written: clear them
 * beforehand), while P, y32, cs, Os, tgt_logit and the workspace are indexed by m; the dWg product runs over K = Mc tokens.  Same sums as the
 * dense calls up to the order of the f32 additions over tokens. */
int vmmt_compact_nonpad(const int64_t* y, int M, int pad, int Mc, int32_t* rows, int32_t* count, void* stream);

/* ---- row gathers / small fused kernels -------------------------------------------------------------------------- */
/* out[r][0:D] = table[ids[r]][0:D]; table f32 (embedding master weights: modules/Embeddings.py:181; or the HBM-resident
 * image-feature table: TrainerMultimodal.py:632-639), out f32 or bf16 per out_dtype. */
int vmmt_gather_rows(int out_dtype, const float* table, int64_t ldt, const int64_t* ids, void* out, int64_t ldo, int R,
                     int D, void* stream);
/* out[c] += sum_r X[r][c], and out2[c] likewise when not NULL (bias gradients; b_ih and b_hh share one).
 * blk > 0: the C columns are blocks of `blk` of which the first `valid` are summed into out[(c / blk) * valid + c % blk]
 * (padded gate blocks -> nn.LSTM's [4H] bias layout, see vmmt_gemm_args.c_row_blk); blk = 0: out[c] */
int vmmt_colsum(int dtype, const void* X, int64_t ld, int R, int C, int blk, int valid, float* out, float* out2, void* stream);
/* out[ids[r]][0:D] += X[r][0:D] (f32; rows with ids[r] == pad_id are dropped): the embedding-gradient scatter-add with padding_idx
 * (modules/Embeddings.py:118) behind the GEMM that produced X = dgates W_ih */
int vmmt_scatter_add_rows(const float* X, int64_t ldx, const int64_t* ids, int64_t pad_id, float* out, int64_t ldo, int R, int D,
                          void* stream);
/* out[r] += sum_c X[r][c]  (generator bias gradient: row sums of G^T) */
int vmmt_rowsum(int dtype, const void* X, int64_t ld, int R, int C, float* out, void* stream);
/* scaled dropout mask: 1/(1-p) with prob 1-p else 0 (counter-based RNG; VI_Model1.py:132, Models.py:124-129) */
int vmmt_dropout_mask(int dtype, void* mask, int64_t n, float p, uint64_t seed, void* stream);
int vmmt_randn(float* out, int64_t n, uint64_t seed, void* stream);
int vmmt_mul(int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int R, int C,
             void* stream);
/* out = dy * mask * act'(y) with act' expressed through the activation output y (mask / y may be NULL);
 * dy_f32 != 0: dy is an f32 buffer (a split-K accumulated gradient) whatever `dtype` is */
int vmmt_act_bwd(int dtype, int act, const void* dy, int64_t lddy, int dy_f32, const void* y, int64_t ldy,
                 const void* mask, int64_t ldm, void* out, int64_t ldo, int R, int C, void* stream);
/* fused mu/sigma -> sample -> KL: z = mu + sigma*eps (training) | mu (eval)  (Models.py:933, Dists.py:21-26, H2);
 * kl_b[b] = sum_k 0.5(mu^2+sigma^2-1) - log sigma (VILoss.py:446-456); stats[KL_SUM] += sum_b kl_b. */
int vmmt_latent_fwd(int dtype, const float* mu, const float* sigma, const float* eps, float* z32, void* zT, int64_t ldz,
                    float* kl_b, float* stats, int B, int Z, int training, void* stream);
/* q(z|x) forward in ONE launch: hbar = masked mean over time of the (detached) encoder memory (NormalVariationalEncoder.py:65-84),
 * h1 = relu(hbar W1^T + b1), out = h1 W2^T + b2 for the location and the scale network (:12-43; scale: Softplus), then
 * vmmt_latent_fwd's sample and KL.  bf16 only; H % 256 == 0, Z % 128 == 0, Z <= 512; returns 1 (invalid argument)
 * otherwise and the caller issues vmmt_masked_mean + 4 vmmt_gemm + vmmt_latent_fwd instead (same results up to f32 summation
 * order).  hbar / h1_* are written for the backward pass.
 * Z is the TILED latent size, Z_valid <= Z the model's (opts.py --z_latent_dim 500 -> Z 512, Z_valid 500): biases, eps, mu, sigma,
 * z32 hold Z_valid entries per row (row stride Z_valid), the weights' rows / columns beyond Z_valid must read as zeros, and the
 * sample / KL cover Z_valid lanes only.
 * split != 0: two workgroups per 16 sentences (location / scale network): mu and sigma are written, the sample and the KL are NOT
 * (z32 / zT / kl_b / stats untouched): follow with vmmt_latent_fwd. */
int vmmt_qnet_fwd(int dtype, const void* ctx, int64_t ldc, const int64_t* lens, const void* w1_loc, const void* w1_scale, int64_t ldw1,
                  const float* b1_loc, const float* b1_scale, const void* w2_loc, const void* w2_scale, int64_t ldw2,
                  const float* b2_loc, const float* b2_scale, const float* eps, void* hbar, int64_t ldh, void* h1_loc, void* h1_scale,
                  int64_t ldh1, float* mu, float* sigma, float* z32, void* zT, int64_t ldz, float* kl_b, float* stats, int B, int S,
                  int H, int Z, int Z_valid, int training, int split, void* stream);
/* d/d(mu, pre-softplus scale) of max(mult * KL_mean, margin) * inv_norm  (VILoss.py:460-473, Loss.py:129).
 * dz / eps (f32 [B][Z], both or NULL): the reparameterised gradient z = mu + sigma * eps NOT detached -- d mu += dz,
 * d sigma += dz * eps.  As executed the reference detaches the sample (hazard H2: modules/Dists.py:21-26, Models.py:930-933),
 * so the default passes NULL. */
int vmmt_latent_bwd(int dtype, const float* mu, const float* sigma, const float* kl_sum, float batch_global, float mult,
                    int use_freebits, float margin, float inv_norm, const float* dz, const float* eps, void* dmu, int64_t ld1,
                    void* dpre, int64_t ld2, int B, int Z, void* stream);
/* dL/dz of the reparameterised sample: sum_t dzrow[t*B + b] (= dgates_t W_z, the decoder-input path: VI_Model1.py:99-100) plus
 * the image network's gate path zt = z * sigmoid(w.z + bias) (NormalVariationalEncoder.py:286-293; dzt = dL/dzt). */
int vmmt_reparam_dz(const float* dzrow, int64_t ldr, int T, const float* dzt, int64_t ldd, const float* z, const float* g,
                    const float* w, float* out, int B, int Z, void* stream);
/* ---- conditional-prior variant (--conditional) -------------------------------------------------
 * p(z|x) = gen_net_global (onmt/Models.py:889), q(z|x,y,v) = GlobalFullInferenceNetwork
 * (onmt/modules/NormalVariationalEncoder.py:164-228), KL between two diagonal Gaussians (onmt/VILoss.py:437-452).
 * latent_cond_fwd: z = mu + sigma*eps (training) | mu_p (evaluation, Models.py:913); kl_b / stats[KL_SUM] as vmmt_latent_fwd.
 * latent_cond_bwd: d/d(mu, pre-softplus scale) of q and of p of max(mult * KL_mean, margin) * inv_norm. */
int vmmt_latent_cond_fwd(int dtype, const float* mu, const float* sigma, const float* mu_p, const float* sigma_p,
                         const float* eps, float* z32, void* zT, int64_t ldz, float* kl_b, float* stats, int B, int Z,
                         int training, void* stream);
int vmmt_latent_cond_bwd(int dtype, const float* mu, const float* sigma, const float* mu_p, const float* sigma_p,
                         const float* kl_sum, float batch_global, float mult, int use_freebits, float margin, float inv_norm,
                         const float* dz, const float* eps, void* dmu, int64_t ld1, void* dpre, int64_t ld2, void* dmu_p, int64_t ld3,
                         void* dpre_p, int64_t ld4, int B, int Z, void* stream);
/* masked mean of BATCH-major rows (row b*T + t): the output of encoder_tgt, which the reference runs over the transposed
 * target (Models.py:892-894, hazard H5); out[b] = mean_{t<len_b} x[b*T+t] (NormalVariationalEncoder.py:65-84) */
int vmmt_masked_mean_bm(int dtype, const void* x, int64_t ldx, const int64_t* lens, void* out, int64_t ldo, int B, int T, int H,
                        void* stream);
/* backward of vmmt_masked_mean (batch_major = 0, rows s*B+b) / vmmt_masked_mean_bm (1): dx[row] (+)= s < len_b ? dh[b]/len_b : 0 */
int vmmt_masked_mean_bwd(int dtype, const void* dh, int64_t lddh, const int64_t* lens, void* dx, int64_t lddx, int B, int S, int H,
                         int batch_major, int accumulate, void* stream);

/* image-network gate (modules/NormalVariationalEncoder.py:286-293) */
int vmmt_gate_fwd(int dtype, const float* z, const float* w, const float* bias, float* g, void* zt, int64_t ldzt, int B,
                  int Z, void* stream);
int vmmt_gate_bwd(const float* dzt, int64_t ldd, const float* z, const float* g, float* dw, float* db, int B, int Z,
                  void* stream);
/* image term of the ELBO as executed (H1: VILoss.py:39-44,321-331,408-415) + its gradient w.r.t. mu_v (dmu may be NULL) */
int vmmt_image_loss(int dtype, const float* mu_v, int64_t ldm, const float* img, int64_t ldi, int B, int D, float inv_norm,
                    void* dmu, int64_t ldd, float* stats, void* stream);
/* fp32 master weight [R][C] -> compute shadow T (optionally transposed; src2 optional second addend, e.g. b_ih + b_hh) */
int vmmt_pack(int dtype, const float* src, const float* src2, int64_t ld_src, void* dst, int64_t ld_dst, int R, int C,
              int transpose, void* stream);

/* many vmmt_pack operations in ONE launch.  `descs` is a device array of n vmmt_pack_desc; `total_chunks` the sum of
 * their `chunks` (2048 elements each: ceil(R*C / 2048), or for transpose != 0 ceil(R/64) * ceil(C/32) tiles of the source);
 * chunk_start must be the exclusive prefix sum of `chunks`. */
typedef struct vmmt_pack_desc {
  const float* src; const float* src2; void* dst;
  int64_t ld_src, ld_dst;
  int R, C, transpose, dtype;
  int chunk_start, chunks;
} vmmt_pack_desc;
int vmmt_pack_multi(const vmmt_pack_desc* descs, int n, int total_chunks, void* stream);

/* many buffers cleared in ONE launch (the gradient arena and the small accumulators of a step; replaces loss.backward()'s
 * implicit zero-filled .grad buffers and model.zero_grad(), TrainerMultimodal.py:628-629).  `descs`: device array of n
 * descriptors; ptr 16-byte aligned, bytes a multiple of 4; chunk_start = exclusive prefix sum of ceil(bytes / 16384);
 * total_chunks = their sum. */
typedef struct vmmt_zero_desc {
  void* ptr; int64_t bytes; int64_t chunk_start;
} vmmt_zero_desc;
int vmmt_zero_multi(const vmmt_zero_desc* descs, int n, int total_chunks, void* stream);

/* batch preparation in one launch: ids -> workspace, tgt[:-1] / tgt[1:] (Models.py:867, VILoss.py:205), lengths, image row
 * indices, statistics reset, optional eps ~ N(0,I) (eps may be NULL).  src [S][B], tgt [T][B], all int64 device arrays.
 * The workspace may hold more positions than the batch (S_ws >= S, T_ws >= T): the rest is filled with `pad`.
 * flags_src / flags_tgt (optional, int32 [2][R_*]): row flags of the lazily updated embedding tables (see vmmt_rows_mark below) -- every
 * row the batch looks up (o_src, o_tin: the pad row of bucketed positions included) is flagged for the update `gen` (>= 1). */
int vmmt_prepare_batch(const int64_t* src, const int64_t* tgt, const int64_t* src_len, const int64_t* idx, int S, int T, int B,
                       int S_ws, int T_ws, int pad, int64_t* o_src, int64_t* o_tin, int64_t* o_y, int64_t* o_len, int64_t* o_idx,
                       float* stats, float* eps, int64_t n_eps, uint64_t seed, int32_t* flags_src, int R_src, int32_t* flags_tgt,
                       int R_tgt, int gen, void* stream);

/* ---- beam search (translation) --------------------------------------------------------------------------------
 * One position of Beam.advance (onmt/translate/Beam.py:63-121) for all B sentences of a decoding batch of K*B rows
 * (row = k*B + b, TranslatorMultimodalVI.py:105-108).  `logits` f32 [K*B][V] (generator output BEFORE log-softmax; the
 * kernel applies it).  `scores` [B][K] in/out (Beam.scores), `cur_tok` [K*B] the tokens just fed (Beam.next_ys[-1]);
 * first != 0: only beam 0 is scored (Beam.py:91-92); mask_eos != 0: word_probs[:, eos] = -1e20 (min_length, Beam.py:77-80).
 * Outputs: next_tok [K*B] (next input rows), sel_rows [K*B] = parent row index prev_k*B + b (for vmmt_rows_select),
 * and this position's history slots hist_score / hist_prev / hist_next [B][K].  Ties: lowest flat index.  K <= 16.
 * `ws`: vmmt_beam_advance_ws_bytes(B, K, V) bytes of device scratch (per-chunk softmax statistics and candidates). */
int64_t vmmt_beam_advance_ws_bytes(int B, int K, int V);
int vmmt_beam_advance(const float* logits, int64_t ld, int B, int K, int V, const int64_t* cur_tok, float* scores, int first,
                      int mask_eos, int eos, int64_t* next_tok, int64_t* sel_rows, float* hist_score, int* hist_prev,
                      int64_t* hist_next, void* ws, int64_t ws_bytes, void* stream);
/* dst[r][0:row_bytes] = src[rows[r]][0:row_bytes] (byte strides; row_bytes even): re-orders decoder state rows by parent
 * beam (RNNDecoderState.beam_update, onmt/Models.py:589-594).  src and dst must not overlap. */
int vmmt_rows_select(const void* src, int64_t ld_src_bytes, const int64_t* rows, void* dst, int64_t ld_dst_bytes, int R,
                     int row_bytes, void* stream);

/* One decoded position's results -> their slot of the per-position history, indexed by a DEVICE counter:
 *   dst_i + counter * stride_bytes_i  <-  src_i[0 : bytes_i]   for the n <= VMMT_HIST_MAX_SEGS segments (4-byte granularity),
 * nothing when counter is outside [0, limit); then counter += bump.  With the position index on the device every position of a
 * decoding loop (variational_mmt_amd/decode.py; reference loop: TranslatorMultimodalVI.py:170-216) is the same launch sequence
 * with the same arguments, captured once as a hipGraph and replayed. */
#define VMMT_HIST_MAX_SEGS 6
typedef struct { const void* src; void* dst; int64_t bytes; int64_t stride_bytes; } vmmt_hist_seg;
int vmmt_history_append(const vmmt_hist_seg* segs, int n, int* counter, int limit, int bump, void* stream);

/* ---- image-feature table ------------------------------------------------------------------------------------------
 * In-place X[r][c] = (X[r][c] - mean[c]) / std[c] over the HBM-resident fp32 table [R][D] (row stride ld): the
 * `-use_standardised_image_features` branch, train_mm_vi_model1.py:499-501 (there: numpy on the host, fp32). */
int vmmt_standardise_rows(float* X, int64_t ld, const float* mean, const float* stdv, int64_t R, int D, void* stream);

/* ---- optimiser: clip_grad_norm + Adam over a flat fp32 arena (onmt/Optim.py:68-70,94-96) -------------------------- */
/* ||g||^2 of one arena segment into slot `slot` of `scratch` (f32[VMMT_SUMSQ_SCRATCH], zeroed once by the caller):
 *   scratch[0 .. SLOTS)            slot totals (plain store by the last workgroup of the launch)
 *   scratch[SLOTS .. 2 SLOTS)      uint32 tickets (self-resetting)
 *   scratch[2 SLOTS ...]           SLOTS x MAXBLOCKS per-workgroup partials
 * The partials are added in index order by the last-arriving workgroup: bit-reproducible, and identical on every
 * data-parallel rank (torch.nn.utils.clip_grad_norm of the reference is a host-side sum: Optim.py:94-95).
 * vmmt_adam_step's `sumsq` is this scratch: it adds the slot totals in index order. */
#define VMMT_SUMSQ_SLOTS 8
#define VMMT_SUMSQ_MAXBLOCKS 768
#define VMMT_SUMSQ_SCRATCH (2 * VMMT_SUMSQ_SLOTS + VMMT_SUMSQ_SLOTS * VMMT_SUMSQ_MAXBLOCKS)
int vmmt_sumsq(const float* g, int64_t n, float* scratch, int slot, void* stream);
/* max_blocks > 0 caps the grid (grid-stride loop).  shadow_bf16 (optional): the bf16 compute copy of THIS parameter range, same
 * flat layout (a 2-D weight whose shadow rows are unpadded): written along with p, so the shadow refresh need not read p back.
 * skip (optional): device int32[2]; skip[0] != 0 -> the launch changes NOTHING (p, m, v, shadow) and adds 1 to skip[1] -- the guard
 * word of the persistent recurrences (VMMT_SEQ_GUARD_WORD): a step whose gradients are known to be garbage is not applied. */
int vmmt_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                   int step, float max_norm, const float* sumsq, float grad_scale, int max_blocks, void* shadow_bf16,
                   const int32_t* skip, void* stream);
/* The same update over a range WITH A HOLE and a shadow that covers a PART of it, in one launch: the elements [0, span) without
 * [hole_lo, hole_lo + hole_n) -- a lazily updated embedding table (vmmt_adam_rows_step) in the middle of an arena segment -- and the bf16 copy
 * shadow_bf16[0 .. shadow_n) of the elements [shadow_lo, shadow_lo + shadow_n), which lie on one side of the hole.  All four offsets / lengths
 * are multiples of 4.  The same bits as one vmmt_adam_step per piece: the step's last update (Optim.py:94-96) is a handful of small pieces
 * whose launches, not their bytes, stood at the end of every step. */
int vmmt_adam_step_ranges(float* p, const float* g, float* m, float* v, int64_t span, int64_t hole_lo, int64_t hole_n, float lr, float beta1,
                          float beta2, float eps, int step, float max_norm, const float* sumsq, float grad_scale, int max_blocks,
                          void* shadow_bf16, int64_t shadow_lo, int64_t shadow_n, const int32_t* skip, void* stream);

/* Data parallelism with the optimiser state sharded over the ranks (one process per GPU; the reference has no multi-GPU path,
 * train_mm_vi_model1.py:73-75 -- the norm reproduced is clip_grad_norm's over ALL gradients of the global batch, Optim.py:94-95):
 * every rank norms its shards with vmmt_sumsq (one slot per arena segment);
 *   vmmt_dp_norm_pack   row[0 .. SLOTS) = the slot totals, row[SLOTS] = the bits of guard[0] (0 without a guard): the rank's 36-byte
 *                       contribution to ONE all-gather
 *   vmmt_dp_norm_fold   rows f32 [world][SLOTS + 1] (the gathered rows) -> sumsq[0] = sum over ranks of (sum over slots), in that fixed
 *                       order on every rank (identical clip coefficient everywhere), sumsq[1 .. SLOTS) = 0, guard[0] = max over ranks */
int vmmt_dp_norm_pack(const float* sumsq, const int32_t* guard, float* row, void* stream);
int vmmt_dp_norm_fold(const float* rows, int world, float* sumsq, int32_t* guard, void* stream);

/* ---- embedding tables: exact lazy Adam by row -----------------------------------------------------------------------------------
 * nn.Embedding tables (modules/Embeddings.py:118,181) receive gradient only in the rows a batch looks up, yet the reference clears,
 * norms and updates every element of each table at every step (loss.backward()'s zero-filled .grad, clip_grad_norm, torch.optim.Adam:
 * TrainerMultimodal.py:628-629, Optim.py:68-70,94-96): 36 B per element.  A row WITHOUT gradient still moves under Adam (m *= beta1,
 * v *= beta2, p -= step(m, v)), but that step depends only on the row's own state and the step's scalars: these entry points apply it
 * later, with the dense kernel's arithmetic in the dense kernel's order -- parameters and moments are BIT-identical to vmmt_adam_step
 * on a gradient that is zero outside the batch's rows (tests/test_gpu_row_adam.py) -- and touch only the rows a batch uses plus a
 * rolling 1 / roll of the table per step (no row is ever more than ~roll steps behind, whatever the distribution of the ids).
 *   flags int32 [2][R]                   generation numbers: row r is flagged for update t when flags[t & 1][r] == t; never cleared
 *                                        (two arrays by parity: vmmt_prepare_batch flags the next batch while a half of the last update,
 *                                        on another stream, still reads its own)
 *   last  int32 [R]                      the optimiser step each row is current for
 *   hist  int32 [VMMT_LAZY_HIST_WORDS]   [0] the last recorded step, [1] error word (0 = fine: a replay never met an overwritten
 *                                        entry), then a ring of VMMT_LAZY_HIST (step_size, 1 / sqrt(bc2), applied, step) entries
 *   vmmt_rows_mark      flags[g & 1][ids[i]] = g with g = hist[0] + 1, the next update (vmmt_prepare_batch does the same for a batch's
 *                       source and target ids with the update number passed by the host)
 *   vmmt_rows_catchup   mode 0: flagged rows are brought up to step hist[0] and their gradient rows cleared (between vmmt_rows_mark and
 *                       the lookup / the backward's scatter-add); mode 1: EVERY row is brought up to hist[0] (flush: checkpoints,
 *                       evaluation, before the dense kernels take over)
 *   vmmt_sumsq_rows     ||g||^2 over the flagged rows into slot `slot` of the norm scratch (deterministic, one launch); rowsq f32 [R]
 *                       scratch: block sums + a self-resetting ticket in its first ceil(R / 64) + 1 words (zero before the first call)
 *   vmmt_adam_rows_step update `step` (= hist[0] + 1): flagged rows with their gradient, rows r % roll == step % roll without one
 *                       (roll = 0: none; <= VMMT_LAZY_HIST / 4), records the step in the ring.  skip: as vmmt_adam_step -- the step
 *                       is recorded as skipped and replays leave it out.  C % 4 == 0, p / g / m / v 16-byte aligned. */
#define VMMT_LAZY_HIST 128
#define VMMT_LAZY_HIST_WORDS (4 + 4 * VMMT_LAZY_HIST)
int vmmt_rows_mark(const int64_t* ids, int64_t n, int32_t* flags, int R, const int32_t* hist, void* stream);
int vmmt_rows_catchup(float* p, float* g, float* m, float* v, int R, int C, const int32_t* flags, int32_t* last, int32_t* hist,
                      float beta1, float beta2, float eps, int mode, void* stream);
/* mode 0 with a bf16 copy of every flagged row written behind its catch-up: shadow [R][ld_shadow] (ld_shadow >= C, a multiple of 8: 16-byte
 * aligned rows; the columns beyond C are never written) -- the table the LSTM's input projection fetches its A operand from by token id
 * (vmmt_gemm_args.a_row_ids): current for exactly the rows the batch looks up. */
int vmmt_rows_catchup_shadow(float* p, float* g, float* m, float* v, int R, int C, const int32_t* flags, int32_t* last, int32_t* hist,
                             float beta1, float beta2, float eps, void* shadow_bf16, int64_t ld_shadow, void* stream);
int vmmt_adam_rows_step(float* p, float* g, float* m, float* v, int R, int C, const int32_t* flags, int32_t* last, int32_t* hist,
                        float lr, float beta1, float beta2, float eps, int step, int roll, float max_norm, const float* sumsq,
                        float grad_scale, const int32_t* skip, void* stream);
int vmmt_sumsq_rows(const float* g, int R, int C, const int32_t* flags, const int32_t* hist, float* rowsq, float* scratch, int slot,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif

/* libmultinn_hip -- C ABI of the MI355X-native LSTM-NADE / LSTM-RBM hot path of ilya16/MultINN.
 *
 * Conventions:
 *   - every function returns 0 (MNN_OK) or a negative error code; mnn_last_error() gives a
 *     thread-local message for the last failure on the calling thread;
 *   - every pointer is a CALLER-OWNED DEVICE pointer (e.g. a torch allocation); the library
 *     never allocates or frees user tensors; scratch is a caller-provided workspace whose size
 *     comes from the matching *_workspace_bytes() query;
 *   - every call takes a hipStream_t (as void*) and is asynchronous on that stream;
 *   - no global mutable state; RNG is passed as (seed, row0, sub) -- never global;
 *   - dtypes are explicit enums; "T" operands are bf16, f16 (IEEE half) or f32 (mnn_dtype).
 *
 * Each entry cites the reference interface it replaces as file:line under
 * /root/reference/multinn (ilya16/MultINN).
 *
 * SURVEY.md 8(b)'s proposed symbol list is exported in full since round 4: mnn_generate_scan runs the whole sampling scan of a generator in one
 * call (the Python mirror captures that call -- or, for the feedback modes, the grouped single-step calls -- into ONE hipGraph per shape:
 * multinn_amd/common.py ScanGraphs).
 * (mnn_comm_init / mnn_allreduce_flat / mnn_comm_destroy ARE exported since round 4 -- see "Data-parallel exchange" below; the Python mirror
 *  still issues its all-reduce through torch.distributed by default, backend "nccl" = the same RCCL.)
 */
#ifndef MULTINN_HIP_H
#define MULTINN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNN_OK 0
#define MNN_ERR_INVALID (-1)
#define MNN_ERR_HIP (-2)

typedef void* mnn_stream_t; /* hipStream_t */

typedef enum { MNN_F32 = 0, MNN_BF16 = 1, MNN_U8 = 2, MNN_F16 = 3 } mnn_dtype;   /* MNN_F16: IEEE half (precision "fp16"), accepted wherever the 16-bit "T" is */

/* RNG streams of the build's Philox4x32-10 contract (DESIGN.md "RNG contract") */
enum { MNN_STREAM_DROPOUT = 0, MNN_STREAM_NADE = 1, MNN_STREAM_RBM_H = 2, MNN_STREAM_RBM_V = 3,
       MNN_STREAM_DBN_ENC = 4, MNN_STREAM_DBN_DEC = 5 };

/* ABI version of THIS header: bumped whenever a signature or a descriptor struct changes.  mnn_version() returns the value the library
 * was built with; a loader must compare the two before its first call (multinn_amd/_lib.py load() does) -- a library built for another
 * version reads garbage arguments without any diagnosis otherwise.  121 (round 6): + mnn_gemm_tn_rows.  120 (round 6): mnn_step_increment gained `ls_dyn`, `ls_good`, `grow_after` (dynamic f16 loss scale).  119 (round 6): mnn_det_dense_job gained `Wp`, + mnn_det_dense_pack / _pack_bytes.  118 (round 6): mnn_det_lstm_job gained `Wp`, + mnn_det_lstm_pack / _pack_bytes, mnn_generate_scan_workspace_bytes gained `n_in`.  117 (round 6): + mnn_lstm_cluster_bwd_ok, + mnn_ragged_index / mnn_rows_gather16 / mnn_rows_scatter_f32, `n_rows_dev` on the gated NADE forwards and mnn_nade_logprob_bwd, `inv` / `hdr` on mnn_pianoroll_shift_timemajor_t, + mnn_lstm_resident_{fwd,bwd}_multi / mnn_lstm_cluster_{fwd,bwd}_multi / _bwd_multi_ok, `unsafe` on mnn_nade_logprob_fwd_gated, `unsafe` on mnn_nade_logprob_bwd.  116: + mnn_lstm_cluster_ok / _fwd / _bwd.  115: mnn_step_increment gained `sumsq`, `clip_norm`.  114: + mnn_lstm_resident_ok / _fwd / _bwd.  113: mnn_pianoroll_shift_timemajor_t gained `count`.  112: + mnn_generate_scan.  111: mnn_rbm_free_energy gained `p_h`.  110: mnn_clip_adam_step gained `skipped`; the dtype arguments of
 * mnn_pianoroll_shift_timemajor_t / mnn_grad_rows_fanout and the `f16` descriptor fields of round 3 are part of it. */
#define MNN_ABI_VERSION 121
int mnn_version(void);
const char* mnn_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Dense contractions on MFMA.   C[M,N] (op)= A[M,K] . B[N,K]^T (+ bias[N])
 * Both operands are K-contiguous ("TN" GEMM): A row-major [M,K] (lda), B row-major [N,K] (ldb).
 * dtype = MNN_BF16 (v_mfma_f32_32x32x16_bf16) or MNN_F32 (v_mfma_f32_32x32x2_f32, exact f32 fma
 * chain).  C is f32 unless c_dtype says bf16.  K, lda, ldb must be multiples of 8 (bf16) / 4 (f32).
 * flags: bit0 accumulate into C (C += ...), bit1 atomic accumulate (split-K, split>1).
 * Replaces: tf.matmul / tf.layers.Dense / LSTMBlockCell GEMMs -- rnn.py:124, rnn_nade.py:54-57,
 *           rnn_rbm.py:252-253, and their autodiff (generator.py:176-205).
 * ------------------------------------------------------------------------------------------ */
#define MNN_GEMM_ACCUMULATE 1
#define MNN_GEMM_ATOMIC 2
#define MNN_GEMM_A_KBLOCK32 8 /* A is stored K-blocked: element (m, k) at ((k >> 5) * lda + m) * 32 + (k & 31), lda = its row count (>= M).
                               * 16-bit operands, K % 64 == 0.  The layout a producer of dz^T writes in contiguous kilobytes (it owns 32 rows
                               * of a timestep = one block of 32 k): mnn_lstm_rowpar_bwd with L->ld_t == 0 emits it. */
/* (flag value 4, a K-major A read through transposing LDS loads, was measured net-neutral in round 3 and removed in round 4.) */
int mnn_gemm_tn(mnn_stream_t s, int dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                void* C, int ldc, int c_dtype, const float* bias, int flags, int split_k);
/* The same product on a COMPACTED ragged batch (mnn_ragged_index: valid rows first): m_rows_dev / k_rows_dev (device words, either may be NULL)
 * say how many rows of A (= of C) / how much of the K dimension carry data.  A hint the large-tile 16-bit kernels act on -- row tiles past
 * the count leave without writing (their C rows are never read), the K loop stops at the count (zeros lie behind it) -- and every other kernel
 * ignores: the result on the valid rows is the same either way. */
int mnn_gemm_tn_rows(mnn_stream_t s, int dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                     void* C, int ldc, int c_dtype, const float* bias, int flags, int split_k, const int* m_rows_dev, const int* k_rows_dev);

/* out[C,R] = in[R,C]^T with dtype conversion (in_dtype in {f32,bf16,f16,u8} -> out_dtype in {f32,bf16,f16});
 * used to build K-contiguous operands for weight-gradient GEMMs and transposed weight copies. */
int mnn_transpose(mnn_stream_t s, const void* in, int in_dtype, int R, int C, int ld_in, void* out, int out_dtype, int ld_out);

/* dst[r, c] = convert(src[r, c]) (2-D strided copy/convert; f32<->bf16/f16, u8->f32/bf16/f16) */
int mnn_convert2d(mnn_stream_t s, const void* src, int src_dtype, int ld_src, void* dst, int dst_dtype, int ld_dst, int R, int C);

/* ------------------------------------------------------------------------------------------
 * Piano-roll plumbing.
 * mnn_pianoroll_shift_timemajor: x u8 [B,T,D] (joint view of [B,T,P,M], feature p*M+m)
 *   -> inputs  T-dtype [T,B,ld_in]   inputs[t,b,:]  = x[b,t-1,:] (zeros at t=0, zero pad to ld_in)
 *   -> targets u8      [T,B,D]       targets[t,b,:] = x[b,t,:]
 *   -> row_weight f32  [T*B]         1/sum(lengths) if t<lengths[b] else 0   (lengths may be NULL)
 * Replaces: multinn_joint.py:83-89,132-139 (zero pad + inputs/targets slicing) and the row
 *           selection of utils/sequences.py:6-37 (as a weight mask; row order is time-major).
 * n_valid_total: sum of lengths over ALL ranks (data parallel); 0 -> computed from this batch.
 * ------------------------------------------------------------------------------------------ */
int mnn_pianoroll_shift_timemajor(mnn_stream_t s, const uint8_t* x, int B, int T, int D, const int32_t* lengths,
                                  void* inputs, int in_dtype, int ld_in, uint8_t* targets, float* row_weight,
                                  long n_valid_total);
/* Same plumbing for the bf16 train step, plus inputs_t[ld_in, ld_t >= B*T]: the transposed copy of `inputs` (feature-major,
 * column t*B+b), the K-major operand of layer 1's weight-gradient GEMM (saves a transpose pass over `inputs`). */
int mnn_pianoroll_shift_timemajor_t(mnn_stream_t s, const uint8_t* x, int B, int T, int D, const int32_t* lengths, void* inputs,
                                    int ld_in, void* inputs_t, int ld_t, uint8_t* targets, float* row_weight, long n_valid_total,
                                    int dtype /* MNN_BF16 or MNN_F16: the flavour of inputs / inputs_t */,
                                    unsigned* count /* optional u32 [MNN_DENSITY_SLOTS]: += set target cells (mnn_density_gate, v = NULL) */,
                                    const int32_t* inv /* optional [B*T]: targets and row_weight are written in COMPACT row order (mnn_ragged_index) */,
                                    const int32_t* hdr /* with inv: row_weight = hdr's 1/n_valid_total on compact rows < hdr[0], 0 behind them */);

/* Ragged windows without their padding (utils/sequences.py:6-37, rnn_nade.py:91-92,225: the reference drops padded rows before the NADE).
 * mnn_ragged_index builds, ON THE DEVICE from lengths[B], the permutation of the T*B time-major rows that puts the valid ones (t < lengths[b],
 * time-major order kept) first and the padding behind them:  idx[k] = row at compact position k,  inv[row] = its compact position, and a header
 *   hdr[0] = n_valid (int32, this rank's valid rows)      hdr[1] = 1 / n_total (f32 bits; n_total = *n_total_dev, or n_valid when NULL: the
 *   hdr[2] = loss scale 2^round(log2(scale_rows n_total))           valid rows of ALL ranks the mean-over-rows loss divides by)
 *   hdr[3] = 1 / hdr[2]                                    (scale_rows <= 0: 1)
 * so that a captured step serves any lengths: nothing of it is read on the host.  The LSTM still steps every row (dynamic_decode with
 * impute_finished=False, rnn_nade.py:204-218); Dense + NADE run on the compact rows: mnn_rows_gather16 copies the LSTM outputs into compact order
 * (rows >= n_valid zeroed; optionally also transposed, the K-major operand of the Dense weight gradient), the scans take n_rows_dev = hdr and
 * skip workgroups of padding (mnn_nade_logprob_*), and mnn_rows_scatter_f32 puts the Dense input gradient back into time-major order (padding
 * rows zeroed) for the LSTM backward. */
int mnn_ragged_index(mnn_stream_t s, const int32_t* lengths, int B, int T, const float* n_total_dev, float scale_rows, int32_t* idx, int32_t* inv,
                     int32_t* hdr /* 4 words */);
int mnn_rows_gather16(mnn_stream_t s, const void* src /* 16-bit [N, C], pitch ld_src */, int ld_src, const int32_t* idx, const int32_t* n_rows_dev,
                      int N, int C, void* dst /* [N, C], pitch ld_dst */, int ld_dst, void* dst_t /* optional [C, ld_t >= N] */, int ld_t);
int mnn_rows_scatter_f32(mnn_stream_t s, const float* src /* compact [N, C] */, const int32_t* inv, const int32_t* n_rows_dev, int N, int C,
                         float* dst /* time-major [N, C] */);

/* per-track variant: targets_tracks u8 [M,T,B,P] from x u8 [B,T,P,M]  (multi_encoder_nn.py:66-76) */
int mnn_pianoroll_split_tracks(mnn_stream_t s, const uint8_t* x, int B, int T, int P, int M, uint8_t* targets_tracks);

/* ------------------------------------------------------------------------------------------
 * LSTM (tf.contrib.cudnn_rnn.CudnnCompatibleLSTMCell == LSTMBlockCell, gate order i,ci,f,o,
 * forget_bias 0; rnn.py:104-145).  Internal pre-activation layout is gate-interleaved:
 * column (unit/32)*128 + gate*32 + unit%32; units must be a multiple of 32.
 *
 * mnn_lstm_pack_weights: natural TF kernel W[(in+u),4u] f32 + bias[4u] ->
 *   wx_t  T [4u, ld_in]   (permuted rows, K-contiguous; input-projection B operand)
 *   wh_t  T [4u, u]       (permuted rows; recurrent B operand)
 *   wh_p  T [u, 4u]       (permuted columns; recurrent dgrad B operand)
 *   wx_p  T [in, 4u]      (permuted columns; input dgrad B operand)  (may be NULL)
 *   bias_p f32 [4u]       (permuted)
 * ------------------------------------------------------------------------------------------ */
int mnn_lstm_pack_weights(mnn_stream_t s, const float* W, const float* bias, int n_in, int units, int dtype, int ld_in,
                          void* wx_t, void* wh_t, void* wh_p, void* wx_p, float* bias_p);

/* scatter-add the gate-interleaved gradient (dWx_t[4u,ld_in] f32, dWh_t[4u,u] f32, db_p[4u]) back
 * into the natural TF layout dW[(in+u),4u], db[4u] (accumulating). */
int mnn_lstm_unpack_grads(mnn_stream_t s, const float* dwx_t, const float* dwh_t, const float* db_p, int n_in, int units,
                          int ld_in, float* dW, float* db);
/* Same, and the three packed sources are zeroed as they are read (each element by the thread that read it): persistent accumulators
 * for the split-K weight-gradient GEMMs (MNN_GEMM_ACCUMULATE) and the recurrence's bias sums, no zero fill per step. */
int mnn_lstm_unpack_grads_consume(mnn_stream_t s, float* dwx_t, float* dwh_t, float* db_p, int n_in, int units, int ld_in, float* dW,
                                  float* db);
/* Consuming form for ONE packed matrix dw_cat f32 [4u, ld_in + u] = [dWx^T | dWh^T]: the output of a single weight-gradient GEMM
 * dz^T . [x^T ; h_prev^T]^T over the concatenated operand (dz^T is then streamed once for both gradients; rnn.py:60-62's kernel is
 * [(in + u), 4u], i.e. the two blocks are rows of one variable). */
int mnn_lstm_unpack_grads_cat(mnn_stream_t s, float* dw_cat, float* db_p, int n_in, int units, int ld_in, float* dW, float* db);

/* One layer over steps [t_begin, t_end) of a sequence, time-major (chunked calls let the layers of a
 * stack run as a wavefront on separate streams).  xproj f32 [T,B,4u] = inputs . Wx + b (already
 * computed with mnn_gemm_tn).  Runs one fused {recurrent GEMM + gate pointwise} kernel per step;
 * step t > 0 reads its previous state from h[t-1], c[t-1].
 *   h0,c0 may be NULL (zero state, rnn.py:155-176).
 *   gates f32 [T,B,4u] (post-activation i,g,f,o; saved for backward; may be NULL for inference)
 *   c f32 [T,B,u], h T [T,B,u]
 * Replaces: dynamic_decode/dynamic_rnn over the cell -- rnn_nade.py:204-218, rnn_rbm.py:217-223. */
/* hT (optional, T dtype [u, ld_hT], ld_hT >= T*B): transposed PREVIOUS-state operand of the recurrent
 * weight gradient, hT[:, (t+1)*B + b] = h[t, b, :] for t < T-1 (columns [0,B) are the caller's h0^T). */
int mnn_lstm_seq_fwd(mnn_stream_t s, int dtype, int T, int B, int units, int t_begin, int t_end, const float* xproj,
                     const void* wh_t, const void* h0, const float* c0, float* gates, float* c, void* h, void* hT, int ld_hT);

/* BPTT for one layer over steps [t_begin, t_end), processed downwards; successive calls must cover
 * the sequence from the top range to 0 and share `workspace` (it carries d c between calls).
 * dh_ext f32 [T,B,u] (gradient arriving at h_t from above), dz f32 [T,B,4u] out (gate-interleaved
 * pre-activation gradient), dh0/dc0 f32 [B,u] out with the t_begin == 0 call (may be NULL).
 * workspace: mnn_lstm_seq_bwd_workspace_bytes(B, units). */
size_t mnn_lstm_seq_bwd_workspace_bytes(int B, int units);
/* dzT_t (optional, T dtype [4u, ld_t], ld_t >= T*B): transposed dz, dzT_t[col, t*B + b] = dz[t, b, col] (the
 * K-contiguous operand of both weight-gradient GEMMs); db_p (optional f32 [4u]): += sum over rows of dz.
 * The bf16 step kernels (units 128/256/512) produce both in their epilogue; then dz (f32) may be NULL. */
int mnn_lstm_fused_outputs(int dtype, int units);   /* 1 if the step kernels emit hT / dzT_t / db_p themselves */
int mnn_lstm_seq_bwd(mnn_stream_t s, int dtype, int T, int B, int units, int t_begin, int t_end, const float* dh_ext,
                     const void* wh_p, const float* gates, const float* c, const float* c0, float* dz,
                     void* dz_T /* T copy of dz or NULL */, float* dh0, float* dc0, void* workspace, void* dzT_t, int ld_t,
                     float* db_p);

/* Two-layer wavefront (bf16, both layers' units in {128,256,512}): ONE launch per timestep for the whole stack,
 * three stages, lag 2.  Forward launch s in [0, T+2): layer-1 step s | layer-2 input projection of step s-1
 * (xproj2 = y1 . Wx2 + b2, written into L2->xproj) | layer-2 step s-2.  Backward launch k in [0, T+2): layer-2 step
 * T-1-k | layer-1 incoming gradient of step T-k (dh1 = (dz2 . Wx2^T) * keep/kp, written into L1->dh_ext) | layer-1
 * step T+1-k.  Dropout (rnn.py:132) uses keep masks precomputed by mnn_dropout_mask (u8 [T,B,u], same Philox
 * counters as mnn_dropout_fwd); y = h/kp*mask is written by the step kernels.  Other pointers as in
 * mnn_lstm_seq_fwd / _bwd; backward needs no f32 dz, no dh0/dc0. */
typedef struct { int units; const float* xproj; const void* wh_t; const void* h0; const float* c0; float* gates; float* c; void* h;
                 void* hT; int ld_hT; void* y; const uint8_t* mask; const void* wx_t; int ld_w; const float* bias_p;
                 void* yT; int ld_yT; /* persistent form only (else NULL): transposed copy of the layer's output (y, or h without
                                         dropout), yT[unit][t*B + row] -- the K-contiguous operand of the next weight gradient */
                 int xproj_bf16;      /* mnn_lstm_rowpar_fwd only: xproj points at 16-bit values [T,B,4u] (gate-minor, bias included) instead of f32 */
                 int f16;             /* persistent forms (mnn_lstm2_persist_*, mnn_lstm_rowpar_*): every 16-bit tensor of this layer (weights, h, y, hT, yT,
                                         16-bit xproj / saved gates) is IEEE half instead of bfloat16 */
               } mnn_lstm_fwd_layer;
typedef struct { int units; const float* dh_ext; const void* wh_p; const float* gates; const float* c; const float* c0; float* dz;
                 void* dz_T; void* workspace; void* dzT_t; int ld_t; float* db_p; const uint8_t* mask; const void* wx_p;
                 int f16;             /* as in mnn_lstm_fwd_layer */
               } mnn_lstm_bwd_layer;
int mnn_lstm2_seq_fwd(mnn_stream_t s, int T, int B, const mnn_lstm_fwd_layer* L1, const mnn_lstm_fwd_layer* L2, float keep_prob,
                      int s_begin, int s_end);
int mnn_lstm2_seq_bwd(mnn_stream_t s, int T, int B, const mnn_lstm_bwd_layer* L1, const mnn_lstm_bwd_layer* L2, float keep_prob,
                      int k_begin, int k_end);

/* Persistent form of the two-layer recurrence: ONE launch runs all T steps of both layers (replaces the T+2 launches of
 * mnn_lstm2_seq_fwd / _bwd; same layer descriptors, same outputs).  A workgroup keeps its slice of the recurrent weights in
 * registers for the whole sequence and hands 32-row state tiles to the workgroups of the same row tile through
 * write-through stores into an exchange area + progress flags (multinn_amd/csrc/lstm_persist.hip).  Layer 2's input
 * projection is folded into its step, so L2->xproj is not used (may be NULL); backward, L1->dh_ext and both dz_T are
 * not used (may be NULL) and dz must be NULL.  Backward, L2->mask (optional): when given, L2->dh_ext is the gradient wrt
 * layer 2's DROPPED output and the launch applies the dropout backward itself (dh_ext / keep_prob * mask, the arithmetic of
 * mnn_dropout_bwd); when NULL, dh_ext is taken as the gradient wrt h2.
 * workspace: mnn_lstm2_persist_workspace_bytes(T,B,u1,u2) bytes of device memory, 256-byte aligned, zeroed ONCE by the
 * caller at allocation (progress flags, a sticky give-up word, the exchange area); each call re-zeroes its progress
 * words itself (a fill kernel: no memset node under hipGraph capture); forward and backward calls may share one workspace.
 * mnn_lstm2_persist_status copies the sticky word to the host (synchronises): non-zero after any launch that gave up
 * on a bounded spin -- its outputs are then garbage.  mnn_lstm2_persist_ok() says whether the grid fits this device at
 * once (one workgroup per CU); the entries refuse shapes for which it does not. */
int mnn_lstm2_persist_ok(int B, int units1, int units2);
/* The persistent form keeps its saved gates [T,B,4u] (both layers) and READS layer 1's xproj GATE-MINOR: column unit*4 + g
 * instead of the gate-interleaved (unit/32)*128 + g*32 + unit%32 -- the four values of a (row, unit) are one 16-byte access.
 * mnn_lstm_rows_gate_minor re-orders the rows of a packed wx_t [4u, ld] (and bias_p) accordingly, so the ordinary projection
 * GEMM produces that xproj.  Gates written by mnn_lstm2_persist_fwd are only meaningful to mnn_lstm2_persist_bwd. */
int mnn_lstm_rows_gate_minor(mnn_stream_t s, int dtype, int units, int ld, const void* wx_t, const float* bias_p, void* wx_gm, float* bias_gm);
size_t mnn_lstm2_persist_workspace_bytes(int T, int B, int units1, int units2);
int mnn_lstm2_persist_status(const void* workspace, int B, int units1, int units2, int* status);
int mnn_lstm2_persist_fwd(mnn_stream_t s, int T, int B, const mnn_lstm_fwd_layer* L1, const mnn_lstm_fwd_layer* L2, float keep_prob,
                          void* workspace);
int mnn_lstm2_persist_bwd(mnn_stream_t s, int T, int B, const mnn_lstm_bwd_layer* L1, const mnn_lstm_bwd_layer* L2, float keep_prob,
                          void* workspace);
/* Row-parallel persistent recurrence of ONE layer (multinn_amd/csrc/lstm_rowpar.hip), the form for B >= 512: a workgroup keeps its 32-unit
 * slice of the recurrent weights in LDS, every WAVE owns one 32-row tile for the whole sequence (no K split, no workgroup barrier in the
 * loop); the layers run as separate launches with the next layer's input projection (mnn_gemm_tn) between them.  Same layer descriptors as
 * above, with: forward -- L->xproj GATE-MINOR f32 (or, with L->xproj_bf16, bf16: half the bytes of the step's largest tensor, written by
 * mnn_gemm_tn with a bf16 C) [T,B,4u] including the bias (mnn_lstm_rows_gate_minor); L->gates points at BF16 [T,B,4u]
 * (gate-minor; half the bytes of the other forms' f32 copy: the saved activations only feed products that are rounded to bf16 anyway, and
 * only mnn_lstm_rowpar_bwd reads them); with a keep mask (L->mask, L->y) the layer's output is y and L->h receives ONLY its last timestep
 * (the final state) -- the rows of h[0 .. T-2] are left untouched; L->wx_t / bias_p unused, no initial state (h0 = c0 = NULL: a window starts from the zero state, train.py:165-173); backward -- L->dh_ext f32 [T,B,u]
 * required (the gradient wrt the layer's output; with L->mask it is taken wrt the DROPPED output and dh_ext / keep_prob * mask is applied
 * here), L->dz_T (optional) receives dz bf16 [T,B,4u] row-major in the gate-interleaved column order (the A operand of the input-gradient
 * GEMM against wx_p), L->dzT_t / db_p as in the persistent form, L->workspace / wx_p / dz unused.  B must be a multiple of 32.
 * workspace: mnn_lstm_rowpar_workspace_bytes(T,B,u) bytes, 256-byte aligned, zeroed ONCE at allocation; forward and backward calls of the
 * same layer may share it.  Up to two row tiles per workgroup every (row tile, unit tile) item is worked by a PAIR of waves (forward: the
 * gate columns split between them, backward: K split; MNN_ROWPAR_NO_PAIR=1 keeps one wave per item).  mnn_lstm_rowpar_status: the sticky give-up word (non-zero: a bounded spin gave up, outputs are garbage). */
int mnn_lstm_rowpar_ok(int B, int units);
size_t mnn_lstm_rowpar_workspace_bytes(int T, int B, int units);
int mnn_lstm_rowpar_status(const void* workspace, int* status);
int mnn_lstm_rowpar_fwd(mnn_stream_t s, int T, int B, const mnn_lstm_fwd_layer* L, float keep_prob, void* workspace);
int mnn_lstm_rowpar_bwd(mnn_stream_t s, int T, int B, const mnn_lstm_bwd_layer* L, float keep_prob, void* workspace);
/* CU-resident recurrence of ONE 256-unit layer (multinn_amd/csrc/lstm_resident.hip; rnn.py:104-145 as above): a workgroup owns four batch
 * rows for the whole sequence and keeps the layer's ENTIRE recurrent matrix on its CU (three quarters in the registers of its four waves,
 * one quarter in LDS), so a timestep hands nothing between workgroups -- no flags, no exchange area, no workspace, no co-residency
 * requirement (any grid size runs).  Same layer descriptors, inputs, outputs and layouts as mnn_lstm_rowpar_fwd (16-bit gate-minor xproj
 * REQUIRED: L->xproj_bf16 != 0; the saved gates, hT and yT come together or not at all); units must be 256 and B a multiple of 4
 * (mnn_lstm_resident_ok); every tensor of the call below 2 GB (buffer descriptors). */
int mnn_lstm_resident_ok(int B, int units);
int mnn_lstm_resident_fwd(mnn_stream_t s, int T, int B, const mnn_lstm_fwd_layer* L, float keep_prob);
/* ... and its backward: descriptor, outputs and layouts of mnn_lstm_rowpar_bwd (dh_ext required; dz_T, dzT_t / ld_t (0 = K-blocked), db_p optional;
 * workspace / wx_p / dz unused).  The saved gates must be the 16-bit gate-minor copy written by mnn_lstm_resident_fwd / mnn_lstm_rowpar_fwd. */
int mnn_lstm_resident_bwd(mnn_stream_t s, int T, int B, const mnn_lstm_bwd_layer* L, float keep_prob);
/* Cluster form of the CU-resident recurrence for ONE 512-unit layer (multinn_amd/csrc/lstm_cluster.hip; rnn.py:104-145 as above): eight
 * workgroups -- one per CU, on one XCD -- share 32 batch rows for the whole sequence; each keeps the recurrent weights of 64 units in its waves'
 * registers and the members exchange h[t] (backward: dz[t]) through a two-deep area in their XCD's L2 (progress flags, bounded spins, the sticky
 * status word of mnn_lstm_rowpar_status).  Same layer descriptors, inputs, outputs, layouts AND workspace as mnn_lstm_rowpar_fwd / _bwd with a
 * 16-bit gate-minor xproj (the caller may use either form on the same buffers).  units == 512, B a multiple of 256 with B / 4 <= the device's
 * CUs (mnn_lstm_cluster_ok; the whole grid must be resident at once: never next to another persistent launch); every tensor below 2 GB. */
int mnn_lstm_cluster_ok(int B, int units);
int mnn_lstm_cluster_fwd(mnn_stream_t s, int T, int B, const mnn_lstm_fwd_layer* L, float keep_prob, void* workspace);
/* ... and its backward: descriptor, outputs and layouts of mnn_lstm_rowpar_bwd (dh_ext required; dz_T, dzT_t / ld_t (0 = K-blocked), db_p optional;
 * wx_p / dz unused).  The contraction over the 2048 gate columns is split the way the columns are owned: a member multiplies its own 256 columns
 * of dz[t+1] (in its LDS: nothing is exchanged in front of the MFMAs) into partial sums for all 512 units, and the members reduce-scatter the
 * partials (16-bit, 32 KB out + 32 KB in per CU and step).  T >= 4.  The two-deep exchange area is only valid while a cluster's eight workgroups
 * share an XCD: mnn_lstm_cluster_bwd_ok answers that ON THE HOST (one probe launch of the kernels' grid per device and batch size, cached; also
 * 0 under MNN_PERSIST_NO_LOCAL) and the caller takes mnn_lstm_rowpar_bwd -- same descriptor, same saved activations -- when it says 0.  A launch
 * that nevertheless finds a cluster off one XCD gives up (status word) instead of computing on stale lines. */
int mnn_lstm_cluster_bwd_ok(int B, int units);
int mnn_lstm_cluster_bwd(mnn_stream_t s, int T, int B, const mnn_lstm_bwd_layer* L, float keep_prob, void* workspace);
/* Several INDEPENDENT layers of one shape in one launch: the M per-track generators of the jamming mode (multinn_jamming.py:40-68,213-221 trains
 * them on one loss; each has its own LSTM).  L: array of njobs (1..8) descriptors sharing T, B, units, keep_prob, precision, dropout and save
 * mode; every job keeps its own tensors and -- cluster form -- its own workspace.  The CU-resident and cluster recurrences own row groups for
 * the whole sequence and hand nothing to another group, so the jobs' groups share one grid (njobs * B / 4 workgroups, resp. 8 * njobs * B / 32 in
 * rounds when that exceeds the device: a cluster's members are neighbours in dispatch order).  mnn_lstm_cluster_bwd_multi_ok: the placement
 * answer of mnn_lstm_cluster_bwd_ok for that larger grid. */
int mnn_lstm_resident_fwd_multi(mnn_stream_t s, int T, int B, int njobs, const mnn_lstm_fwd_layer* L, float keep_prob);
int mnn_lstm_resident_bwd_multi(mnn_stream_t s, int T, int B, int njobs, const mnn_lstm_bwd_layer* L, float keep_prob);
int mnn_lstm_cluster_fwd_multi(mnn_stream_t s, int T, int B, int njobs, const mnn_lstm_fwd_layer* L, float keep_prob, void* const* workspaces);
int mnn_lstm_cluster_bwd_multi(mnn_stream_t s, int T, int B, int njobs, const mnn_lstm_bwd_layer* L, float keep_prob, void* const* workspaces);
int mnn_lstm_cluster_bwd_multi_ok(int B, int units, int njobs);
int mnn_dropout_mask(mnn_stream_t s, uint8_t* mask, int T, int B, int units, float keep_prob, uint64_t seed, const int32_t* step_dev,
                     uint32_t row0, int layer);

/* Output dropout of DropoutWrapper (rnn.py:132): y = h/kp * floor(kp+u), u = Philox(stream 0,
 * row = row0+b, sub = (t<<8)|layer, elem = unit).  h,y T [T,B,u].  kp>=1 -> copy.
 * step_dev (optional device int32): the effective seed is seed + *step_dev, so a captured hipGraph
 * of the train step draws fresh masks on every replay.  t_offset: absolute time index of h[0] (chunked calls). */
int mnn_dropout_fwd(mnn_stream_t s, int dtype, const void* h, void* y, int T, int B, int units, float keep_prob,
                    uint64_t seed, const int32_t* step_dev, uint32_t row0, int layer, int t_offset);
/* dh[t,b,j] (+)= dy[t,b,j]/kp*keep  (f32 in/out, mask recomputed) */
int mnn_dropout_bwd(mnn_stream_t s, const float* dy, float* dh, int T, int B, int units, float keep_prob, uint64_t seed,
                    const int32_t* step_dev, uint32_t row0, int layer, int accumulate, int t_offset);

/* NADE log-prob on the matrix cores (multinn_amd/csrc/nade_mfma.hip), Hn == 256 only (mnn_nade_mfma_ok): same contract and
 * outputs as mnn_nade_logprob_fwd, but the D x Hn decoder dot products of a row run as a block-sparse bf16 GEMM over the
 * row's 1 + nnz(v) distinct hidden states (f32 accumulation; the encoder sums stay f32).  w_dec_bf16: bf16 copy of w_dec,
 * [tracks,D,Hn] (mnn_convert2d).  Used by the bf16 compute mode; the f32 entries above remain the parity path. */
int mnn_nade_mfma_ok(int Hn);
int mnn_nade_logprob_fwd_mfma(mnn_stream_t s, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride, const float* bias,
                              int ld_bias, const float* w_enc, const void* w_dec_bf16, const float* row_weight, float* nll, float* cond_p,
                              float* d_bias, float* a_final);
/* Density-gated pair (bf16 compute mode, nade.py:155-229 unchanged): the matrix-core form's cost grows with the number of ACTIVE visibles
 * (one hidden state per active visible and row), the f32 vector form's hardly does (MI355X, [1024,256,88,5]: 2.4 vs 3.6 ms at density 0.03,
 * 19.7 vs 6.6 ms at 0.5).  mnn_density_gate counts the non-zero bytes of v ON THE DEVICE and writes gate[0] = (count > threshold); `count` is
 * a zeroed scratch array of MNN_DENSITY_SLOTS u32 partial counts (spread so that the adds do not serialise on one address), left zero.
 * v == NULL: the partial counts are already there (mnn_pianoroll_shift_timemajor_t counted while it wrote v) and only the decision runs.
 * The two *_gated entries are then both launched and each returns at once unless gate[0] == run_if
 * (gate NULL: always runs).  A captured step so takes the cheaper form at every replay, whatever batch it is fed.
 * mnn_nade_logprob_fwd_gated with a gate and run_if = 1 (the DENSE launch of the pair) and Hn > 128 advances the hidden states multiplicatively
 * (u = exp(-a), one multiply by exp(-w_enc[i]) per flip, h = 1 / (1 + u); `a` -- and a_final -- is still the exact sum and re-derives u whenever
 * |a| passes 40): conditionals and NLL within 2e-5 of the direct-sigmoid form, which every other call of this entry point (and
 * mnn_nade_logprob_fwd, the f32 parity path) keeps. */
#define MNN_DENSITY_SLOTS 256
int mnn_density_gate(mnn_stream_t s, const uint8_t* v, long n, long threshold, int* gate, unsigned* count);
int mnn_nade_logprob_fwd_gated(mnn_stream_t s, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride,
                               const float* bias, int ld_bias, const float* w_enc, const float* w_dec, const float* row_weight,
                               float* nll, float* cond_p, float* d_bias, float* a_final, const int* gate, int run_if,
                               const int* n_rows_dev /* optional: see mnn_ragged_index */,
                               int* unsafe /* optional int32[1]: cleared by this entry point, then counted up by every wave of the multiplicative
                                              form (below) a row of which passed |a| = 40 (and once by any other form or a gated-out launch) -- 0 is
                                              mnn_nade_logprob_bwd's licence to carry exp(-a) */);
int mnn_nade_logprob_fwd_mfma_gated(mnn_stream_t s, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride, const float* bias,
                                    int ld_bias, const float* w_enc, const void* w_dec_bf16, const float* row_weight, float* nll, float* cond_p,
                                    float* d_bias, float* a_final, const int* gate, int run_if, const int* n_rows_dev);
/* Split-operand form of the matrix-core scan (precision "fp16": nade.py:199-221 within 1e-4 on every conditional, which 8- or 11-bit operands
 * of the decoder dot products miss by 2-10x): every hidden state and decoder weight is carried as an IEEE-half pair hi + lo (22 significant
 * bits) and a logit is the f32 sum of the three 16-bit MFMA products hi.hi + hi.lo + lo.hi (~1e-6 of the f32 vector scan).  w_dec_packed:
 * mnn_nade_f32_pack's output, f16 [tracks*D][hi | lo][Hn] in a buffer of the f32 original's size (rows * Hn * 4 bytes, 16-byte aligned).
 * Same arguments and gate convention as mnn_nade_logprob_fwd_mfma_gated (gate NULL: always runs). */
int mnn_nade_f32_pack(mnn_stream_t s, const float* w_dec, long rows, int Hn, float* w_dec_packed);
int mnn_nade_logprob_fwd_mfma_f32(mnn_stream_t s, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride, const float* bias,
                                  int ld_bias, const float* w_enc, const void* w_dec_packed, const float* row_weight, float* nll, float* cond_p,
                                  float* d_bias, float* a_final, const int* gate, int run_if, const int* n_rows_dev);

/* ------------------------------------------------------------------------------------------
 * NADE (models/common/nade.py).  Weights w_enc,w_dec f32 [tracks,D,Hn].  Rows: v u8
 * [tracks][N][D] (track stride v_track_stride, row stride D); biases come from one Dense output
 * matrix `bias` f32 [N, ld_bias]: b_enc of track m at column m*Hn, b_dec at tracks*Hn + m*D
 * (rnn_nade.py:245; rnn_multinade.py:242-249).  Hn <= 256.
 *
 * mnn_nade_logprob_fwd (nade.py:155-229): nll f32 [tracks,N], cond_p f32 [tracks,N,D];
 *   if row_weight != NULL also dl f32 [N, ld_bias] columns tracks*Hn.. := d(sum_n w[n]*nll)/d b_dec;
 *   a_final f32 [tracks,N,Hn] (optional) receives the final hidden pre-activation for the backward.
 * mnn_nade_logprob_bwd: reverse scan from a_final -> d_bias[:, m*Hn..] (= d b_enc), d_w_enc, d_w_dec f32
 *   [tracks,D,Hn] (ACCUMULATED with atomics: zero them first).  Exploits v sparsity exactly: h is
 *   recomputed only where v = 1.
 * mnn_nade_sample (nade.py:231-308): deterministic-order kernel, Bernoulli u < sigmoid(l/T);
 *   u = Philox(stream 1, row = row0+n, sub, elem = m*D+i); temperature <= 0 -> threshold 0.5.
 *   D <= 1536 (a row's logits, b_dec and draws are parked in LDS during the scan); Hn <= 256.
 * ------------------------------------------------------------------------------------------ */
int mnn_nade_logprob_fwd(mnn_stream_t s, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride,
                         const float* bias, int ld_bias, const float* w_enc, const float* w_dec, const float* row_weight,
                         float* nll, float* cond_p, float* d_bias, float* a_final);
int mnn_nade_logprob_bwd(mnn_stream_t s, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride,
                         const float* bias, int ld_bias, const float* w_enc, const float* w_dec, const float* a_final,
                         float* d_bias, float* d_w_enc, float* d_w_dec, const int* n_rows_dev /* optional: see mnn_ragged_index */,
                         const int* unsafe /* optional: the counter the forward's density-gated dense launch left (mnn_nade_logprob_fwd_gated).
                                              0 = a dense batch every row of which the forward's multiplicative form vouched for: the reverse scan
                                              then advances exp(-a) multiplicatively like the dense forward (no exponential per flip); both
                                              instantiations are launched, one leaves at once.  NULL: the direct form */);
int mnn_nade_sample(mnn_stream_t s, int tracks, int N, int D, int Hn, const float* bias, int ld_bias, const float* w_enc,
                    const float* w_dec, float temperature, uint64_t seed, uint32_t row0, uint32_t sub, uint8_t* samples,
                    long s_track_stride, int s_row_stride, int s_elem_stride, float* nll);
/* mnn_nade_sample for SEVERAL single-NADE generators in ONE launch (multinn_feedback.py:196: every per-track generator's sample_single in a
 * step of the feedback scan).  `jobs`: HOST array of 1..8 descriptors (passed to the kernel by value): the generator's Dense output matrix
 * bias [N, ld_bias] (b_enc at column 0, b_dec at column Hn), its weights [D, Hn], its Philox seed and where its samples go
 * (samples[row * s_row_stride + i * s_elem_stride]: e.g. straight into track m of a [B, steps, P, M] piano-roll). */
typedef struct {
    const float* bias; int ld_bias;
    const float* w_enc; const float* w_dec;
    uint64_t seed;
    uint8_t* samples; float* nll;            /* nll [N] or NULL */
} mnn_nade_sample_job;
int mnn_nade_sample_multi(mnn_stream_t s, int njobs, const mnn_nade_sample_job* jobs, int N, int D, int Hn, float temperature,
                          uint32_t row0, uint32_t sub, long s_row_stride, int s_elem_stride);

/* ------------------------------------------------------------------------------------------
 * RBM (models/common/rbm.py).  W f32 [D,Hn]; bh f32 [N or 1, Hn] (ld_bh = 0 broadcasts one row);
 * bv likewise.  Deterministic summation order (ascending index) for bit-exact sampling.
 * mnn_rbm_gibbs (rbm.py:192-231): k steps from v0 u8 [N,D]; p_v f32 [N,D], v_out u8 [N,D];
 *   uniforms Philox(stream 2/3, row = row_ids ? row_ids[n] : row0+n, sub = sub0+it, elem = j / d).
 * mnn_rbm_hidden (rbm.py:148-167,337-353): p_h f32 [N,Hn]; h u8 [N,Hn] sampled with `stream`
 *   (either may be NULL).  mnn_rbm_visible (rbm.py:169-190,355-373) likewise.
 * mnn_rbm_free_energy (rbm.py:256-258, per-row, R4): F f32 [N].
 * workspace for gibbs/visible: mnn_rbm_workspace_bytes(D,Hn) (transposed W copy).
 * ------------------------------------------------------------------------------------------ */
size_t mnn_rbm_workspace_bytes(int D, int Hn);
int mnn_rbm_gibbs(mnn_stream_t s, int N, int D, int Hn, int k, const uint8_t* v0, const float* W, const float* bh, int ld_bh,
                  const float* bv, int ld_bv, uint64_t seed, uint32_t row0, const uint32_t* row_ids, uint32_t sub0,
                  float* p_v, uint8_t* v_out, void* workspace);
/* The same chain with the step counter of the optimiser read ON THE DEVICE: effective seed = seed + *seed_step (NULL: seed).  A launch
 * captured in a hipGraph then draws new uniforms at every replay, like the dropout masks (mnn_dropout_mask's step pointer); the
 * reference re-runs its random ops at every sess.run (rbm.py:222-226). */
int mnn_rbm_gibbs_stepped(mnn_stream_t s, int N, int D, int Hn, int k, const uint8_t* v0, const float* W, const float* bh, int ld_bh,
                  const float* bv, int ld_bv, uint64_t seed, uint32_t row0, const uint32_t* row_ids, uint32_t sub0,
                  float* p_v, uint8_t* v_out, void* workspace, const int* seed_step);
int mnn_rbm_hidden(mnn_stream_t s, int N, int D, int Hn, const void* v, int v_dtype, const float* W, const float* bh,
                   int ld_bh, int stream_id, uint64_t seed, uint32_t row0, uint32_t sub, float* p_h, uint8_t* h);
int mnn_rbm_visible(mnn_stream_t s, int N, int D, int Hn, const void* h, int h_dtype, const float* W, const float* bv,
                    int ld_bv, int stream_id, uint64_t seed, uint32_t row0, uint32_t sub, float* p_v, uint8_t* v,
                    void* workspace);
int mnn_rbm_free_energy(mnn_stream_t s, int N, int D, int Hn, const uint8_t* v, const float* W, const float* bh, int ld_bh,
                        const float* bv, int ld_bv, float* F, float* p_h /* optional f32 [N, Hn]: sigmoid(v W + bh), = -dF/dz, for the backward pass */);
/* CD-k bias deltas (rbm.py:318-327): dbv[d] += scale * sum_n (v - p_v)[n,d], dbh[j] += scale * sum_n (h - p_h)[n,j] (f32 atomics: zero
 * the outputs first).  The weight delta is two mnn_gemm_tn products combined by mnn_axpby_f32, which is also the `assign_add` of
 * rbm.py:329-333: out[i] = a*x[i] + b*y[i] (out may alias x or y; y may be NULL when b == 0).  Under data parallelism the flat
 * [dW | dbv | dbh] buffer is what the ranks all-reduce. */
int mnn_rbm_cd_bias_delta(mnn_stream_t s, int N, int D, int Hn, const uint8_t* v, const float* p_v, const uint8_t* h, const float* p_h,
                          float scale, float* dbv, float* dbh);
int mnn_axpby_f32(mnn_stream_t s, long n, float a, const float* x, float b, const float* y, float* out);
/* rbm.py:286-297 visible_bias_init_ops: bv[d] = log(1e-6 + p/(1-p)), p = colsum[d] / count (colsum: mnn_bias_grad over the batch). */
int mnn_rbm_visible_bias_init(mnn_stream_t s, int D, const float* colsum, float count, float* bv);
/* Rows of the LSTM-RBM cost gradient (rnn_rbm.py:113-126 over rbm.py:229, cost = F(v) - F(v_s)): d_out[n, :Hn] = w (ss - sv), d_out[n, Hn:Hn+D] =
 * w (v_s - v), zero padding up to ld, w = row_weight[n] * scale; pos = w ss and neg = -w sv [N,Hn] are the scaled hidden blocks of the two
 * ACCUMULATING weight-gradient products v_s^T pos + v^T neg.  v / v_s u8 [N,D]; sv / ss = sigmoid(bh + v W), sigmoid(bh + v_s W) f32 [N,Hn]. */
int mnn_rbm_cd_rows(mnn_stream_t s, int N, int D, int Hn, int ld, const uint8_t* v, const uint8_t* v_s, const float* sv, const float* ss,
                    const float* row_weight, float scale, float* d_out, float* pos, float* neg);

/* dz = dy * y * (1 - y) over n f32 words (dz may alias dy): backward of the sigmoid Dense layers of the feedback module (dnn.py:60-76). */
int mnn_sigmoid_grad_f32(mnn_stream_t s, long n, const float* dy, const float* y, float* dz);

/* ------------------------------------------------------------------------------------------
 * Reductions / optimiser on the flat parameter buffer.
 * mnn_sumsq: out[0] += sum(x^2) (f32 atomic; zero first).  mnn_weighted_sum: out[0] += sum w*x.
 * mnn_clip_adam_step (utils/training.py:163-175 + train.py:64): scale = clip*min(1/gn,1/clip) with
 *   gn = sqrt(*sumsq) read ON DEVICE; TF Adam with epsilon outside the bias correction;
 *   `step` is the 1-based step count; if step_dev (device int32) is given the count is *step_dev + 1,
 *   read ON DEVICE (hipGraph replay) and mnn_step_increment advances it.  sgd != 0 -> plain SGD (train.py:61-62).
 *   A gradient norm that is not finite (*sumsq is inf / NaN: an overflow of the loss-scaled f16 backward pass) SKIPS the update on the
 *   device -- theta, m, v unchanged -- and adds 1 to *skipped (device int32, may be NULL); clip_norm <= 0 disables the test.
 *   mnn_step_increment takes the same (sumsq, clip_norm) and leaves the counter alone on such a step (sumsq may be NULL: always advance),
 *   so the bias correction counts APPLIED steps.
 * ------------------------------------------------------------------------------------------ */
int mnn_sumsq(mnn_stream_t s, const float* x, long n, float* out);
int mnn_weighted_sum(mnn_stream_t s, const float* x, const float* w, long n, float* out);
int mnn_clip_adam_step(mnn_stream_t s, float* theta, const float* grad, float* m, float* v, long n, const float* sumsq,
                       float clip_norm, float lr, float beta1, float beta2, float eps, int step, const int32_t* step_dev, int sgd,
                       int32_t* skipped);
int mnn_step_increment(mnn_stream_t s, int32_t* step_dev, const float* sumsq, float clip_norm,
                       float* ls_dyn /* optional [m, 1 / m]: the dynamic part of the f16 loss scale -- halved by a skipped step, doubled (up to 1) after
                                        `grow_after` applied steps in a row; the owner of the backward pass multiplies its gradient seed by m and the
                                        finished gradient by 1 / m (both read on the device: a captured step follows it) */,
                       int32_t* ls_good /* [1]: applied steps since the last change */, int grow_after);
int mnn_bias_grad(mnn_stream_t s, const float* dY, int rows, int cols, int ld, float* db, int accumulate);
int mnn_fill_f32(mnn_stream_t s, float* x, long n, float value);
/* One pass over the f32 gradient block dY[rows, cols_c] of the dense layer (rnn_estimator.py:205-215's tf.gradients through the
 * Dense layer): out_c = bf16 copy [rows, ld_c] with the padding columns [cols_t, cols_c) written as zeros (dY's are not read); out_t = bf16 transpose of the first cols_t columns [cols_t, ld_t >= rows];
 * db[c] += column sums for c < cols_t.  Replaces convert2d + transpose + bias_grad (three reads of dY) in bf16 mode. */
int mnn_grad_rows_fanout(mnn_stream_t s, const float* dY, int rows, int cols_c, int cols_t, int ld, void* out_c, int ld_c, void* out_t,
                         int ld_t, float* db, int dtype /* MNN_BF16 or MNN_F16: the flavour of out_c / out_t */);

/* ------------------------------------------------------------------------------------------
 * Musical sample metrics (metrics/musical.py:45-275): integer passes over a sampled piano-roll
 * x u8 [B, bars, steps, P, M] (any non-zero byte = note on).  The host turns the tables into EB/UP/UPC/QN/PR/DP/TD in float64.
 * mnn_musical_bar_stats: one workgroup per (sample, bar), nbars = B*bars; all outputs int32 [nbars, M] except
 *   beat_chroma int32 [nbars, 4, 12, M] (notes per quarter-bar and chroma class; class of pitch p = p / (ceil(P/12)) as
 *   _to_chroma's reshape does, musical.py:36-41).  poly_steps counts steps with MORE than poly_threshold pitches (:130);
 *   pattern_class u8 [steps] (1 = weight 1, 2 = weight `tolerance`, 0 = none; NULL = all 0) feeds pat_on / pat_tol (:148-175).
 *   steps: multiple of 4, <= 192; M <= 8.
 * mnn_musical_note_stats: notes are maximal runs along the bars*steps time axis per (sample, pitch, track) (:93-106);
 *   onsets[m] += notes, qualified[m] += notes longer than `threshold` steps (caller zeroes both int32 [M]). */
int mnn_musical_bar_stats(mnn_stream_t s, const uint8_t* x, int nbars, int steps, int P, int M, int poly_threshold,
                          const uint8_t* pattern_class, int32_t* notes, int32_t* used_pitches, int32_t* used_classes,
                          int32_t* poly_steps, int32_t* pat_on, int32_t* pat_tol, int32_t* beat_chroma);
int mnn_musical_note_stats(mnn_stream_t s, const uint8_t* x, int B, int T, int P, int M, int threshold, int32_t* onsets, int32_t* qualified);

/* Evaluation statistics (metrics/statistical.py:6-47).
 * mnn_eval_counts: counts u64[4] += {true positives, false positives, false negatives, equal cells} over n cells of
 *   targets / predictions (u8, non-zero = 1): the raw sums of tf.metrics.accuracy / precision / recall (:28-32).
 * mnn_log_loss_rows: out[row] = sum_d -(t log(p+1e-7) + (1-t) log(1-p+1e-7)): the encoders' reconstruction cost
 *   (tf.losses.log_loss, pass_encoder.py:81-86, rbm.py:124-129); probs f32 [N, ld_probs]. */
int mnn_eval_counts(mnn_stream_t s, const uint8_t* targets, const uint8_t* predictions, long n, unsigned long long* counts);
int mnn_log_loss_rows(mnn_stream_t s, const uint8_t* targets, const float* probs, int N, int D, int ld_probs, float* out);

/* ------------------------------------------------------------------------------------------
 * Deterministic f32 single steps of the SAMPLING scan (rnn_estimator.py:293-323 `_generate_recurrence`: sample_single -> single_step;
 * rnn_nade.py:253-277; multinn_feedback.py:175-218; the LSTM cell of rnn.py:124, the Dense of rnn_nade.py:54-57 / rnn_rbm.py:252-253).
 * Every output is a fixed sequence of IEEE f32 operations -- z = ((p_0 + p_1) + (p_2 + p_3)) + bias with p_s the ascending-k fmaf chain
 * (from 0) over quarter s of xh = [x | x2 | h_prev] (quarters: the k-pairs of every 1024-column chunk cut into four runs of ceil(pairs / 4));
 * i, f, o = det_sigmoid(z), ci = det_tanh(z) = 2 det_sigmoid(2 z) - 1; c = round(ci i) + round(c_prev f); h = det_tanh(c) o; Dense: the
 * same four quarter chains, then + bias -- which oracle/det_ref.c restates, so a whole generate() scan is checked bit for bit.
 * Weights are the f32 MASTER weights in their TF layout (LSTM kernel [(n_x + n_x2 + units), 4 units], column blocks i | ci | f | o;
 * Dense kernel [K, N] with row pitch ld_w).  `jobs` is a HOST array of 1..MNN_DET_MAX_JOBS descriptors (passed to the kernel by
 * value): the jobs of one call run as ONE launch -- the M per-track generators of a feedback-scan step.  x: u8 cells (0 / 1) or f32.
 * ------------------------------------------------------------------------------------------ */
#define MNN_DET_MAX_JOBS 8
typedef struct {
    const void* x; int x_dtype; int n_x; int ld_x; int es_x; /* first input block: element (row, k) at x[row * ld_x + k * es_x] (es_x >= 1: e.g. one
                                                              * track of a [B, P, M] step); MNN_U8 or MNN_F32; n_x may be 0 */
    const float* x2; int n_x2; int ld_x2;                   /* optional second block (the feedback vector), concatenated behind x */
    const float* h_prev; const float* c_prev;               /* [B, units]; both NULL = zero state */
    const float* W; const float* bias;                      /* [(n_x + n_x2 + units), 4 units], [4 units] */
    float* c_out; float* h_out;                             /* [B, units] (may not alias c_prev / h_prev) */
    int units;                                              /* multiple of 32 */
    const float* Wp;                                        /* optional: W repacked by mnn_det_lstm_pack (the same numbers in the step kernel's load
                                                             * order: a quarter chain is then ~30 16-byte loads, all in flight at once, instead of 119
                                                             * dependent-latency 4-byte ones); NULL: W is read in place */
} mnn_det_lstm_job;
typedef struct {
    const float* x; int ld_x; int K;                        /* [B, ld_x] */
    const float* W; int ld_w; int N;                        /* [K, ld_w >= N] */
    const float* bias;                                      /* [N] or NULL */
    float* out; int ld_out;                                 /* [B, ld_out >= N] */
    const float* Wp;                                        /* optional: W repacked by mnn_det_dense_pack (see mnn_det_lstm_job.Wp) */
} mnn_det_dense_job;
int mnn_lstm_step_det(mnn_stream_t s, int B, int njobs, const mnn_det_lstm_job* jobs);
/* W f32 [K = inputs + units, 4 units] (TF layout) -> Wp (mnn_det_lstm_pack_bytes(K, units) bytes, 16-byte aligned) for mnn_det_lstm_job.Wp; to be
 * redone whenever W changes (mnn_generate_scan does it once per scan, into its workspace) */
size_t mnn_det_lstm_pack_bytes(int K, int units);
int mnn_det_lstm_pack(mnn_stream_t s, const float* W, int K, int units, float* Wp);
size_t mnn_det_dense_pack_bytes(int K, int N);
int mnn_det_dense_pack(mnn_stream_t s, const float* W, int K, int N, int ld_w, float* Wp);
int mnn_dense_det(mnn_stream_t s, int B, int njobs, const mnn_det_dense_job* jobs);

/* mnn_generate_scan (rnn_estimator.py:271-323 `generate`): the WHOLE sampling scan of an LSTM-NADE / LSTM-MultiNADE generator in
 * one call -- the intro pass (n_intro deterministic LSTM steps from a zero state over intro u8 [B, n_intro, n_in], Dense on the last output),
 * then num_steps x { mnn_nade_sample with Philox sub-counter = generated step -> LSTM step on the sample -> Dense } -- enqueued on stream s from
 * one host loop (nothing synchronised: capturable into a hipGraph).  layers: HOST array of the stack's master weights (TF layout); Dense W
 * [units_last, n_out], n_out = tracks * (Hn + D) (b_enc blocks, then b_dec blocks); n_in = tracks * D (a sample is the next input).
 * samples u8 [B, num_steps, tracks * D] (feature m D + i for one NADE, i tracks + m for several: rnn_multinade.py:313-314).  Same bits as the
 * single-step entry points; workspace: mnn_generate_scan_workspace_bytes(), 256-byte aligned, caller-owned. */
#define MNN_SCAN_MAX_LAYERS 8
typedef struct { const float* W; const float* bias; int units; } mnn_scan_lstm_layer;
size_t mnn_generate_scan_workspace_bytes(int B, int n_in, int n_layers, const mnn_scan_lstm_layer* layers, int n_out);
int mnn_generate_scan(mnn_stream_t s, int B, int n_intro, int num_steps, const uint8_t* intro, int n_in, int n_layers,
                      const mnn_scan_lstm_layer* layers, const float* dense_W, const float* dense_bias, int n_out, int tracks, int D, int Hn,
                      const float* w_enc, const float* w_dec, float temperature, uint64_t seed, uint32_t row0, uint8_t* samples,
                      void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------
 * Data-parallel exchange (the gradients of utils/training.py:151-177 as ONE flat f32 buffer): RCCL over xGMI, one
 * communicator per process = per GPU, created, passed and destroyed by the CALLER (the library keeps no communicator).
 *   mnn_comm_unique_id: rank 0 fills 128 bytes (MNN_COMM_ID_BYTES) and ships them to the other ranks through any host channel;
 *   mnn_comm_init: collective over all `world` ranks, binds to the calling thread's current HIP device;
 *   mnn_allreduce_flat: buf[0..n) := sum over ranks, in place, asynchronous on stream s (the step's ONE collective);
 *   mnn_comm_destroy.
 * RCCL is bound at run time (dlopen librccl.so.1 -- the copy already in the process if there is one); without it these calls return an
 * error and every other entry point works.
 * ------------------------------------------------------------------------------------------ */
#define MNN_COMM_ID_BYTES 128
typedef struct mnn_comm_s* mnn_comm_t;
int mnn_comm_unique_id(void* id_out);
int mnn_comm_init(mnn_comm_t* comm, int rank, int world, const void* id);
int mnn_allreduce_flat(mnn_comm_t comm, mnn_stream_t s, float* buf, long n);
int mnn_comm_destroy(mnn_comm_t comm);

/* ------------------------------------------------------------------------------------------
 * Measurement support (not on the product path).  mnn_probe_sigmoid: `blocks` x 256 threads x 8 independent chains x `iters`
 * sigmoids (the NADE kernels' v_exp_f32 + v_rcp_f32 form), out f32 [blocks*256]: bench.py times it with HIP events to get the
 * device's measured transcendental peak, the bound of the NADE phases in the step roofline.
 * ------------------------------------------------------------------------------------------ */
int mnn_probe_sigmoid(mnn_stream_t s, int blocks, int iters, float* out);

#ifdef __cplusplus
}
#endif
#endif /* MULTINN_HIP_H */

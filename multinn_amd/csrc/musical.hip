// Musical sample metrics over sampled piano-rolls: the integer passes.
// Reference: /root/reference/multinn/metrics/musical.py:45-275 (NumPy).  Byte/integer work, HBM-bound: every byte of
// the piano-roll u8 [B, bars, steps, P, M] is read once per pass; the outputs are small int32 tables that the host mirror
// (multinn_amd/metrics.py) turns into the reference's rates in float64.
#include "common.h"

// One workgroup per (sample, bar).  Outputs, each int32 [B*bars, M] unless noted:
//   notes        number of note cells (count_nonzero)                       musical.py:57 (any), :176 (num_notes)
//   used_pitches pitches with at least one note in the bar                  musical.py:73
//   used_classes chroma classes with at least one note                      musical.py:73 on _to_chroma (:16-41): class = p / (Ppad/12)
//   poly_steps   time steps with MORE than `poly_threshold` pitches on      musical.py:130
//   pat_on / pat_tol   notes on steps whose drum-pattern weight is 1 / `tolerance` (pattern_class[step] = 1 / 2, else 0)   musical.py:148-175
//   beat_chroma  int32 [B*bars, 4, 12, M]: notes per (quarter of the bar, chroma class)   musical.py:205-207
#define MUS_MAX_STEPS 192
#define MUS_MAX_TRACKS 8
__global__ void __launch_bounds__(256)
musical_bar_kernel(const uint8_t* __restrict__ x, int steps, int P, int M, int poly_threshold, const uint8_t* __restrict__ pattern_class,
                   int32_t* __restrict__ notes, int32_t* __restrict__ used_pitches, int32_t* __restrict__ used_classes,
                   int32_t* __restrict__ poly_steps, int32_t* __restrict__ pat_on, int32_t* __restrict__ pat_tol, int32_t* __restrict__ beat_chroma) {
    __shared__ int s_cnt[MUS_MAX_STEPS][MUS_MAX_TRACKS];        // notes per (step, track)
    __shared__ int s_bc[4][12][MUS_MAX_TRACKS];                 // notes per (beat, class, track)
    __shared__ int s_up[MUS_MAX_TRACKS];                        // used pitches per track
    const int bar = blockIdx.x, tid = threadIdx.x;
    const uint8_t* xb = x + (size_t)bar * steps * P * M;
    for (int i = tid; i < MUS_MAX_STEPS * MUS_MAX_TRACKS; i += 256) (&s_cnt[0][0])[i] = 0;
    for (int i = tid; i < 4 * 12 * MUS_MAX_TRACKS; i += 256) (&s_bc[0][0][0])[i] = 0;
    if (tid < MUS_MAX_TRACKS) s_up[tid] = 0;
    __syncthreads();
    const int per = (P + 11) / 12;                               // pitches per chroma class after zero padding to a multiple of 12
    const int spb = steps / 4;                                   // steps per beat
    for (int col = tid; col < P * M; col += 256) {               // one thread per (pitch, track) column, consecutive bytes across threads
        const int p = col / M, m = col - p * M, cls = p / per;
        int used = 0;
        for (int st = 0; st < steps; ++st) {
            if (xb[(size_t)st * P * M + col] != 0) {
                used = 1;
                atomicAdd(&s_cnt[st][m], 1);
                atomicAdd(&s_bc[min(st / spb, 3)][cls][m], 1);
            }
        }
        if (used) atomicAdd(&s_up[m], 1);
    }
    __syncthreads();
    if (tid < M) {
        const int m = tid;
        int n = 0, poly = 0, on = 0, tol = 0;
        for (int st = 0; st < steps; ++st) {
            const int c = s_cnt[st][m];
            n += c;
            poly += c > poly_threshold ? 1 : 0;
            const int pc = pattern_class != nullptr ? pattern_class[st] : 0;
            on += pc == 1 ? c : 0;
            tol += pc == 2 ? c : 0;
        }
        int classes = 0;
        for (int cc = 0; cc < 12; ++cc) classes += (s_bc[0][cc][m] + s_bc[1][cc][m] + s_bc[2][cc][m] + s_bc[3][cc][m]) > 0 ? 1 : 0;
        const size_t o = (size_t)bar * M + m;
        notes[o] = n; used_pitches[o] = s_up[m]; used_classes[o] = classes; poly_steps[o] = poly; pat_on[o] = on; pat_tol[o] = tol;
    }
    for (int i = tid; i < 4 * 12 * M; i += 256) {
        const int m = i % M, cc = (i / M) % 12, bt = i / (12 * M);
        beat_chroma[(size_t)bar * 4 * 12 * M + i] = s_bc[bt][cc][m];
    }
}

extern "C" int mnn_musical_bar_stats(mnn_stream_t s, const uint8_t* x, int nbars, int steps, int P, int M, int poly_threshold,
                                     const uint8_t* pattern_class, int32_t* notes, int32_t* used_pitches, int32_t* used_classes,
                                     int32_t* poly_steps, int32_t* pat_on, int32_t* pat_tol, int32_t* beat_chroma) {
    MNN_REQUIRE(x && notes && used_pitches && used_classes && poly_steps && pat_on && pat_tol && beat_chroma, "mnn_musical_bar_stats: null pointer");
    MNN_REQUIRE(nbars > 0 && P > 0 && M > 0 && M <= MUS_MAX_TRACKS && steps >= 4 && steps % 4 == 0 && steps <= MUS_MAX_STEPS,
                "mnn_musical_bar_stats: need nbars,P>0, 0<M<=%d, steps a multiple of 4 in [4,%d] (steps=%d M=%d)", MUS_MAX_TRACKS, MUS_MAX_STEPS, steps, M);
    hipLaunchKernelGGL(musical_bar_kernel, dim3(nbars), dim3(256), 0, (hipStream_t)s, x, steps, P, M, poly_threshold, pattern_class, notes,
                       used_pitches, used_classes, poly_steps, pat_on, pat_tol, beat_chroma);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// Notes = maximal runs of non-zero cells along time for a fixed (sample, pitch, track); bars are concatenated (musical.py:93-96).
// One thread per (sample, pitch, track) walks the T steps (consecutive threads read consecutive bytes of a step).
// onsets[m] += number of notes, qualified[m] += notes LONGER than `threshold` steps (musical.py:103-106).
__global__ void __launch_bounds__(256)
musical_notes_kernel(const uint8_t* __restrict__ x, int Bn, int T, int P, int M, int threshold, int32_t* __restrict__ onsets, int32_t* __restrict__ qualified) {
    __shared__ int s_on[MUS_MAX_TRACKS], s_q[MUS_MAX_TRACKS];
    if (threadIdx.x < MUS_MAX_TRACKS) { s_on[threadIdx.x] = 0; s_q[threadIdx.x] = 0; }
    __syncthreads();
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id < (long)Bn * P * M) {
        const int col = (int)(id % ((long)P * M));
        const long b = id / ((long)P * M);
        const int m = col % M;
        const uint8_t* xb = x + (size_t)b * T * P * M + col;
        int run = 0, n_on = 0, n_q = 0;
        for (int t = 0; t < T; ++t) {
            if (xb[(size_t)t * P * M] != 0) {
                n_on += run == 0 ? 1 : 0;
                ++run;
            } else {
                n_q += run > threshold ? 1 : 0;
                run = 0;
            }
        }
        n_q += run > threshold ? 1 : 0;
        if (n_on) atomicAdd(&s_on[m], n_on);
        if (n_q) atomicAdd(&s_q[m], n_q);
    }
    __syncthreads();
    if (threadIdx.x < M) {
        if (s_on[threadIdx.x]) atomicAdd(onsets + threadIdx.x, s_on[threadIdx.x]);
        if (s_q[threadIdx.x]) atomicAdd(qualified + threadIdx.x, s_q[threadIdx.x]);
    }
}

extern "C" int mnn_musical_note_stats(mnn_stream_t s, const uint8_t* x, int B, int T, int P, int M, int threshold, int32_t* onsets,
                                      int32_t* qualified) {
    MNN_REQUIRE(x && onsets && qualified, "mnn_musical_note_stats: null pointer");
    MNN_REQUIRE(B > 0 && T > 0 && P > 0 && M > 0 && M <= MUS_MAX_TRACKS, "mnn_musical_note_stats: need B,T,P>0 and 0<M<=%d", MUS_MAX_TRACKS);
    const long n = (long)B * P * M;
    hipLaunchKernelGGL(musical_notes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, x, B, T, P, M, threshold, onsets, qualified);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// ------------------------------------------------------------------------------------------------
// Evaluation statistics: the passes behind metrics/statistical.py:6-47 and the encoders' reconstruction cost
// (pass_encoder.py:81-86, rbm.py:124-129: tf.losses.log_loss summed over the visibles, epsilon 1e-7).
// ------------------------------------------------------------------------------------------------
// counts[0..3] += true positives, false positives, false negatives, equal cells over n cells (targets / predictions: non-zero = 1)
__global__ void __launch_bounds__(256) eval_counts_kernel(const uint8_t* __restrict__ targets, const uint8_t* __restrict__ predictions, long n,
                                                          unsigned long long* __restrict__ counts) {
    unsigned tp = 0, fp = 0, fn = 0, eq = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const bool t = targets[i] != 0, p = predictions[i] != 0;
        tp += (t && p) ? 1u : 0u; fp += (!t && p) ? 1u : 0u; fn += (t && !p) ? 1u : 0u; eq += (t == p) ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { tp += __shfl_xor(tp, o); fp += __shfl_xor(fp, o); fn += __shfl_xor(fn, o); eq += __shfl_xor(eq, o); }
    if ((threadIdx.x & 63) == 0) {
        if (tp) atomicAdd(counts + 0, (unsigned long long)tp);
        if (fp) atomicAdd(counts + 1, (unsigned long long)fp);
        if (fn) atomicAdd(counts + 2, (unsigned long long)fn);
        if (eq) atomicAdd(counts + 3, (unsigned long long)eq);
    }
}

extern "C" int mnn_eval_counts(mnn_stream_t s, const uint8_t* targets, const uint8_t* predictions, long n, unsigned long long* counts) {
    MNN_REQUIRE(targets && predictions && counts && n > 0, "mnn_eval_counts: bad arguments");
    const int blocks = (int)min((long)2048, (n + 255) / 256);
    hipLaunchKernelGGL(eval_counts_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, targets, predictions, n, counts);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// out[row] = sum_d -( t log(p + 1e-7) + (1 - t) log(1 - p + 1e-7) ), one wave per row (ascending-index partial sums per lane, xor tree)
__global__ void __launch_bounds__(256) log_loss_rows_kernel(const uint8_t* __restrict__ targets, const float* __restrict__ probs, int N, int D,
                                                            int ld_p, float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    float acc = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float p = probs[(size_t)row * ld_p + d];
        acc += targets[(size_t)row * D + d] != 0 ? -logf(p + 1e-7f) : -logf(1.0f - p + 1e-7f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[row] = acc;
}

extern "C" int mnn_log_loss_rows(mnn_stream_t s, const uint8_t* targets, const float* probs, int N, int D, int ld_probs, float* out) {
    MNN_REQUIRE(targets && probs && out && N > 0 && D > 0 && ld_probs >= D, "mnn_log_loss_rows: bad arguments");
    hipLaunchKernelGGL(log_loss_rows_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)s, targets, probs, N, D, ld_probs, out);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

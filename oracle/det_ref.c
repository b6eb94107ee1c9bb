/* Deterministic float32 CPU restatement of the sampling kernels' arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  "Bit-exact Bernoulli sampling under a fixed
 * RNG" (BASELINE.json north_star) needs the CPU and the GPU to evaluate u < sigmoid(logit) on
 * bit-identical probabilities.  This file restates, independently of multinn_amd/csrc, the
 * specification in DESIGN.md ("Deterministic sampling"):
 *   det_exp(x): clamp to [-87,87]; n = floor(x*log2e + 0.5); r = x - n*ln2 (hi/lo split, fma);
 *               degree-6 Taylor polynomial by Horner with fma; result * 2^n.
 *   det_sigmoid(x) = 1 / (1 + det_exp(-x))            (IEEE division)
 *   NADE logit (models/common/nade.py:326-328): 64 partial sums, partial l = fma-chain over
 *               hidden j = l, l+64, l+128, l+192 ; xor-butterfly 32,16,8,4,2,1 ; + b_dec.
 *   RBM logits (models/common/rbm.py:351,370): ascending-index fma chain from 0, then + bias.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static float det_exp(float x) {
    if (x < -87.0f) x = -87.0f;
    if (x > 87.0f) x = 87.0f;
    const float n = floorf(fmaf(x, 1.4426950408889634f, 0.5f));
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.3888889225e-3f;
    p = fmaf(p, r, 8.3333337680e-3f);
    p = fmaf(p, r, 4.1666667908e-2f);
    p = fmaf(p, r, 1.6666667163e-1f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    uint32_t bits = ((uint32_t)((int)n + 127)) << 23;
    float s;
    memcpy(&s, &bits, 4);
    return p * s;
}

static float det_sigmoid(float x) { return 1.0f / (1.0f + det_exp(-x)); }

void det_sigmoid_array(const float* x, float* out, long n) {
    for (long i = 0; i < n; ++i) out[i] = det_sigmoid(x[i]);
}

/* nade.py:231-308 for one track m.  u[N*D] uniforms; temperature <= 0 -> threshold 0.5.
 * bias: [N, ld_bias], b_enc at column m*Hn, b_dec at tracks*Hn + m*D.  w_enc,w_dec: [D,Hn] of track m. */
void nade_sample_det(int N, int D, int Hn, int tracks, int m, const float* bias, int ld_bias, const float* w_enc, const float* w_dec,
                     float temperature, const float* u, uint8_t* samples, float* p_out) {
    for (int n = 0; n < N; ++n) {
        float a[256];
        for (int j = 0; j < 256; ++j) a[j] = j < Hn ? bias[(long)n * ld_bias + m * Hn + j] : 0.0f;
        const float* bd = bias + (long)n * ld_bias + tracks * Hn + m * D;
        for (int i = 0; i < D; ++i) {
            float part[64];
            for (int l = 0; l < 64; ++l) {
                float acc = 0.0f;
                for (int q = 0; q < 4; ++q) {
                    const int j = l + 64 * q;
                    const float wd = j < Hn ? w_dec[(long)i * Hn + j] : 0.0f;
                    acc = fmaf(det_sigmoid(a[j]), wd, acc);
                }
                part[l] = acc;
            }
            for (int o = 32; o > 0; o >>= 1) {
                float nx[64];
                for (int l = 0; l < 64; ++l) nx[l] = part[l] + part[l ^ o];
                memcpy(part, nx, sizeof(part));
            }
            const float logit = bd[i] + part[0];
            const float p = det_sigmoid(logit);
            int on;
            if (temperature > 0.0f) {
                const float ps = temperature == 1.0f ? p : det_sigmoid(logit / temperature);
                on = u[(long)n * D + i] < ps;
            } else {
                on = p >= 0.5f;
            }
            samples[(long)n * D + i] = (uint8_t)on;
            if (p_out) p_out[(long)n * D + i] = p;
            if (on)
                for (int j = 0; j < Hn; ++j) a[j] = a[j] + w_enc[(long)i * Hn + j];
        }
    }
}

/* rbm.py:337-353: p_h[n,j] = sigmoid(sum_d v[n,d] W[d,j] + bh[n,j]) */
void rbm_hidden_det(int N, int D, int Hn, const float* v, const float* W, const float* bh, int ld_bh, float* p_h) {
    for (int n = 0; n < N; ++n)
        for (int j = 0; j < Hn; ++j) {
            float acc = 0.0f;
            for (int d = 0; d < D; ++d) acc = fmaf(v[(long)n * D + d], W[(long)d * Hn + j], acc);
            p_h[(long)n * Hn + j] = det_sigmoid(acc + bh[(long)n * ld_bh + j]);
        }
}

/* rbm.py:355-373: p_v[n,d] = sigmoid(sum_j h[n,j] W[d,j] + bv[n,d]) */
void rbm_visible_det(int N, int D, int Hn, const float* h, const float* W, const float* bv, int ld_bv, float* p_v) {
    for (int n = 0; n < N; ++n)
        for (int d = 0; d < D; ++d) {
            float acc = 0.0f;
            for (int j = 0; j < Hn; ++j) acc = fmaf(h[(long)n * Hn + j], W[(long)d * Hn + j], acc);
            p_v[(long)n * D + d] = det_sigmoid(acc + bv[(long)n * ld_bv + d]);
        }
}

/* rbm.py:192-231: k Gibbs steps; u_h[k,N,Hn], u_v[k,N,D]; outputs last p_v and v_k. */
void rbm_gibbs_det(int N, int D, int Hn, int k, const uint8_t* v0, const float* W, const float* bh, int ld_bh, const float* bv, int ld_bv,
                   const float* u_h, const float* u_v, float* p_v, uint8_t* v_out, float* vbuf, float* hbuf, float* pbuf) {
    for (long e = 0; e < (long)N * D; ++e) { vbuf[e] = (float)v0[e]; p_v[e] = (float)v0[e]; }
    for (int it = 0; it < k; ++it) {
        rbm_hidden_det(N, D, Hn, vbuf, W, bh, ld_bh, pbuf);
        for (long e = 0; e < (long)N * Hn; ++e) hbuf[e] = u_h[(long)it * N * Hn + e] < pbuf[e] ? 1.0f : 0.0f;
        rbm_visible_det(N, D, Hn, hbuf, W, bv, ld_bv, p_v);
        for (long e = 0; e < (long)N * D; ++e) vbuf[e] = u_v[(long)it * N * D + e] < p_v[e] ? 1.0f : 0.0f;
    }
    for (long e = 0; e < (long)N * D; ++e) v_out[e] = (uint8_t)vbuf[e];
}

/* ---- deterministic single steps of the sampling scan (DESIGN.md "Deterministic sampling"; the device side is csrc/det_step.hip) ----
 * det_tanh(x) = 2 det_sigmoid(2 x) - 1 (both multiplications by 2 are exact; one rounding in the subtraction). */
static float det_tanh(float x) { return 2.0f * det_sigmoid(2.0f * x) - 1.0f; }

/* Summation order of the deterministic Dense / LSTM contractions (the specification csrc/det_step.hip implements on the matrix cores): the K
 * input columns are taken in chunks of 1024; the k-PAIRS (2p, 2p + 1) of a chunk (its column count rounded up to even; a missing last column
 * counts as input 0) are cut into FOUR contiguous quarters of q = ceil(pairs / 4) pairs; quarter s has its own ascending fmaf chain p_s from
 * 0, running on over the chunks; the result is ((p_0 + p_1) + (p_2 + p_3)) + bias. */
#define DET_KC 1024
static int det_quarter(int k, int K) {
    const int k0 = (k / DET_KC) * DET_KC;
    int kc = K - k0 < DET_KC ? K - k0 : DET_KC;
    kc = (kc + 1) & ~1;
    const int pairs = kc / 2, q = (pairs + 3) / 4;
    return ((k - k0) / 2) / q;
}

/* rnn.py:124 (CudnnCompatibleLSTMCell = LSTMBlockCell, forget_bias 0): xh = [x | h_prev]; z = xh . W + b with W [(n_in + u), 4u], column
 * blocks i | ci | f | o, summed in the quartered order above.  h_prev / c_prev NULL = zero state. */
void lstm_step_det(int B, int n_in, int u, const float* x, int ld_x, const float* h_prev, const float* c_prev, const float* W,
                   const float* bias, float* c_out, float* h_out) {
    static float z[4][4 * 1024];                     /* u <= 1024 */
    const int K = n_in + u;
    for (int n = 0; n < B; ++n) {
        /* the chains of the 4u columns advance together, k ascending (each partial sum is still ONE chain per column: the loop order only
         * lets the compiler use vector fma instructions across columns) */
        for (int s = 0; s < 4; ++s)
            for (int col = 0; col < 4 * u; ++col) z[s][col] = 0.0f;
        for (int k = 0; k < K; ++k) {
            const float xv = k < n_in ? x[(long)n * ld_x + k] : (h_prev ? h_prev[(long)n * u + (k - n_in)] : 0.0f);
            const float* w = W + (long)k * 4 * u;
            float* zs = z[det_quarter(k, K)];
            for (int col = 0; col < 4 * u; ++col) zs[col] = fmaf(xv, w[col], zs[col]);
        }
        for (int col = 0; col < 4 * u; ++col) {
            const float s01 = z[0][col] + z[1][col], s23 = z[2][col] + z[3][col];
            const float t = s01 + s23;
            z[0][col] = t + bias[col];
        }
        for (int j = 0; j < u; ++j) {
            const float gi = det_sigmoid(z[0][j]), gc = det_tanh(z[0][u + j]), gf = det_sigmoid(z[0][2 * u + j]), go = det_sigmoid(z[0][3 * u + j]);
            const float cp = c_prev ? c_prev[(long)n * u + j] : 0.0f;
            const float t1 = gc * gi, t2 = cp * gf;
            const float c = t1 + t2;
            c_out[(long)n * u + j] = c;
            h_out[(long)n * u + j] = det_tanh(c) * go;
        }
    }
}

/* tf.layers.Dense (rnn_nade.py:54-57, rnn_rbm.py:252-253): out = x . W + b, W [K, N], summed in the quartered order above. */
void dense_det(int B, int K, int N, const float* x, int ld_x, const float* W, const float* bias, float* out, int ld_out) {
    static float z[4][8192];                         /* N <= 8192 */
    for (int n = 0; n < B; ++n) {
        float* o = out + (long)n * ld_out;
        for (int s = 0; s < 4; ++s)
            for (int j = 0; j < N; ++j) z[s][j] = 0.0f;
        for (int k = 0; k < K; ++k) {
            const float xv = x[(long)n * ld_x + k];
            const float* w = W + (long)k * N;
            float* zs = z[det_quarter(k, K)];
            for (int j = 0; j < N; ++j) zs[j] = fmaf(xv, w[j], zs[j]);
        }
        for (int j = 0; j < N; ++j) {
            const float s01 = z[0][j] + z[1][j], s23 = z[2][j] + z[3][j];
            const float t = s01 + s23;
            o[j] = bias ? t + bias[j] : t;
        }
    }
}

// NADE log-prob backward with BOTH row sums on the matrix cores (round 6; precision "fp16": the caller's loss scale keeps d nll / d logit
// in IEEE-half range).  Reference: /root/reference/multinn/models/common/nade.py:199-229 (autograd of log_prob).
//
// The vector scan (nade.hip: nade_bwd_kernel) spends its time ISSUING: per visible and wave 16 packed FMAs, 8 v_readlane, an LDS exchange.  Both
// of its products are row sums with a shared operand once the rows' hidden states are written down per SEGMENT (`a` only moves where v = 1,
// nade.py:219, so a row has 1 + nnz distinct states):
//     S[slot]        = sum over the segment's visibles of  dl[row, i] * w_dec[i]          (what the scan calls c: closed with one h(1-h) per flip)
//     d w_dec[i]    += sum over slots                       dl[row, i] [i in the slot's segment] * h[slot]
//     d w_enc[f]    += sum over flip slots                  [f is the slot's flip] * G[slot]
// with slot = (row, segment).  A workgroup owns 64 rows x a 128-unit hidden slice and walks the visibles top down in chunks of 32; a chunk is
// processed for 16 rows at a time (a "sub-block": 16 base slots + up to 48 flip slots = one 64-deep K range), in three phases all eight waves
// take part in, separated by workgroup barriers:
//   A  every thread = one (row, visible) cell: scatter dl (as f16) into the masked operand images in LDS -- AS [slot][visible], AD [visible][slot],
//      A1 [visible][slot] (one-hot of the flips) -- then S = AS . w_dec on v_mfma_f32_16x16x32_f16 (wave = 16 hidden units) into LDS;
//   B  the STATE machine in the vector scan's layout (lane = hidden unit, the row wave-uniform: a flip is a scalar branch): wave (hidden half,
//      row quad) adds S[base] to the open segment, and per flip closes it (G += c h(1-h)), publishes G and the new h = sigmoid(a - w_enc[f]) as
//      f16 snapshots [slot][hidden], and opens the next segment with S[slot];
//   C  d w_dec += AD . H and d w_enc += A1 . G on the matrix cores, the snapshots read through ds_read_b64_tr_b16 (the B operand straight from
//      the row-major [slot][hidden] image).  Accumulators stay in registers for the 64 rows of the workgroup and leave as one f32 atomic per
//      (visible, unit) -- from the accumulator layout, no cross-wave exchange.
// More flips than 48 in a sub-block's chunk (dense patches) split the chunk's visible range until they fit (a range of one visible has at most
// 16).  f32: a, h, G, c, w_enc, every accumulator; f16: dl, w_dec, the h / G snapshots as MFMA operands (G scaled by 1/16).
#include "common.h"

typedef _Float16 nb2_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 nb2_h4 __attribute__((ext_vector_type(4)));
typedef short nb2_s4 __attribute__((ext_vector_type(4)));
typedef float nb2_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) nb2_s4* nb2_lds_s4;

#define NB2_SP 132            // floats per S row (slot): 128 hidden + pad
#define NB2_HP 144            // halves per snapshot row (slot): 128 hidden + pad (rows 288 bytes apart: the 4 rows of a transposed read on distinct banks)
#define NB2_ASP 40            // halves per AS row (slot): 32 visibles + pad
#define NB2_ADP 72            // halves per AD / A1 row (visible): 64 slots + pad
#define NB2_WDP 40            // halves per staged w_dec row (hidden unit): 32 visibles + pad
#define NB2_GS 0.0625f        // scale of the G snapshots (f16 range)
#define NB2_OFF_S 0
#define NB2_OFF_H (NB2_OFF_S + 64 * NB2_SP * 4)
#define NB2_OFF_G (NB2_OFF_H + 64 * NB2_HP * 2)
#define NB2_OFF_OPS (NB2_OFF_G + 64 * NB2_HP * 2)          // two buffers of 16 KB: AS | AD | A1
#define NB2_OPS_BYTES 16384
#define NB2_OPS_AD (64 * NB2_ASP * 2)
#define NB2_OPS_A1 (NB2_OPS_AD + 32 * NB2_ADP * 2)
#define NB2_OFF_WD (NB2_OFF_OPS + 2 * NB2_OPS_BYTES)
#define NB2_OFF_WE (NB2_OFF_WD + 128 * NB2_WDP * 2)
#define NB2_OFF_VB (NB2_OFF_WE + 32 * 128 * 4)
#define NB2_OFF_PF (NB2_OFF_VB + 8 * 16 * 4)
#define NB2_LDS (NB2_OFF_PF + 8 * 32 * 4)
static_assert(NB2_OPS_A1 + 32 * NB2_ADP * 2 <= NB2_OPS_BYTES, "operand images fit their buffer");
static_assert(NB2_LDS <= 160 * 1024, "LDS budget");

#ifdef NB2_TRACE     // development only (profiles/tools/nade_bwd2_trace.py): shader-clock cycles per phase of wave 0 of workgroup 0
__device__ long long nb2_trace[16];
extern "C" int mnn_nade_bwd2_trace(void* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(nb2_trace), sizeof(nb2_trace)) == hipSuccess ? 0 : -2; }
#define NB2_T(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { const long long now_ = __builtin_readcyclecounter(); nb2_trace[k] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define NB2_T(k) do { } while (0)
#endif

__global__ void __launch_bounds__(512)
nade_bwd2_kernel(int tracks, int N, int D, int HnT, int nslice, const uint8_t* __restrict__ v, long v_track_stride, int ld_bias,
                 const float* __restrict__ w_enc, const float* __restrict__ w_dec, const float* __restrict__ a_final, float* __restrict__ d_bias,
                 float* __restrict__ d_w_enc, float* __restrict__ d_w_dec, const int* __restrict__ n_rows_dev, const int* __restrict__ gate, int run_if) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (gate != nullptr && *gate != run_if) return;                       // density-gated pair with the vector scan: uniform exit
    float* sS = reinterpret_cast<float*>(smem + NB2_OFF_S);
    _Float16* sH = reinterpret_cast<_Float16*>(smem + NB2_OFF_H);
    _Float16* sG = reinterpret_cast<_Float16*>(smem + NB2_OFF_G);
    _Float16* sWD = reinterpret_cast<_Float16*>(smem + NB2_OFF_WD);
    float* sWE = reinterpret_cast<float*>(smem + NB2_OFF_WE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned* sVB = reinterpret_cast<unsigned*>(smem + NB2_OFF_VB) + 16 * w;      // this wave's copy of the sub-block's v bits (one word per row)
    int* sPF = reinterpret_cast<int*>(smem + NB2_OFF_PF) + 32 * w;               // ... and of the exclusive prefix of their flip counts
    const int m = blockIdx.y / nslice, hb = (blockIdx.y - m * nslice) * 128;
    const int rb = blockIdx.x * 64;
    if (n_rows_dev != nullptr && rb >= *n_rows_dev) {                      // compacted ragged batch: padding rows only -- d b_enc = 0
        for (int e = tid; e < 64 * 128; e += 512) {
            const int n = e >> 7, j = e & 127;
            if (rb + n < N) d_bias[(size_t)(rb + n) * ld_bias + m * HnT + hb + j] = 0.f;
        }
        return;
    }
    const int Nv = n_rows_dev != nullptr ? min(*n_rows_dev, N) : N;
    const uint8_t* __restrict__ vm = v + (size_t)m * v_track_stride;
    const float* __restrict__ we = w_enc + (size_t)m * D * HnT + hb;
    const float* __restrict__ wd = w_dec + (size_t)m * D * HnT + hb;
    const int dl_off = tracks * HnT + m * D;
    // roles
    const int hh = w & 1, rq = w >> 1, hidl = 64 * hh + lane;               // state role: rows 4 rq .. +3 of a sub-block, hidden unit hidl of the slice
    const int mn = lane & 15, kg = lane >> 4;                               // matrix-core role: hidden tile 16 w .. +15; lane = (row / column mn, k group kg)
    const int cr = tid >> 5, ci = tid & 31;                                 // cell role: (row cr of the sub-block, visible ci of the chunk)

    float a[4][4], h[4][4], G[4][4], c[4][4];                               // [sub-block][row of the quad]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = rb + 16 * j + 4 * rq + rr;
            a[j][rr] = a_final[((size_t)m * N + min(row, N - 1)) * HnT + hb + hidl];
        }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            if (!(rb + 16 * j + 4 * rq + rr < Nv)) a[j][rr] = 0.f;
            h[j][rr] = fast_sigmoid(a[j][rr]);
            G[j][rr] = 0.f;
            c[j][rr] = 0.f;
        }
    // both snapshot images start as zeros: a slot the current unit does not write keeps an older (finite) snapshot and meets a zero operand --
    // uninitialised LDS could hold a NaN pattern there (0 x NaN); the base rows of the G image are never written at all (a base slot has no flip)
    for (int e = tid; e < 2 * 64 * NB2_HP * 2 / 16; e += 512) reinterpret_cast<uint4*>(smem + NB2_OFF_H)[e] = make_uint4(0u, 0u, 0u, 0u);
    // both operand buffers start zeroed
    for (int e = tid; e < 2 * NB2_OPS_BYTES / 16; e += 512) reinterpret_cast<uint4*>(smem + NB2_OFF_OPS)[e] = make_uint4(0u, 0u, 0u, 0u);

    const int nch = (D + 31) / 32;
    // weights of a chunk: thread e -> (visible e >> 7, hidden e & 127), eight per thread; fetched one chunk ahead
    float rwd[8], rwe[8];
    auto wfetch = [&](int c0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = tid + 512 * k, vi = e >> 7, hj = e & 127;
            const int gi = min(max(c0 + vi, 0), D - 1);
            rwd[k] = wd[(size_t)gi * HnT + hj];
            rwe[k] = we[(size_t)gi * HnT + hj];
        }
    };
    auto wstore = [&](int c0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = tid + 512 * k, vi = e >> 7, hj = e & 127;
            const bool ok = c0 + vi < D;
            sWD[hj * NB2_WDP + vi] = (_Float16)(ok ? rwd[k] : 0.f);
            sWE[vi * 128 + hj] = ok ? rwe[k] : 0.f;
        }
    };
    wfetch((nch - 1) * 32);
    // this thread's cell of dl and its share of the v bytes of a (chunk, sub-block), requested ONE unit ahead: unconditional loads from clamped
    // addresses, looked at a unit later (a global round trip in front of every unit's first barrier was 5 of the first version's 6.7 ms)
    float dl_n;
    unsigned char vb_n[8];
    auto cfetch = [&](int c0, int r0) {
        dl_n = d_bias[(size_t)min(r0 + cr, N - 1) * ld_bias + dl_off + min(c0 + ci, D - 1)];
#pragma unroll
        for (int k = 0; k < 8; ++k) vb_n[k] = vm[(size_t)min(r0 + 2 * k + (lane >> 5), N - 1) * D + min(c0 + (lane & 31), D - 1)];
    };
    cfetch((nch - 1) * 32, rb);
    int sbc = 0;                                                            // running index of (sub-block, range) units: selects the operand buffer
#ifdef NB2_TRACE
    long long tprev_ = __builtin_readcyclecounter();
#endif
    for (int cc = nch - 1; cc >= 0; --cc) {
        const int c0 = 32 * cc;
        __syncthreads();                                                    // every wave is done with the previous chunk's weights and snapshots
        wstore(c0);
        if (cc > 0) wfetch(c0 - 32);
        NB2_T(8);
        nb2_f4 accD[2], accE[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) { accD[mt] = (nb2_f4){0.f, 0.f, 0.f, 0.f}; accE[mt] = (nb2_f4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r0 = rb + 16 * j;
            // ---- this thread's cell of dl, and the sub-block's v bits (every wave takes all 16 ballots: no hand-off between waves) ----
            const float dlv = (r0 + cr < Nv && c0 + ci < D) ? dl_n : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = r0 + 2 * k + (lane >> 5), gi = c0 + (lane & 31);
                const unsigned long long bal = __ballot(row < Nv && gi < D && vb_n[k] != 0);
                if (lane == 0) { sVB[2 * k] = (unsigned)bal; sVB[2 * k + 1] = (unsigned)(bal >> 32); }
            }
            if (j < 3) cfetch(c0, r0 + 16);                                 // the next unit's cell and v bytes
            else if (cc > 0) cfetch(c0 - 32, rb);
            NB2_T(0);
            int hi = 32;
            while (hi > 0) {                                                // ranges of visibles [lo, hi), top down; ONE range unless a dense patch overflows
                int lo = 0, F = 0;
                for (;;) {
                    const unsigned msk = (hi == 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
                    // exclusive prefix of the rows' flip counts inside the range (lanes 0..15: one row each, then a 16-lane scan)
                    int cnt = lane < 16 ? __popc(sVB[lane & 15] & msk) : 0;
                    int inc = cnt;
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        const int up = __shfl_up(inc, o);
                        if ((lane & 15) >= o) inc += up;
                    }
                    F = __builtin_amdgcn_readlane(inc, 15);
                    if (F <= 48 || hi - lo == 1) {
                        if (lane < 16) sPF[lane] = inc - cnt;
                        break;
                    }
                    lo = hi - ((hi - lo) >> 1);                             // too many flips: take the upper half of the range
                }
                const unsigned msk = (hi == 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
                NB2_T(1);
                char* ops = smem + NB2_OFF_OPS + (sbc & 1) * NB2_OPS_BYTES;
                _Float16* sAS = reinterpret_cast<_Float16*>(ops);
                _Float16* sAD = reinterpret_cast<_Float16*>(ops + NB2_OPS_AD);
                _Float16* sA1 = reinterpret_cast<_Float16*>(ops + NB2_OPS_A1);
                // ---- A: scatter the masked operands (thread = cell) ----
                if (ci >= lo && ci < hi) {
                    const unsigned bits = sVB[cr] & msk;
                    const int seg = __popc(bits >> ci);                     // flips of the row at visibles >= this one, inside the range
                    const int slot = seg == 0 ? cr : 16 + sPF[cr] + seg - 1;
                    const _Float16 x = (_Float16)dlv;
                    sAS[slot * NB2_ASP + ci] = x;
                    sAD[ci * NB2_ADP + slot] = x;
                    if ((bits >> ci) & 1u) sA1[ci * NB2_ADP + slot] = (_Float16)1.f;
                }
                NB2_T(2);
                __syncthreads();                                            // B1: operands complete (and the previous unit's phase C has read its snapshots)
                NB2_T(3);
                // ---- A: segment sums S = AS . w_dec (wave = 16 hidden units), into LDS for the state machine ----
                {
                    const nb2_h8 Bw = *reinterpret_cast<const nb2_h8*>(&sWD[(16 * w + mn) * NB2_WDP + 8 * kg]);
                    const int ntile = (16 + F + 15) >> 4;
                    for (int mt = 0; mt < ntile; ++mt) {
                        const nb2_h8 A = *reinterpret_cast<const nb2_h8*>(&sAS[(16 * mt + mn) * NB2_ASP + 8 * kg]);
                        const nb2_f4 s4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, Bw, (nb2_f4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) sS[(16 * mt + 4 * kg + i) * NB2_SP + 16 * w + mn] = s4[i];
                    }
                    // the OTHER operand buffer (read last by the previous unit's phase C, in front of B1) is cleared for the next unit
                    uint4* z = reinterpret_cast<uint4*>(smem + NB2_OFF_OPS + ((sbc + 1) & 1) * NB2_OPS_BYTES);
                    z[tid] = make_uint4(0u, 0u, 0u, 0u);
                    z[tid + 512] = make_uint4(0u, 0u, 0u, 0u);
                }
                NB2_T(4);
                __syncthreads();                                            // B2: S complete
                NB2_T(5);
                // ---- B: the state machine (lane = hidden unit; the row is wave-uniform) ----
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = 4 * rq + rr;
                    unsigned bits = __builtin_amdgcn_readfirstlane(sVB[r] & msk);
                    int slot = 16 + __builtin_amdgcn_readfirstlane(sPF[r]);
                    c[j][rr] += sS[r * NB2_SP + hidl];
                    sH[r * NB2_HP + hidl] = (_Float16)h[j][rr];
                    while (bits != 0u) {
                        const int f = 31 - __builtin_clz(bits);            // flips top down
                        bits ^= 1u << f;
                        G[j][rr] = fmaf(c[j][rr], fmaf(-h[j][rr], h[j][rr], h[j][rr]), G[j][rr]);       // close the segment that used a_{f+1}
                        sG[slot * NB2_HP + hidl] = (_Float16)(G[j][rr] * NB2_GS);                       // d w_enc[f] += G_{f+1}
                        a[j][rr] -= sWE[f * 128 + hidl];                                                 // a_f = a_{f+1} - w_enc[f]
                        h[j][rr] = fast_sigmoid(a[j][rr]);
                        sH[slot * NB2_HP + hidl] = (_Float16)h[j][rr];
                        c[j][rr] = sS[slot * NB2_SP + hidl];                                            // the segment that starts at f
                        ++slot;
                    }
                }
                NB2_T(6);
                __syncthreads();                                            // B3: snapshots complete
                NB2_T(7);
                // ---- C: d w_dec += AD . H, d w_enc += A1 . G (wave = 16 hidden units; B operands by transposed reads of [slot][hidden]) ----
                {
                    const int nks = (16 + F > 32) ? 2 : 1;
                    for (int ks = 0; ks < nks; ++ks) {
                        const int rowb = 32 * ks + 8 * kg + (mn >> 2), colb = 16 * w + 4 * (mn & 3);
                        const nb2_s4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((nb2_lds_s4)(sH + (rowb) * NB2_HP + colb));
                        const nb2_s4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((nb2_lds_s4)(sH + (rowb + 4) * NB2_HP + colb));
                        const nb2_s4 g0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((nb2_lds_s4)(sG + (rowb) * NB2_HP + colb));
                        const nb2_s4 g1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((nb2_lds_s4)(sG + (rowb + 4) * NB2_HP + colb));
                        const nb2_h8 BH = __builtin_bit_cast(nb2_h8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                        const nb2_h8 BG = __builtin_bit_cast(nb2_h8, __builtin_shufflevector(g0, g1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            const nb2_h8 AD = *reinterpret_cast<const nb2_h8*>(&sAD[(16 * mt + mn) * NB2_ADP + 32 * ks + 8 * kg]);
                            const nb2_h8 A1 = *reinterpret_cast<const nb2_h8*>(&sA1[(16 * mt + mn) * NB2_ADP + 32 * ks + 8 * kg]);
                            accD[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AD, BH, accD[mt], 0, 0, 0);
                            accE[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, BG, accE[mt], 0, 0, 0);
                        }
                    }
                }
                NB2_T(9);
                ++sbc;
                hi = lo;
            }
        }
        // ---- the chunk's sums over the workgroup's 64 rows leave: one f32 atomic per (visible, hidden unit), straight from the accumulators ----
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int vi = c0 + 16 * mt + 4 * kg + i;
                if (vi < D) {
                    const size_t o = ((size_t)m * D + vi) * HnT + hb + 16 * w + mn;
                    atomicAdd(d_w_dec + o, accD[mt][i]);
                    atomicAdd(d_w_enc + o, accE[mt][i] * (1.0f / NB2_GS));
                }
            }
        NB2_T(10);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = rb + 16 * j + 4 * rq + rr;
            const float g = fmaf(c[j][rr], fmaf(-h[j][rr], h[j][rr], h[j][rr]), G[j][rr]);
            if (row < N) __builtin_nontemporal_store(g, &d_bias[(size_t)row * ld_bias + m * HnT + hb + hidl]);
        }
}

// internal (nade.hip dispatches here): not part of the C ABI
int mnn_nade_bwd2_ok(int Hn) { return (Hn == 128 || Hn == 256) ? 1 : 0; }
int mnn_nade_bwd2_launch(hipStream_t st, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride, int ld_bias, const float* w_enc,
                         const float* w_dec, const float* a_final, float* d_bias, float* d_w_enc, float* d_w_dec, const int* n_rows_dev, const int* gate,
                         int run_if) {
    static bool raised[64];
    int dev = 0;
    MNN_HIP(hipGetDevice(&dev));
    MNN_REQUIRE(dev >= 0 && dev < 64, "mnn_nade_logprob_bwd: device index %d", dev);
    if (!raised[dev]) {
        MNN_HIP(hipFuncSetAttribute((const void*)nade_bwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NB2_LDS));
        raised[dev] = true;
    }
    const int nslice = Hn / 128;
    hipLaunchKernelGGL(nade_bwd2_kernel, dim3(cdiv(N, 64), tracks * nslice), dim3(512), NB2_LDS, st, tracks, N, D, Hn, nslice, v, v_track_stride, ld_bias,
                       w_enc, w_dec, a_final, d_bias, d_w_enc, d_w_dec, n_rows_dev, gate, run_if);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// NADE log-prob on the matrix cores (bf16 operands, f32 accumulation) for Hn = 256: forward scan.
// Reference: /root/reference/multinn/models/common/nade.py:155-229.
//
// The conditional of visible d is  p_d = sigmoid(b_dec[d] + w_dec[d] . h_d),  h_d = sigmoid(a_d),
// a_{d+1} = a_d + v_d w_enc[d]  (nade.py:206-219).  `a` only moves at visibles with v = 1, so a row has
// 1 + nnz(v) distinct hidden STATES and the D dot products are a block-sparse GEMM
//     logits[row, d] = state(row, d) . w_dec[d]
// A workgroup owns 32 rows and walks the visibles in tiles of 32 columns.  Per tile:
//   * base states (one per row, as of the tile's first column) x w_dec tile  -> 32 x 32 logits on MFMA
//   * every flip (v = 1) inside the tile creates a new state: a += w_enc[d] in f32 (thread = hidden unit),
//     h = sigmoid(a) -> bf16 into a flip-state tile; flip states x w_dec tile -> 32-slot x 32 logits on MFMA
//   * each (row, d) picks the logit of the latest state created before d (popcount of the row's v bits), adds
//     b_dec, and finishes p, the NLL term and d nll / d b_dec.
// The dense work (D x Hn MACs per row) runs on v_mfma_f32_16x16x32_bf16; the VALU keeps only the sparse encoder
// adds, one sigmoid per state element and the per-(row, d) pointwise.  w_enc stays f32 (a is exact up to f32
// summation order and is handed to the backward pass as a_final); w_dec is read as a bf16 copy.
#include "common.h"
typedef float nm_f4 __attribute__((ext_vector_type(4)));
// results (conditionals, d nll / d logit, a_final) are written once and read by later kernels: stored non-temporal, they leave the decoder / encoder
// weights every workgroup re-reads in the L2 (2.19 -> 2.17 ms at [262144, 440, 256]; same in the backward's d b_enc store)
#define NM_ST4(dst, a) __builtin_nontemporal_store((nm_f4){a[0], a[1], a[2], a[3]}, reinterpret_cast<nm_f4*>(dst))
#define NM_ST1(dst, x) __builtin_nontemporal_store((float)(x), dst)
#include <algorithm>
#include <type_traits>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gas_ptr_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#ifdef NM_TRACE      // development only: per-phase clocks of one workgroup (profiles/tools/nade_trace.hip)
__device__ long long nm_trace[16];
#define NM_T(k) do { if (tid == 0 && blockIdx.x == NM_TRACE) { const long long now_ = wall_clock64(); nm_trace[k] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define NM_T(k) do { } while (0)
#endif
#define NADE_EPS 1e-6f
#define LN2F 0.6931471805599453f
#define NM_H 256          // hidden width this kernel is specialised for
#define NM_PITCH 264      // bf16 elements per state-tile row (256 + 8 pad: rows 16 B aligned, 4-bank skew)
#define NM_LP 36          // floats per logit-tile row

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt(0), i.e. every
// prefetch load and every result store in flight would be waited for at each of the tile's barriers.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float nm_ln(float x) { return __builtin_amdgcn_logf(x) * LN2F; }

// SPLIT = the SPLIT-OPERAND form (precision "fp16": BASELINE.json's 1e-4 on every conditional, which 8- or 11-bit operands of the decoder dot
// products miss).  Every hidden state h and every decoder weight w is carried as TWO IEEE halves, x = hi + lo with hi = f16(x) and
// lo = f16(x - hi) (22 significant bits), and a logit is the f32 sum of three 16-bit matrix-core products  hi.hi + hi.lo + lo.hi  (the
// dropped lo.lo term is 2^-22 of a product): ~1e-6 of the vector scan's f32 arithmetic at the 16-bit MFMA rate.  (Round 3 first shipped
// this form on v_mfma_f32_16x16x4_f32 -- exact f32 products at 1/16 of the 16-bit rate: its two logit phases cost 0.9 of the launch's
// 2.8 ms at [1024,256,88,5]; measured by ablation, profiles/round3_d_nade_fwd_ablation.md.)  A state element is ONE 32-bit LDS word
// (hi in the low half, lo in the high half), unpacked into the two A fragments with v_perm_b32; the pre-activations live in registers
// (indirect register addressing), which keeps two workgroups on a CU.  w_dec comes from mnn_nade_f32_pack as f16 [d][hi | lo][Hn].
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
struct NmPairB { f16x8_t hi, lo; };                       // decoder fragment of one k-step: the weights' high and low halves
template <bool SPLIT> struct NmState;
template <> struct NmState<false> { typedef bf16_t T; static constexpr int PITCH = NM_PITCH; };
template <> struct NmState<true> { typedef unsigned T; static constexpr int PITCH = 260; };  // 260 words: rows 4 banks apart
// state word of x in (0, 1): hi = x truncated to f16's 11 significant bits (a mask: exact), lo = x - hi (exact in f32, at most 13 significant
// bits) rounded toward zero to f16 -- one v_and, one v_sub, one v_cvt_pkrtz; |x - (hi + lo)| < 2^-21 x (f16 subnormals: 2^-24 absolute)
__device__ __forceinline__ unsigned nm_pack_hl(float x) {
    const float hi = __uint_as_float(__float_as_uint(x) & 0xFFFFE000u);
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(hi, x - hi));
}

template <bool SPLIT>
struct NadeFwdSmemT {
    float sA[SPLIT ? 1 : 32][NM_H];     // pre-activations a[row][hidden] (f32; thread = hidden unit owns a column); split form: in registers
    typename NmState<SPLIT>::T sH[32][NmState<SPLIT>::PITCH];            // current state of every row (as of the tile being processed)
    typename NmState<SPLIT>::T sF[32][NmState<SPLIT>::PITCH];            // states created by the flips of the tile (one chunk of 32 slots)
    float sLb[32][NM_LP];               // logits of the base states  [row][column]
    float sLf[32][NM_LP];               // logits of the flip states  [slot][column]
    unsigned sMask[2][32];              // v bits of the tile, per row (double buffered: tile c, tile c+1)
    unsigned sSb[2][33];                // exclusive prefix of popcounts (slot of a row's first flip); [32] = flips in the tile
    unsigned short sFl[2][1024];        // flips in pass order: row << 5 | column
    unsigned short sPs[2][34];          // first slot of pass j (the j-th flips of the rows)
    unsigned sAct[2][32];               // rows taking part in pass j (bit n: row n has more than j flips in the tile)
};

// 16 rows x K=256 of a state tile as MFMA A fragments (lane: row l & 15, k = 32 s + 8 (l >> 4) + j)
__device__ __forceinline__ void nm_load_a(const bf16_t (*tile)[NM_PITCH], int mi, int lane, bf16x8_t (&a)[8]) {
    const bf16_t* p = &tile[16 * mi + (lane & 15)][8 * (lane >> 4)];
#pragma unroll
    for (int s = 0; s < 8; ++s) a[s] = *reinterpret_cast<const bf16x8_t*>(p + 32 * s);
}
template <bool SPLIT> __device__ __forceinline__ typename NmState<SPLIT>::T nm_state(float h) {
    if constexpr (SPLIT) return nm_pack_hl(h);
    else return f32_to_bf16(h);
}
// 16 x 16 logits of one state tile against the tile's decoder rows
__device__ __forceinline__ f32x4_t nm_dot(const bf16x8_t (&af)[8], const bf16x8_t (&bfr)[8]) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s], bfr[s], acc, 0, 0, 0);
    return acc;
}
// split-operand form: 16 rows x K = 256 of a packed state tile against the wave's 16 decoder columns.  Lane (row l & 15, k group l >> 4) reads
// its 8 packed words of a k-step with two 16-byte LDS loads (rows 4 banks apart: conflict-free) and splits them into the hi and lo fragments.
__device__ __forceinline__ f32x4_t nm_dot_hl(const unsigned (*tile)[260], int mi, int lane, const NmPairB (&bw)[8]) {
    const unsigned* p = &tile[16 * mi + (lane & 15)][8 * (lane >> 4)];
    f32x4_t accm = {0.f, 0.f, 0.f, 0.f}, accs = {0.f, 0.f, 0.f, 0.f};     // main term | the two small terms: two independent chains
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const uint4 q0 = *reinterpret_cast<const uint4*>(p + 32 * s), q1 = *reinterpret_cast<const uint4*>(p + 32 * s + 4);
        uint4 uh, ul;
        uh.x = __builtin_amdgcn_perm(q0.y, q0.x, 0x05040100u); ul.x = __builtin_amdgcn_perm(q0.y, q0.x, 0x07060302u);
        uh.y = __builtin_amdgcn_perm(q0.w, q0.z, 0x05040100u); ul.y = __builtin_amdgcn_perm(q0.w, q0.z, 0x07060302u);
        uh.z = __builtin_amdgcn_perm(q1.y, q1.x, 0x05040100u); ul.z = __builtin_amdgcn_perm(q1.y, q1.x, 0x07060302u);
        uh.w = __builtin_amdgcn_perm(q1.w, q1.z, 0x05040100u); ul.w = __builtin_amdgcn_perm(q1.w, q1.z, 0x07060302u);
        const f16x8_t ah = __builtin_bit_cast(f16x8_t, uh), al = __builtin_bit_cast(f16x8_t, ul);
        accs = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bw[s].lo, accs, 0, 0, 0);
        accm = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bw[s].hi, accm, 0, 0, 0);
        accs = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bw[s].hi, accs, 0, 0, 0);
    }
    return accm + accs;
}
// One 32 x 32 logit tile of a state tile: wave (mi, ni) = w & 1, w >> 1 writes its 16 x 16 quadrant.
template <bool SPLIT, typename Tile, typename B>
__device__ __forceinline__ void nm_logit_tile(const Tile& tile, int w, int lane, const B& bfr, float (*out)[NM_LP]) {
    const int ni = w >> 1, mi = w & 1;
    f32x4_t acc;
    if constexpr (SPLIT) {
        acc = nm_dot_hl(tile, mi, lane, bfr);
    } else {
        bf16x8_t af[8];
        nm_load_a(tile, mi, lane, af);
        acc = nm_dot(af, bfr);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[16 * mi + 4 * (lane >> 4) + i][16 * ni + (lane & 15)] = acc[i];
}

// thread = hidden unit: the 32 rows' pre-activations a[32] live in registers and are indexed by the (wave-uniform) row of
// each flip -- the compiler lowers that to s_set_gpr_idx, no scratch.  Per flip the VALU does one add, one sigmoid,
// one bf16 convert and two LDS writes; everything it needs (the flip list, the flips' w_enc values) was fetched a tile ahead.
template <bool SPLIT>
__device__ __forceinline__ void
nade_fwd_mfma_body(int tracks, int N, int D, const uint8_t* __restrict__ v, long v_track_stride, const float* __restrict__ bias, int ld_bias,
                     const float* __restrict__ w_enc, const bf16_t* __restrict__ w_dec_bf, const float* __restrict__ row_weight,
                     float* __restrict__ nll, float* __restrict__ cond_p, float* __restrict__ d_bias, float* __restrict__ a_final,
                     const int* __restrict__ gate, int run_if, const int* __restrict__ n_rows_dev) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    typedef NadeFwdSmemT<SPLIT> NadeFwdSmem;
    typedef typename NmState<SPLIT>::T state_t;
    typedef typename std::conditional<SPLIT, NmPairB, bf16x8_t>::type bfrag_t;
    constexpr int KS = 8;                                    // k-steps of 32 hidden units
    NadeFwdSmem& S = *reinterpret_cast<NadeFwdSmem*>(smem_raw);
    if (gate != nullptr && *gate != run_if) return;          // density-gated pair of launches: uniform exit
    constexpr int Hn = NM_H;
    const int m = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ni = w >> 1;                                   // this wave's column half of a 32 x 32 logit tile (w & 1: its row half / K half)
    const int rb = blockIdx.x * 32;
    const uint8_t* __restrict__ vm = v + (size_t)m * v_track_stride;
    const float* __restrict__ we = w_enc + (size_t)m * D * Hn;
    const bf16_t* __restrict__ wd = w_dec_bf + (size_t)m * D * Hn * (SPLIT ? 2 : 1);        // (split form: f16 [d][hi | lo][Hn] behind the 16-bit pointer type)
    const int bd_off = tracks * Hn + m * D;
    const int ntile = (D + 31) / 32;
    // compacted ragged batch (mnn_ragged_index; see nade.hip): a workgroup whose 32 rows are all padding writes the zeros later kernels read and leaves
    if (n_rows_dev != nullptr && rb >= *n_rows_dev) {
        if (d_bias != nullptr)
            for (int e = tid; e < 32 * D; e += 256) {
                const int n = e / D, i = e - n * D;
                if (rb + n < N) d_bias[(size_t)(rb + n) * ld_bias + bd_off + i] = 0.f;
            }
        if (nll != nullptr && tid < 32 && rb + tid < N) nll[(size_t)m * N + rb + tid] = 0.f;
        return;
    }

    // split form: the pre-activations of this thread's hidden unit for the 32 rows live in REGISTERS, indexed by the wave-uniform row of each
    // flip (indirect register addressing); that frees 32 KiB of LDS (two workgroups per CU with the doubled state tiles) and takes the
    // LDS round trip out of every flip's read-modify-write
    float areg[SPLIT ? 32 : 1];
    {
        // all 32 encoder-bias loads in flight before the first is used (unconditional, from clamped rows): with `row < N ? load : 0` every
        // load sat behind its own branch and s_waitcnt vmcnt(0) -- 32 memory round trips in a row at the start of every workgroup
        float av[32];
#pragma unroll
        for (int n = 0; n < 32; ++n) av[n] = bias[(size_t)min(rb + n, N - 1) * ld_bias + m * Hn + tid];
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            const float x = rb + n < N ? av[n] : 0.f;
            if constexpr (SPLIT) areg[n] = x;
            else S.sA[n][tid] = x;
            S.sH[n][tid] = nm_state<SPLIT>(fast_sigmoid(x));
        }
    }
    // v bytes of rows 8w .. 8w+7 of a tile: lane -> (row 8w + 2i + (lane >> 5), column lane & 31)
    auto load_v = [&](int c, unsigned char (&vb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = min(rb + 8 * w + 2 * i + (lane >> 5), N - 1), dd = min(32 * c + (lane & 31), D - 1);
            vb[i] = vm[(size_t)row * D + dd];
        }
    };
    auto ballots = [&](int c, const unsigned char (&vb)[4], unsigned (&msk)[32]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rb + 8 * w + 2 * i + (lane >> 5), dd = 32 * c + (lane & 31);
            const unsigned long long bal = __ballot(row < N && dd < D && vb[i] != 0);
            if (lane == 0) { msk[8 * w + 2 * i] = (unsigned)bal; msk[8 * w + 2 * i + 1] = (unsigned)(bal >> 32); }
        }
    };
    // wave 0: flip list of the tile whose masks are in sMask[buf], in PASS order (all first flips of the rows, then all second
    // flips, ...): consecutive entries touch different rows, so their read-modify-writes of sA are independent.
    // sSb[buf][n] keeps the rank-0 slot of row n; the slot of its j-th flip is looked up through sPs (start of pass j).
    auto build_list = [&](int buf) {
        if (w == 0) {
            const unsigned mk = lane < 32 ? S.sMask[buf][lane] : 0u;
            const int cnt = __popc(mk);
            unsigned rest = mk;
            int base = 0;
            for (int j = 0; j < 32; ++j) {                   // pass j: rows with more than j flips, in row order
                const unsigned act = (unsigned)__ballot(cnt > j);
                if (lane == 0) { S.sPs[buf][j] = (unsigned short)base; S.sAct[buf][j] = act; }
                if (act == 0u) break;
                if (cnt > j) {
                    const int pos = base + __popc(act & ((1u << lane) - 1u));
                    const int dpos = __builtin_ctz(rest);
                    rest &= rest - 1;
                    S.sFl[buf][pos] = (unsigned short)((lane << 5) | dpos);
                }
                base += __popc(act);
            }
            if (lane == 0) S.sSb[buf][32] = (unsigned)base;
        }
    };
    auto load_b = [&](int c, bfrag_t (&b)[KS]) {            // lane: column l & 15 of the wave's column half, k = 32 s + 8 (l >> 4) + j
        const int d = min(32 * c + 16 * ni + (lane & 15), D - 1);
        if constexpr (SPLIT) {                                // [d][hi | lo][Hn] f16
            const f16_t* p = reinterpret_cast<const f16_t*>(wd) + (size_t)d * 2 * Hn + 8 * (lane >> 4);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                b[s].hi = *reinterpret_cast<const f16x8_t*>(p + 32 * s);
                b[s].lo = *reinterpret_cast<const f16x8_t*>(p + Hn + 32 * s);
            }
        } else {
            const bf16_t* p = wd + (size_t)d * Hn + 8 * (lane >> 4);
#pragma unroll
            for (int s = 0; s < KS; ++s) b[s] = *reinterpret_cast<const bf16x8_t*>(p + 32 * s);
        }
    };
    // w_enc[column of flip k0+u][this hidden unit] for the 32 flips of a chunk (entries past the tile's count re-read flip 0's row)
    const __amdgpu_buffer_rsrc_t we_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(we), 0, D * Hn * 4, 0x00020000);
    auto load_we = [&](int c, int buf, int k0, float (&wv)[32]) {
        const int F = (int)S.sSb[buf][32];
        // ONE list read per lane (lane u holds entry k0 + u), the entries then come out of the lanes with v_readlane: a uniform LDS read per
        // entry followed by v_readfirstlane is compiled into one LDS round trip per entry, 32 in a row
        const int entv = (int)S.sFl[buf][(k0 + (lane & 31) < F) ? k0 + (lane & 31) : 0];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            // buffer load: the row offset rides in the instruction's SCALAR offset operand, the lane offset (4 tid) is one loop-invariant
            // register -- no vector ALU work per load (a global load took a 64-bit vector add per row: 32 per tile and thread)
            const int e = __builtin_amdgcn_readlane(entv, u);
            wv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(we_rs, tid * 4, min(32 * c + (e & 31), D - 1) * (Hn * 4), 0));
        }
    };
    unsigned char vb[4];
    bfrag_t bfr[KS];
    float wev[32];
    load_v(0, vb);
    load_b(0, bfr);
    ballots(0, vb, S.sMask[0]);
    if (ntile > 1) load_v(1, vb);
    __syncthreads();
    build_list(0);
    __syncthreads();
    load_we(0, 0, 0, wev);

    // epilogue ownership: thread -> row n = tid >> 3, columns 4 (tid & 7) + k
    const int en = tid >> 3, ed0 = 4 * (tid & 7), erow = rb + en;
    const bool evalid = erow < N;
    const int err = evalid ? erow : N - 1;
    const float rw = (row_weight != nullptr && evalid) ? row_weight[erow] : 0.f;
    float lp = 0.f;
#ifdef NM_TRACE
    long long tprev_ = wall_clock64();
#endif
    for (int c = 0; c < ntile; ++c) {
        const int buf = c & 1, nbuf = buf ^ 1;
        // ---- S0: base logits; the next tile's operands start flying ----
        float bdec[4];
        {
            const float* bp = bias + (size_t)err * ld_bias + bd_off + 32 * c + ed0;
            if (32 * c + ed0 + 3 < D && (((size_t)bp) & 15) == 0) {             // one 16-byte request instead of four
                const float4 b4 = *reinterpret_cast<const float4*>(bp);
                bdec[0] = b4.x; bdec[1] = b4.y; bdec[2] = b4.z; bdec[3] = b4.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) bdec[k] = bias[(size_t)err * ld_bias + bd_off + min(32 * c + ed0 + k, D - 1)];
            }
        }
        // (Requesting these and the v bytes in the middle of the tile before instead -- so that the flip pass, whose generated waits drain the
        // memory queue by its last flip, never waits at HBM latency -- was measured: 2.18 -> 2.25 ms, not kept.)
        __builtin_amdgcn_sched_barrier(0);
        nm_logit_tile<SPLIT>(S.sH, w, lane, bfr, S.sLb);
        if (c + 1 < ntile) {
            ballots(c + 1, vb, S.sMask[nbuf]);               // v bytes requested one tile ago
            if (c + 2 < ntile) load_v(c + 2, vb);
        }
        NM_T(0);
        lds_barrier();                                     // B1: sLb complete, sH consumed, next masks written
        NM_T(1);
        const int F = (int)S.sSb[buf][32];
        const unsigned mk = S.sMask[buf][en];
        float lsel[4];                                       // the logit of each of this thread's 4 (row, column) pairs
        int jj[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            jj[k] = __popc(mk & ((1u << (ed0 + k)) - 1u));   // flips of this row strictly before the column
            lsel[k] = S.sLb[en][ed0 + k];
        }
        int slot_of[4];                                      // slot of the state each pair uses: pass jj-1, rank of the row among that pass's rows
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pj = max(jj[k] - 1, 0);
            slot_of[k] = jj[k] > 0 ? (int)S.sPs[buf][pj] + __popc(S.sAct[buf][pj] & ((1u << en) - 1u)) : -1;
        }
        for (int k0 = 0; k0 < F; k0 += 32) {
            // ---- S1: the chunk's flips in pass order: a[row] += w_enc[d]; new state -> sF[slot] and the row's current state.
            //      Eight at a time: within a pass the rows are distinct (and ascending), so the eight read-modify-writes of
            //      sA are issued together and the in-order VALU sees eight independent chains.
            if (k0 > 0) load_we(c, buf, k0, wev);            // rare: more than 32 flips in one tile
            const int cnt = min(32, F - k0);
            const int flv = (int)S.sFl[buf][min(k0 + (lane & 31), F - 1)];       // the chunk's list entries, one per lane (see load_we)
#pragma unroll
            for (int u0 = 0; u0 < 32; u0 += 8) {
                if (u0 < cnt) {
                    int n[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) n[i] = __builtin_amdgcn_readlane(flv, u0 + i) >> 5;
                    if constexpr (SPLIT) {
                        // pre-activations in registers: the read-modify-writes of the batch run in list order (a row may appear twice in a
                        // batch that straddles two passes), each a register move with a wave-uniform index and one add; the eight sigmoids
                        // are independent of each other; entries past the count add 0 to the row of the last flip.  Kept free of branches
                        // so that the indexed writes happen in place (a conditional write made the compiler copy all 32 registers per flip)
                        float x[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            x[i] = areg[n[i]] + ((u0 + i < cnt) ? wev[u0 + i] : 0.f);
                            areg[n[i]] = x[i];
                        }
                        state_t hb[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) hb[i] = nm_state<SPLIT>(fast_sigmoid(x[i]));
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            if (u0 + i < cnt) {
                                S.sF[u0 + i][tid] = hb[i];
                                S.sH[n[i]][tid] = hb[i];
                            }
                        }
                    } else {
                        bool indep = u0 + 8 <= cnt;          // a full batch inside one pass: rows strictly ascending, hence distinct
#pragma unroll
                        for (int i = 0; i < 7; ++i) indep = indep && n[i] < n[i + 1];
                        if (indep) {
                            float x[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) x[i] = S.sA[n[i]][tid];
#pragma unroll
                            for (int i = 0; i < 8; ++i) x[i] += wev[u0 + i];
                            state_t hb[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) hb[i] = nm_state<SPLIT>(fast_sigmoid(x[i]));
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                S.sA[n[i]][tid] = x[i];
                                S.sF[u0 + i][tid] = hb[i];
                                S.sH[n[i]][tid] = hb[i];
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                if (u0 + i < cnt) {
                                    const float av = S.sA[n[i]][tid] + wev[u0 + i];
                                    S.sA[n[i]][tid] = av;
                                    const state_t hb1 = nm_state<SPLIT>(fast_sigmoid(av));
                                    S.sF[u0 + i][tid] = hb1;
                                    S.sH[n[i]][tid] = hb1;
                                }
                            }
                        }
                    }
                }
            }
            NM_T(2);
            if (k0 == 0 && c + 1 < ntile) build_list(nbuf);
            NM_T(3);
            lds_barrier();                                 // B2: sF complete (and the next tile's list)
            NM_T(4);
            // ---- S2: flip logits; each pair whose state sits in this chunk picks its logit ----
            nm_logit_tile<SPLIT>(S.sF, w, lane, bfr, S.sLf);
            if (k0 == 0 && F <= 32 && c + 1 < ntile) load_we(c + 1, nbuf, 0, wev);     // behind the MFMAs: the next tile's encoder rows
            NM_T(5);
            lds_barrier();                                 // B3: sLf ready
            NM_T(6);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int slot = slot_of[k] - k0;
                if (slot_of[k] >= 0 && slot >= 0 && slot < 32) lsel[k] = S.sLf[slot][ed0 + k];
            }
        }
        if (F == 0 && c + 1 < ntile) {                       // no flip in this tile: the bookkeeping of the loop body still has to happen
            build_list(nbuf);
            lds_barrier();
            load_we(c + 1, nbuf, 0, wev);
        } else if (F > 32 && c + 1 < ntile) {
            load_we(c + 1, nbuf, 0, wev);                    // the extra chunks overwrote the prefetched values
        }
        // b_dec of this tile (requested at its top) is waited for HERE, in front of the w_dec request: the wait the compiler puts in front of
        // the pointwise must hold on every path that reaches it (also the last tile's, which requests nothing), so it is `vmcnt(3)` -- placed
        // behind the 16 w_dec loads it made the pointwise wait for all of them (2.47 -> 2.18 ms with the batched bias loads above)
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("" :: "v"(bdec[k]));
        if (c + 1 < ntile) load_b(c + 1, bfr);               // the tile's MFMAs are issued: fetch the next tile's w_dec fragments under the pointwise
        NM_T(7);
        // ---- S3: pointwise of this thread's 4 pairs, once per tile; the four results of a matrix leave as ONE 16-byte store where the
        //      four columns exist and the address is aligned (one store instruction instead of four per matrix and tile)
        float prv[4], dbv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int dpos = ed0 + k, gd = 32 * c + dpos;
            prv[k] = 0.f; dbv[k] = 0.f;
            if (gd >= D) continue;
            const float l = bdec[k] + lsel[k];
            const bool on = (mk >> dpos) & 1u;
            const float pr = fast_sigmoid(l);
            const float qr = fast_sigmoid(-l);               // 1-p without cancellation
            lp += on ? nm_ln(NADE_EPS + pr) : nm_ln(NADE_EPS + qr);
            prv[k] = pr;
            const float dnll_dp = on ? -fast_rcp(NADE_EPS + pr) : fast_rcp(NADE_EPS + qr);
            dbv[k] = rw * dnll_dp * pr * qr;
        }
        if (evalid) {
            const int gd0 = 32 * c + ed0;
            const bool full = gd0 + 3 < D;
            if (cond_p != nullptr) {
                float* dst = cond_p + ((size_t)m * N + erow) * D + gd0;
                if (full && (((size_t)dst) & 15) == 0) NM_ST4(dst, prv);
                else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (gd0 + k < D) dst[k] = prv[k];
                }
            }
            if (d_bias != nullptr) {
                float* dst = d_bias + (size_t)erow * ld_bias + bd_off + gd0;
                if (full && (((size_t)dst) & 15) == 0) NM_ST4(dst, dbv);
                else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (gd0 + k < D) dst[k] = dbv[k];
                }
            }
        }
        NM_T(8);
    }
    lp += __shfl_xor(lp, 1);
    lp += __shfl_xor(lp, 2);
    lp += __shfl_xor(lp, 4);
    if ((tid & 7) == 0 && evalid && nll != nullptr) nll[(size_t)m * N + erow] = -lp;
    if (a_final != nullptr) {
#pragma unroll
        for (int n = 0; n < 32; ++n)
            if (rb + n < N) NM_ST1(&a_final[((size_t)m * N + rb + n) * Hn + tid], SPLIT ? areg[SPLIT ? n : 0] : S.sA[SPLIT ? 0 : n][tid]);
    }
}

// two entry points over the one body: the split form is held to 256 registers (two workgroups per CU), the bf16 form keeps its own allocation
template <bool SPLIT> __global__ void nade_fwd_mfma_kernel(int, int, int, const uint8_t*, long, const float*, int, const float*, const bf16_t*, const float*,
                                                           float*, float*, float*, float*, const int*, int, const int*);
template <>
__global__ void __launch_bounds__(256)
nade_fwd_mfma_kernel<false>(int tracks, int N, int D, const uint8_t* __restrict__ v, long v_track_stride, const float* __restrict__ bias, int ld_bias,
                            const float* __restrict__ w_enc, const bf16_t* __restrict__ w_dec_bf, const float* __restrict__ row_weight,
                            float* __restrict__ nll, float* __restrict__ cond_p, float* __restrict__ d_bias, float* __restrict__ a_final,
                            const int* __restrict__ gate, int run_if, const int* __restrict__ n_rows_dev) {
    nade_fwd_mfma_body<false>(tracks, N, D, v, v_track_stride, bias, ld_bias, w_enc, w_dec_bf, row_weight, nll, cond_p, d_bias, a_final, gate, run_if, n_rows_dev);
}
template <>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
nade_fwd_mfma_kernel<true>(int tracks, int N, int D, const uint8_t* __restrict__ v, long v_track_stride, const float* __restrict__ bias, int ld_bias,
                           const float* __restrict__ w_enc, const bf16_t* __restrict__ w_dec_bf, const float* __restrict__ row_weight,
                           float* __restrict__ nll, float* __restrict__ cond_p, float* __restrict__ d_bias, float* __restrict__ a_final,
                           const int* __restrict__ gate, int run_if, const int* __restrict__ n_rows_dev) {
    nade_fwd_mfma_body<true>(tracks, N, D, v, v_track_stride, bias, ld_bias, w_enc, w_dec_bf, row_weight, nll, cond_p, d_bias, a_final, gate, run_if, n_rows_dev);
}

extern "C" int mnn_nade_mfma_ok(int Hn) { return Hn == NM_H ? 1 : 0; }

extern "C" int mnn_nade_logprob_fwd_mfma(mnn_stream_t s, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride,
                                         const float* bias, int ld_bias, const float* w_enc, const void* w_dec_bf16, const float* row_weight,
                                         float* nll, float* cond_p, float* d_bias, float* a_final) {
    return mnn_nade_logprob_fwd_mfma_gated(s, tracks, N, D, Hn, v, v_track_stride, bias, ld_bias, w_enc, w_dec_bf16, row_weight, nll, cond_p, d_bias,
                                           a_final, nullptr, 0, nullptr);
}

template <bool SPLIT>
static int nm_launch(mnn_stream_t s, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride, const float* bias, int ld_bias,
                     const float* w_enc, const void* w_dec16, const float* row_weight, float* nll, float* cond_p, float* d_bias, float* a_final,
                     const int* gate, int run_if, const int* n_rows_dev) {
    MNN_REQUIRE(tracks > 0 && N > 0 && D > 0 && Hn == NM_H, "mnn_nade_logprob_fwd_mfma: need tracks,N,D>0 and Hn == 256 (Hn=%d)", Hn);
    MNN_REQUIRE(v && bias && w_enc && w_dec16, "mnn_nade_logprob_fwd_mfma: null pointer");
    MNN_REQUIRE(ld_bias >= tracks * (Hn + D), "mnn_nade_logprob_fwd_mfma: ld_bias %d < tracks*(Hn+D)", ld_bias);
    MNN_REQUIRE(d_bias == nullptr || row_weight != nullptr, "mnn_nade_logprob_fwd_mfma: d_bias needs row_weight");
    MNN_REQUIRE(((uintptr_t)w_dec16 & 15) == 0, "mnn_nade_logprob_fwd_mfma: the 16-bit decoder weights must be 16-byte aligned");
    static bool attr_set[64];
    int dev = 0;
    MNN_HIP(hipGetDevice(&dev));
    MNN_REQUIRE(dev >= 0 && dev < 64, "mnn_nade_logprob_fwd_mfma: device index %d", dev);
    if (!attr_set[dev]) {
        MNN_HIP(hipFuncSetAttribute((const void*)nade_fwd_mfma_kernel<SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(NadeFwdSmemT<SPLIT>)));
        attr_set[dev] = true;
    }
    dim3 grid(cdiv(N, 32), tracks);
    hipLaunchKernelGGL(nade_fwd_mfma_kernel<SPLIT>, grid, dim3(256), sizeof(NadeFwdSmemT<SPLIT>), (hipStream_t)s, tracks, N, D, v, v_track_stride, bias,
                       ld_bias, w_enc, (const bf16_t*)w_dec16, row_weight, nll, cond_p, d_bias, a_final, gate, run_if, n_rows_dev);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

extern "C" int mnn_nade_logprob_fwd_mfma_gated(mnn_stream_t s, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride,
                                               const float* bias, int ld_bias, const float* w_enc, const void* w_dec_bf16, const float* row_weight,
                                               float* nll, float* cond_p, float* d_bias, float* a_final, const int* gate, int run_if, const int* n_rows_dev) {
    return nm_launch<false>(s, tracks, N, D, Hn, v, v_track_stride, bias, ld_bias, w_enc, w_dec_bf16, row_weight, nll, cond_p, d_bias, a_final, gate, run_if, n_rows_dev);
}

extern "C" int mnn_nade_logprob_fwd_mfma_f32(mnn_stream_t s, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride,
                                             const float* bias, int ld_bias, const float* w_enc, const void* w_dec_packed, const float* row_weight,
                                             float* nll, float* cond_p, float* d_bias, float* a_final, const int* gate, int run_if, const int* n_rows_dev) {
    return nm_launch<true>(s, tracks, N, D, Hn, v, v_track_stride, bias, ld_bias, w_enc, w_dec_packed, row_weight, nll, cond_p, d_bias, a_final, gate, run_if, n_rows_dev);
}

// w_dec f32 [rows, Hn] -> f16 [rows][hi | lo][Hn]: the B operand of the split-operand form (4 bytes per weight, as the f32 original)
__global__ void __launch_bounds__(256) nade_f32_pack_kernel(const float* __restrict__ w, long n, int Hn, float* __restrict__ out) {
    f16_t* o = reinterpret_cast<f16_t*>(out);
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const long row = e / Hn;
        const int k = (int)(e - row * Hn);
        const float x = w[e];
        const f16_t hi = (f16_t)x;
        o[row * 2 * Hn + k] = hi;
        o[row * 2 * Hn + Hn + k] = (f16_t)(x - (float)hi);
    }
}
extern "C" int mnn_nade_f32_pack(mnn_stream_t s, const float* w_dec, long rows, int Hn, float* out) {
    MNN_REQUIRE(w_dec && out && rows > 0 && Hn > 0 && Hn % 8 == 0, "mnn_nade_f32_pack: bad arguments (Hn must be a multiple of 8)");
    MNN_REQUIRE(((uintptr_t)out & 15) == 0, "mnn_nade_f32_pack: the packed buffer must be 16-byte aligned");
    const long n = rows * Hn;
    hipLaunchKernelGGL(nade_f32_pack_kernel, dim3((int)std::min(2048L, (n + 255) / 256)), dim3(256), 0, (hipStream_t)s, w_dec, n, Hn, out);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

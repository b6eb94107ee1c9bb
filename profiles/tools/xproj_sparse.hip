// Layer 1's input projection of a PIANO-ROLL batch without the matrix cores: xproj[n, :] = bias + sum over the notes i that are ON in row n of
// Wx[i, :]  (rnn.py:124's x . W for a binary x; rnn_nade.py:204-218 feeds the shifted piano-roll).  At the density of music (13 of 440 cells)
// the dense product spends 97 % of its MFMA work, its operand stream and its 24 us tiles on zeros: [262144 x 448] . [448 x 2048] is 0.48
// TFLOP and 0.73 ms as a GEMM, and 7 G f32 additions here.  The same exact-sparsity idea as the NADE kernels; dense
// batches take the GEMM (the density gate decides on the device, both launches are issued, one returns at once).
//
// A workgroup (16 waves) owns a 128-column slice of the gate columns: the slice of Wx^T -- [K notes + a zero row][128] in the 16-bit compute
// type, 113 KB -- is loaded into LDS ONCE and serves thousands of rows.  A wave takes 4 rows at a time, one per 16-lane group, a lane 8
// columns:  the row's note mask (K / 32 words, written by the piano-roll pass) is compacted into an index list by the group's lanes in parallel
// (one mask word per lane, popcount prefix over the DPP row of 16, each lane then writes the indices of its own bits); the main loop reads
// one index per group (LDS broadcast), one 16-byte piece of that note's weight row per lane, and adds it in f32 (two
// conversions + one packed f32 add per column pair).  Rows with fewer notes than the wave's longest read the zero row.  One 16-byte store per lane
// writes 4 rows x 256 bytes of the 16-bit xproj.
#include "common.h"
#include <algorithm>

#define XS_COLS 128
#define XS_WAVES 16
#define XS_LIST 128                      // list entries per row kept in LDS; a row with more notes sends its round down the direct path

struct XsArgs {
    int N, K, nw, ld_mask;               // K: rows of Wt (notes incl. padding), nw = ceil(K / 32) mask words per row, ld_mask in words
    const uint32_t* mask;
    const h16_t* Wt; int ld_w;           // Wt [K][ld_w]: Wx^T, column = gate column of xproj
    const float* bias;
    h16_t* out; int ld_out;
    int rows_per_wg;
    const int* gate; int run_if;
};

typedef float xs_f2 __attribute__((ext_vector_type(2)));
template <typename F>
__device__ __forceinline__ void xs_add8(xs_f2 (&acc)[4], const uint4 w) {
    // acc pair p += the two 16-bit values of word p: two conversions + ONE packed f32 add (v_pk_add_f32 runs two lanes' worth per issue slot).
    // (v_fma_mix_f32, which converts inside the FMA, was measured at a quarter of the plain rate here: 707 us for the bench shape.)
    const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        xs_f2 v;
        v.x = F::lo(u[p]);
        v.y = F::hi(u[p]);
        acc[p] += v;
    }
}

template <typename F>
__global__ void __launch_bounds__(XS_WAVES * 64)
xproj_sparse_kernel(XsArgs A) {
    extern __shared__ __attribute__((aligned(16))) char xs_smem[];
    if (A.gate != nullptr && *A.gate != A.run_if) return;                    // uniform
    char* wl = xs_smem;                                                      // [(K + 1) rows][128 cols] 16-bit: 256 bytes per note
    uint16_t* lists = reinterpret_cast<uint16_t*>(xs_smem + (size_t)(A.K + 1) * 256);      // [wave][4 rows][XS_LIST]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c16 = lane & 15;
    const int col0 = blockIdx.x * XS_COLS;
    // the slice of Wx^T -> LDS (16-byte pieces; the zero row behind it)
    for (int idx = threadIdx.x; idx < A.K * 16; idx += XS_WAVES * 64) {
        const int row = idx >> 4, piece = idx & 15;
        *reinterpret_cast<uint4*>(wl + row * 256 + piece * 16) = *reinterpret_cast<const uint4*>(A.Wt + (size_t)row * A.ld_w + col0 + piece * 8);
    }
    if (threadIdx.x < 16) *reinterpret_cast<uint4*>(wl + A.K * 256 + threadIdx.x * 16) = make_uint4(0u, 0u, 0u, 0u);
    float bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = A.bias != nullptr ? A.bias[col0 + c16 * 8 + j] : 0.f;
    __syncthreads();
    uint16_t* lw = lists + (w * 4 + g) * XS_LIST;
    const char* wlc = wl + c16 * 16;
    const int r_begin = blockIdx.y * A.rows_per_wg, r_end = min(A.N, r_begin + A.rows_per_wg);
    for (int r0 = r_begin + w * 4; r0 < r_end; r0 += XS_WAVES * 4) {
        const int row = r0 + g;
        const bool ok = row < r_end;
        // one mask word per lane of the group
        const uint32_t mw0 = (ok && c16 < A.nw) ? A.mask[(size_t)row * A.ld_mask + c16] : 0u;
        int incl = __builtin_popcount(mw0);
        // inclusive prefix over the DPP row (16 lanes): row_shr 1, 2, 4, 8 with zero fill
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);
        const int cnt = __builtin_amdgcn_ds_bpermute((lane | 15) << 2, incl);             // the group's note count
        int pos = incl - __builtin_popcount(mw0);
        for (uint32_t mw = mw0; mw != 0u; mw &= mw - 1u) {                   // this lane's bits -> their places in the row's list
            if (pos < XS_LIST) lw[pos] = (uint16_t)(32 * c16 + __builtin_ctz(mw));
            ++pos;
        }
        const int c0 = __builtin_amdgcn_readlane(cnt, 0), c1 = __builtin_amdgcn_readlane(cnt, 16), c2 = __builtin_amdgcn_readlane(cnt, 32),
                  c3 = __builtin_amdgcn_readlane(cnt, 48);
        const int kmax = max(max(c0, c1), max(c2, c3));
        xs_f2 acc[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) { acc[p].x = bv[2 * p]; acc[p].y = bv[2 * p + 1]; }
        if (kmax <= XS_LIST) {
            // (the list writes above and the reads below are LDS operations of ONE wave: executed in order)
            asm volatile("" ::: "memory");
            // four notes per turn: ONE unconditional 8-byte read of the list (entries past the row's count are stale and selected away, not
            // branched around: a conditional LDS read is a branch + a wait per entry), four weight reads in flight together
            for (int k0 = 0; k0 < kmax; k0 += 4) {
                const uint2 q = *reinterpret_cast<const uint2*>(lw + k0);
                const int i0 = k0 + 0 < cnt ? (int)(q.x & 0xffffu) : A.K, i1 = k0 + 1 < cnt ? (int)(q.x >> 16) : A.K;
                const int i2 = k0 + 2 < cnt ? (int)(q.y & 0xffffu) : A.K, i3 = k0 + 3 < cnt ? (int)(q.y >> 16) : A.K;
                const uint4 w0 = *reinterpret_cast<const uint4*>(wlc + i0 * 256), w1 = *reinterpret_cast<const uint4*>(wlc + i1 * 256);
                const uint4 w2 = *reinterpret_cast<const uint4*>(wlc + i2 * 256), w3 = *reinterpret_cast<const uint4*>(wlc + i3 * 256);
                xs_add8<F>(acc, w0); xs_add8<F>(acc, w1); xs_add8<F>(acc, w2); xs_add8<F>(acc, w3);
            }
        } else {
            // a row with more than XS_LIST notes: walk the mask words directly (word d of the group's row sits in lane 16 g + d)
            for (int d = 0; d < A.nw; ++d) {
                uint32_t word = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane & 48) | d) << 2, (int)mw0);
                while (__builtin_amdgcn_ballot_w64(word != 0u) != 0ull) {
                    const int idx = word != 0u ? 32 * d + __builtin_ctz(word) : A.K;
                    xs_add8<F>(acc, *reinterpret_cast<const uint4*>(wlc + idx * 256));
                    word &= word - 1u;
                }
            }
        }
        if (ok) {
            uint4 o;
            o.x = pack2<F>(acc[0].x, acc[0].y); o.y = pack2<F>(acc[1].x, acc[1].y); o.z = pack2<F>(acc[2].x, acc[2].y); o.w = pack2<F>(acc[3].x, acc[3].y);
            *reinterpret_cast<uint4*>(A.out + (size_t)row * A.ld_out + col0 + c16 * 8) = o;
        }
    }
}

extern "C" int mnn_xproj_sparse_ok(int K, int ncols) { return (K > 0 && K <= 512 && ncols > 0 && ncols % XS_COLS == 0) ? 1 : 0; }

extern "C" int mnn_xproj_sparse(mnn_stream_t s, int dtype, int N, int K, const uint32_t* mask, int ld_mask, const void* Wt, int ld_w, const float* bias,
                                int ncols, void* out, int ld_out, const int* gate, int run_if) {
    MNN_REQUIRE(dtype == MNN_BF16 || dtype == MNN_F16, "mnn_xproj_sparse: 16-bit weights / output (dtype %d)", dtype);
    MNN_REQUIRE(mnn_xproj_sparse_ok(K, ncols), "mnn_xproj_sparse: K in 1..512 and ncols a multiple of %d (K=%d ncols=%d)", XS_COLS, K, ncols);
    MNN_REQUIRE(N > 0 && mask && Wt && out && ld_mask >= (K + 31) / 32 && ld_w >= ncols && ld_out >= ncols, "mnn_xproj_sparse: null pointer / pitch too small");
    MNN_REQUIRE(ld_w % 8 == 0 && ld_out % 8 == 0 && ((uintptr_t)Wt & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)mask & 3) == 0,
                "mnn_xproj_sparse: Wt / out 16-byte aligned with pitches that are multiples of 8 elements");
    const size_t lds = (size_t)(K + 1) * 256 + (size_t)XS_WAVES * 4 * XS_LIST * 2;
    MNN_REQUIRE(lds <= 160 * 1024, "mnn_xproj_sparse: K too large for the LDS-resident weight slice");
    static bool raised_[64][2];
    int dev = 0;
    MNN_HIP(hipGetDevice(&dev));
    MNN_REQUIRE(dev >= 0 && dev < 64, "mnn_xproj_sparse: device index %d", dev);
    const int fi = dtype == MNN_F16 ? 1 : 0;
    if (!raised_[dev][fi]) {
        MNN_HIP(hipFuncSetAttribute(fi ? (const void*)xproj_sparse_kernel<Fp16F> : (const void*)xproj_sparse_kernel<Bf16F>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        raised_[dev][fi] = true;
    }
    const int nsl = ncols / XS_COLS;
    // about two workgroups per CU and slice-load amortised over >= 1024 rows each
    int chunks = std::max(1, std::min(cdiv(512, nsl), cdiv(N, 1024)));
    const int rows_per_wg = cdiv(cdiv(N, chunks), 64) * 64;
    chunks = cdiv(N, rows_per_wg);
    XsArgs a{N, K, (K + 31) / 32, ld_mask, mask, (const h16_t*)Wt, ld_w, bias, (h16_t*)out, ld_out, rows_per_wg, gate, run_if};
    if (fi) hipLaunchKernelGGL(xproj_sparse_kernel<Fp16F>, dim3(nsl, chunks), dim3(XS_WAVES * 64), lds, (hipStream_t)s, a);
    else hipLaunchKernelGGL(xproj_sparse_kernel<Bf16F>, dim3(nsl, chunks), dim3(XS_WAVES * 64), lds, (hipStream_t)s, a);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// MFMA GEMM core for gfx950 + the fused LSTM step kernels built on it.
//
//   C[M,N] (op)= A[M,K] . B[N,K]^T        both operands K-contiguous
//
// T = bf16 -> v_mfma_f32_32x32x16_bf16 ; T = f32 -> v_mfma_f32_32x32x2_f32 (exact f32 fma chain).
// Tiles are staged global -> registers -> LDS (double buffered); an LDS row holds one 64-byte
// K-chunk (32 bf16 / 16 f32) padded to 80 bytes so that the 16 rows a ds_read_b128 lane group
// touches fall on 16 distinct 16-byte slots (conflict-free, MI355X_MICROARCH "LDS").
#include "common.h"
typedef unsigned gm_u4 __attribute__((ext_vector_type(4)));
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// (LDS tile rows: GemmCore::ROW = bytes of K per stage + 16 B pad)
#define CHUNK_B 64   // bytes of K per row per stage

template <typename T> struct Elem;
template <> struct Elem<bf16_t> { static constexpr int PER16 = 8; static constexpr int PER_CHUNK = 32; };
template <> struct Elem<f16_t>  { static constexpr int PER16 = 8; static constexpr int PER_CHUNK = 32; };
template <> struct Elem<float>  { static constexpr int PER16 = 4; static constexpr int PER_CHUNK = 32; };      // f32 stages 128 B of K per row (below)
// 16-bit C outputs: the mnn_dtype code of C (MNN_F32 = plain f32) travels as `c16`; a 16-bit value is converted by its own flavour
__device__ __forceinline__ h16_t cvt_c16(float v, int c16) { return c16 == MNN_F16 ? f32_to_f16(v) : f32_to_bf16(v); }

// KH = 2 (f32 only): the workgroup has a second set of WM x WN waves that takes the other half of every K chunk (the step kernels of the f32
// recurrence: one wave per 32 x 128 tile put 512 waves on 1024 SIMDs -- half of the f32 matrix-core rate at best); reduce_kh() adds the halves.
template <typename T, int BM, int BN, int WM, int WN, int KH = 1>
struct GemmCore {
    static constexpr int NT = WM * WN * KH * 64;
    static constexpr int TM = BM / WM / 32;
    static constexpr int TN = BN / WN / 32;
    // bytes of K per tile row and stage: 64 for the 16-bit types (32 k), 128 for f32 (32 k: with 16 k a wave had 4 .. 32 MFMAs between two
    // workgroup barriers and the step kernels of the f32 recurrence were barrier-bound); ROW = its LDS pitch (+ 16 B pad), P = 16-byte pieces
    static constexpr int RB = sizeof(T) == 4 ? 128 : 64, ROW = RB + 16, P = RB / 16;
    static constexpr int A_CHUNKS = (BM * P + NT - 1) / NT;   // 16-byte pieces per thread per stage
    static constexpr int B_CHUNKS = (BN * P + NT - 1) / NT;
    static constexpr int LDS_BYTES = 2 * (BM + BN) * ROW;
    static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "tile/wave mismatch");
    static_assert(KH == 1 || (KH == 2 && sizeof(T) == 4), "the in-workgroup K split is built for f32 operands");
    static_assert(KH == 1 || WM * WN * 16 * 64 * 4 <= LDS_BYTES, "reduce_kh stages one tile per wave in the stage buffers");

    uint4 ra[A_CHUNKS], rb[B_CHUNKS];

    __device__ __forceinline__ void gload(const T* __restrict__ A, int lda, int M, int m0, const T* __restrict__ B, int ldb,
                                          int N, int n0, int K, int k0, int tid) {
#pragma unroll
        for (int s = 0; s < A_CHUNKS; ++s) {
            const int idx = tid + s * NT, row = min(idx / P, BM - 1), c = idx % P;     // (BM * P < NT: the surplus threads repeat the last row)
            const int gr = m0 + row, gk = k0 + c * Elem<T>::PER16;
            ra[s] = (gr < M && gk < K) ? *reinterpret_cast<const uint4*>(A + (size_t)gr * lda + gk) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < B_CHUNKS; ++s) {
            const int idx = tid + s * NT, row = min(idx / P, BN - 1), c = idx % P;
            const int gr = n0 + row, gk = k0 + c * Elem<T>::PER16;
            rb[s] = (gr < N && gk < K) ? *reinterpret_cast<const uint4*>(B + (size_t)gr * ldb + gk) : make_uint4(0, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void lstore(char* sA, char* sB, int tid) {
#pragma unroll
        for (int s = 0; s < A_CHUNKS; ++s) {
            const int idx = tid + s * NT, row = min(idx / P, BM - 1), c = idx % P;     // (surplus threads store the same bytes again)
            *reinterpret_cast<uint4*>(sA + row * ROW + c * 16) = ra[s];
        }
#pragma unroll
        for (int s = 0; s < B_CHUNKS; ++s) {
            const int idx = tid + s * NT, row = min(idx / P, BN - 1), c = idx % P;
            *reinterpret_cast<uint4*>(sB + row * ROW + c * 16) = rb[s];
        }
    }
    // one 64-byte K-chunk of MFMAs for this wave
    __device__ __forceinline__ void compute(const char* sA, const char* sB, int wm, int wn, int lane, f32x16_t (&acc)[TM][TN], int kh = 0) {
        const int r = lane & 31, h = lane >> 5;
        const char* pa = sA + (wm * (BM / WM) + r) * ROW;
        const char* pb = sB + (wn * (BN / WN) + r) * ROW;
        if constexpr (sizeof(T) == 2) {
            using F = typename FlavorOf<T>::type;
            using V8 = typename F::x8;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                V8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const V8*>(pa + i * 32 * ROW + ks * 32 + h * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const V8*>(pb + j * 32 * ROW + ks * 32 + h * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = F::mfma32(a[i], b[j], acc[i][j]);
            }
        } else {
            // K order inside the 32-k chunk is permuted (lane half h owns k = 16 h .. 16 h + 15) so each lane reads contiguous bytes; A and B use
            // the same permutation, so the product is unchanged.  KH == 2: this wave multiplies half of that, k = 16 h + 8 kh .. + 7.
            constexpr int NV = 4 / KH;                   // 16-byte pieces per lane, tile row and chunk
            f32x4_t a[TM][NV], b[TN][NV];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int v = 0; v < NV; ++v) a[i][v] = *reinterpret_cast<const f32x4_t*>(pa + i * 32 * ROW + h * 64 + kh * (NV * 16) + v * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int v = 0; v < NV; ++v) b[j][v] = *reinterpret_cast<const f32x4_t*>(pb + j * 32 * ROW + h * 64 + kh * (NV * 16) + v * 16);
#pragma unroll
            for (int ks = 0; ks < 4 * NV; ++ks)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][ks >> 2][ks & 3], b[j][ks >> 2][ks & 3], acc[i][j], 0, 0, 0);
        }
    }

    // full K loop [kc0, kc1) in chunk units; acc must be initialised by the caller
    __device__ __forceinline__ void run(const T* __restrict__ A, int lda, int M, int m0, const T* __restrict__ B, int ldb, int N,
                                        int n0, int K, int kc0, int kc1, char* smem, f32x16_t (&acc)[TM][TN]) {
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int kh = wave / (WM * WN), wq = wave % (WM * WN);
        const int wm = wq / WN, wn = wq % WN;
        char* sA[2] = {smem, smem + (BM + BN) * ROW};
        char* sB[2] = {smem + BM * ROW, smem + (BM + BN) * ROW + BM * ROW};
        if (kc0 >= kc1) return;
        gload(A, lda, M, m0, B, ldb, N, n0, K, kc0 * Elem<T>::PER_CHUNK, tid);
        lstore(sA[0], sB[0], tid);
        __syncthreads();
        int cur = 0;
        for (int kc = kc0; kc < kc1; ++kc) {
            const bool more = kc + 1 < kc1;
            if (more) gload(A, lda, M, m0, B, ldb, N, n0, K, (kc + 1) * Elem<T>::PER_CHUNK, tid);
            compute(sA[cur], sB[cur], wm, wn, lane, acc, kh);
            if (more) lstore(sA[cur ^ 1], sB[cur ^ 1], tid);
            __syncthreads();
            cur ^= 1;
        }
    }
    // KH == 2, after run() (its last barrier has freed the stage buffers): the second wave set's partial tiles are added into the first set's,
    // tile by tile through `smem`.  Returns true for the waves that now hold the sums (the epilogue is theirs; wave % (WM WN) is their tile slot).
    __device__ __forceinline__ bool reduce_kh(char* smem, f32x16_t (&acc)[TM][TN]) {
        if constexpr (KH == 1) return true;
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int kh = wave / (WM * WN), wq = wave % (WM * WN);
        float* red = reinterpret_cast<float*>(smem) + wq * 16 * 64;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (i + j > 0) __syncthreads();
                if (kh == 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[r * 64 + lane] = acc[i][j][r];
                }
                __syncthreads();
                if (kh == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += red[r * 64 + lane];
                }
            }
        return kh == 0;
    }
};

// C/D fragment coordinates of a 32x32 MFMA tile: reg r of lane l -> (row, col)
__device__ __forceinline__ int frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ----------------------------------------------------------------------------------------------
// plain GEMM
// ----------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, int KH = 1>
__global__ void __launch_bounds__(WM * WN * KH * 64)
gemm_tn_kernel(const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb, void* __restrict__ Cv, int ldc, int c_bf16,
               const float* __restrict__ bias, int M, int N, int K, int flags, int split_k, int ntm, int ntn) {
    using Core = GemmCore<T, BM, BN, WM, WN, KH>;
    extern __shared__ __attribute__((aligned(16))) char gemm_tn_smem[];      // Core::LDS_BYTES (f32: 72 KB, past the static limit)
    char* smem = gemm_tn_smem;
    // XCD-aware tile order: blocks b and b+8 share an XCD (speed only); give each XCD one m-panel
    // for all n-tiles so the A panel stays in that XCD's L2.
    const int bid = blockIdx.x;
    const int grp = bid / (8 * ntn), within = bid % (8 * ntn);
    const int mt = grp * 8 + (within & 7), nt = within >> 3;
    if (mt >= ntm) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int nchunks = (K + Elem<T>::PER_CHUNK - 1) / Elem<T>::PER_CHUNK;
    const int z = blockIdx.y;
    const int per = (nchunks + split_k - 1) / split_k;
    const int kc0 = z * per, kc1 = min(nchunks, kc0 + per);
    f32x16_t acc[Core::TM][Core::TN];
#pragma unroll
    for (int i = 0; i < Core::TM; ++i)
#pragma unroll
        for (int j = 0; j < Core::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    Core core;
    core.run(A, lda, M, m0, B, ldb, N, n0, K, kc0, kc1, smem, acc);
    if (kc0 >= kc1 && z > 0) return;                     // (uniform over the workgroup)
    if (!core.reduce_kh(smem, acc)) return;              // KH == 2: the second wave set's halves are added in; the first set owns the epilogue

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    float* Cf = reinterpret_cast<float*>(Cv);
    bf16_t* Cb = reinterpret_cast<bf16_t*>(Cv);
#pragma unroll
    for (int i = 0; i < Core::TM; ++i)
#pragma unroll
        for (int j = 0; j < Core::TN; ++j) {
            const int col = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
            if (col >= N) continue;
            const float bv = (bias != nullptr && z == 0) ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / WM) + i * 32 + frag_row(r, lane);
                if (row >= M) continue;
                const float val = acc[i][j][r] + bv;
                const size_t o = (size_t)row * ldc + col;
                if (c_bf16) {
                    Cb[o] = cvt_c16(val, c_bf16);
                } else if (flags & MNN_GEMM_ATOMIC) {
                    atomicAdd(Cf + o, val);
                } else if (flags & MNN_GEMM_ACCUMULATE) {
                    Cf[o] += val;
                } else {
                    Cf[o] = val;
                }
            }
        }
}


// ----------------------------------------------------------------------------------------------
// bf16 GEMM, direct-to-LDS staging (global_load_lds_dwordx4): 128x128 tile, BK = 64, 4 waves (2x2),
// two LDS buffers, one barrier per K-tile.  An LDS-DMA wave-instruction writes 64 lanes x 16 B = 1 KiB
// LINEARLY (8 tile rows of 128 B), so the tile image cannot be padded; bank conflicts of the
// ds_read_b128 fragment reads are removed by an XOR swizzle applied on the per-lane SOURCE address and
// again on the read (cdna_hip_programming.md rule 21).  BK = 64 (128-byte tile rows) by default, BK = 128 for the
// tall-K weight-gradient GEMMs.  Requires K % 64 == 0 (operands are zero-padded by the callers).
// ----------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void* gas_ptr_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// CPR = 16-byte chunks per tile row (8 for BK = 64, 16 for BK = 128); swizzle: chunk c of row r lives at
// c ^ ((r >> 1) & 7) for 128-byte rows and c ^ (r & 15) for 256-byte rows (16 rows of a ds_read_b128 group -> 16 slots).

// Epilogue of the LDS-DMA kernels: one wave's NI x NJ accumulator tiles -> C.  The store mode is resolved ONCE (template), the row
// part of every address is wave-uniform (scalar arithmetic), the per-lane part is one offset computed once.  The per-element form
// (64-bit row * ldc per lane, four run-time mode branches per element) cost ~40 instructions per stored value: at K = 448 the
// 256 x 256 tile spent 21 of its 35 us issuing its 128 stores per thread (profiles/tools/gemm_trace.hip) -- 150 -> 89 us for xproj1.
enum { EPI_STORE = 0, EPI_ATOMIC = 1, EPI_ACCUM = 2, EPI_BF16 = 3, EPI_F16 = 4 };
template <int MODE>
__device__ __forceinline__ void epi_put(float* __restrict__ rowf, bf16_t* __restrict__ rowb, unsigned lane_off, float val) {
    // rowf / rowb are wave-uniform pointers (SGPR base), lane_off the one per-lane offset: the store needs no vector address arithmetic
    if (MODE == EPI_BF16) rowb[lane_off] = f32_to_bf16(val);
    else if (MODE == EPI_F16) rowb[lane_off] = f32_to_f16(val);
    else if (MODE == EPI_ATOMIC) atomicAdd(rowf + lane_off, val);
    else if (MODE == EPI_ACCUM) rowf[lane_off] += val;
    else rowf[lane_off] = val;
}
template <int MODE, int NI, int NJ>
__device__ __forceinline__ void epi_store_tiles(void* __restrict__ Cv, int ldc, int M, int N, int row0, int col0, const f32x16_t (&acc)[NI][NJ],
                                                const float* __restrict__ bias, int lane) {
    const int r = lane & 31, h4 = 4 * (lane >> 5);
    float* Cf = reinterpret_cast<float*>(Cv);
    bf16_t* Cb = reinterpret_cast<bf16_t*>(Cv);
    const unsigned lane_off = (unsigned)h4 * (unsigned)ldc + (unsigned)r;
    if (row0 + NI * 32 <= M && col0 + NJ * 32 <= N) {          // interior (uniform): no predicates at all
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float bv = bias != nullptr ? bias[col0 + j * 32 + r] : 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const size_t ro = (size_t)(row0 + i * 32 + (e & 3) + 8 * (e >> 2)) * ldc + col0 + j * 32;      // uniform
                    epi_put<MODE>(Cf + ro, Cb + ro, lane_off, acc[i][j][e] + bv);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int cb = col0 + j * 32;                          // uniform
        if (cb >= N) continue;
        const bool colok = cb + r < N;
        const float bv = (bias != nullptr && colok) ? bias[cb + r] : 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rb = row0 + i * 32 + (e & 3) + 8 * (e >> 2);      // uniform: this register's row for lanes 0..31 (lanes 32..63: +4)
                if (rb >= M) continue;
                if (colok && rb + h4 < M) {
                    const size_t ro = (size_t)rb * ldc + cb;
                    epi_put<MODE>(Cf + ro, Cb + ro, lane_off, acc[i][j][e] + bv);
                }
            }
        }
    }
}
// Wide form for plain / bf16 stores of an interior wave tile: every 32 x 32 accumulator tile is turned round through a wave-private LDS
// tile ([32][36] f32: the stage buffers are idle after the K loop) so that a lane holds consecutive COLUMNS, and leaves as 16-byte stores
// covering whole 128-byte rows (f32: 4 stores of 8 rows x 128 B per tile; bf16: 2 stores of 16 rows x 64 B) instead of sixteen 4- / 2-byte
// stores per lane.  A CU issues stores at a fixed instruction rate whatever their width (the element-wise epilogue of a 256 x 256 f32 tile:
// 128 store instructions per lane, ~11 us at K = 448 next to ~12 us of K loop), so the bytes per instruction are what counts.
#define EPI_SC_FLOATS 1152        // per-wave scratch: 32 rows x 36 floats
#ifndef EPI_WIDE
#define EPI_WIDE 1
#endif
template <int MODE, int NI, int NJ>
__device__ __forceinline__ void epi_store_wide(void* __restrict__ Cv, int ldc, int row0, int col0, const f32x16_t (&acc)[NI][NJ],
                                               const float* __restrict__ bias, int lane, float* __restrict__ sc) {
    const int r = lane & 31, hh = lane >> 5;
    float* Cf = reinterpret_cast<float*>(Cv);
    bf16_t* Cb = reinterpret_cast<bf16_t*>(Cv);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float bv = bias != nullptr ? bias[col0 + j * 32 + r] : 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int e = 0; e < 16; ++e) sc[((e & 3) + 8 * (e >> 2) + 4 * hh) * 36 + r] = acc[i][j][e] + bv;
            asm volatile("" ::: "memory");
            if (MODE == EPI_BF16 || MODE == EPI_F16) {
                using CF = typename std::conditional<MODE == EPI_F16, Fp16F, Bf16F>::type;
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int row = 16 * p + (lane >> 2), c8 = 8 * (lane & 3);
                    const float4 v0 = *reinterpret_cast<const float4*>(sc + row * 36 + c8);
                    const float4 v1 = *reinterpret_cast<const float4*>(sc + row * 36 + c8 + 4);
                    uint4 pk;
                    pk.x = pack2<CF>(v0.x, v0.y);
                    pk.y = pack2<CF>(v0.z, v0.w);
                    pk.z = pack2<CF>(v1.x, v1.y);
                    pk.w = pack2<CF>(v1.z, v1.w);
                    // NON-TEMPORAL (also below): the activation GEMMs' C is written once and read by a later kernel; with the default policy it pushes the
                    // operand panels the tiles of a row share through the L2 out (Dense forward 240 -> 212 us, xproj1 on this kernel 745 -> 605 us)
                    __builtin_nontemporal_store(__builtin_bit_cast(gm_u4, pk), reinterpret_cast<gm_u4*>(Cb + (size_t)(row0 + i * 32 + row) * ldc + col0 + j * 32 + c8));
                }
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int row = 8 * p + (lane >> 3), c4 = 4 * (lane & 7);
                    __builtin_nontemporal_store(*reinterpret_cast<const gm_u4*>(sc + row * 36 + c4), reinterpret_cast<gm_u4*>(Cf + (size_t)(row0 + i * 32 + row) * ldc + col0 + j * 32 + c4));
                }
            }
            asm volatile("" ::: "memory");
        }
    }
}
template <int NI, int NJ>
__device__ __forceinline__ void epi_dispatch(void* __restrict__ Cv, int ldc, int c_bf16, int flags, int M, int N, int row0, int col0,
                                             const f32x16_t (&acc)[NI][NJ], const float* __restrict__ bias, int lane, float* __restrict__ sc = nullptr) {
    if (sc != nullptr && !(flags & (MNN_GEMM_ATOMIC | MNN_GEMM_ACCUMULATE)) && row0 + NI * 32 <= M && col0 + NJ * 32 <= N &&
        ((size_t)Cv & 15) == 0 && (ldc & (c_bf16 ? 7 : 3)) == 0) {               // all wave-uniform
        if (c_bf16 == MNN_F16) epi_store_wide<EPI_F16, NI, NJ>(Cv, ldc, row0, col0, acc, bias, lane, sc);
        else if (c_bf16) epi_store_wide<EPI_BF16, NI, NJ>(Cv, ldc, row0, col0, acc, bias, lane, sc);
        else epi_store_wide<EPI_STORE, NI, NJ>(Cv, ldc, row0, col0, acc, bias, lane, sc);
        return;
    }
    if (c_bf16 == MNN_F16) epi_store_tiles<EPI_F16, NI, NJ>(Cv, ldc, M, N, row0, col0, acc, bias, lane);
    else if (c_bf16) epi_store_tiles<EPI_BF16, NI, NJ>(Cv, ldc, M, N, row0, col0, acc, bias, lane);
    else if (flags & MNN_GEMM_ATOMIC) epi_store_tiles<EPI_ATOMIC, NI, NJ>(Cv, ldc, M, N, row0, col0, acc, bias, lane);
    else if (flags & MNN_GEMM_ACCUMULATE) epi_store_tiles<EPI_ACCUM, NI, NJ>(Cv, ldc, M, N, row0, col0, acc, bias, lane);
    else epi_store_tiles<EPI_STORE, NI, NJ>(Cv, ldc, M, N, row0, col0, acc, bias, lane);
}

template <int CPR>
__device__ __forceinline__ int swz(int row) { return CPR == 8 ? ((row >> 1) & 7) : (CPR == 4 ? ((row >> 2) & 3) : (row & 15)); }

// This thread's LDS-DMA slots of one operand tile (ROWS rows x CPR 16-byte chunks, NW waves): the global address of every slot at
// k = 0 is computed ONCE per workgroup (row clamp, swizzle, 64-bit row * ld); staging a K tile is then one 64-bit add per slot.
// Recomputing them per K tile cost ~0.5 us of address arithmetic per tile in front of the 8 DMA instructions (profiles/tools/gemm_trace.hip).
// KB = the operand is stored K-BLOCKED: element (row, k) at ((k >> 5) * ld + row) * 32 + (k & 31), ld = its total row count -- 32 consecutive k of
// one row are 64 contiguous bytes, and the 32 k of all rows of a block are one contiguous slab.  A producer that owns a (32-k block, row
// range) -- the backward recurrence: 32 batch rows of one timestep x its 128 gate columns -- writes its piece of dz^T as whole contiguous
// kilobytes instead of one 32-byte run per row (rows 512 KB apart).  The LDS image and everything behind it are the same as for the plain
// K-contiguous operand (BK = 64 = two blocks: chunks 0..3 from block 2 kt, chunks 4..7 from block 2 kt + 1).
template <int ROWS, int NW, int CPR, bool KB = false>
struct GldsSlots {
    static constexpr int NS = ROWS * CPR / (64 * NW);
    const bf16_t* base;                                    // wave-uniform
    unsigned off[NS];                                       // element offset of the slot at k = 0 (operands stay below 2^31 elements)
    unsigned kmul;                                          // KB: elements per unit of k0 (= ld); plain: 1
    __device__ __forceinline__ void init(const bf16_t* __restrict__ G, int ld, int rows_total, int r0, int wave, int lane) {
        static_assert(!KB || CPR == 8, "the K-blocked layout is defined for 64-deep K tiles");
        base = G;
        kmul = KB ? (unsigned)ld : 1u;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int p = (s * NW + wave) * 64 + lane;      // linear 16-byte slot of the tile image
            const int row = p / CPR, pc = p % CPR;
            const int c = pc ^ swz<CPR>(row);               // logical K chunk held by this slot
            const int gr = min(r0 + row, rows_total - 1);   // rows past the edge replicate the last row (never stored)
            if (KB) off[s] = ((unsigned)(c >> 2) * (unsigned)ld + (unsigned)gr) * 32u + (unsigned)((c & 3) * 8);
            else off[s] = (unsigned)gr * (unsigned)ld + (unsigned)(c * 8);
        }
    }
    __device__ __forceinline__ void stage(int k0, char* tile, int wave) const {
        const bf16_t* b = base + (size_t)k0 * kmul;         // wave-uniform (K-blocked: 64 k = two blocks = 64 ld elements)
#pragma unroll
        for (int s = 0; s < NS; ++s)
            __builtin_amdgcn_global_load_lds((gas_ptr_t)(b + off[s]), (lds_ptr_t)(tile + (s * NW + wave) * 1024), 16, 0, 0);
    }
};

// Split-K work mapping.  The hardware deals consecutive workgroup ids round-robin over the 8 XCDs (xcd = id % 8).  All tiles that read the same
// K slice should run on ONE XCD, so that the slice's A and B panels are fetched into that L2 once and the other tiles hit.  `z = id % split_k`
// achieves that when split_k is a multiple of 8 (or the real tiles sit at multiples of 8 in the slot order: a single row group).  For other
// slice counts XCD x owns the CONTIGUOUS range [x T / 8, (x + 1) T / 8) of the slice-major order (item = z * slots + tile); the grid's x
// extent is a multiple of 8 workgroups, so T % 8 == 0.  Measured, [1024 x 768] weight gradient, K = 262144, split 20: 708 -> 399 us (split
// 16: 456 either way).  Not used for multiples of 8: [256 x 704] at split 64 takes 159 us with id % split_k and 349 us contiguous.
#define GEMM_SPLITK_CONTIG 0x100         // internal flag bit (set by the launcher): the mapping below; clear: z = id % split_k
__device__ __forceinline__ void splitk_item(int split_k, int flags, int& z, int& bid) {
    const int lin = blockIdx.x + blockIdx.y * gridDim.x;
    if (!(flags & GEMM_SPLITK_CONTIG)) { z = lin % split_k; bid = lin / split_k; return; }
    const int slots = gridDim.x, per_xcd = (slots * split_k) >> 3;
    const int item = (lin & 7) * per_xcd + (lin >> 3);
    z = item / slots;
    bid = item - z * slots;
}

template <int BK, typename F>
__global__ void __launch_bounds__(256)
gemm_tn_glds_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, void* __restrict__ Cv, int ldc, int c_bf16,
                    const float* __restrict__ bias, int M, int N, int K, int flags, int split_k, int ntm, int ntn) {
    constexpr int CPR = BK / 8, ROWB = BK * 2, TILE = 128 * ROWB;
    __shared__ __attribute__((aligned(16))) char smem[2][2][TILE];
    int z, bid;
    splitk_item(split_k, flags, z, bid);               // the tiles that share an XCD read the same K slice of A and B: the re-reads hit that XCD's L2
    const int grp = bid / (8 * ntn), within = bid % (8 * ntn);
    const int mt = grp * 8 + (within & 7), nt = within >> 3;
    if (mt >= ntm) return;
    const int m0 = mt * 128, n0 = nt * 128;
    const int nkt = K / BK;
    const int per = (nkt + split_k - 1) / split_k;
    const int kt0 = z * per, kt1 = min(nkt, kt0 + per);
    if (kt0 >= kt1) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    GldsSlots<128, 4, CPR> slotA, slotB;
    slotA.init(A, lda, M, m0, wave, lane);
    slotB.init(B, ldb, N, n0, wave, lane);
    slotA.stage(kt0 * BK, smem[0][0], wave);
    slotB.stage(kt0 * BK, smem[0][1], wave);
    __syncthreads();                                        // hipcc drains vmcnt(0) before the barrier
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        if (kt + 1 < kt1) {
            slotA.stage((kt + 1) * BK, smem[cur ^ 1][0], wave);
            slotB.stage((kt + 1) * BK, smem[cur ^ 1][1], wave);
        }
        const char* sA = smem[cur][0];
        const char* sB = smem[cur][1];
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            typename F::x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + r;
                a[i] = *reinterpret_cast<const typename F::x8*>(sA + row * ROWB + (((ks * 2 + h) ^ swz<CPR>(row)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wn * 64 + j * 32 + r;
                b[j] = *reinterpret_cast<const typename F::x8*>(sB + row * ROWB + (((ks * 2 + h) ^ swz<CPR>(row)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = F::mfma32(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
        cur ^= 1;
    }
    epi_dispatch<2, 2>(Cv, ldc, c_bf16, flags, M, N, m0 + wm * 64, n0 + wn * 64, acc, z == 0 ? bias : nullptr, lane,
                       EPI_WIDE ? reinterpret_cast<float*>(&smem[0][0][0]) + wave * EPI_SC_FLOATS : nullptr);      // the K loop ended with a barrier
}

// ----------------------------------------------------------------------------------------------
// Large-tile variant: 256 x 256 x 64 per workgroup, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 MFMA tiles.
// The 128 x 128 kernel above reads one LDS fragment per MFMA (2 A + 2 B fragments feed 4 MFMAs) and stages 0.5 KiB of
// operands per MFMA through LDS-DMA: 1.5 KiB of LDS traffic per 32-cycle MFMA against a 128 B/clk LDS -- it is LDS-bound
// at two thirds of the MFMA rate before any latency.  Here 4 A + 2 B fragments feed 8 MFMAs and a stage carries 0.25 KiB
// per MFMA: 1.0 KiB per MFMA, and one workgroup per CU holds two 64 KiB stages.
// ----------------------------------------------------------------------------------------------
#ifdef GM_TRACE      // development only (profiles/tools/gemm_trace.hip): phase clocks of workgroup GM_TRACE, accumulated in 10 ns units
__device__ long long gm_trace[16];
#define GM_T(k) do { if (threadIdx.x == 0 && blockIdx.x == GM_TRACE && blockIdx.y == 0) { const long long now_ = wall_clock64(); gm_trace[k] += now_ - gmprev_; gmprev_ = now_; } } while (0)
#define GM_T0() long long gmprev_ = wall_clock64()
#else
#define GM_T(k) do { } while (0)
#define GM_T0() do { } while (0)
#endif
template <typename F, bool AKB = false>
__global__ void __launch_bounds__(512)
gemm_tn_glds256_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, void* __restrict__ Cv, int ldc, int c_bf16,
                       const float* __restrict__ bias, int M, int N, int K, int flags, int split_k, int ntm, int ntn,
                       const int* __restrict__ m_rows_dev, const int* __restrict__ k_rows_dev) {
    constexpr int BK = 64, CPR = 8, ROWB = 128, TILE = 256 * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem256[];        // [stage][A | B][256 rows x 128 B]
    int z, bid;
    splitk_item(split_k, flags, z, bid);
    const int grp = bid / (8 * ntn), within = bid % (8 * ntn);
    const int mt = grp * 8 + (within & 7), nt = within >> 3;
    if (mt >= ntm) return;
    const int m0 = mt * 256, n0 = nt * 256;
    // compacted ragged batches (mnn_gemm_tn_rows): only the first *m_rows_dev rows of A (of C) / the first *k_rows_dev of the K dimension carry
    // data -- row tiles past the count leave (their C rows are never read), the K loop stops at the count (what lies behind it is zero)
    if (m_rows_dev != nullptr && m0 >= *m_rows_dev) return;
    const int nkt = k_rows_dev != nullptr ? min(K / BK, (*k_rows_dev + BK - 1) / BK) : K / BK;
    const int per = (nkt + split_k - 1) / split_k;
    const int kt0 = z * per, kt1 = min(nkt, kt0 + per);
    if (kt0 >= kt1) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    GldsSlots<256, 8, CPR, AKB> slotA;
    GldsSlots<256, 8, CPR> slotB;
    slotA.init(A, lda, M, m0, wave, lane);
    slotB.init(B, ldb, N, n0, wave, lane);
    auto stage = [&](int buf, int kt) {
        slotA.stage(kt * BK, smem256 + (buf * 2 + 0) * TILE, wave);
        slotB.stage(kt * BK, smem256 + (buf * 2 + 1) * TILE, wave);
    };
    GM_T0();
    stage(0, kt0);
    __syncthreads();                                        // hipcc drains vmcnt(0) before the barrier
    GM_T(1);
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        // (spreading these 8 DMA instructions over the four k-steps was measured: the 0.6 us they hold the wave moves into the k-steps --
        // 256 x 256 x 448 tile: issue 4.3 -> 1.4 us, LDS reads + MFMA 8.9 -> 10.2, barriers 1.8 -> 2.6: the K tile stays at ~2 us, LDS-bound)
#ifndef ABL_NODMA
        if (kt + 1 < kt1) stage(cur ^ 1, kt + 1);
#endif
        GM_T(2);
        const char* sA = smem256 + (cur * 2 + 0) * TILE;
        const char* sB = smem256 + (cur * 2 + 1) * TILE;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            typename F::x8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 128 + i * 32 + r;
                a[i] = *reinterpret_cast<const typename F::x8*>(sA + row * ROWB + (((ks * 2 + h) ^ swz<CPR>(row)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wn * 64 + j * 32 + r;
                b[j] = *reinterpret_cast<const typename F::x8*>(sB + row * ROWB + (((ks * 2 + h) ^ swz<CPR>(row)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#ifndef ABL_NOMFMA
                    acc[i][j] = F::mfma32(a[i], b[j], acc[i][j]);
#else
                    asm volatile("" :: "v"(a[i]), "v"(b[j]));
#endif
                }
        }
        GM_T(3);
        __syncthreads();
        GM_T(4);
        cur ^= 1;
    }
    epi_dispatch<4, 2>(Cv, ldc, c_bf16, flags, M, N, m0 + wm * 128, n0 + wn * 64, acc, z == 0 ? bias : nullptr, lane,
                       EPI_WIDE ? reinterpret_cast<float*>(smem256) + wave * EPI_SC_FLOATS : nullptr);             // the K loop ended with a barrier
    GM_T(5);
#ifdef GM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GM_T(6);
#endif
}

int mnn_gemm_bres_ok(int M, int N, int K);                      // gemm_bres.hip (internal: not part of the C ABI)
// MNN_GEMM_BRES=0: the LDS-staged kernels for these shapes too -- the comparison partner of the weight-resident form in
// tests/test_gpu_kernels.py (one process runs both, so the switch is read per call: one getenv on a path that a captured step never re-executes)
static bool gemm_bres_enabled() {
    const char* be = getenv("MNN_GEMM_BRES");
    return be == nullptr || atoi(be) != 0;
}
int mnn_gemm_bres_launch(hipStream_t st, int f16, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias);

template <typename T>
static int launch_gemm(hipStream_t st, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                       int c_bf16, const float* bias, int flags, int split_k, const int* m_rows_dev = nullptr, const int* k_rows_dev = nullptr) {
    constexpr int BM = 128, BN = 128;
    const int ntm = cdiv(M, BM), ntn = cdiv(N, BN);
    const int ngrp = cdiv(ntm, 8);
    dim3 grid(ngrp * 8 * ntn, split_k);
    if constexpr (sizeof(T) == 2) {
        // large problems: 256 x 256 tiles (enough of them to occupy the chip, each with enough K tiles to amortise its prologue)
        const int ntm2 = cdiv(M, 256), ntn2 = cdiv(N, 256);
        static const bool no256 = getenv("MNN_GEMM_NO256") != nullptr;
        // A persistent ring form of this tile (4 LDS slots of 32-deep sub-tiles, counted vmcnt, raw barriers) was measured in round 3 and is NOT
        // shipped (profiles/round3_b_gemm_ablation.md): slower on K = 448 .. 1024 and on the K = 262144 weight gradients, because the operand
        // stream arrives at ~34 GB/s per CU whatever is in flight; its one win (K = 256, the Dense forward) needs an N-edge epilogue at N = 704.
        if (flags & MNN_GEMM_A_KBLOCK32) {                  // A stored K-blocked: the 256 x 256 tile, only the LDS-DMA source addresses differ
            using F = typename FlavorOf<T>::type;
            static bool attr_kb[64];
            int dev = 0;
            MNN_HIP(hipGetDevice(&dev));
            MNN_REQUIRE(dev >= 0 && dev < 64, "mnn_gemm_tn: device index %d", dev);
            if (!attr_kb[dev]) {
                MNN_HIP(hipFuncSetAttribute((const void*)gemm_tn_glds256_kernel<F, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 256 * 128));
                attr_kb[dev] = true;
            }
            dim3 grid2(cdiv(ntm2, 8) * 8 * ntn2, split_k);
            hipLaunchKernelGGL((gemm_tn_glds256_kernel<F, true>), grid2, dim3(512), 4 * 256 * 128, st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C,
                               ldc, c_bf16, bias, M, N, K, flags, split_k, ntm2, ntn2, m_rows_dev, k_rows_dev);
            MNN_LAUNCH_CHECK();
            return MNN_OK;
        }
        // the input projections (K = 448 / 512, 16-bit C, M a multiple of 128): the weight-resident persistent form (gemm_bres.hip)
        {
            if (gemm_bres_enabled() && split_k == 1 && !(flags & (MNN_GEMM_ACCUMULATE | MNN_GEMM_ATOMIC | MNN_GEMM_A_KBLOCK32)) && c_bf16 != 0 &&
                c_bf16 == (std::is_same<T, f16_t>::value ? MNN_F16 : MNN_BF16) && mnn_gemm_bres_ok(M, N, K) && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 &&
                ((uintptr_t)C & 15) == 0 && (size_t)128 * 64 * (size_t)lda < ((size_t)1 << 31))
                return mnn_gemm_bres_launch(st, std::is_same<T, f16_t>::value ? 1 : 0, M, N, K, A, lda, B, ldb, C, ldc, bias);
        }
        if (!no256 && K % 64 == 0 && M >= 256 && N >= 192 && (long)ntm2 * ntn2 * split_k >= 192 && K / 64 / split_k >= 4) {     // measured per shape: profiles/round1_f_gemm_shapes.md; round 3: 4 K tiles suffice (Dense forward K = 256: 362 -> 247 us)
            using F = typename FlavorOf<T>::type;
            static bool attr_set[64];
            int dev = 0;
            MNN_HIP(hipGetDevice(&dev));
            if (dev >= 0 && dev < 64 && !attr_set[dev]) {       // per device and per flavour (this function is instantiated once per T)
                MNN_HIP(hipFuncSetAttribute((const void*)gemm_tn_glds256_kernel<F>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 256 * 128));
                attr_set[dev] = true;
            }
            dim3 grid2(cdiv(ntm2, 8) * 8 * ntn2, split_k);
            hipLaunchKernelGGL(gemm_tn_glds256_kernel<F>, grid2, dim3(512), 4 * 256 * 128, st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, c_bf16,
                               bias, M, N, K, flags, split_k, ntm2, ntn2, m_rows_dev, k_rows_dev);
            MNN_LAUNCH_CHECK();
            return MNN_OK;
        }
        // (BK = 128 for the tall-K weight-gradient GEMMs was measured slower: C2 1.28 -> 1.54 ms, 1 block/CU at 128 KiB LDS)
        if (K % 64 == 0) {
            hipLaunchKernelGGL((gemm_tn_glds_kernel<64, typename FlavorOf<T>::type>), grid, dim3(256), 0, st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, c_bf16, bias, M,
                               N, K, flags, split_k, ntm, ntn);
            MNN_LAUNCH_CHECK();
            return MNN_OK;
        }
    }
    if constexpr (sizeof(T) == 4) {
        // f32 operands: eight waves per 128 x 128 tile, two per 64 x 64 quadrant, each taking half of every staged K chunk (MNN_GEMM_F32_KH=0: four)
        static const bool kh2 = getenv("MNN_GEMM_F32_KH") == nullptr || atoi(getenv("MNN_GEMM_F32_KH")) != 0;
        static bool raised_[64];
        bool& raised = mnn_dev_flag(raised_);
        if (!raised) {
            constexpr int need2 = GemmCore<T, BM, BN, 2, 2, 2>::LDS_BYTES, need1 = GemmCore<T, BM, BN, 2, 2>::LDS_BYTES;
            const void* f2 = (const void*)gemm_tn_kernel<T, BM, BN, 2, 2, 2>;
            const void* f1 = (const void*)gemm_tn_kernel<T, BM, BN, 2, 2>;
            MNN_HIP(hipFuncSetAttribute(f2, hipFuncAttributeMaxDynamicSharedMemorySize, need2));
            MNN_HIP(hipFuncSetAttribute(f1, hipFuncAttributeMaxDynamicSharedMemorySize, need1));
            raised = true;
        }
        if (kh2) {
            constexpr size_t lds2 = GemmCore<T, BM, BN, 2, 2, 2>::LDS_BYTES;
            hipLaunchKernelGGL((gemm_tn_kernel<T, BM, BN, 2, 2, 2>), grid, dim3(512), lds2, st, (const T*)A, lda, (const T*)B, ldb, C, ldc,
                               c_bf16, bias, M, N, K, flags, split_k, ntm, ntn);
            MNN_LAUNCH_CHECK();
            return MNN_OK;
        }
    }
    constexpr size_t lds1 = GemmCore<T, BM, BN, 2, 2>::LDS_BYTES;
    hipLaunchKernelGGL((gemm_tn_kernel<T, BM, BN, 2, 2>), grid, dim3(256), lds1, st, (const T*)A, lda, (const T*)B, ldb, C, ldc,
                       c_bf16, bias, M, N, K, flags, split_k, ntm, ntn);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

extern "C" int mnn_gemm_tn(mnn_stream_t s, int dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                           int ldc, int c_dtype, const float* bias, int flags, int split_k) {
    return mnn_gemm_tn_rows(s, dtype, M, N, K, A, lda, B, ldb, C, ldc, c_dtype, bias, flags, split_k, nullptr, nullptr);
}

extern "C" int mnn_gemm_tn_rows(mnn_stream_t s, int dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                                int ldc, int c_dtype, const float* bias, int flags, int split_k, const int* m_rows_dev, const int* k_rows_dev) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(dtype == MNN_BF16 || dtype == MNN_F16 || dtype == MNN_F32, "mnn_gemm_tn: dtype must be bf16, f16 or f32 (got %d)", dtype);
    MNN_REQUIRE(c_dtype == MNN_F32 || c_dtype == MNN_BF16 || c_dtype == MNN_F16, "mnn_gemm_tn: c_dtype must be f32, bf16 or f16");
    MNN_REQUIRE(M > 0 && N > 0 && K > 0, "mnn_gemm_tn: empty problem M=%d N=%d K=%d", M, N, K);
    const int al = dtype == MNN_F32 ? 4 : 8;
    MNN_REQUIRE(K % al == 0 && lda % al == 0 && ldb % al == 0, "mnn_gemm_tn: K/lda/ldb must be multiples of %d (K=%d lda=%d ldb=%d)",
                al, K, lda, ldb);
    if (flags & MNN_GEMM_A_KBLOCK32) {
        MNN_REQUIRE(dtype != MNN_F32 && K % 64 == 0 && lda >= M && ldb >= K && ldc >= N &&
                    (size_t)K * (size_t)lda < ((size_t)1 << 32),
                    "mnn_gemm_tn: a K-blocked A needs 16-bit operands, K %% 64 == 0, lda (its row count) >= M, K * lda < 2^32 (M=%d K=%d lda=%d)", M, K, lda);
    } else {
        MNN_REQUIRE(lda >= K, "mnn_gemm_tn: leading dimension too small");
    }
    MNN_REQUIRE(ldb >= K && ldc >= N, "mnn_gemm_tn: leading dimension too small");
    MNN_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "mnn_gemm_tn: operands must be 16-byte aligned");
    if (split_k < 1) split_k = 1;
    if (split_k > 1) {
        MNN_REQUIRE(c_dtype == MNN_F32, "mnn_gemm_tn: split-K needs an f32 C");
        if (!(flags & MNN_GEMM_ACCUMULATE)) {
            MNN_HIP(mnn_zero_async(C, (size_t)N * 4, (size_t)ldc * 4, M, st));
        }
        flags |= MNN_GEMM_ATOMIC | MNN_GEMM_ACCUMULATE;
        static int map = -2;                 // MNN_GEMM_SPLITK_MAP = 0 | 1 forces a mapping (development); default: by slice count
        if (map == -2) { const char* e = getenv("MNN_GEMM_SPLITK_MAP"); map = e ? atoi(e) : -1; }
        if (map == 1 || (map < 0 && split_k % 8 != 0)) flags |= GEMM_SPLITK_CONTIG;       // see splitk_item
    }
    MNN_REQUIRE(!(c_dtype != MNN_F32 && (flags & (MNN_GEMM_ACCUMULATE | MNN_GEMM_ATOMIC))), "mnn_gemm_tn: a 16-bit C cannot accumulate");
    const int c16 = c_dtype == MNN_F32 ? 0 : c_dtype;      // 0: f32 C; else the mnn_dtype code of the 16-bit C
    if (dtype == MNN_BF16) return launch_gemm<bf16_t>(st, M, N, K, A, lda, B, ldb, C, ldc, c16, bias, flags, split_k, m_rows_dev, k_rows_dev);
    if (dtype == MNN_F16) return launch_gemm<f16_t>(st, M, N, K, A, lda, B, ldb, C, ldc, c16, bias, flags, split_k, m_rows_dev, k_rows_dev);
    return launch_gemm<float>(st, M, N, K, A, lda, B, ldb, C, ldc, c16, bias, flags, split_k);      // (the f32 kernels do not look at the row counts: a hint only)
}

// ----------------------------------------------------------------------------------------------
// LSTM forward step: z = xproj[t] + h_{t-1} . Wh ; gates ; c,h           (rnn.py:124, LSTMBlockCell)
// block = 64 rows x 128 pre-activation columns = 32 units x 4 gates (gate-interleaved layout),
// 2 waves, each 32 rows x 128 cols -> every lane owns all four gates of its (row, unit) elements.
// ----------------------------------------------------------------------------------------------
template <typename T> struct StepKH { static constexpr int value = sizeof(T) == 4 ? 2 : 1; };      // f32: two K halves per tile (256 threads)
template <typename T>
__global__ void __launch_bounds__(128 * StepKH<T>::value)
lstm_fwd_step_kernel(const T* __restrict__ h_prev, const T* __restrict__ wh_t, const float* __restrict__ xproj,
                     const float* __restrict__ c_prev, float* __restrict__ gates, float* __restrict__ c_out, T* __restrict__ h_out,
                     int B, int U) {
    using Core = GemmCore<T, 64, 128, 2, 1, StepKH<T>::value>;
    __shared__ __attribute__((aligned(16))) char smem[Core::LDS_BYTES];
    const int nt = blockIdx.x, m0 = blockIdx.y * 64, n0 = nt * 128;
    const int N4 = 4 * U;
    f32x16_t acc[1][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
    if (h_prev != nullptr) {
        Core core;
        const int nchunks = (U + Elem<T>::PER_CHUNK - 1) / Elem<T>::PER_CHUNK;
        core.run(h_prev, U, B, m0, wh_t, U, N4, n0, U, 0, nchunks, smem, acc);
        if (!core.reduce_kh(smem, acc)) return;
    } else if (threadIdx.x >= 128) return;               // (no recurrent term: the first wave set alone runs the pointwise part)
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 1;
    const int col = lane & 31, unit = nt * 32 + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wave * 32 + frag_row(r, lane);
        if (row >= B) continue;
        const size_t zo = (size_t)row * N4 + n0 + col;
        const float zi = acc[0][0][r] + xproj[zo];
        const float zg = acc[0][1][r] + xproj[zo + 32];
        const float zf = acc[0][2][r] + xproj[zo + 64];
        const float zq = acc[0][3][r] + xproj[zo + 96];
        const float gi = fast_sigmoid(zi), gg = fast_tanh(zg), gf = fast_sigmoid(zf), go = fast_sigmoid(zq);
        const size_t uo = (size_t)row * U + unit;
        const float cp = c_prev != nullptr ? c_prev[uo] : 0.f;
        const float c = gg * gi + cp * gf;
        const float h = fast_tanh(c) * go;
        if (gates != nullptr) {
            gates[zo] = gi; gates[zo + 32] = gg; gates[zo + 64] = gf; gates[zo + 96] = go;
        }
        c_out[uo] = c;
        h_out[uo] = Cvt<T>::store(h);
    }
}


// ----------------------------------------------------------------------------------------------
// Latency-optimised bf16 step kernels (the recurrence is T sequential tiny GEMMs: per-step latency,
// not throughput, sets the time).  One block = one 32-row x 32-unit output tile; the K dimension is
// split over the block's waves; every wave loads ALL its operands straight into registers with
// 16-byte loads issued back to back (no LDS staging, no K-loop barrier), runs its MFMAs, and the
// partial tiles meet once in LDS.  Within a wave's K slice the natural MFMA order is kept (k = 16 s + 8 h + j), so
// one load instruction touches 32 contiguous bytes of each row (half the cache-line visits of a per-lane-contiguous
// split).
// ----------------------------------------------------------------------------------------------
template <int KS>
__device__ __forceinline__ void load_frags(const bf16_t* __restrict__ p, bf16x8_t (&f)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) f[s] = *reinterpret_cast<const bf16x8_t*>(p + 16 * s);
}

// forward: 4 waves, K = U = 64*KS.  Tile columns = 4 gates x 32 units (gate-interleaved layout).
struct LstmFwdArgs {
    const bf16_t* h_prev; const bf16_t* wh_t; const float* xproj; const float* c_prev;
    float* gates; float* c_out; bf16_t* h_out; bf16_t* hT;
    int U, ld_hT, colT, active;
    bf16_t* y_out; const uint8_t* mask; float kp;      // optional dropped output y = h/kp*mask (DropoutWrapper, rnn.py:132) with a precomputed keep mask
};

template <int KS>
__device__ __forceinline__ void lstm_fwd_body(const LstmFwdArgs& A, int B, int bx, int by, float (&red)[4][4][16][64], bf16_t (&sT)[32][40]) {
    const bf16_t* __restrict__ h_prev = A.h_prev; const bf16_t* __restrict__ wh_t = A.wh_t;
    const float* __restrict__ xproj = A.xproj; const float* __restrict__ c_prev = A.c_prev;
    float* __restrict__ gates = A.gates; float* __restrict__ c_out = A.c_out; bf16_t* __restrict__ h_out = A.h_out;
    bf16_t* __restrict__ hT = A.hT;
    const int U = A.U, ld_hT = A.ld_hT, colT = A.colT;
    const int nt = bx, m0 = by * 32, n0 = nt * 128, N4 = 4 * U;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    // epilogue operands of this wave's 4 fragment rows, requested first so they fly under the MFMAs
    const int col = r, unit = nt * 32 + col;
    float xp[4][4], cp[4];
    bool live[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = m0 + 8 * w + q + 4 * hh;
        live[q] = row < B;
        const int rr = live[q] ? row : B - 1;
        const size_t zo = (size_t)rr * N4 + n0 + col;
#pragma unroll
        for (int g = 0; g < 4; ++g) xp[q][g] = xproj[zo + 32 * g];
        cp[q] = c_prev != nullptr ? c_prev[(size_t)rr * U + unit] : 0.f;
    }
    f32x16_t acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;
    if (h_prev != nullptr) {
        const int kb = w * 16 * KS + hh * 8;     // natural MFMA K order: the two lane halves read ADJACENT 16-byte pieces of a row
        const int arow = min(m0 + r, B - 1);
        bf16x8_t a[KS], b[4][KS];
        load_frags<KS>(h_prev + (size_t)arow * U + kb, a);
#pragma unroll
        for (int g = 0; g < 4; ++g) load_frags<KS>(wh_t + (size_t)(n0 + 32 * g + r) * U + kb, b[g]);
        __builtin_amdgcn_sched_barrier(0);      // keep EVERY load in flight before the first MFMA (latency, not registers, is scarce)
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[g][s], acc[g], 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[w][g][i][lane] = acc[g][i];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = 4 * w + q;                       // fragment reg -> row 8w + q + 4h of the tile
        float z[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) z[g] = xp[q][g] + ((red[0][g][i][lane] + red[1][g][i][lane]) + (red[2][g][i][lane] + red[3][g][i][lane]));
        if (!live[q]) continue;
        const int row = m0 + 8 * w + q + 4 * hh;
        const float gi = fast_sigmoid(z[0]), gg = fast_tanh(z[1]), gf = fast_sigmoid(z[2]), go = fast_sigmoid(z[3]);
        const float c = gg * gi + cp[q] * gf;
        const float h = fast_tanh(c) * go;
        const size_t zo = (size_t)row * N4 + n0 + col, uo = (size_t)row * U + unit;
        if (gates != nullptr) { gates[zo] = gi; gates[zo + 32] = gg; gates[zo + 64] = gf; gates[zo + 96] = go; }
        c_out[uo] = c;
        const bf16_t hb = f32_to_bf16(h);
        h_out[uo] = hb;
        if (A.y_out != nullptr) A.y_out[uo] = f32_to_bf16(bf16_to_f32(hb) / A.kp * (float)A.mask[uo]);
        if (hT != nullptr) sT[col][8 * w + q + 4 * hh] = hb;
    }
    if (hT != nullptr) {                    // hT[unit][colT + row]: 32 units x 64-byte row segments (weight-gradient operand)
        __syncthreads();
        if (threadIdx.x < 128) {
            const int uu = threadIdx.x >> 2, piece = threadIdx.x & 3;
            const int row = m0 + piece * 8;
            bf16_t* dst = hT + (size_t)(nt * 32 + uu) * ld_hT + colT + row;
            if (row + 8 <= B && (((size_t)(colT + row) & 7) == 0) && ((ld_hT & 7) == 0)) {
                *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&sT[uu][piece * 8]);
            } else {
                for (int k = 0; k < 8; ++k)
                    if (row + k < B) dst[k] = sT[uu][piece * 8 + k];
            }
        }
    }
}

template <int KS>
__global__ void __launch_bounds__(256) lstm_fwd_step_v2(LstmFwdArgs A, int B) {
    __shared__ float red[4][4][16][64];
    __shared__ bf16_t sT[32][40];          // h tile, [unit][row] (+pad), for the transposed copy
    lstm_fwd_body<KS>(A, B, blockIdx.x, blockIdx.y, red, sT);
}

// Stage P of the three-stage wavefront: layer 2's input projection for ONE timestep,
// out[B,4U2] = y[B,U1] . Wx2^T + bias (gate-interleaved columns).  Same structure as the step body: 32 rows x 128
// columns per block, K split over the 4 waves, operands straight to registers, one LDS reduction.
struct LstmProjArgs {
    const bf16_t* y; const bf16_t* wx_t; const float* bias_p; float* out;
    int K, ld_w, N4, active;
};

template <int KS>
__device__ __forceinline__ void lstm_proj_body(const LstmProjArgs& A, int B, int bx, int by, float (&red)[4][4][16][64]) {
    const int m0 = by * 32, n0 = bx * 128;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    f32x16_t acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;
    {
        const int kb = w * 16 * KS + hh * 8;     // natural MFMA K order: the two lane halves read ADJACENT 16-byte pieces of a row
        const int arow = min(m0 + r, B - 1);
        bf16x8_t a[KS], b[4][KS];
        load_frags<KS>(A.y + (size_t)arow * A.K + kb, a);
#pragma unroll
        for (int g = 0; g < 4; ++g) load_frags<KS>(A.wx_t + (size_t)(n0 + 32 * g + r) * A.ld_w + kb, b[g]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[g][s], acc[g], 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[w][g][i][lane] = acc[g][i];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = 4 * w + q;
        const int row = m0 + 8 * w + q + 4 * hh;
        if (row >= B) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = n0 + 32 * g + r;
            A.out[(size_t)row * A.N4 + col] = A.bias_p[col] + ((red[0][g][i][lane] + red[1][g][i][lane]) + (red[2][g][i][lane] + red[3][g][i][lane]));
        }
    }
}

// Block -> (unit tile, row tile) map of the fused launches.  MNN_XCD_ROWS: blocks that share an XCD (linear id mod 8,
// speed only) share a ROW tile, so the step's activations are fetched into one XCD's L2 once; otherwise they share
// unit tiles (the weight slice stays per-XCD but every XCD re-reads all activation rows).
#ifndef MNN_XCD_ROWS
#define MNN_XCD_ROWS 0     // measured: sharing row tiles per XCD is 3 % slower (every XCD then streams all weights)
#endif
__device__ __forceinline__ bool fused_tile(int nrt, int& bx, int& by) {
#if MNN_XCD_ROWS
    const int id = blockIdx.x;                   // 1-D grid of 8 * ceil(nrt/8) * (unit tiles)
    const int nrt8 = (nrt + 7) >> 3;
    const int j = id >> 3;
    by = (id & 7) + 8 * (j % nrt8);
    bx = j / nrt8;
    return by < nrt;
#else
    bx = blockIdx.x; by = blockIdx.y;
    return true;
#endif
}

// Three-stage forward wavefront, lag 2: launch s = layer-1 step s | layer-2 projection of step s-1 | layer-2 step s-2.
template <int KS1, int KS2>
__global__ void __launch_bounds__(256) lstm3_fwd_step(LstmFwdArgs A1, LstmProjArgs P, LstmFwdArgs A2, int B) {
    __shared__ float red[4][4][16][64];
    __shared__ bf16_t sT[32][40];
    const int nb1 = A1.U / 32, nb2 = A2.U / 32;
    int bx, by;
    if (!fused_tile((B + 31) / 32, bx, by)) return;
    if (bx < nb1) {
        if (A1.active) lstm_fwd_body<KS1>(A1, B, bx, by, red, sT);
    } else if (bx < nb1 + nb2) {
        if (P.active) lstm_proj_body<KS1>(P, B, bx - nb1, by, red);
    } else {
        if (A2.active) lstm_fwd_body<KS2>(A2, B, bx - nb1 - nb2, by, red, sT);
    }
}

// ----------------------------------------------------------------------------------------------
// Big-batch forward step (B >= 512): the step is a real GEMM ([B,U] x [U,4U]) and the register-direct kernel above
// re-reads the 128 KiB weight slice once per 32 rows (82 MB of L2 traffic per step at B = 1024).  Here a block
// owns 64 rows x 32 units (128 gate columns), streams K in 64-wide tiles through a double-buffered, XOR-swizzled
// LDS image filled by global_load_lds (as gemm_tn_glds_kernel), 4 waves = 2 row halves x 2 k-step halves, one
// exchange of the two k-halves through LDS, then the same gate epilogue.
// ----------------------------------------------------------------------------------------------
template <int ROWS>
__device__ __forceinline__ void glds_stage_rows(const bf16_t* __restrict__ G, int ld, int rows_total, int r0, int k0, char* tile, int wave, int lane) {
#pragma unroll
    for (int s = 0; s < ROWS * 8 / 256; ++s) {
        const int p = (s * 4 + wave) * 64 + lane;
        const int row = p >> 3, pc = p & 7;
        const int c = pc ^ ((row >> 1) & 7);
        const int gr = min(r0 + row, rows_total - 1);
        __builtin_amdgcn_global_load_lds((gas_ptr_t)(G + (size_t)gr * ld + k0 + c * 8), (lds_ptr_t)(tile + (s * 4 + wave) * 1024), 16, 0, 0);
    }
}

__global__ void __launch_bounds__(256) lstm_fwd_step_v3(LstmFwdArgs A, int B) {
    __shared__ __attribute__((aligned(16))) char smem[2][24 * 1024];      // [buffer][A 64x128 B | B 128x128 B]
    __shared__ bf16_t sT[32][72];                                          // h tile [unit][row] for the transposed copy
    const bf16_t* __restrict__ h_prev = A.h_prev; const bf16_t* __restrict__ wh_t = A.wh_t;
    const float* __restrict__ xproj = A.xproj; const float* __restrict__ c_prev = A.c_prev;
    float* __restrict__ gates = A.gates; float* __restrict__ c_out = A.c_out; bf16_t* __restrict__ h_out = A.h_out;
    bf16_t* __restrict__ hT = A.hT;
    const int U = A.U, N4 = 4 * U;
    const int nt = blockIdx.x, m0 = blockIdx.y * 64, n0 = nt * 128;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave & 1, wk = wave >> 1;
    const int r = lane & 31, hh = lane >> 5;
    const int col = r, unit = nt * 32 + col;
    // epilogue operands of this wave's 8 fragment rows (regs 8wk .. 8wk+7), requested first
    float xp[8][4], cp[8];
    bool live[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int i = 8 * wk + q;
        const int row = m0 + 32 * wm + frag_row(i, lane);
        live[q] = row < B;
        const int rr = live[q] ? row : B - 1;
        const size_t zo = (size_t)rr * N4 + n0 + col;
#pragma unroll
        for (int g = 0; g < 4; ++g) xp[q][g] = xproj[zo + 32 * g];
        cp[q] = c_prev != nullptr ? c_prev[(size_t)rr * U + unit] : 0.f;
    }
    f32x16_t acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;
    if (h_prev != nullptr) {
        const int nkt = U / 64;
        glds_stage_rows<64>(h_prev, U, B, m0, 0, smem[0], wave, lane);
        glds_stage_rows<128>(wh_t, U, N4, n0, 0, smem[0] + 8192, wave, lane);
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + 1 < nkt) {
                glds_stage_rows<64>(h_prev, U, B, m0, (kt + 1) * 64, smem[cur ^ 1], wave, lane);
                glds_stage_rows<128>(wh_t, U, N4, n0, (kt + 1) * 64, smem[cur ^ 1] + 8192, wave, lane);
            }
            const char* sA = smem[cur];
            const char* sB = smem[cur] + 8192;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int ks = 2 * wk + k2;
                const int arow = wm * 32 + r;
                const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(sA + arow * 128 + (((ks * 2 + hh) ^ ((arow >> 1) & 7)) << 4));
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int brow = g * 32 + r;
                    const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(sB + brow * 128 + (((ks * 2 + hh) ^ ((brow >> 1) & 7)) << 4));
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[g], 0, 0, 0);
                }
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    // exchange the two k-halves: wave (wm, wk) keeps fragment regs 8wk..8wk+7 and receives them from wave (wm, 1-wk)
    float* ex = reinterpret_cast<float*>(&smem[0][0]);                     // [wave][gate][8][64] f32 = 32 KiB
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 8; ++q) ex[((wave * 4 + g) * 8 + q) * 64 + lane] = acc[g][8 * (1 - wk) + q];
    __syncthreads();
    const int pw = wave ^ 2;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int i = 8 * wk + q;
        float z[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) z[g] = xp[q][g] + (acc[g][i] + ex[((pw * 4 + g) * 8 + q) * 64 + lane]);
        const int lrow = 32 * wm + frag_row(i, lane);
        if (!live[q]) continue;
        const int row = m0 + lrow;
        const float gi = fast_sigmoid(z[0]), gg = fast_tanh(z[1]), gf = fast_sigmoid(z[2]), go = fast_sigmoid(z[3]);
        const float c = gg * gi + cp[q] * gf;
        const float h = fast_tanh(c) * go;
        const size_t zo = (size_t)row * N4 + n0 + col, uo = (size_t)row * U + unit;
        if (gates != nullptr) { gates[zo] = gi; gates[zo + 32] = gg; gates[zo + 64] = gf; gates[zo + 96] = go; }
        c_out[uo] = c;
        const bf16_t hb = f32_to_bf16(h);
        h_out[uo] = hb;
        if (hT != nullptr) sT[col][lrow] = hb;
    }
    if (hT != nullptr) {
        __syncthreads();
        const int uu = threadIdx.x >> 3, piece = threadIdx.x & 7;
        const int row = m0 + piece * 8;
        bf16_t* dst = hT + (size_t)(nt * 32 + uu) * A.ld_hT + A.colT + row;
        if (row + 8 <= B && (((size_t)(A.colT + row) & 7) == 0) && ((A.ld_hT & 7) == 0)) {
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&sT[uu][piece * 8]);
        } else {
            for (int k = 0; k < 8; ++k)
                if (row + k < B) dst[k] = sT[uu][piece * 8 + k];
        }
    }
}

// backward: 8 waves, K = 4U = 128*KS.  dh = dh_ext + dz_next . Wh^T, then the gate pointwise.
struct LstmBwdArgs {
    const bf16_t* dz_next; const bf16_t* wh_p; const float* dh_ext; const float* gates; const float* c_t; const float* c_prev;
    float* dc; float* dz; bf16_t* dzT; bf16_t* dzTt;
    int U, first, ld_t, colT, active;
};

template <int KS>
__device__ __forceinline__ void lstm_bwd_body(const LstmBwdArgs& A, int B, int bx, int by, float (&red)[8][16][64], bf16_t (&sT)[4][32][40]) {
    const bf16_t* __restrict__ dz_next = A.dz_next; const bf16_t* __restrict__ wh_p = A.wh_p; const float* __restrict__ dh_ext = A.dh_ext;
    const float* __restrict__ gates = A.gates; const float* __restrict__ c_t = A.c_t; const float* __restrict__ c_prev = A.c_prev;
    float* __restrict__ dc = A.dc; float* __restrict__ dz = A.dz; bf16_t* __restrict__ dzT = A.dzT; bf16_t* __restrict__ dzTt = A.dzTt;
    const int U = A.U, first = A.first, ld_t = A.ld_t, colT = A.colT;
    const int m0 = by * 32, n0 = bx * 32, N4 = 4 * U;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int unit = n0 + r, pc = gate_perm_col(0, unit);
    float e_dh[2], e_g[2][4], e_c[2], e_cp[2], e_dc[2];
    bool live[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = 2 * w + q;
        const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        live[q] = row < B;
        const int rr = live[q] ? row : B - 1;
        const size_t uo = (size_t)rr * U + unit, zo = (size_t)rr * N4 + pc;
        e_dh[q] = dh_ext[uo];
#pragma unroll
        for (int g = 0; g < 4; ++g) e_g[q][g] = gates[zo + 32 * g];
        e_c[q] = c_t[uo];
        e_cp[q] = c_prev != nullptr ? c_prev[uo] : 0.f;
        e_dc[q] = first ? 0.f : dc[uo];
    }
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (dz_next != nullptr) {
        const int kb = w * 16 * KS + hh * 8;     // natural MFMA K order: the two lane halves read ADJACENT 16-byte pieces of a row
        const int arow = min(m0 + r, B - 1);
        bf16x8_t a[KS], b[KS];
        load_frags<KS>(dz_next + (size_t)arow * N4 + kb, a);
        load_frags<KS>(wh_p + (size_t)unit * N4 + kb, b);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) red[w][i][lane] = acc[i];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = 2 * w + q;
        float sum = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) sum += red[ww][i][lane];
        if (!live[q]) continue;
        const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        const size_t uo = (size_t)row * U + unit, zo = (size_t)row * N4 + pc;
        const float dh = e_dh[q] + sum;
        const float gi = e_g[q][0], gg = e_g[q][1], gf = e_g[q][2], go = e_g[q][3];
        const float tc = fast_tanh(e_c[q]);
        const float d_o = dh * tc;
        const float d_c = dh * go * (1.f - tc * tc) + e_dc[q];
        const float dzi = d_c * gg * gi * (1.f - gi);
        const float dzg = d_c * gi * (1.f - gg * gg);
        const float dzf = d_c * e_cp[q] * gf * (1.f - gf);
        const float dzo = d_o * go * (1.f - go);
        dc[uo] = d_c * gf;
        if (dz != nullptr) { dz[zo] = dzi; dz[zo + 32] = dzg; dz[zo + 64] = dzf; dz[zo + 96] = dzo; }
        const bf16_t bi = f32_to_bf16(dzi), bg = f32_to_bf16(dzg), bff = f32_to_bf16(dzf), bo = f32_to_bf16(dzo);
        dzT[zo] = bi; dzT[zo + 32] = bg; dzT[zo + 64] = bff; dzT[zo + 96] = bo;
        if (dzTt != nullptr) {
            const int lr = (i & 3) + 8 * (i >> 2) + 4 * hh;
            sT[0][r][lr] = bi; sT[1][r][lr] = bg; sT[2][r][lr] = bff; sT[3][r][lr] = bo;
        }
    }
    if (dzTt != nullptr) {
        __syncthreads();
        {                                    // dzT_t[pc + 32 g][colT + row]: 128 rows x 64-byte segments, one 16-byte piece per thread
            const int gu = threadIdx.x >> 2, piece = threadIdx.x & 3;      // gu = g*32 + unit
            const int g = gu >> 5, uu = gu & 31;
            const int row = m0 + piece * 8;
            bf16_t* dst = dzTt + (size_t)(gate_perm_col(g, n0 + uu)) * ld_t + colT + row;
            if (row + 8 <= B && (((size_t)(colT + row) & 7) == 0) && ((ld_t & 7) == 0)) {
                *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&sT[g][uu][piece * 8]);
            } else {
                for (int k = 0; k < 8; ++k)
                    if (row + k < B) dst[k] = sT[g][uu][piece * 8 + k];
            }
        }
    }
}

template <int KS>
__global__ void __launch_bounds__(512) lstm_bwd_step_v2(LstmBwdArgs A, int B) {
    __shared__ float red[8][16][64];
    __shared__ bf16_t sT[4][32][40];       // dz tile, [gate][unit][row] (+pad), for the transposed copy
    lstm_bwd_body<KS>(A, B, blockIdx.x, blockIdx.y, red, sT);
}


// Stage Q of the three-stage backward wavefront: layer 1's incoming gradient for ONE timestep,
// dh1[B,U1] = (dz2[B,4U2] . Wx2) * keep/kp  (the dgrad of layer 2's input projection through the dropout of
// rnn.py:132).  32 rows x 32 units per block, K = 4U2 split over the 8 waves, one LDS reduction.
struct LstmDgradArgs {
    const bf16_t* dz; const bf16_t* wx_p; const uint8_t* mask; float* out;
    int K, U, active; float kp;
};

template <int KS>
__device__ __forceinline__ void lstm_dgrad_body(const LstmDgradArgs& A, int B, int bx, int by, float (&red)[8][16][64]) {
    const int m0 = by * 32, n0 = bx * 32;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int unit = n0 + r;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    {
        const int kb = w * 16 * KS + hh * 8;     // natural MFMA K order: the two lane halves read ADJACENT 16-byte pieces of a row
        const int arow = min(m0 + r, B - 1);
        bf16x8_t a[KS], b[KS];
        load_frags<KS>(A.dz + (size_t)arow * A.K + kb, a);
        load_frags<KS>(A.wx_p + (size_t)unit * A.K + kb, b);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) red[w][i][lane] = acc[i];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = 2 * w + q;
        const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        float sum = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) sum += red[ww][i][lane];
        if (row >= B) continue;
        const size_t uo = (size_t)row * A.U + unit;
        A.out[uo] = A.mask != nullptr ? sum / A.kp * (float)A.mask[uo] : sum;
    }
}

// Three-stage backward wavefront, lag 2: launch k = layer-2 step T-1-k | layer-1 incoming gradient of step T-k |
// layer-1 step T+1-k.
template <int KS1, int KS2>
__global__ void __launch_bounds__(512) lstm3_bwd_step(LstmBwdArgs A1, LstmDgradArgs Q, LstmBwdArgs A2, int B) {
    __shared__ float red[8][16][64];
    __shared__ bf16_t sT[4][32][40];
    const int nb1 = A1.U / 32, nb2 = A2.U / 32;
    int bx, by;
    if (!fused_tile((B + 31) / 32, bx, by)) return;
    if (bx < nb2) {
        if (A2.active) lstm_bwd_body<KS2>(A2, B, bx, by, red, sT);
    } else if (bx < nb2 + nb1) {
        if (Q.active) lstm_dgrad_body<KS2>(Q, B, bx - nb2, by, red);
    } else {
        if (A1.active) lstm_bwd_body<KS1>(A1, B, bx - nb2 - nb1, by, red, sT);
    }
}

// db_p[c] += sum over columns [c0, c1) of row c of dzT (contiguous bf16 rows): the LSTM bias gradient, one pass over the
// transposed dz the step kernels already wrote (no per-step reduction on the latency-critical chain).
__global__ void __launch_bounds__(256) rowsum_bf16_kernel(const bf16_t* __restrict__ X, int ld, int c0, int c1, float* __restrict__ out) {
    __shared__ float part[4];
    const bf16_t* row = X + (size_t)blockIdx.x * ld;
    float acc = 0.f;
    for (int c = c0 + threadIdx.x; c < c1; c += 256) acc += bf16_to_f32(row[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] += (part[0] + part[1]) + (part[2] + part[3]);
}

extern "C" int mnn_transpose(mnn_stream_t s, const void* in, int in_dtype, int R, int C, int ld_in, void* out, int out_dtype, int ld_out);
extern "C" int mnn_bias_grad(mnn_stream_t s, const float* dY, int rows, int cols, int ld, float* db, int accumulate);

static bool lstm_v2_ok(int dtype, int units) { return dtype == MNN_BF16 && (units == 128 || units == 256 || units == 512); }
extern "C" int mnn_lstm_fused_outputs(int dtype, int units) { return lstm_v2_ok(dtype, units) ? 1 : 0; }

extern "C" int mnn_lstm_seq_fwd(mnn_stream_t s, int dtype, int T, int B, int units, int t_begin, int t_end, const float* xproj,
                                const void* wh_t, const void* h0, const float* c0, float* gates, float* c, void* h, void* hT, int ld_hT) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(dtype == MNN_BF16 || dtype == MNN_F16 || dtype == MNN_F32, "mnn_lstm_seq_fwd: dtype must be bf16, f16 or f32");
    MNN_REQUIRE(T > 0 && B > 0 && units > 0 && units % 32 == 0, "mnn_lstm_seq_fwd: units must be a positive multiple of 32 (T=%d B=%d u=%d)",
                T, B, units);
    MNN_REQUIRE(xproj && wh_t && c && h, "mnn_lstm_seq_fwd: null pointer");
    MNN_REQUIRE(0 <= t_begin && t_begin < t_end && t_end <= T, "mnn_lstm_seq_fwd: bad step range [%d,%d) of %d", t_begin, t_end, T);
    MNN_REQUIRE(hT == nullptr || ld_hT >= T * B, "mnn_lstm_seq_fwd: ld_hT %d < T*B", ld_hT);
    const size_t esz = dtype == MNN_F32 ? 4 : 2;
    dim3 grid(units / 32, cdiv(B, 64));
    const bool v2 = lstm_v2_ok(dtype, units);
    for (int t = t_begin; t < t_end; ++t) {
        const float* xp = xproj + (size_t)t * B * 4 * units;
        float* gt = gates ? gates + (size_t)t * B * 4 * units : nullptr;
        float* ct = c + (size_t)t * B * units;
        const float* cp = t == 0 ? c0 : c + (size_t)(t - 1) * B * units;
        char* ht = (char*)h + (size_t)t * B * units * esz;
        const char* hp = t == 0 ? (const char*)h0 : (const char*)h + (size_t)(t - 1) * B * units * esz;
        if (v2) {
            dim3 g2(units / 32, cdiv(B, 32));
            LstmFwdArgs fa{(const bf16_t*)hp, (const bf16_t*)wh_t, xp, cp, gt, ct, (bf16_t*)ht, (bf16_t*)(t + 1 < T ? hT : nullptr), units, ld_hT,
                           (t + 1) * B, 1};
#define FWD2(KS) hipLaunchKernelGGL(lstm_fwd_step_v2<KS>, g2, dim3(256), 0, st, fa, B)
            if (B >= 512) hipLaunchKernelGGL(lstm_fwd_step_v3, dim3(units / 32, cdiv(B, 64)), dim3(256), 0, st, fa, B);   // GEMM-shaped step
            else if (units == 512) FWD2(8); else if (units == 256) FWD2(4); else FWD2(2);
#undef FWD2
        } else if (dtype == MNN_BF16)
            hipLaunchKernelGGL(lstm_fwd_step_kernel<bf16_t>, grid, dim3(128), 0, st, (const bf16_t*)hp, (const bf16_t*)wh_t, xp, cp, gt, ct,
                               (bf16_t*)ht, B, units);
        else if (dtype == MNN_F16)
            hipLaunchKernelGGL(lstm_fwd_step_kernel<f16_t>, grid, dim3(128), 0, st, (const f16_t*)hp, (const f16_t*)wh_t, xp, cp, gt, ct,
                               (f16_t*)ht, B, units);
        else
            hipLaunchKernelGGL(lstm_fwd_step_kernel<float>, grid, dim3(256), 0, st, (const float*)hp, (const float*)wh_t, xp, cp, gt, ct,
                               (float*)ht, B, units);
    }
    MNN_LAUNCH_CHECK();
    if (hT != nullptr && !v2) {            // generic path: hT[:, (t+1)*B + b] = h[t, b, :] for the steps of this call
        const int rows = (min(t_end, T - 1) - t_begin) * B;     // h[T-1] is nobody's previous state
        if (rows > 0)
            return mnn_transpose(s, (const char*)h + (size_t)t_begin * B * units * esz, dtype, rows, units, units,
                                 (char*)hT + (size_t)(t_begin + 1) * B * esz, dtype, ld_hT);
    }
    return MNN_OK;
}

// ----------------------------------------------------------------------------------------------
// LSTM backward step: dh = dh_ext[t] + dz[t+1] . Wh^T ; pointwise -> dz[t], dc (in place)
// block = 64 rows x BU units (64; f32: 32), 2 waves, each 32 rows x BU units (f32: a second wave pair takes the other half of every K chunk --
// 64 x 64 tiles on two waves put 256 waves on 1024 SIMDs: 109 us per step at B = 1024, U = 512).
// ----------------------------------------------------------------------------------------------
template <typename T> struct StepBU { static constexpr int value = sizeof(T) == 4 ? 32 : 64; };
template <typename T>
__global__ void __launch_bounds__(128 * StepKH<T>::value)
lstm_bwd_step_kernel(const T* __restrict__ dz_next, const T* __restrict__ wh_p, const float* __restrict__ dh_ext,
                     const float* __restrict__ gates, const float* __restrict__ c_t, const float* __restrict__ c_prev,
                     float* __restrict__ dc, float* __restrict__ dz, T* __restrict__ dzT, float* __restrict__ dh_out, int B, int U,
                     int first) {
    constexpr int BU = StepBU<T>::value, TNU = BU / 32;
    using Core = GemmCore<T, 64, BU, 2, 1, StepKH<T>::value>;
    __shared__ __attribute__((aligned(16))) char smem[Core::LDS_BYTES];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * BU;
    const int N4 = 4 * U;
    f32x16_t acc[1][TNU];
#pragma unroll
    for (int j = 0; j < TNU; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
    if (dz_next != nullptr) {
        Core core;
        const int nchunks = (N4 + Elem<T>::PER_CHUNK - 1) / Elem<T>::PER_CHUNK;
        core.run(dz_next, N4, B, m0, wh_p, N4, U, n0, N4, 0, nchunks, smem, acc);
        if (!core.reduce_kh(smem, acc)) return;
    } else if (threadIdx.x >= 128) return;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 1;
#pragma unroll
    for (int j = 0; j < TNU; ++j) {
        const int unit = n0 + j * 32 + (lane & 31);
        if (unit >= U) continue;
        const int pc = gate_perm_col(0, unit);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wave * 32 + frag_row(r, lane);
            if (row >= B) continue;
            const size_t uo = (size_t)row * U + unit;
            const size_t zo = (size_t)row * N4 + pc;
            if (dh_out != nullptr) {          // pseudo-step "t = -1": only the recurrent gradient into h0
                dh_out[uo] = acc[0][j][r];
                continue;
            }
            const float dh = dh_ext[uo] + acc[0][j][r];
            const float gi = gates[zo], gg = gates[zo + 32], gf = gates[zo + 64], go = gates[zo + 96];
            const float c = c_t[uo];
            const float cp = c_prev != nullptr ? c_prev[uo] : 0.f;
            const float tc = fast_tanh(c);
            const float dcn = first ? 0.f : dc[uo];
            const float d_o = dh * tc;
            const float d_c = dh * go * (1.f - tc * tc) + dcn;
            const float dzi = d_c * gg * gi * (1.f - gi);
            const float dzg = d_c * gi * (1.f - gg * gg);
            const float dzf = d_c * cp * gf * (1.f - gf);
            const float dzo = d_o * go * (1.f - go);
            dc[uo] = d_c * gf;
            if (dz != nullptr) { dz[zo] = dzi; dz[zo + 32] = dzg; dz[zo + 64] = dzf; dz[zo + 96] = dzo; }
            if ((void*)dzT != (void*)dz) {
                dzT[zo] = Cvt<T>::store(dzi); dzT[zo + 32] = Cvt<T>::store(dzg);
                dzT[zo + 64] = Cvt<T>::store(dzf); dzT[zo + 96] = Cvt<T>::store(dzo);
            }
        }
    }
}

extern "C" size_t mnn_lstm_seq_bwd_workspace_bytes(int B, int units) { return (size_t)B * units * sizeof(float); }

extern "C" int mnn_lstm_seq_bwd(mnn_stream_t s, int dtype, int T, int B, int units, int t_begin, int t_end, const float* dh_ext,
                                const void* wh_p, const float* gates, const float* c, const float* c0, float* dz, void* dz_T, float* dh0,
                                float* dc0, void* workspace, void* dzT_t, int ld_t, float* db_p) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(dtype == MNN_BF16 || dtype == MNN_F16 || dtype == MNN_F32, "mnn_lstm_seq_bwd: dtype must be bf16, f16 or f32");
    MNN_REQUIRE(T > 0 && B > 0 && units > 0 && units % 32 == 0, "mnn_lstm_seq_bwd: units must be a positive multiple of 32");
    MNN_REQUIRE(dh_ext && wh_p && gates && c && workspace, "mnn_lstm_seq_bwd: null pointer");
    MNN_REQUIRE(dtype == MNN_F32 ? (dz != nullptr) : (dz_T != nullptr), "mnn_lstm_seq_bwd: dz (f32) / dz_T (bf16) output required");
    MNN_REQUIRE(0 <= t_begin && t_begin < t_end && t_end <= T, "mnn_lstm_seq_bwd: bad step range [%d,%d) of %d", t_begin, t_end, T);
    MNN_REQUIRE(dzT_t == nullptr || ld_t >= T * B, "mnn_lstm_seq_bwd: ld_t %d < T*B", ld_t);
    const bool v2 = lstm_v2_ok(dtype, units);
    MNN_REQUIRE(v2 || db_p == nullptr || dz != nullptr, "mnn_lstm_seq_bwd: the generic path needs the f32 dz for the bias gradient");
    if (dtype == MNN_F32) dz_T = dz;
    const size_t esz = dtype == MNN_F32 ? 4 : 2;
    float* dc = (float*)workspace;        // carried d c between calls: process [t_begin,t_end) from the top range downwards
    dim3 grid(cdiv(units, 64), cdiv(B, 64));
    const size_t zs = (size_t)B * 4 * units, us = (size_t)B * units;
    for (int t = t_end - 1; t >= t_begin - 1; --t) {
        if (t < t_begin && !(t_begin == 0 && dh0 != nullptr)) break;
        const char* dzn = t == T - 1 ? nullptr : (const char*)dz_T + (size_t)(t + 1) * zs * esz;
        const int tt = t < 0 ? 0 : t;
        const float* cp = t <= 0 ? c0 : c + (size_t)(t - 1) * us;
        float* dzt = dz ? dz + (size_t)tt * zs : nullptr;
        char* dzTt = (char*)dz_T + (size_t)tt * zs * esz;
        float* dho = t < 0 ? dh0 : nullptr;
        if (v2 && t >= 0) {
            dim3 g2(units / 32, cdiv(B, 32));
            LstmBwdArgs ba{(const bf16_t*)dzn, (const bf16_t*)wh_p, dh_ext + (size_t)tt * us, gates + (size_t)tt * zs, c + (size_t)tt * us, cp, dc,
                           dzt, (bf16_t*)dzTt, (bf16_t*)dzT_t, units, t == T - 1 ? 1 : 0, ld_t, t * B, 1};
#define BWD2(KS) hipLaunchKernelGGL(lstm_bwd_step_v2<KS>, g2, dim3(512), 0, st, ba, B)
            // (a GEMM-shaped 64x64-tile variant for B >= 512 measured SLOWER: 25 vs 17 us/step at B = 1024 -- K = 4U makes a
            //  32-tile barrier loop on half the CUs; see profiles/round1_d_notes.md)
            if (units == 512) BWD2(16); else if (units == 256) BWD2(8); else BWD2(4);
#undef BWD2
        } else if (dtype == MNN_BF16)
            hipLaunchKernelGGL(lstm_bwd_step_kernel<bf16_t>, grid, dim3(128), 0, st, (const bf16_t*)dzn, (const bf16_t*)wh_p,
                               dh_ext + (size_t)tt * us, gates + (size_t)tt * zs, c + (size_t)tt * us, cp, dc, dzt, (bf16_t*)dzTt, dho, B,
                               units, t == T - 1 ? 1 : 0);
        else if (dtype == MNN_F16)
            hipLaunchKernelGGL(lstm_bwd_step_kernel<f16_t>, grid, dim3(128), 0, st, (const f16_t*)dzn, (const f16_t*)wh_p,
                               dh_ext + (size_t)tt * us, gates + (size_t)tt * zs, c + (size_t)tt * us, cp, dc, dzt, (f16_t*)dzTt, dho, B,
                               units, t == T - 1 ? 1 : 0);
        else
            hipLaunchKernelGGL(lstm_bwd_step_kernel<float>, dim3(cdiv(units, 32), cdiv(B, 64)), dim3(256), 0, st, (const float*)dzn, (const float*)wh_p,
                               dh_ext + (size_t)tt * us, gates + (size_t)tt * zs, c + (size_t)tt * us, cp, dc, dzt, (float*)dzTt, dho, B,
                               units, t == T - 1 ? 1 : 0);
    }
    MNN_LAUNCH_CHECK();
    if (t_begin == 0 && dc0 != nullptr) MNN_HIP(mnn_copy_async(dc0, dc, us * sizeof(float), st));
    if (v2 && db_p != nullptr) {
        MNN_REQUIRE(dzT_t != nullptr, "mnn_lstm_seq_bwd: db_p needs dzT_t on the fused path");
        if (t_begin == 0)      // calls cover the sequence from the top down: the t_begin == 0 call is the last one
            hipLaunchKernelGGL(rowsum_bf16_kernel, dim3(4 * units), dim3(256), 0, st, (const bf16_t*)dzT_t, ld_t, 0, T * B, db_p);
        MNN_LAUNCH_CHECK();
    }
    if (!v2) {                             // generic path: separate transpose / column-sum kernels for the steps of this call
        const int rows = (t_end - t_begin) * B;
        if (dzT_t != nullptr) {
            int rc = mnn_transpose(s, (const char*)dz_T + (size_t)t_begin * zs * esz, dtype, rows, 4 * units, 4 * units,
                                   (char*)dzT_t + (size_t)t_begin * B * esz, dtype, ld_t);
            if (rc != MNN_OK) return rc;
        }
        if (db_p != nullptr) return mnn_bias_grad(s, dz + (size_t)t_begin * zs, rows, 4 * units, 4 * units, db_p, 1);
    }
    return MNN_OK;
}

// ----------------------------------------------------------------------------------------------
// Two-layer wavefront entries (bf16, units in {128,256,512}), three stages per launch, lag 2:
//   forward  launch s: layer-1 step s | layer-2 input projection of step s-1 | layer-2 step s-2      s in [0, T+2)
//   backward launch k: layer-2 step T-1-k | layer-1 incoming gradient of step T-k | layer-1 step T+1-k
// One launch per timestep for the whole stack: half the kernel boundaries of the per-layer form, the second chain
// hidden behind the first, no per-chunk host work.
// ----------------------------------------------------------------------------------------------
template <int K1>
static void launch_fwd3(hipStream_t st, dim3 grid, const LstmFwdArgs& a1, const LstmProjArgs& p, const LstmFwdArgs& a2, int B, int u2) {
    if (u2 == 512) hipLaunchKernelGGL((lstm3_fwd_step<K1, 8>), grid, dim3(256), 0, st, a1, p, a2, B);
    else if (u2 == 256) hipLaunchKernelGGL((lstm3_fwd_step<K1, 4>), grid, dim3(256), 0, st, a1, p, a2, B);
    else hipLaunchKernelGGL((lstm3_fwd_step<K1, 2>), grid, dim3(256), 0, st, a1, p, a2, B);
}
template <int K1>
static void launch_bwd3(hipStream_t st, dim3 grid, const LstmBwdArgs& a1, const LstmDgradArgs& q, const LstmBwdArgs& a2, int B, int u2) {
    if (u2 == 512) hipLaunchKernelGGL((lstm3_bwd_step<K1, 16>), grid, dim3(512), 0, st, a1, q, a2, B);
    else if (u2 == 256) hipLaunchKernelGGL((lstm3_bwd_step<K1, 8>), grid, dim3(512), 0, st, a1, q, a2, B);
    else hipLaunchKernelGGL((lstm3_bwd_step<K1, 4>), grid, dim3(512), 0, st, a1, q, a2, B);
}

static LstmFwdArgs make_fwd_args(const mnn_lstm_fwd_layer* L, int T, int B, int t, float kp) {
    LstmFwdArgs a{};
    a.U = L->units;
    a.active = (t >= 0 && t < T) ? 1 : 0;
    if (!a.active) return a;
    const size_t us = (size_t)B * L->units, zs = 4 * us;
    a.h_prev = t == 0 ? (const bf16_t*)L->h0 : (const bf16_t*)L->h + (size_t)(t - 1) * us;
    a.wh_t = (const bf16_t*)L->wh_t;
    a.xproj = L->xproj + (size_t)t * zs;
    a.c_prev = t == 0 ? L->c0 : L->c + (size_t)(t - 1) * us;
    a.gates = L->gates ? L->gates + (size_t)t * zs : nullptr;
    a.c_out = L->c + (size_t)t * us;
    a.h_out = (bf16_t*)L->h + (size_t)t * us;
    a.hT = (t + 1 < T) ? (bf16_t*)L->hT : nullptr;
    a.ld_hT = L->ld_hT;
    a.colT = (t + 1) * B;
    if (L->mask != nullptr) {
        a.y_out = (bf16_t*)L->y + (size_t)t * us;
        a.mask = L->mask + (size_t)t * us;
        a.kp = kp;
    }
    return a;
}

extern "C" int mnn_lstm2_seq_fwd(mnn_stream_t s, int T, int B, const mnn_lstm_fwd_layer* L1, const mnn_lstm_fwd_layer* L2, float keep_prob,
                                 int s_begin, int s_end) {
    MNN_REQUIRE(L1 && L2 && !L1->xproj_bf16 && !L2->xproj_bf16, "this form reads f32 input projections (xproj_bf16 is for mnn_lstm_rowpar_fwd)");
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(L1 && L2 && T > 0 && B > 0 && keep_prob > 0.f, "mnn_lstm2_seq_fwd: bad arguments");
    MNN_REQUIRE(lstm_v2_ok(MNN_BF16, L1->units) && lstm_v2_ok(MNN_BF16, L2->units), "mnn_lstm2_seq_fwd: units must be 128/256/512 (bf16)");
    MNN_REQUIRE(!L1->f16 && !L2->f16, "mnn_lstm2_seq_fwd: bf16 only (f16 layers run mnn_lstm2_persist_* / mnn_lstm_rowpar_* / mnn_lstm_seq_*)");
    MNN_REQUIRE(0 <= s_begin && s_begin < s_end && s_end <= T + 2, "mnn_lstm2_seq_fwd: bad launch range [%d,%d) of %d", s_begin, s_end, T + 2);
    for (const mnn_lstm_fwd_layer* L : {L1, L2}) {
        MNN_REQUIRE(L->xproj && L->wh_t && L->c && L->h, "mnn_lstm2_seq_fwd: null pointer");
        MNN_REQUIRE(L->hT == nullptr || L->ld_hT >= T * B, "mnn_lstm2_seq_fwd: ld_hT too small");
        MNN_REQUIRE((L->mask == nullptr) == (keep_prob >= 1.0f) && (L->mask == nullptr || L->y != nullptr),
                    "mnn_lstm2_seq_fwd: a keep mask and a y buffer are needed exactly when keep_prob < 1");
    }
    MNN_REQUIRE(L2->wx_t && L2->bias_p && L2->ld_w >= L1->units, "mnn_lstm2_seq_fwd: layer 2 needs its input-projection weights");
    const size_t us1 = (size_t)B * L1->units;
    const int ut_f = L1->units / 32 + 2 * (L2->units / 32);
    dim3 grid = MNN_XCD_ROWS ? dim3(8 * cdiv(cdiv(B, 32), 8) * ut_f) : dim3(ut_f, cdiv(B, 32));
    for (int si = s_begin; si < s_end; ++si) {
        const LstmFwdArgs a1 = make_fwd_args(L1, T, B, si, keep_prob), a2 = make_fwd_args(L2, T, B, si - 2, keep_prob);
        LstmProjArgs p{};
        const int tp = si - 1;
        p.active = (tp >= 0 && tp < T) ? 1 : 0;
        if (p.active) {
            p.y = (const bf16_t*)(L1->mask ? L1->y : L1->h) + (size_t)tp * us1;
            p.wx_t = (const bf16_t*)L2->wx_t;
            p.bias_p = L2->bias_p;
            p.out = const_cast<float*>(L2->xproj) + (size_t)tp * B * 4 * L2->units;
            p.K = L1->units; p.ld_w = L2->ld_w; p.N4 = 4 * L2->units;
        }
        if (L1->units == 512) launch_fwd3<8>(st, grid, a1, p, a2, B, L2->units);
        else if (L1->units == 256) launch_fwd3<4>(st, grid, a1, p, a2, B, L2->units);
        else launch_fwd3<2>(st, grid, a1, p, a2, B, L2->units);
    }
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

static LstmBwdArgs make_bwd_args(const mnn_lstm_bwd_layer* L, int T, int B, int t) {
    LstmBwdArgs a{};
    a.U = L->units;
    a.active = (t >= 0 && t < T) ? 1 : 0;
    if (!a.active) return a;
    const size_t us = (size_t)B * L->units, zs = 4 * us;
    a.dz_next = t == T - 1 ? nullptr : (const bf16_t*)L->dz_T + (size_t)(t + 1) * zs;
    a.wh_p = (const bf16_t*)L->wh_p;
    a.dh_ext = L->dh_ext + (size_t)t * us;
    a.gates = L->gates + (size_t)t * zs;
    a.c_t = L->c + (size_t)t * us;
    a.c_prev = t == 0 ? L->c0 : L->c + (size_t)(t - 1) * us;
    a.dc = (float*)L->workspace;
    a.dz = L->dz ? L->dz + (size_t)t * zs : nullptr;
    a.dzT = (bf16_t*)L->dz_T + (size_t)t * zs;
    a.dzTt = (bf16_t*)L->dzT_t;
    a.first = t == T - 1 ? 1 : 0;
    a.ld_t = L->ld_t;
    a.colT = t * B;
    return a;
}

extern "C" int mnn_lstm2_seq_bwd(mnn_stream_t s, int T, int B, const mnn_lstm_bwd_layer* L1, const mnn_lstm_bwd_layer* L2, float keep_prob,
                                 int k_begin, int k_end) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(L1 && L2 && T > 0 && B > 0 && keep_prob > 0.f, "mnn_lstm2_seq_bwd: bad arguments");
    MNN_REQUIRE(lstm_v2_ok(MNN_BF16, L1->units) && lstm_v2_ok(MNN_BF16, L2->units), "mnn_lstm2_seq_bwd: units must be 128/256/512 (bf16)");
    MNN_REQUIRE(!L1->f16 && !L2->f16, "mnn_lstm2_seq_bwd: bf16 only");
    MNN_REQUIRE(0 <= k_begin && k_begin < k_end && k_end <= T + 2, "mnn_lstm2_seq_bwd: bad launch range [%d,%d) of %d", k_begin, k_end, T + 2);
    for (const mnn_lstm_bwd_layer* L : {L1, L2}) {
        MNN_REQUIRE(L->dh_ext && L->wh_p && L->gates && L->c && L->dz_T && L->workspace, "mnn_lstm2_seq_bwd: null pointer");
        MNN_REQUIRE(L->dzT_t == nullptr || L->ld_t >= T * B, "mnn_lstm2_seq_bwd: ld_t too small");
        MNN_REQUIRE(L->db_p == nullptr || L->dzT_t != nullptr, "mnn_lstm2_seq_bwd: db_p needs dzT_t");
    }
    MNN_REQUIRE(L2->wx_p != nullptr, "mnn_lstm2_seq_bwd: layer 2 needs wx_p (its input weights, [u1, 4u2])");
    MNN_REQUIRE((L1->mask == nullptr) == (keep_prob >= 1.0f), "mnn_lstm2_seq_bwd: layer 1's keep mask is needed exactly when keep_prob < 1");
    const size_t us1 = (size_t)B * L1->units;
    const int ut_b = 2 * (L1->units / 32) + L2->units / 32;
    dim3 grid = MNN_XCD_ROWS ? dim3(8 * cdiv(cdiv(B, 32), 8) * ut_b) : dim3(ut_b, cdiv(B, 32));
    for (int k = k_begin; k < k_end; ++k) {
        const LstmBwdArgs a2 = make_bwd_args(L2, T, B, T - 1 - k), a1 = make_bwd_args(L1, T, B, T + 1 - k);
        LstmDgradArgs q{};
        const int tq = T - k;
        q.active = (tq >= 0 && tq < T) ? 1 : 0;
        if (q.active) {
            q.dz = (const bf16_t*)L2->dz_T + (size_t)tq * B * 4 * L2->units;
            q.wx_p = (const bf16_t*)L2->wx_p;
            q.mask = L1->mask ? L1->mask + (size_t)tq * us1 : nullptr;
            q.out = const_cast<float*>(L1->dh_ext) + (size_t)tq * us1;
            q.K = 4 * L2->units; q.U = L1->units; q.kp = keep_prob;
        }
        if (L1->units == 512) launch_bwd3<16>(st, grid, a1, q, a2, B, L2->units);
        else if (L1->units == 256) launch_bwd3<8>(st, grid, a1, q, a2, B, L2->units);
        else launch_bwd3<4>(st, grid, a1, q, a2, B, L2->units);
    }
    MNN_LAUNCH_CHECK();
    // bias gradients: ONE row-sum pass over dz^T per layer, when that layer has finished its last (t = 0) step
    if (L2->db_p && k_begin < T && k_end >= T)
        hipLaunchKernelGGL(rowsum_bf16_kernel, dim3(4 * L2->units), dim3(256), 0, st, (const bf16_t*)L2->dzT_t, L2->ld_t, 0, T * B, L2->db_p);
    if (L1->db_p && k_end == T + 2)
        hipLaunchKernelGGL(rowsum_bf16_kernel, dim3(4 * L1->units), dim3(256), 0, st, (const bf16_t*)L1->dzT_t, L1->ld_t, 0, T * B, L1->db_p);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

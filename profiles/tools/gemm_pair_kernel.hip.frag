// gemm_tn_pair_kernel -- the "pair" form of mnn_gemm_tn's short-K activation GEMMs (two de-phased 4-wave workgroups of 256 x 128 per CU), REMOVED
// from the product library in round 6: it was built and measured in round 5 (profiles/round5_a_gemm_pair_notes.md: -2..-5 % on xproj1, +3..+17 %
// elsewhere), never became the default on a BASELINE shape, and its six development variants (MNN_GEMM_PAIR_VAR) plus four getenv calls per GEMM
// stayed compiled into libmultinn_hip.so.  Kept here for the record.  This is a FRAGMENT of multinn_amd/csrc/gemm.hip as of the round-5 tree: it
// uses that file's GldsSlot / mfma fragments / GM_T macros; to rebuild it, paste it back in front of `launch_gemm` of that revision
// (git show 7346180:multinn_amd/csrc/gemm.hip has the dispatch code, MNN_GEMM_PAIR = 0 | 1 | 2).

// Epilogues of the pair kernel.  The wave's 4 x 2 accumulator tiles are TRANSPOSED (lane = row of C, an accumulator quad = 4 consecutive
// columns): acc[i][j][4 q + t] = C[row0 + 32 i + r][colw + 32 j + 8 q + 4 h + t], r = lane & 31, h = lane >> 5.
// Every bias quad is loaded FIRST (one wait; a load between the stores would make its vmcnt wait cover the stores in front of it too).
// EDGE: per-lane row / column predicates (N is a multiple of the store width, so a piece is inside or outside as a whole).
// NOSTORE (development, ablation builds): everything but the store instruction.
template <bool NOSTORE>
__device__ __forceinline__ void pair_store16(void* p, uint4 v) {
    if (!NOSTORE) *reinterpret_cast<uint4*>(p) = v;
    else asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
}
// (1) straight from the registers: f32 C = one 16-byte store per quad; 16-bit C = two quads joined across the lane halves by
// v_permlane32_swap into one 16-byte store (8 columns).  A store instruction covers 32 rows x 32 contiguous bytes.
template <int C16, bool EDGE, bool NOSTORE>
__device__ __forceinline__ void pair_epilogue(const f32x16_t (&acc)[4][2], void* __restrict__ Cv, int ldc, const float* __restrict__ bias, int M, int N,
                                              int row, int colw, int h) {
    if (C16 == 0) {
        float* Cf = reinterpret_cast<float*>(Cv);
        float4 bv[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = colw + j * 32 + 8 * q + 4 * h;
                bv[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bias != nullptr) bv[j][q] = *reinterpret_cast<const float4*>(bias + (EDGE ? min(col, N - 4) : col));      // (an outside quad is never stored)
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rw = row + i * 32;
            float* crow = Cf + (size_t)rw * ldc;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = colw + j * 32 + 8 * q + 4 * h;
                    float4 v;
                    v.x = acc[i][j][4 * q + 0] + bv[j][q].x; v.y = acc[i][j][4 * q + 1] + bv[j][q].y;
                    v.z = acc[i][j][4 * q + 2] + bv[j][q].z; v.w = acc[i][j][4 * q + 3] + bv[j][q].w;
                    if (!EDGE || (col < N && rw < M)) pair_store16<NOSTORE>(crow + col, make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)));
                }
        }
    } else {
        using CF = typename std::conditional<C16 == MNN_F16, Fp16F, Bf16F>::type;
        h16_t* Cb = reinterpret_cast<h16_t*>(Cv);
        float4 b0[2][2], b1[2][2];                                 // the bias of a lane's OWN quads 2 qq and 2 qq + 1 (before the exchange)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int c0 = colw + j * 32 + 16 * qq + 4 * h;
                b0[j][qq] = make_float4(0.f, 0.f, 0.f, 0.f);
                b1[j][qq] = b0[j][qq];
                if (bias != nullptr) {
                    b0[j][qq] = *reinterpret_cast<const float4*>(bias + (EDGE ? min(c0, N - 4) : c0));
                    b1[j][qq] = *reinterpret_cast<const float4*>(bias + (EDGE ? min(c0 + 8, N - 4) : c0 + 8));
                }
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rw = row + i * 32;
            h16_t* crow = Cb + (size_t)rw * ldc;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {                   // quads 2 qq (-> lanes 0..31) and 2 qq + 1 (-> lanes 32..63) become one 8-column piece each
                    const int col = colw + j * 32 + 16 * qq + 8 * h;
                    const int e0 = 8 * qq, e1 = 8 * qq + 4;
                    const float4 p0 = b0[j][qq], p1 = b1[j][qq];
                    const uint32_t x0 = pack2<CF>(acc[i][j][e0 + 0] + p0.x, acc[i][j][e0 + 1] + p0.y), x1 = pack2<CF>(acc[i][j][e0 + 2] + p0.z, acc[i][j][e0 + 3] + p0.w);
                    const uint32_t y0 = pack2<CF>(acc[i][j][e1 + 0] + p1.x, acc[i][j][e1 + 1] + p1.y), y1 = pack2<CF>(acc[i][j][e1 + 2] + p1.z, acc[i][j][e1 + 3] + p1.w);
                    // lanes 0..31 keep X (their quad 2 qq) and receive the upper lanes' X; lanes 32..63 receive the lower lanes' Y and keep theirs
                    const auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                    if (!EDGE || (col < N && rw < M)) pair_store16<NOSTORE>(crow + col, make_uint4(s0[0], s1[0], s0[1], s1[1]));
                }
        }
    }
}
// (2) whole lines: 32 rows x 64 columns of the wave tile at a time take a turn through a wave-private LDS tile (the ring slots are idle
// behind the K loop) in the TYPE of C, written in the accumulator layout (a quad = 8 / 16 contiguous bytes of a row) and read back
// row-contiguous: a store instruction covers 8 rows x 128 bytes (16-bit C) or 4 rows x 256 bytes (f32 C) -- full cache lines instead of
// 32-byte pieces four instructions apart (stores alone, xproj1 [262144 x 2048] f16: 3.3 TB/s with the pieces).
#define PAIR_SC_BYTES 8704          // per wave: 32 rows x 272 bytes (f32: 256 + 16 pad; 16-bit: 144-byte rows = 128 + 16 pad)
template <int C16, bool EDGE, bool NOSTORE>
__device__ __forceinline__ void pair_epilogue_lines(const f32x16_t (&acc)[4][2], void* __restrict__ Cv, int ldc, const float* __restrict__ bias, int M,
                                                    int N, int row0, int colw, int lane, char* __restrict__ sc) {
    const int r = lane & 31, h = lane >> 5;
    float4 bv[2][4];
    if (bias != nullptr) {                                      // ONE uniform branch around all eight loads (a branch per load waits per load)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = colw + j * 32 + 8 * q + 4 * h;
                bv[j][q] = *reinterpret_cast<const float4*>(bias + (EDGE ? min(col, N - 4) : col));      // (an outside quad is never stored)
            }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (C16 == 0) {
        constexpr int PITCH = 272;
        float* Cf = reinterpret_cast<float*>(Cv);
        const int rp = lane >> 4, piece = lane & 15;               // read-back: 4 rows x 16 pieces of 16 bytes per instruction
        const int col = colw + piece * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v;
                    v.x = acc[i][j][4 * q + 0] + bv[j][q].x; v.y = acc[i][j][4 * q + 1] + bv[j][q].y;
                    v.z = acc[i][j][4 * q + 2] + bv[j][q].z; v.w = acc[i][j][4 * q + 3] + bv[j][q].w;
                    *reinterpret_cast<float4*>(sc + r * PITCH + (j * 32 + 8 * q + 4 * h) * 4) = v;
                }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int rl = p * 4 + rp, rw = row0 + i * 32 + rl;
                const uint4 v = *reinterpret_cast<const uint4*>(sc + rl * PITCH + piece * 16);
                if (!EDGE || (col < N && rw < M)) pair_store16<NOSTORE>(Cf + (size_t)rw * ldc + col, v);
            }
        }
    } else {
        using CF = typename std::conditional<C16 == MNN_F16, Fp16F, Bf16F>::type;
        constexpr int PITCH = 144;
        h16_t* Cb = reinterpret_cast<h16_t*>(Cv);
        const int rp = lane >> 3, piece = lane & 7;                // read-back: 8 rows x 8 pieces of 16 bytes per instruction
        const int col = colw + piece * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint2 pk;
                    pk.x = pack2<CF>(acc[i][j][4 * q + 0] + bv[j][q].x, acc[i][j][4 * q + 1] + bv[j][q].y);
                    pk.y = pack2<CF>(acc[i][j][4 * q + 2] + bv[j][q].z, acc[i][j][4 * q + 3] + bv[j][q].w);
                    *reinterpret_cast<uint2*>(sc + r * PITCH + (j * 32 + 8 * q + 4 * h) * 2) = pk;
                }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int rl = p * 8 + rp, rw = row0 + i * 32 + rl;
                const uint4 v = *reinterpret_cast<const uint4*>(sc + rl * PITCH + piece * 16);
                if (!EDGE || (col < N && rw < M)) pair_store16<NOSTORE>(Cb + (size_t)rw * ldc + col, v);
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Short-K form ("pair" kernel): the step's activation GEMMs have M = B*T rows and K = 256 .. 1024, so a 256 x 256 tile is 4 .. 16 K tiles of
// matrix-core work between a prologue (first operand round trip) and an epilogue (128 .. 256 KB of C through the CU's store path) that its
// eight waves run in lockstep: 8 us of MFMA inside a 24 us tile (profiles/round4_e_resident_recurrence_notes.md).  Here a workgroup is FOUR
// waves on a 256 x 128 tile (the same 128 x 64 wave tile: 4 A + 2 B fragments feed 8 MFMAs) and holds 72 KiB of LDS -- three 24 KiB slots of
// a 32-deep K step -- so that TWO workgroups share a CU, one wave of each per SIMD.  The hardware interleaves them: one workgroup's
// prologue, LDS-DMA issue and C stores run under the other's MFMAs.  Inside a workgroup the LDS-DMA of K step s + 2 is issued behind ONE raw
// barrier per step and waited for with a counted vmcnt (a __syncthreads() would drain it: cdna_hip_programming.md "Pipelining across
// barriers"); all 12 fragment reads of a step are requested before its first MFMA, so the second half's reads land under the first half's MFMAs.
// The MFMA operands are SWAPPED (weights as the A operand): a lane holds 4 consecutive columns of ONE row of C per accumulator quad, which
// is what both epilogues above start from.  Measurement trail: profiles/round5_a_gemm_pair_notes.md.
// ----------------------------------------------------------------------------------------------
// EPI: 1 = C straight from the registers, 2 = whole lines through LDS.
// VAR (development, MNN_GEMM_PAIR_VAR): 0 = product; 3 = no stores; 4 = no LDS-DMA in the loop; 5 = no MFMA; 6 = neither MFMA nor stores; 7 = stores only; 8 = MFMA only
template <typename F, int C16, int EPI = 2, int VAR = 0>      // C16: 0 = f32 C, else the mnn_dtype code of the 16-bit C
__global__ void __launch_bounds__(256, 2)
gemm_tn_pair_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, void* __restrict__ Cv, int ldc,
                    const float* __restrict__ bias, int M, int N, int K, int ntm, int ntn) {
    constexpr int BK = 32, CPR = 4, ROWB = 64, A_T = 256 * ROWB, SLOT = (256 + 128) * ROWB;
    constexpr bool NODMA = VAR == 4 || VAR == 7 || VAR == 8, NOMFMA = VAR == 5 || VAR == 6 || VAR == 7, NOSTORE = VAR == 3 || VAR == 6 || VAR == 8;
    extern __shared__ __attribute__((aligned(16))) char smem_pair[];       // [3 slots][A 256 rows x 64 B | B 128 rows x 64 B]
    const int bid = blockIdx.x;
    const int grp = bid / (8 * ntn), within = bid % (8 * ntn);
    const int mt = grp * 8 + (within & 7), nt = within >> 3;                // the column tiles of one row panel run on one XCD, back to back
    if (mt >= ntm) return;
    const int m0 = mt * 256, n0 = nt * 128;
    const int nks = K / BK;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    GldsSlots<256, 4, CPR> slotA;
    GldsSlots<128, 4, CPR> slotB;
    slotA.init(A, lda, M, m0, wave, lane);
    slotB.init(B, ldb, N, n0, wave, lane);
    auto stage = [&](int slot, int s) {
        slotA.stage(s * BK, smem_pair + slot * SLOT, wave);
        slotB.stage(s * BK, smem_pair + slot * SLOT + A_T, wave);
    };
    GM_T0();
    stage(0, 0);
    if (nks > 1) stage(1, 1);
    GM_T(1);
    // per-lane read offsets: row (.. + r) * 64 + ((2 ks + h) ^ swz(r)) * 16; i * 32 rows and the wave's base are multiples of 16 rows (swz unchanged)
    const int sw = (r >> 2) & 3;
    const int offA = (wm * 128 + r) * ROWB, offB = A_T + (wn * 64 + r) * ROWB;
    const int kx0 = ((0 + h) ^ sw) << 4, kx1 = ((2 + h) ^ sw) << 4;
    int cur = 0;
    for (int s = 0; s < nks; ++s) {
        // this wave's pieces of step s have landed (the 6 of step s + 1 may still fly); behind the barrier every wave's have, and every wave
        // has finished reading the slot of step s - 1, which step s + 2 overwrites
        if (NODMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (s + 1 < nks) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GM_T(2);
        __builtin_amdgcn_s_barrier();
        GM_T(3);
        if (!NODMA && s + 2 < nks) stage(cur >= 1 ? cur - 1 : 2, s + 2);
        GM_T(4);
        const char* sl = smem_pair + cur * SLOT;
        // The twelve fragment reads of the step are inline assembly with hand-counted waits: hipcc waits lgkmcnt(0) in front of the first MFMA
        // of every group in this loop (also in the 256 x 256 kernel), i.e. for ALL reads in flight.  Here the k-half 1 reads land under
        // the MFMAs of k-half 0.  (LDS returns a wave's reads in order; sched_barrier: hipcc moves MFMAs across an asm wait otherwise.)
        typename F::x8 a[2][4], b[2][2];
        const uint32_t la = (uint32_t)(uintptr_t)(sl + offA), lb = (uint32_t)(uintptr_t)(sl + offB);      // LDS byte addresses (the low 32 bits of a generic LDS pointer)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint32_t kx = ks ? kx1 : kx0;
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[ks][j]) : "v"(lb + kx), "n"(j * 32 * ROWB));
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[ks][i]) : "v"(la + kx), "n"(i * 32 * ROWB));
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // a[ks][i] is read number 6 ks + 2 + i of the twelve: wait until only the younger ones are in flight
                if (ks == 0 && i == 0) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
                else if (ks == 0 && i == 1) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                else if (ks == 0 && i == 2) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                else if (ks == 0 && i == 3) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                else if (ks == 1 && i == 0) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                else if (ks == 1 && i == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                else if (ks == 1 && i == 2) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (NOMFMA) asm volatile("" :: "v"(a[ks][i]), "v"(b[ks][j]));
                    else acc[i][j] = F::mfma32(b[ks][j], a[ks][i], acc[i][j]);       // swapped: the tile is C^T, lane = row of C
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        GM_T(5);
        cur = cur == 2 ? 0 : cur + 1;
    }
    const int colw = n0 + wn * 64;
    const bool interior = m0 + 256 <= M && colw + 64 <= N;     // uniform: an interior wave tile has no predicates
    if (EPI == 1) {
        const int row = m0 + wm * 128 + r;
        if (interior) pair_epilogue<C16, false, NOSTORE>(acc, Cv, ldc, bias, M, N, row, colw, h);
        else pair_epilogue<C16, true, NOSTORE>(acc, Cv, ldc, bias, M, N, row, colw, h);
    } else {
        __builtin_amdgcn_s_barrier();                            // every wave is done with the last slot (no LDS-DMA is pending: vmcnt(0) above)
        GM_T(6);
        char* sc = smem_pair + wave * PAIR_SC_BYTES;
        if (interior) pair_epilogue_lines<C16, false, NOSTORE>(acc, Cv, ldc, bias, M, N, m0 + wm * 128, colw, lane, sc);
        else pair_epilogue_lines<C16, true, NOSTORE>(acc, Cv, ldc, bias, M, N, m0 + wm * 128, colw, lane, sc);
    }
    GM_T(7);
#ifdef GM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GM_T(8);
#endif
}

// NOT part of the library: the two-half-tile, flag-less (tagged granule) form of the cluster backward, measured in round 5 and not shipped
// (profiles/round5_g_cluster_notes.md: 6.8 us per timestep against 5.1 for the one-tile form at [1024, 256], first layer).  It was built inside
// multinn_amd/csrc/lstm_cluster.hip (same file-level helpers: cl_probe_xcd, cl_ld / cl_st, CL_* macros, ClBwdArgs with the extra fields
// `int ablate; int nonce;`), selected by MNN_CLUSTER_PIPE, with the workspace of mnn_lstm_rowpar_workspace_bytes (1 MB per cluster).
// Host side: a.nonce = a per-launch counter & 0xfffff; T >= 4; launch with ClBwd2Geom::LDS of dynamic LDS.

// ------------------------------------------------------------------------------------------------------------------
// backward, two half tiles in a software pipeline, hand-offs WITHOUT flags.  The kernel above runs  MFMAs -> partial stores -> wait for them
// -> flag -> poll -> partial loads -> pointwise  strictly one after the other: three memory round trips (~0.6 us each) and the cluster's skew
// sit on every step (stage clocks: profiles/round5_g_cluster_notes.md).  Here
//   * the cluster's 32 rows are two HALVES of 16 (v_mfma_f32_16x16x32: A = 16 units of Wh x 32 of the member's gate columns, B = those columns
//     of dz[t+1] for the half's 16 rows; 8 unit tiles x 8 k-steps per wave and half) that run half a step apart:
//         products A(k) | partial sums + pointwise B(k-1) | barrier | products B(k) | partial sums + pointwise A(k) | barrier
//   * a partial-sum granule carries its own step number: 16 bytes per lane and destination wave = four 16-bit sums + a 32-bit tag (the
//     guide's data-tagged granule, "handoff-1to1"): the producer only stores (no wait for the stores, no flag), the consumer loads its eight
//     sources' granules EARLY -- in the middle of the other half's product phase -- and checks the tags when it needs the values; a granule
//     that had not landed yet is simply loaded again.  A half's partial sums so travel, and their loads fly, under the other half's MFMAs.
// Accumulator layout: lane l holds units 4 (l >> 4) + r of its tile for row l & 15 -- on the consumer side FOUR consecutive units of one row
// per half (one group of operands: saved gates 32 bytes, c[t-1] 16, dh_ext 16, keep bytes 4).
// Two exchange buffers per half: a producer re-writes a buffer two steps later, behind its own pointwise of the step between, whose
// granules come from a wave of EVERY member, each of which has passed its workgroup's barrier behind the pointwise that read the old ones.
// ------------------------------------------------------------------------------------------------------------------
struct ClBwd2Geom {
    static constexpr int U = 512;
    static constexpr int PZ = 256 * 2 + 16;          // pitch of a row of the member's dz columns in LDS (rows 4 banks apart)
    static constexpr int ZB = 16 * PZ;               // one half tile
    static constexpr int OFF_Z = 0;                  // [half][buffer]
    static constexpr int OFF_L = OFF_Z + 4 * ZB;
    static constexpr int LDS = OFF_L + 16;
    static constexpr int XBUF = 8 * 4 * 8 * 1024;    // one exchange buffer of a half: [destination member][destination wave][source member][lane] x 16 bytes
    static constexpr int XCL = 4 * XBUF;             // per cluster: [half][buffer] = 1 MB (eight timesteps' room in the row-parallel workspace)
};

// ABL (development builds with -DCL_ABLATE, env MNN_CLUSTER_ABL; timing only -- results are wrong): 1 no output emit, 2 no partial stores / tag checks,
// 4 no granule loads / tag checks, 8 no MFMAs, 16 no pointwise arithmetic
template <typename F, bool DROP, int ABL = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) lstm_cl_bwd2_kernel(ClBwdArgs A) {
    typedef ClBwd2Geom G;
    typedef typename F::x8 frag_t;
    constexpr int U = G::U;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g4 = lane >> 4;
    const int T = A.T, B = A.B;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3, mem = seq & 7, cl = xcd * (A.ncl >> 3) + (seq >> 3);
    const int row0 = 32 * cl;
    const size_t us = (size_t)B * U;
    const float ikp = 1.0f / A.kp;
    unsigned* status = A.sync;
    int* s_local = reinterpret_cast<int*>(smem + G::OFF_L);
    cl_probe_xcd(A.sync + CL_FLAGS_OFF + 32 * A.ncl + 32 * cl, status, mem, s_local);

    // ---- producer side: Wh[unit][the member's 256 gate columns] for the units [128 w, 128 w + 128): eight tiles x 8 k-steps of A fragments, in AGPRs ----
    frag_t wr[8][8];
    {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const h16_t* src = A.wh_p + (size_t)(128 * w + 16 * i + n) * (4 * U) + 256 * mem + 8 * g4;
#pragma unroll
            for (int s = 0; s < 8; ++s) wr[i][s] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(src + 32 * s));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int s = 0; s < 8; ++s) asm volatile("" : "+a"(wr[i][s]));
    }
    for (int i = tid; i < 4 * G::ZB / 4; i += 256) reinterpret_cast<unsigned*>(smem + G::OFF_Z)[i] = 0u;       // dz[T] = 0

    // ---- consumer side: this lane's four units, for row n of either half; operands a step ahead ----
    const int u0 = 64 * mem + 16 * w + 4 * g4;
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)A.gates, 0, (int)min((size_t)T * us * 8, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)A.c, 0, (int)min((size_t)T * us * 4, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)A.dh_ext, 0, (int)min((size_t)T * us * 4, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(DROP ? (void*)A.mask : (void*)A.c, 0, (int)min((size_t)T * us, (size_t)0x7fffffff), 0x00020000);
    const unsigned vo_e = (unsigned)(n * U + u0);                               // element offset inside a step's 32 x u block (half X: + 16 X rows)
    u32x4_t gq0[2], gq1[2], cq[2], dq[2];
    unsigned mq[2] = {0u, 0u};
    auto request = [&](int X, int t) {                                          // operands of step t (t >= 0) of half X
        const unsigned so = (unsigned)((size_t)t * us + (size_t)(row0 + 16 * X) * U);
        // (readfirstlane: the compiler turns the clamped t - 1 into a VECTOR saturating subtract, and the load into a waterfall loop over its "divergent" offset)
        const unsigned sp = __builtin_amdgcn_readfirstlane((unsigned)((size_t)(t > 0 ? t - 1 : 0) * us + (size_t)(row0 + 16 * X) * U));
        gq0[X] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, vo_e * 8, so * 8, 0);
        gq1[X] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, vo_e * 8 + 16, so * 8, 0);
        cq[X] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, vo_e * 4, sp * 4, 0);                          // c[t-1] (t = 0: read and ignored)
        dq[X] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, vo_e * 4, so * 4, 0);
        if (DROP) mq[X] = __builtin_amdgcn_raw_buffer_load_b32(rs_m, vo_e, so, 0);
    };
    float cnext[2][4], dcreg[2][4], dbv[2][4][4];
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        const u32x4_t c_last = __builtin_amdgcn_raw_buffer_load_b128(rs_c, vo_e * 4, (unsigned)((size_t)(T - 1) * us + (size_t)(row0 + 16 * X) * U) * 4, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cnext[X][r] = __uint_as_float(c_last[r]);
            dcreg[X][r] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) dbv[X][g][r] = 0.f;
        }
    }
    request(0, T - 1);
    request(1, T - 1);
    CL_BARRIER();
    const bool local = A.allow_local != 0 && *s_local != 0;
    if (*s_local == 0) {
        // the two-deep exchange area is re-written every other step: only valid while the cluster shares one L2.  Give up loudly (sticky status word).
        if (tid == 0) { cl_st(status, 1u); cl_st(status + 1, 1u); }
        return;
    }

    // ---- the exchange area of this cluster: [half][buffer]; granule tags count from 1, and a launch starts from a clean area only in its first
    // two steps' view: the tag of step kk is (launch nonce << 12 | kk + 1) so that a granule of an EARLIER launch never matches ----
    char* xch = A.xchg + (size_t)cl * (size_t)G::XCL;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)xch, 0, G::XCL, 0x00020000);
    const unsigned vo_xl = (unsigned)(((mem * 4 + w) * 8) * 1024 + lane * 16);   // loads: + 1024 source member
    const unsigned nonce = (unsigned)A.nonce << 12;
    constexpr int abl = ABL;
    CL_TR_DECL;

    // ---- outputs of a half step (the member's 16 x 256 slice of dz, from the LDS tile): row-major 16-byte pieces and the transposed copy ----
    constexpr unsigned OOB = 0x80000000u;
    const size_t N = (size_t)T * B;
    const bool kb = A.ld_t == 0;
    const __amdgpu_buffer_rsrc_t rs_zc = __builtin_amdgcn_make_buffer_rsrc(A.dzc ? (void*)A.dzc : (void*)A.c, 0, A.dzc ? (int)min(N * 4 * U * 2, (size_t)0x7fffffff) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_zt = __builtin_amdgcn_make_buffer_rsrc(A.dzT ? (void*)A.dzT : (void*)A.c, 0,
                                                                           A.dzT ? (int)min(kb ? N * 4 * U * 2 : (size_t)4 * U * A.ld_t * 2, (size_t)0x7fffffff) : 0, 0x00020000);
    // piece j (0, 1) of this thread: p = tid + 256 j;  row-major: row p >> 5, 16-byte piece p & 31;  transposed: column p >> 1, rows 8 (p & 1) ..+7 of the half
    u32x4_t e_row;
    unsigned e_col[8];
    auto emit_read = [&](const char* zb, int j) {
        const int p = tid + 256 * j;
        e_row = *reinterpret_cast<const u32x4_t*>(zb + (p >> 5) * G::PZ + (p & 31) * 16);
#pragma unroll
        for (int k = 0; k < 8; ++k) e_col[k] = *reinterpret_cast<const h16_t*>(zb + (8 * (p & 1) + k) * G::PZ + (p >> 1) * 2);
    };
    auto emit_store = [&](int X, int tt, int j) {   // dz[tt] of half X (tt >= T: nothing)
        const unsigned none = tt >= T ? OOB : 0u;
        const int tc = tt >= T ? 0 : tt;
        const int p = tid + 256 * j;
        const unsigned so_c = (unsigned)(((size_t)tc * B + row0 + 16 * X) * 4 * U * 2);
        __builtin_amdgcn_raw_buffer_store_b128(e_row, rs_zc, (unsigned)(((p >> 5) * 4 * U + 256 * mem) * 2 + (p & 31) * 16) | none, so_c, 0);
        u32x4_t v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = e_col[2 * k] | (e_col[2 * k + 1] << 16);
        const int col = 256 * mem + (p >> 1);
        const unsigned vo_t = kb ? (unsigned)(col * 64 + 32 * X + (p & 1) * 16) : (unsigned)(((size_t)col * A.ld_t + 16 * X + 8 * (p & 1)) * 2);
        const unsigned so_t = kb ? (unsigned)(((size_t)tc * (B >> 5) + cl) * (4 * U) * 64) : (unsigned)(((size_t)tc * B + row0) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_zt, vo_t | none, so_t, 0);
    };

    // the eight sources' granules of half Y, step kq (requested early: see above)
    u32x4_t pq[8];
    auto fetch = [&](int Y, int kq) {
        const unsigned xb = (unsigned)((2 * Y + (kq & 1)) * G::XBUF);
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) pq[s8] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo_xl + 1024 * s8, xb, CL_SC1);
    };
    // products of half X at step kk (reads the half's dz[t+1] tile, emits that tile's outputs in the MFMA shadow) and their granules;
    // half way through, the granules of the OTHER half's pending pointwise (half Y, step kq; kq < 0: none) are requested
    auto produce = [&](int X, int kk, int Y, int kq) {
        const int t = T - 1 - kk;
        const char* zb = smem + G::OFF_Z + (2 * X + (kk & 1)) * G::ZB;
        const char* zin = zb + n * G::PZ + g4 * 16;
        const unsigned xb = (unsigned)((2 * X + (kk & 1)) * G::XBUF);
        const unsigned tag = nonce | (unsigned)(kk + 1);
        frag_t bq[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) bq[s] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(zin + 64 * s));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mnn_f32x4 ac = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (!(abl & 8)) ac = F::mfma16(wr[i][s], bq[s], ac);
                if (!(abl & 1)) {
                    if ((i == 0 || i == 4) && s == 1) emit_read(zb, i >> 2);
                    if ((i == 2 || i == 6) && s == 1) emit_store(X, t + 1, i >> 2);
                }
                __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400 | 0x40 | 0x200);
            }
            // destination of this tile: member 2 w + (i >> 2), wave i & 3; its lane l reads what this lane l stores
            u32x4_t v;                                  // each 8-byte half of the granule carries the tag
            v[0] = pack2<F>(ac[0], ac[1]);
            v[1] = tag;
            v[2] = pack2<F>(ac[2], ac[3]);
            v[3] = tag;
            const unsigned vo = (unsigned)(((((2 * w + (i >> 2)) * 4 + (i & 3)) * 8 + mem) * 1024) + lane * 16);
            if (!(abl & 2)) __builtin_amdgcn_raw_buffer_store_b128(v, rs_x, vo, xb, 0);      // plain: stays in the XCD's L2 (the kernel has refused any other placement)
            if (i == 3 && kq >= 0 && !(abl & 4)) fetch(Y, kq);
        }
        CL_FENCE();
        __builtin_amdgcn_sched_barrier(0);              // the tag checks of the pointwise behind stay behind (hoisted, they wait for the granules inside the MFMA stream)
    };
    // partial sums and pointwise of half X at step kk (its granules were requested by the product phase in front); writes the half's dz[t] tile.
    // false: the launch is aborting
    auto consume = [&](int X, int kk) -> bool {
        const int t = T - 1 - kk;
        const unsigned tag = nonce | (unsigned)(kk + 1);
        {
            unsigned bad = 0u;
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) bad |= (pq[s8][1] ^ tag) | (pq[s8][3] ^ tag);
            if (__any(bad != 0u) && !(abl & 6)) {                                                    // a granule had not landed: load again (bounded)
                const long long t0 = wall_clock64();
                for (unsigned spins = 1;; ++spins) {
#ifdef CL_TRACE
                    tr_[8 + X] += 1;                    // reloads of half X in this step
#endif
                    CL_FENCE();
                    fetch(X, kk);
                    bad = 0u;
#pragma unroll
                    for (int s8 = 0; s8 < 8; ++s8) bad |= (pq[s8][1] ^ tag) | (pq[s8][3] ^ tag);
                    if (!__any(bad != 0u)) break;
                    if ((spins & 63u) == 0u && (__builtin_amdgcn_readfirstlane(cl_ld(status)) != 0u || wall_clock64() - t0 > CL_LIMIT)) {
                        if (lane == 0) { cl_st(status, 1u); cl_st(status + 1, 1u); }
                        return false;
                    }
                }
            }
        }
        char* zout = smem + G::OFF_Z + (2 * X + ((kk + 1) & 1)) * G::ZB + n * G::PZ;
        float dhr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sum = 0.f;
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) sum += (r & 1) ? F::hi(pq[s8][r & 2]) : F::lo(pq[s8][r & 2]);
            dhr[r] = sum;
        }
        h16_t b4[4][4];                                 // [gate][unit]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (abl & 16) {
#pragma unroll
                for (int g = 0; g < 4; ++g) b4[g][r] = (h16_t)(gq0[X][r] ^ gq1[X][r] ^ cq[X][r] ^ dq[X][r] ^ mq[X] ^ __float_as_uint(dhr[r]));
                continue;
            }
            const unsigned g01 = r < 2 ? gq0[X][2 * r] : gq1[X][2 * r - 4], g23 = r < 2 ? gq0[X][2 * r + 1] : gq1[X][2 * r - 3];
            const float gi = F::lo(g01), gg = F::hi(g01), gf = F::lo(g23), go = F::hi(g23);
            const float dv = __uint_as_float(dq[X][r]);
            const float dh = (DROP ? dv * ikp * (float)((mq[X] >> (8 * r)) & 0xffu) : dv) + dhr[r];
            const float tc = fast_tanh(cnext[X][r]);
            const float d_o = dh * tc;
            const float d_c = dh * go * (1.f - tc * tc) + dcreg[X][r];
            const float cprev = t > 0 ? __uint_as_float(cq[X][r]) : 0.f;
            const float dzv[4] = {d_c * gg * gi * (1.f - gi), d_c * gi * (1.f - gg * gg), d_c * cprev * gf * (1.f - gf), d_o * go * (1.f - go)};
            dcreg[X][r] = d_c * gf;
            cnext[X][r] = cprev;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                b4[g][r] = F::cvt(dzv[g]);
                dbv[X][g][r] += F::f32(b4[g][r]);       // the (16-bit) values the weight-gradient GEMMs see
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2_t v;
            v[0] = (unsigned)b4[g][0] | ((unsigned)b4[g][1] << 16);
            v[1] = (unsigned)b4[g][2] | ((unsigned)b4[g][3] << 16);
            *reinterpret_cast<u32x2_t*>(zout + (gate_perm_col(g, u0) - 256 * mem) * 2) = v;
        }
        request(X, t > 0 ? t - 1 : 0);                  // unconditional (clamped): behind a condition the compiler COPIES the freshly requested registers at the join -- and waits for the loads to do it
        return true;
    };

    // (the first step is peeled: a conditional pointwise inside the loop makes the compiler copy the operand registers at the join -- copies of
    // registers whose loads are still in flight, i.e. a full memory wait in front of every product phase)
    produce(0, 0, 1, -1);
    CL_BARRIER();
    produce(1, 0, 0, 0);
    if (!consume(0, 0)) return;
    CL_BARRIER();
    for (int kk = 1; kk < T; ++kk) {
        CL_TR(0);
#ifdef CL_TRACE
        tr_[8] = tr_[9] = 0;
#endif
        produce(0, kk, 1, kk - 1);
        CL_TR(1);
        if (!consume(1, kk - 1)) return;
        CL_TR(2);
        CL_BARRIER();                                                            // half 1's dz tile of the previous step is complete
        CL_TR(3);
        produce(1, kk, 0, kk);
        CL_TR(4);
        if (!consume(0, kk)) return;
        CL_TR(5);
        CL_BARRIER();                                                            // half 0's dz tile of this step is complete
        CL_TR(6);
        CL_TR_FLUSH(1, kk);
    }
    CL_FENCE();
    fetch(1, T - 1);
    if (!consume(1, T - 1)) return;
    CL_BARRIER();
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            emit_read(smem + G::OFF_Z + (2 * X + (T & 1)) * G::ZB, j);
            emit_store(X, 0, j);
        }
    if (A.db_p != nullptr) {                            // bias gradient: sums over the cluster's 32 rows and all steps
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = dbv[0][g][r] + dbv[1][g][r];
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) v += __shfl_xor(v, d);
                if (n == 0) atomicAdd(A.db_p + gate_perm_col(g, u0 + r), v);
            }
    }
}


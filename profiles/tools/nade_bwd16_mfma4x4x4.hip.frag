// Round 6 prototype (not shipped, never timed end to end): nade_bwd_kernel<2,1> with the flip-free 4 x 4 (row, visible) quads of a chunk multiplied by
// v_mfma_f32_4x4x4_16B_f16.  A FRAGMENT of multinn_amd/csrc/nade.hip (it sat in front of mnn_nade_logprob_bwd and uses that file's helpers); notes:
// profiles/round6_a_nade_bwd_bpermute.md, section "4 x 4 x 4 matrix-core products".

// ----------------------------------------------------------------------------------------------
// backward, f16-product form (precision "fp16": the caller's loss scale keeps |d nll / d logit| in IEEE-half range).
// Same scan, same layout (8 waves x 8 rows, lane = two hidden units of a 128-wide slice, the same exchange and atomics) as nade_bwd_kernel<2, 1>;
// what changes is WHO multiplies.  The kernel above is bound by vector issue: per visible and wave 16 packed FMAs + 8 v_readlane.  Here the 8 x 8
// (row, visible) cells of a chunk are cut into four QUADS of 4 rows x 4 visibles, and a quad in which no row has an active visible (61 % of them
// at rho = 0.03) is TWO v_mfma_f32_4x4x4_16B_f16 per hidden half instead of 32 packed FMAs + 16 readlanes:
//     d w_dec[4 visibles][lane's hidden unit] += dl^T[4 vis x 4 rows] . h[4 rows x hidden]        (A = dl^T, B = the rows' h as IEEE halves)
//     c[4 rows][lane's hidden unit]           += dl[4 rows x 4 vis]  . w_dec[4 vis x hidden]      (A = dl,   B = the staged w_dec as halves)
// The instruction's sixteen 4 x 4 x 4 blocks are the sixteen lane quads of the wave: lane 4b + j supplies column j of block b's B operand
// and receives column j of its result in four registers -- i.e. "lane = hidden unit, registers = visibles (resp. rows)", exactly the layout
// the scan's accumulators already have, so quads with a flip simply run the scalar code on the same registers.  Measured on MI355X
// (scratch probe of round 6): one such MFMA costs ~9 cycles of the SIMD and does not overlap other waves' vector work -- a 3 x cheaper
// multiply, not a free one.  Operands: dl as f16 through a 256-byte per-wave LDS tile in both orientations (two ds_write_b16 per chunk, one
// ds_read_b64 per operand), w_dec staged as f16 [hidden][8 visibles] (one ds_read_b128 per chunk and hidden half), h packed on the fly.
// f32 everywhere else: a, h, G, c, the encoder rows, every sum.
// ----------------------------------------------------------------------------------------------
typedef _Float16 nb_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 nb_h8 __attribute__((ext_vector_type(8)));
typedef float nb_f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))       // two workgroups per CU, as nade_bwd_kernel<2, 1>
nade_bwd16_kernel(int tracks, int N, int D, int HnT, int nslice, const uint8_t* __restrict__ v, long v_track_stride, const float* __restrict__ bias,
                  int ld_bias, const float* __restrict__ w_enc, const float* __restrict__ w_dec, const float* __restrict__ a_final,
                  float* __restrict__ d_bias, float* __restrict__ d_w_enc, float* __restrict__ d_w_dec, const int* __restrict__ n_rows_dev) {
    constexpr int HQ = 2, W = 128;
    __shared__ __attribute__((aligned(16))) float wle[2][8 * W];              // [buffer][visible][lane-major hidden pair]: w_enc, f32
    __shared__ __attribute__((aligned(16))) _Float16 wld[2][HQ][64][8];       // [buffer][hidden half][lane][visible]: w_dec as IEEE halves
    __shared__ __attribute__((aligned(16))) _Float16 dls[8][2][64];           // per wave: dl of the chunk as [row][visible] and as [visible][row]
    __shared__ __attribute__((aligned(16))) float red_[2][8][4][2][W];
    const int m = blockIdx.y / nslice, hb = (blockIdx.y - m * nslice) * W;
    const int Hn = min(W, HnT - hb);
    if (nade_rows_beyond(n_rows_dev, blockIdx.x * 64)) {
        nade_zero_rows(d_bias, ld_bias, m * HnT + hb, Hn, blockIdx.x * 64, 64, N, 512);
        return;
    }
    const int Nv = n_rows_dev != nullptr ? min(*n_rows_dev, N) : N;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rbase = blockIdx.x * 64 + w * 8;
    const uint8_t* __restrict__ vm = v + (size_t)m * v_track_stride;
    const float* __restrict__ we = w_enc + (size_t)m * D * HnT + hb;
    const float* __restrict__ wd = w_dec + (size_t)m * D * HnT + hb;
    const int dl_off = tracks * HnT + m * D;

    float a[8][HQ], h[8][HQ], G[8][HQ];
    nb_f4 c4[2][HQ];                                                          // c of rows 4 g + 0..3 (vector elements), hidden half q
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < HQ; ++q)
            a[r][q] = a_final[((size_t)m * N + min(rbase + r, N - 1)) * HnT + hb + min(lane + 64 * q, Hn - 1)];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            if (!(rbase + r < Nv && lane + 64 * q < Hn)) a[r][q] = 0.f;
            h[r][q] = fast_sigmoid(a[r][q]);
            G[r][q] = 0.f;
        }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int q = 0; q < HQ; ++q) c4[g][q] = (nb_f4){0.f, 0.f, 0.f, 0.f};
    const int fi = lane & 7, frow = rbase + (lane >> 3);
    const bool fvalid = frow < N;
    const int frr = fvalid ? frow : N - 1;
    const int nch = (D + 7) / 8;
    // staging: thread (visible sv of the chunk, lane) fetches its two hidden units' w_dec / w_enc (clamped, unconditional: see nade_bwd_kernel)
    const int sv = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float rd[HQ], re[HQ];
    auto gload = [&](int i0) {
        const int ic = min(max(i0 + sv, 0), D - 1);
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const int col = min(lane + 64 * q, Hn - 1);
            rd[q] = wd[(size_t)ic * HnT + col];
            re[q] = we[(size_t)ic * HnT + col];
        }
    };
    auto lstore = [&](int buf) {
        *reinterpret_cast<float2*>(&wle[buf][sv * W + HQ * lane]) = make_float2(re[0], re[1]);
#pragma unroll
        for (int q = 0; q < HQ; ++q) wld[buf][q][lane][sv] = (_Float16)rd[q];
    };
    gload((nch - 1) * 8);
    lstore(0);
    const int icur = (nch - 1) * 8 + fi;
    bool vcur = fvalid && icur < D && vm[(size_t)frr * D + icur] != 0;
    float dcur = (fvalid && icur < D) ? d_bias[(size_t)frr * ld_bias + dl_off + icur] : 0.f;
    __syncthreads();
    for (int cc = 0; cc < nch; ++cc) {
        const int i0 = (nch - 1 - cc) * 8;
        const int inext = i0 - 8 + fi;
        gload(i0 - 8);
        const int inc = max(inext, 0);
        const uint8_t vraw = vm[(size_t)frr * D + inc];
        const float draw = d_bias[(size_t)frr * ld_bias + dl_off + inc];
        const unsigned long long mask = __ballot(vcur);                        // bit 8 row + visible
        // the chunk's dl as halves, both orientations (this wave's own tile: LDS is in order per wave, no barrier)
        dls[w][0][lane] = (_Float16)dcur;
        dls[w][1][(lane & 7) * 8 + (lane >> 3)] = (_Float16)dcur;
        const float* __restrict__ se = wle[cc & 1];
#pragma unroll
        for (int half = 1; half >= 0; --half) {
            nb_h4 wd4[HQ];                                                      // the half's four decoder rows of this lane's two hidden units
#pragma unroll
            for (int q = 0; q < HQ; ++q) wd4[q] = *reinterpret_cast<const nb_h4*>(&wld[cc & 1][q][lane][4 * half]);
            nb_f4 accd4[HQ];
            float acce[4][HQ];
#pragma unroll
            for (int q = 0; q < HQ; ++q) {
                accd4[q] = (nb_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) acce[k][q] = 0.f;
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const unsigned cell = ((unsigned)(mask >> (32 * g)) >> (4 * half)) & 0x0F0F0F0Fu;      // active visibles of rows 4g..4g+3 in this half
                if (cell == 0u) {
                    // ---- no flip in the quad: two 4x4x4 matrix-core products per hidden half ----
                    const nb_h4 Ac = *reinterpret_cast<const nb_h4*>(&dls[w][0][(4 * g + (lane & 3)) * 8 + 4 * half]);     // dl[row 4g + j][visibles of the half]
                    const nb_h4 Ad = *reinterpret_cast<const nb_h4*>(&dls[w][1][(4 * half + (lane & 3)) * 8 + 4 * g]);     // dl[rows 4g..][visible 4 half + j]
#pragma unroll
                    for (int q = 0; q < HQ; ++q) {
                        const nb_h4 Bd = {(_Float16)h[4 * g][q], (_Float16)h[4 * g + 1][q], (_Float16)h[4 * g + 2][q], (_Float16)h[4 * g + 3][q]};
                        accd4[q] = __builtin_amdgcn_mfma_f32_4x4x4f16(Ad, Bd, accd4[q], 0, 0, 0);
                        c4[g][q] = __builtin_amdgcn_mfma_f32_4x4x4f16(Ac, wd4[q], c4[g][q], 0, 0, 0);
                    }
                } else {
                    // ---- the scalar scan of nade_bwd_kernel on the quad's four rows ----
#pragma unroll
                    for (int k = 3; k >= 0; --k) {
                        const int ii = half * 4 + k;
                        if (i0 + ii >= D) continue;
                        float wdv[HQ];
#pragma unroll
                        for (int q = 0; q < HQ; ++q) wdv[q] = (float)wd4[q][k];
                        if ((cell >> k) & 0x01010101u) {
                            float wev[HQ];
                            lv_load<HQ>(se + ii * W + HQ * lane, wev);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int r = 4 * g + j;
                                if ((cell >> (8 * j + k)) & 1u) {
#pragma unroll
                                    for (int q = 0; q < HQ; ++q) {
                                        G[r][q] = fmaf(c4[g][q][j], fmaf(-h[r][q], h[r][q], h[r][q]), G[r][q]);
                                        c4[g][q][j] = 0.f;
                                        acce[k][q] += G[r][q];
                                        a[r][q] -= wev[q];
                                        h[r][q] = fast_sigmoid(a[r][q]);
                                    }
                                }
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int r = 4 * g + j;
                            const float dl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dcur), r * 8 + ii));
#pragma unroll
                            for (int q = 0; q < HQ; ++q) {
                                accd4[q][k] = fmaf(dl, h[r][q], accd4[q][k]);
                                c4[g][q][j] = fmaf(dl, wdv[q], c4[g][q][j]);
                            }
                        }
                    }
                }
            }
            if (half == 1) lstore((cc + 1) & 1);
            float (*red)[4][2][W] = red_[(2 * cc + 1 - half) & 1];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                *reinterpret_cast<float2*>(&red[w][k][0][HQ * lane]) = make_float2(accd4[0][k], accd4[1][k]);
                *reinterpret_cast<float2*>(&red[w][k][1][HQ * lane]) = make_float2(acce[k][0], acce[k][1]);
            }
            __syncthreads();
            {
                const int k = threadIdx.x >> 7, which = (threadIdx.x >> 6) & 1;
                const int i = i0 + half * 4 + k;
                float sum[HQ] = {0.f, 0.f};
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) {
                    const float2 pq = *reinterpret_cast<const float2*>(&red[ww][k][which][HQ * lane]);
                    sum[0] += pq.x;
                    sum[1] += pq.y;
                }
                if (i < D) {
                    float* dst = (which == 0 ? d_w_dec : d_w_enc) + ((size_t)m * D + i) * HnT + hb + lane;
#pragma unroll
                    for (int q = 0; q < HQ; ++q)
                        if (lane + 64 * q < Hn) atomicAdd(dst + 64 * q, sum[q]);
                }
            }
        }
        const bool ok = fvalid && inext >= 0;
        vcur = ok && vraw != 0;
        dcur = ok ? draw : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            G[r][q] = fmaf(c4[r >> 2][q][r & 3], fmaf(-h[r][q], h[r][q], h[r][q]), G[r][q]);
            const int j = lane + 64 * q, row = rbase + r;
            if (row < N && j < Hn) __builtin_nontemporal_store(G[r][q], &d_bias[(size_t)row * ld_bias + m * HnT + hb + j]);
        }
}


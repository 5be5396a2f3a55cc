// NOT PART OF THE PRODUCT (round 3 experiment): the 16x16x4 weight gradient of gen_train.hip with its operands streamed straight from
// L1 / L2 instead of staged through LDS (kernel body as it was inside gen_train.hip, same GenWgradParams / epilogue).  Correct (<= 7e-7 of
// float64 autograd) but slower: chfak 5, n = 512, eight layer shapes: 3809 vs 2407 us (features.3 876 vs 606, masker.0 with its uint8
// source 1448 vs 496): 12-15 four-piece gather loads per 27 matrix instructions keep the texture addresser busier than the LDS path
// keeps the barriers.
// ------------------------------------------------------------------------------------------------
// The same weight gradient with NO LDS and no barriers (round 3): every wave streams its operands straight from L1 / L2.
// The A operand of v_mfma_f32_16x16x4_f32 is one dword per lane -- lane (row = input channel of the 16-channel block, k = one of 4
// consecutive pixels of a row) reads X[p_k + tap][channel]: 16 consecutive channels of a pixel are 64 contiguous bytes of the NHWC
// tensor, a wave-load is four such pieces; the B operand likewise dY[p_k][column].  27 matrix instructions (864 cycles) per 12
// loads and ~45 address / mask instructions; the 3x3 window's overlap is served by the caches (each input pixel is fetched from L2
// about three times, ~7 bytes per clock and CU).  A wave owns a 16 x (NCOB x 16) x 9-tap block for a share of the pixel groups; the
// four waves of a workgroup are independent until the final reduction; operands of the next group are requested before the
// current group multiplies.  The staged form above spends as long staging a 256-pixel tile (global -> registers -> LDS, two
// barriers) as multiplying it and runs at 2 waves per SIMD; this form runs at 3 and waits for nothing but its own loads.
// Sources: fp32 or uint8 A, optional nearest-upsampled fp32 B; dY fp32 or pooled gradient + argmax bytes (POOLED).
// ------------------------------------------------------------------------------------------------
template <int NCOB, bool POOLED>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NCOB == 1 ? 3 : 2, NCOB == 1 ? 3 : 2))) gen_wgrad_stream_kernel(GenWgradParams P) {
    extern __shared__ __attribute__((aligned(16))) float4 gsm[];
    constexpr int NT = 9, NV = 4 * NT + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int H = P.hw, W = P.hw;
    const int lw = W == 64 ? 6 : (W == 32 ? 5 : (W == 16 ? 4 : (W == 8 ? 3 : 2)));
    const int cog = blockIdx.x % P.ncob, cib = (blockIdx.x / P.ncob) % P.ncib, g = blockIdx.x / (P.ncob * P.ncib);
    const int co = P.dy.ca, ci_total = P.in.ca + P.in.cb;
    const int nblk = (co + 15) / 16;

    // ---- this lane's input channel: which source, its pixel stride, its resolution ----
    const int pa4 = gen_pa4(P.in), kc = cib * GEN_KC + l15;
    const int real = gen_real_channel(P.in, kc);
    const bool chok = real >= 0, isb = chok && kc >= pa4, a_u8 = P.in.mode == GEN_SRC_U8;      // (padding lanes read source A's element 0)
    const int ush = isb ? (P.in.ups == 4 ? 2 : (P.in.ups == 2 ? 1 : 0)) : 0;
    const int Cs = isb ? P.in.cb : P.in.ca, Hs = H >> ush, Ws = W >> ush;
    const int cofs = chok ? (isb ? kc - pa4 : kc) : 0;
    const char* srcp = isb ? (const char*)P.in.b : (const char*)P.in.a;
    const bool ld8 = a_u8 && !isb;                     // this lane reads bytes (/255)
    const int esz = ld8 ? 1 : 4;
    // ---- this lane's output columns ----
    int colc[NCOB];
    bool cok[NCOB];
#pragma unroll
    for (int c = 0; c < NCOB; ++c) {
        colc[c] = (cog * NCOB + c) * 16 + l15;
        cok[c] = cog * NCOB + c < nblk && colc[c] < co;
        colc[c] = cok[c] ? colc[c] : 0;
    }

    frag4 acc[NT][NCOB];
    float bsum[NCOB];
#pragma unroll
    for (int c = 0; c < NCOB; ++c) {
        bsum[c] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t][c] = frag4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- pixel groups (4 consecutive pixels of a row) of this share's images, dealt round-robin to the four waves ----
    const int U = (H * W) >> 2, lg = lw - 2;           // groups per image; log2(groups per row)
    const int nimg = g < P.n ? (P.n - g + P.G - 1) / P.G : 0;
    const long total = (long)nimg * U;                 // groups of this share
    float av[2][NT], bv[2][NCOB];
    [[maybe_unused]] uint32_t bm[2][POOLED ? NCOB : 1];
    auto issue = [&](long u, float (&a)[NT], float (&b)[NCOB], uint32_t (&m)[POOLED ? NCOB : 1]) {
        const int ii = (int)(u / U), q = (int)(u - (long)ii * U), img = g + ii * P.G;
        const int y = q >> lg, x = ((q & ((1 << lg) - 1)) << 2) + kq;
        int colo[3];
        bool cv[3];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
            const int xx = x + tx - 1, xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            cv[tx] = chok && xx >= 0 && xx < W;
            colo[tx] = ((xc >> ush) * Cs + cofs) * esz;
        }
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const int yy = y + ty - 1, yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            const bool rv = yy >= 0 && yy < H;
            const char* rp = srcp + (size_t)((img * Hs + (yc >> ush)) * Ws) * Cs * esz;
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const char* pp = rp + colo[tx];
                float v;
                if (ld8) v = (float)*(const uint8_t*)pp * (1.f / 255.f);
                else v = *(const float*)pp;
                a[ty * 3 + tx] = (rv && cv[tx]) ? v : 0.f;
            }
        }
        if constexpr (POOLED) {
            const size_t e0 = ((size_t)(img * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * co;
#pragma unroll
            for (int c = 0; c < NCOB; ++c) {
                b[c] = ((const float*)P.dy.a)[e0 + colc[c]];
                m[c] = P.dy.am[e0 + colc[c]];
            }
        } else {
            const size_t e0 = ((size_t)(img * H + y) * W + x) * co;
#pragma unroll
            for (int c = 0; c < NCOB; ++c) b[c] = ((const float*)P.dy.a)[e0 + colc[c]];
        }
    };
    auto mac = [&](long u, const float (&a)[NT], float (&b)[NCOB], const uint32_t (&m)[POOLED ? NCOB : 1]) {
        if constexpr (POOLED) {
            const int q = (int)(u % U), y = q >> lg, x = ((q & ((1 << lg) - 1)) << 2) + kq;
            const uint32_t pos = (uint32_t)(((y & 1) << 1) | (x & 1));
#pragma unroll
            for (int c = 0; c < NCOB; ++c) b[c] = (cok[c] && m[c] == pos) ? b[c] : 0.f;
        } else {
#pragma unroll
            for (int c = 0; c < NCOB; ++c) b[c] = cok[c] ? b[c] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < NCOB; ++c) bsum[c] += b[c];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int c = 0; c < NCOB; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[c], acc[t][c], 0, 0, 0);
    };
    long u = wave;
    if (u < total) issue(u, av[0], bv[0], bm[0]);
#pragma unroll 1
    for (; u < total; u += 8) {
        if (u + 4 < total) issue(u + 4, av[1], bv[1], bm[1]);
        __builtin_amdgcn_sched_barrier(0);
        mac(u, av[0], bv[0], bm[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 4 < total) {
            if (u + 8 < total) issue(u + 8, av[0], bv[0], bm[0]);
            __builtin_amdgcn_sched_barrier(0);
            mac(u + 4, av[1], bv[1], bm[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- the four waves' partial blocks summed through LDS (fixed order), one column block at a time; then the slab row ----
    float* red = (float*)gsm;                                  // [3 waves][NV][64]
    float* row = P.slab + (size_t)g * (9 * ci_total * co + co);
#pragma unroll
    for (int c = 0; c < NCOB; ++c) {
        if (wave > 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) red[((wave - 1) * NV + 4 * t + j) * 64 + lane] = acc[t][c][j];
            red[((wave - 1) * NV + 4 * NT) * 64 + lane] = bsum[c];
        }
        __syncthreads();
        if (wave == 0) {
            float bs = bsum[c];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][c][j] += red[(w * NV + 4 * t + j) * 64 + lane];
                bs += red[(w * NV + 4 * NT) * 64 + lane];
            }
            const int col = (cog * NCOB + c) * 16 + l15;
            if (col < co && cog * NCOB + c < nblk) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ci = gen_real_channel(P.in, cib * GEN_KC + 4 * kq + j);
                    if (ci < 0) continue;
#pragma unroll
                    for (int t = 0; t < 9; ++t) row[((size_t)t * ci_total + ci) * co + col] = acc[t][c][j];
                }
            }
            bs += __shfl_xor(bs, 16, 64);
            bs += __shfl_xor(bs, 32, 64);
            if (cib == 0 && kq == 0 && col < co && cog * NCOB + c < nblk) row[(size_t)9 * ci_total * co + col] = bs;
        }
        __syncthreads();
    }
}


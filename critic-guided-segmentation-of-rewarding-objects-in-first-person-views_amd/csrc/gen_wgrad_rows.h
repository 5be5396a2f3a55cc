// Shape-generic 3x3 weight gradient, second form (included by gen_train.hip inside its anonymous namespace):
//   dW[tap][ci][co] = sum over images, pixels q of in[q][ci] dY[q - tap][co]   (nets.py:166-190, 453-492 under autograd)
// as an MFMA GEMM (v_mfma_f32_16x16x4_f32) whose ROWS are the flattened (tap, ci) pairs of one input-channel slice and whose columns
// are one output-channel slice; the pixels are the K dimension.  What it changes against gen_conv3x3_wgrad_kernel:
//   * rows = 9 * cs flattened: 40 channels are 23 row blocks of 16 (98 % useful) instead of 9 taps x 3 blocks of 16 (83 %);
//   * the 8 waves of a workgroup are (at >= 15 row blocks) 4 row groups x 2 pixel phases: a wave owns up to 7 row blocks x NCOB column
//     blocks (18-21 matrix instructions per 9-12 LDS reads; fp32 MFMA and vector ALU instructions do not overlap on a SIMD, so the
//     address / select work per matrix instruction is what the loop is built to minimise), and one staged chunk feeds (9 cs / 16) x NCOB
//     accumulator blocks instead of 9 x NCOB: a staged byte is used 2-3 times as often;
//   * the next chunk's global loads are issued into registers before the matrix instructions of the current chunk and written to a
//     second LDS buffer after them: one barrier per chunk of 128-512 pixels, loads never waited for in the open;
//   * the gradient of a pooled layer is staged RAW (dE at the pooled resolution + the argmax bytes, a quarter of the elements)
//     and expanded when the B operand is read;
//   * one workgroup per CU, persistent over a contiguous range of chunks: ~256 slab rows per layer instead of 512-1536.
// Sources: A fp32 or uint8 NHWC of any width (padded to whole quads in the tile; a lone A of fewer than 4 channels keeps its width) +
// nearest-upsampled B (cb % 4 == 0); co % 4 == 0 (single-channel outputs stay on gen_conv3x3_wgrad_kernel).  Few-channel layers (the
// frames: 9 x 3 rows = 2 row blocks) give all eight waves the same rows and an eighth of the pixels each: nrg row groups x 8 / nrg
// pixel phases.
#pragma once

struct GenWrParams {
    const void* a; const float* b;        // the layer's input cat(A [ca] fp32 or uint8 (/255), nearest-up(B [cb]))
    const float* dy; const uint8_t* am;   // dY [n,hw,hw,co], or (am != NULL) dE [n,hw/2,hw/2,co] + argmax bytes of the pooled layer
    float* slab;                          // [G][9 * (ca + cb) * co + co]
    int n, hw, lw, ca, cb, ush, co;
    int a_u8;                             // A is uint8
    int nrg;                              // row groups (1, 2 or 4) x 8 / nrg pixel phases = the 8 waves
    int G, nsl, ncs, cs, cw;              // chunk shares; input-channel slices of cs channels; output-channel slices of cw channels
    int imgs, th, parts, units;           // chunk = imgs images x th rows (128 .. 512 pixels); parts = hw / th; units = chunks in the job
    int buf_floats;                       // floats per LDS buffer
    int ci_stride;                        // > 0: the slab rows are those of a layer with ci_stride input channels (this launch writes its first ca + cb)
};

template <int NCOB, int RBW, bool POOLED>
__global__ void __launch_bounds__(512) gen_wgrad_rows_kernel(GenWrParams P) {
    if (CGS_KARG_PREFETCH) cgs_kernarg_prefetch<sizeof(GenWrParams)>();
    extern __shared__ __attribute__((aligned(16))) float4 gsm[];
    float* const sm = (float*)gsm;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, kq = lane >> 4;
    const int nrg = P.nrg, nph = 8 / nrg, lnrg = nrg >> 1;      // (nrg = 1, 2, 4 -> log2 = 0, 1, 2)
    const int rg = wave & (nrg - 1), ph = wave >> lnrg;        // row group (row blocks rg + nrg r), pixel phase (groups nph i + ph)
    const int W = P.hw, H = P.hw, PW = W + 2, th = P.th, imgs = P.imgs, lw = P.lw;
    const int cos = blockIdx.x % P.ncs, sl = (blockIdx.x / P.ncs) % P.nsl, g = blockIdx.x / (P.ncs * P.nsl);
    // (A's channels padded to whole quads, then B's; a lone A of fewer than 4 channels -- the frames -- keeps its width: 27 rows
    //  = 2 row blocks instead of 36 = 3)
    const bool narrow = P.cb == 0 && P.ca < 4;
    const int ca4 = narrow ? P.ca : (P.ca + 3) & ~3, ci_total = P.ci_stride > 0 ? P.ci_stride : P.ca + P.cb, ci_pad = ca4 + P.cb;
    const int ks0 = sl * P.cs, csl = min(P.cs, ci_pad - ks0);             // this workgroup's (padded) input channels [ks0, ks0 + csl)
    const int cs0 = cos * P.cw, cwl = min(P.cw, P.co - cs0);               // and output channels [cs0, cs0 + cwl)
    const int IMS = (th + 2) * PW * csl;                                   // in-tile floats per image slot
    const int INF = (imgs * IMS + 3) & ~3;                                 // (dY behind it stays 16-byte aligned)
    const int DPX = POOLED ? (th >> 1) * (W >> 1) : th * W;                // dY tile pixels per image slot
    const int DYF = imgs * DPX * cwl;
    const int BUF = P.buf_floats;
    const int q4 = narrow ? 1 : csl >> 2, qd = cwl >> 2;                   // staged elements per pixel (quads; narrow: the pixel's csl floats)
    const int ush = P.ush, HB = H >> ush, WB = W >> ush;
    const int u0 = (int)((long)g * P.units / P.G), u1 = (int)((long)(g + 1) * P.units / P.G);
    const bool do_bias = sl == 0 && rg == 0;

    // ---- this lane's A-operand offsets: row m = 16 (rg + 4 r) + l15 = (tap, ci) of the slice ----
    int aoff[RBW];
#pragma unroll
    for (int r = 0; r < RBW; ++r) {
        const int m = 16 * (rg + nrg * r) + l15;
        int o = kq * csl;
        if (m < 9 * csl) {
            const int tap = m / csl, ci = m - tap * csl;
            o += ((tap / 3) * PW + tap % 3) * csl + ci;
        }
        aoff[r] = 4 * o;                 // BYTE offsets: the loop adds the group's base with one instruction per read
    }
    frag4 acc[RBW][NCOB];
    float bsum[NCOB];
#pragma unroll
    for (int c = 0; c < NCOB; ++c) {
        bsum[c] = 0.f;
#pragma unroll
        for (int r = 0; r < RBW; ++r) acc[r][c] = frag4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- the halo columns of both buffers are zero for every chunk: written once ----
    if (narrow) {
        for (int e = tid; e < 2 * imgs * (th + 2) * 2 * csl; e += 512) {
            const int j = e % csl, side = (e / csl) & 1, rr = (e / (2 * csl)) % (imgs * (th + 2)), bf = e / (2 * csl * imgs * (th + 2));
            const int t = rr / (th + 2), r = rr - t * (th + 2);
            sm[bf * BUF + t * IMS + (r * PW + (side ? W + 1 : 0)) * csl + j] = 0.f;
        }
    } else {
        for (int e = tid; e < 2 * imgs * (th + 2) * 2 * q4; e += 512) {
            const int q = e % q4, side = (e / q4) & 1, rr = (e / (2 * q4)) % (imgs * (th + 2)), bf = e / (2 * q4 * imgs * (th + 2));
            const int t = rr / (th + 2), r = rr - t * (th + 2);
            *(float4*)(sm + bf * BUF + t * IMS + (r * PW + (side ? W + 1 : 0)) * csl + 4 * q) = f4zero();
        }
    }

    // Staging of chunk u: every thread owns up to KI quads (16 bytes) of the in-tile and KD of dY (+ KD argmax words of a pooled
    // layer).  issue() starts the global loads into registers -- they fly during the matrix instructions of the current chunk --,
    // store() writes them to the other LDS buffer afterwards (rows outside the image and absent images as zeros).
    // (LDS-DMA -- global_load_lds, 16 bytes per lane -- needs no registers but measured ~10 bytes per cycle and CU on this path:
    //  6.5 k of a chunk's 25 k cycles went into issuing 7 such instructions per wave.)
    constexpr int KI = NCOB == 1 ? 9 : 7, KD = 3, KO = 2;      // (a single column block leaves the registers for a larger chunk)
    // Thread -> element maps without divisions in the chunk loop (vector ALU work is not hidden behind the matrix instructions):
    //   main quads (16-byte loads: an fp32 A of whole quads, B): thread = (pixel lane pl, quad column qc), qc fastest; its k-th
    //     element is pixel pl + PL k of the chunk's NPX = imgs (th + 2) W in-tile pixels;
    //   odd quads (a uint8 / odd-width A, converted component by component: qo per pixel, own elements so that a wave does not
    //     run that path beside every 16-byte load): element = tid + 512 k over NPX qo;
    //   dY quads: thread = (pixel lane pld, quad column qcd).
    const bool odd_a = (P.ca & 3) || P.a_u8;
    const int qA = narrow ? 1 : min(max((ca4 - ks0) >> 2, 0), q4);   // A quads per pixel in this slice
    const int qo = odd_a ? qA : 0, qm = q4 - qo;                      // odd / main quads per pixel
    const int NPX = imgs * (th + 2) * W;
    const int PL = qm ? 512 / qm : 0, qc = qm ? tid % qm : 0, pl = qm ? tid / qm : 512;
    const int PLD = 512 / qd, qcd = tid % qd, pld = tid / qd, NPD = imgs * DPX;
    const uint32_t mT = imgs > 1 ? 0xFFFFFFFFu / (uint32_t)(th + 2) + 1u : 0u;       // x / (th + 2) == umulhi(x, mT), x < 65536
    const uint32_t mTD = imgs > 1 ? 0xFFFFFFFFu / (uint32_t)DPX + 1u : 0u, mQO = qo > 1 ? 0xFFFFFFFFu / (uint32_t)qo + 1u : 0u;
    float4 sti[KI], std_[KD], sto[KO];
    uint32_t sam[POOLED ? KD : 1];
    // pixel p of the in-tile -> image slot t, tile row r, column px
    auto pixel = [&](int p, int& t, int& r, int& px) __attribute__((always_inline)) {
        px = p & (W - 1);
        const int rr = p >> lw;
        t = mT ? (int)__umulhi((uint32_t)rr, mT) : 0;
        r = rr - t * (th + 2);
    };
    auto issue = [&](int u) __attribute__((always_inline)) {
        int lpl = pl, lpld = pld, ltid = tid;      // opaque: addresses are recomputed per chunk instead of living in registers
        asm volatile("" : "+v"(lpl), "+v"(lpld), "+v"(ltid));
        const int img0 = (u / P.parts) * imgs, row0 = (u % P.parts) * th;
        const int kc = ks0 + 4 * (qo + qc);                       // this thread's main quad: (padded) channel
        const bool isb = kc >= ca4;
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            const int p = lpl + PL * k;
            sti[k] = f4zero();
            if (lpl < PL && p < NPX) {
                int t, r, px;
                pixel(p, t, r, px);
                const int img = img0 + t, y = row0 + r - 1;
                if (img < P.n && y >= 0 && y < H) {
                    const float* src = isb ? P.b + ((size_t)((img * HB + (y >> ush)) * WB + (px >> ush)) * P.cb + (kc - ca4))
                                           : (const float*)P.a + ((size_t)((img * H + y) * W + px) * P.ca + kc);
                    sti[k] = *(const float4*)src;
                }
            }
        }
        if (qo) {             // odd width / uint8 frames: component by component, the padding channels are zero
#pragma unroll
            for (int k = 0; k < KO; ++k) {
                const int e = ltid + 512 * k;
                sto[k] = f4zero();
                if (e < NPX * qo) {
                    const int p = mQO ? (int)__umulhi((uint32_t)e, mQO) : e, ko = ks0 + 4 * (e - p * qo);
                    int t, r, px;
                    pixel(p, t, r, px);
                    const int img = img0 + t, y = row0 + r - 1;
                    if (img < P.n && y >= 0 && y < H) {
                        const size_t o = (size_t)((img * H + y) * W + px) * P.ca;
                        const int c1 = min(ko + 1, P.ca - 1), c2 = min(ko + 2, P.ca - 1), c3 = min(ko + 3, P.ca - 1);
                        float4 v;
                        if (P.a_u8) {
                            const uint8_t* s8 = (const uint8_t*)P.a + o;
                            v = make_float4((float)s8[ko], (float)s8[c1], (float)s8[c2], (float)s8[c3]);
                            v.x *= 1.f / 255.f; v.y *= 1.f / 255.f; v.z *= 1.f / 255.f; v.w *= 1.f / 255.f;
                        } else {
                            const float* s32 = (const float*)P.a + o;
                            v = make_float4(s32[ko], s32[c1], s32[c2], s32[c3]);
                        }
                        v.y = ko + 1 < P.ca ? v.y : 0.f; v.z = ko + 2 < P.ca ? v.z : 0.f; v.w = ko + 3 < P.ca ? v.w : 0.f;
                        sto[k] = v;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KD; ++k) {
            const int p = lpld + PLD * k;
            std_[k] = f4zero();
            if constexpr (POOLED) sam[k] = 0u;
            if (lpld < PLD && p < NPD) {
                const int t = mTD ? (int)__umulhi((uint32_t)p, mTD) : 0, pp = p - t * DPX;
                const int img = img0 + t;
                if (img < P.n) {
                    const int pix = POOLED ? (img * (H >> 1) + (row0 >> 1)) * (W >> 1) + pp : (img * H + row0) * W + pp;
                    const size_t off = (size_t)pix * P.co + cs0 + 4 * qcd;
                    std_[k] = *(const float4*)(P.dy + off);
                    if constexpr (POOLED) sam[k] = *(const uint32_t*)(P.am + off);
                }
            }
        }
    };
    auto store = [&](int bf) __attribute__((always_inline)) {
        int lpl = pl, lpld = pld, ltid = tid;
        asm volatile("" : "+v"(lpl), "+v"(lpld), "+v"(ltid));
        float* tin = sm + bf * BUF;
        float* tdy = tin + INF;
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            const int p = lpl + PL * k;
            if (lpl < PL && p < NPX) {
                int t, r, px;
                pixel(p, t, r, px);
                *(float4*)(tin + t * IMS + (r * PW + 1 + px) * csl + 4 * (qo + qc)) = sti[k];
            }
        }
        if (qo) {
#pragma unroll
            for (int k = 0; k < KO; ++k) {
                const int e = ltid + 512 * k;
                if (e < NPX * qo) {
                    const int p = mQO ? (int)__umulhi((uint32_t)e, mQO) : e;
                    int t, r, px;
                    pixel(p, t, r, px);
                    float* dst = tin + t * IMS + (r * PW + 1 + px) * csl;
                    if (narrow) {
                        dst[0] = sto[k].x;
                        if (csl > 1) dst[1] = sto[k].y;
                        if (csl > 2) dst[2] = sto[k].z;
                    } else {
                        *(float4*)(dst + 4 * (e - p * qo)) = sto[k];
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KD; ++k) {
            const int p = lpld + PLD * k;
            if (lpld < PLD && p < NPD) {
                *(float4*)(tdy + (p * qd + qcd) * 4) = std_[k];
                if constexpr (POOLED) ((uint32_t*)(tdy + DYF))[p * qd + qcd] = sam[k];
            }
        }
    };

    // the operands of pixel group grp (4 consecutive pixels of one row = one k-step)
    const int gpi = (th * W) >> 2, lgpi = __builtin_ctz(gpi), lgw = lw - 2, nit = (imgs * gpi) >> (3 - lnrg);      // pixel groups per phase (even)
    const int bl = (POOLED ? (kq >> 1) * cwl : kq * cwl) + l15;          // this lane's part of the B-operand offset
    const uint32_t posx = (uint32_t)(kq & 1);
    // (a pooled layer's B operand is loaded raw -- value, argmax byte -- and selected just before it is multiplied, so that the
    //  loads stay in flight behind the other operand set's matrix instructions)
    auto load_ops = [&](const float* tin, int it, float (&a)[RBW], float (&b)[NCOB], uint32_t (&bm)[NCOB + 1]) __attribute__((always_inline)) {
        const float* tdy = tin + INF;
        const int grp = nph * it + ph;
        const int t = grp >> lgpi, rem = grp & (gpi - 1), y = rem >> lgw, x0 = (rem & ((W >> 2) - 1)) << 2;
        const float* ap = tin + (t * IMS + (y * PW + x0) * csl);
#pragma unroll
        for (int r = 0; r < RBW; ++r) a[r] = *(const float*)((const char*)ap + aoff[r]);
        if constexpr (POOLED) {
            const int pp = (t * (th >> 1) + (y >> 1)) * (W >> 1) + (x0 >> 1);                  // (uniform)
            const float* bp = tdy + pp * cwl + bl;
            const uint8_t* mp = (const uint8_t*)(tdy + DYF) + pp * cwl + bl;
            bm[NCOB] = (uint32_t)((y & 1) << 1) | posx;
#pragma unroll
            for (int c = 0; c < NCOB; ++c) {
                b[c] = bp[16 * c];
                bm[c] = mp[16 * c];
            }
        } else {
            const float* bp = tdy + ((t * th + y) * W + x0) * cwl + bl;
#pragma unroll
            for (int c = 0; c < NCOB; ++c) b[c] = bp[16 * c];
        }
    };
    auto select = [&](float (&b)[NCOB], const uint32_t (&bm)[NCOB + 1]) __attribute__((always_inline)) {
        if constexpr (POOLED) {
#pragma unroll
            for (int c = 0; c < NCOB; ++c) b[c] = bm[c] == bm[NCOB] ? b[c] : 0.f;
        }
    };
    auto mfmas = [&](const float (&a)[RBW], const float (&b)[NCOB]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < RBW; ++r)
#pragma unroll
            for (int c = 0; c < NCOB; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[c], acc[r][c], 0, 0, 0);
    };

    if (u0 < u1) { issue(u0); store(0); }
    __syncthreads();
    // (two copies of the loop, chosen once: only the waves of row group 0 in slice 0 add up the bias gradient)
    auto chunks = [&](auto bias_tag) __attribute__((always_inline)) {
        constexpr bool BIAS = decltype(bias_tag)::value;
        for (int u = u0; u < u1; ++u) {
            const int bf = (u - u0) & 1;
            const bool more = u + 1 < u1;
            if (more) issue(u + 1);
            const float* tin = sm + bf * BUF;
            // two operand sets: the loads of one are in flight behind the matrix instructions of the other (nit is even)
            float a0[RBW], b0[NCOB], a1[RBW], b1[NCOB];
            uint32_t m0[NCOB + 1], m1[NCOB + 1];
            load_ops(tin, 0, a0, b0, m0);
#pragma unroll 1
            for (int it = 0; it < nit; it += 2) {
                load_ops(tin, it + 1, a1, b1, m1);
                __builtin_amdgcn_sched_barrier(0);
                select(b0, m0);
                mfmas(a0, b0);
                if constexpr (BIAS) {
#pragma unroll
                    for (int c = 0; c < NCOB; ++c) bsum[c] += b0[c];
                }
                __builtin_amdgcn_sched_barrier(0);
                load_ops(tin, it + 2 < nit ? it + 2 : it, a0, b0, m0);
                __builtin_amdgcn_sched_barrier(0);
                select(b1, m1);
                mfmas(a1, b1);
                if constexpr (BIAS) {
#pragma unroll
                    for (int c = 0; c < NCOB; ++c) bsum[c] += b1[c];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) store(bf ^ 1);
            __syncthreads();
        }
    };
    if (do_bias) chunks(std::true_type{});
    else chunks(std::false_type{});

    // ---- the pixel phases summed through LDS (fixed order), one column block at a time; then the slab row: the waves of phase 0
    //      own their rows ----
    constexpr int NV = 4 * RBW + 1;
    float* row = P.slab + (size_t)g * (9 * ci_total * P.co + P.co);
#pragma unroll
    for (int c = 0; c < NCOB; ++c) {
        if (ph > 0) {
            float* red = sm + (size_t)((ph - 1) * nrg + rg) * NV * 64;
#pragma unroll
            for (int r = 0; r < RBW; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) red[(4 * r + j) * 64 + lane] = acc[r][c][j];
            red[4 * RBW * 64 + lane] = bsum[c];
        }
        __syncthreads();
        const int col = 16 * c + l15;
        if (ph == 0) {
            float bs = bsum[c];
            for (int p = 1; p < nph; ++p) {
                const float* red = sm + (size_t)((p - 1) * nrg + rg) * NV * 64;
#pragma unroll
                for (int r = 0; r < RBW; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[r][c][j] += red[(4 * r + j) * 64 + lane];
                bs += red[4 * RBW * 64 + lane];
            }
            if (col < cwl) {
#pragma unroll
                for (int r = 0; r < RBW; ++r) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int m = 16 * (rg + nrg * r) + 4 * kq + j;
                        if (m < 9 * csl) {
                            const int tap = m / csl, kp = ks0 + (m - tap * csl);        // padded channel -> real channel (or padding)
                            const int ci = kp < ca4 ? (kp < P.ca ? kp : -1) : P.ca + (kp - ca4);
                            if (ci >= 0) row[((size_t)tap * ci_total + ci) * P.co + cs0 + col] = acc[r][c][j];
                        }
                    }
                }
            }
            if (do_bias) {
                bs += __shfl_xor(bs, 16, 64);
                bs += __shfl_xor(bs, 32, 64);
                if (kq == 0 && col < cwl) row[(size_t)9 * ci_total * P.co + cs0 + col] = bs;
            }
        }
        __syncthreads();
    }
}

struct GenWrPlan { int ok, nsl, cs, ncs, cw, ncob, rbw, nrg, G; };
// row groups and row blocks per wave for a slice of cs <= 48 (padded) channels: the instantiated block counts are 2, 3, 6, 7
static void gen_wr_rows(int cs, int& nrg, int& rbw) {
    const int nrb = (9 * cs + 15) / 16;
    nrg = nrb > 14 ? 4 : (nrb > 7 ? 2 : 1);
    const int r = (nrb + nrg - 1) / nrg;
    rbw = r <= 2 ? 2 : (r <= 3 ? 3 : (r <= 6 ? 6 : 7));
}

// slicing of the channels (depends on the channel counts only: the slab count must be known without the map size)
static GenWrPlan gen_wr_plan(int n, int ca, int cb, int co) {
    GenWrPlan p{};
    const bool narrow = cb == 0 && ca < 4;            // the frames: 9 x 3 rows, unpadded
    const int ci = narrow ? ca : ((ca + 3) & ~3) + cb; // padded channel space
    if ((cb & 3) || (co & 3) || ci < 1) return p;
    if (narrow) {
        p.nsl = 1; p.cs = ca;
        gen_wr_rows(p.cs, p.nrg, p.rbw);
    } else {
        // input-channel slices: <= 48 channels (27 row blocks, 7 per row group); the candidate with the fewest computed rows wins
        int best = 0, bestrows = 1 << 30;
        const int nmin = (ci + 47) / 48;
        for (int nsl = nmin; nsl <= nmin + 2 && nsl * 4 <= ci + 3; ++nsl) {
            const int cs = (((ci + nsl - 1) / nsl) + 3) & ~3;
            if ((nsl - 1) * cs >= ci) continue;
            int nrg, rbw;
            gen_wr_rows(cs, nrg, rbw);
            const int rows = nsl * nrg * rbw;
            if (rows < bestrows) { bestrows = rows; best = nsl; }
        }
        if (!best) return p;
        p.nsl = best;
        p.cs = (((ci + best - 1) / best) + 3) & ~3;
        gen_wr_rows(p.cs, p.nrg, p.rbw);
    }
    // output-channel slices: <= 48 channels (the LDS budget of the 4x4 maps: 8 image slots of 6 x 6 x 48 floats + dY, twice)
    p.ncs = (co + 47) / 48;
    p.cw = (((co + p.ncs - 1) / p.ncs) + 3) & ~3;
    p.ncob = p.cw <= 16 ? 1 : 3;
    int G = 256 / (p.nsl * p.ncs);
    if (G < 1) G = 1;
    if (G > n) G = n;
    p.G = G;
    p.ok = 1;
    return p;
}

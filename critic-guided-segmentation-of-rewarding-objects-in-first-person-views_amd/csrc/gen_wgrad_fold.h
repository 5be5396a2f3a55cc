// Weight gradient of the NEAREST-UPSAMPLED source of a decoder layer, folded (included by gen_train.hip inside its anonymous namespace):
//   layer input = cat(A [ca], nearest-up_2(B [cb])) (nets.py:480-489, 501-513 under autograd); this kernel writes dW[tap][ca + ci][co] for B's channels,
//   gen_wgrad_rows_kernel (cb = 0, the slab strides of the whole layer) writes A's rows and the bias row of the same slab rows.
// A pixel of parity (py, px) reads up(B) only through the 2 x 2 low-resolution cells (a, b) around it, so the nine taps collapse to four folds
// per parity class:   dWf[py][px][a][b][ci][co] = sum over the class's pixels of B[(y >> 1) + a - 1 + py][(x >> 1) + b - 1 + px][ci] dY[y][x][co],
//                     dW[ky][kx] = sum_{py, px} dWf[py][px][a(py, ky)][b(px, kx)],   a(0, .) = {0, 1, 1}, a(1, .) = {0, 0, 1}
// -- 4 cb instead of 9 cb GEMM rows per pixel, B staged at its OWN resolution (a quarter of the elements, no upsampling in the loader), and
// 4 cb is a whole number of 16-row blocks for every cb % 4 == 0 (40 channels: 10 blocks per class, nothing padded).
// v_mfma_f32_16x16x4_f32: rows = 16 (fold, ci) pairs, columns = 16 output channels, K = 4 pixels OF ONE CLASS (the same position of four
// consecutive cells of a row).  Workgroup = 8 waves = 4 parity classes x 2 pixel phases; a wave owns all RB = cb / 4 row blocks x NCOB column
// blocks of its class (RB reads of A + NCOB of B per RB x NCOB matrix instructions).  One workgroup per CU, persistent over a contiguous
// range of chunks (th rows of one image); the next chunk's global loads fly during the matrix instructions of the current one.
// NARROW (ca <= 3: masker.0's frames): A's 9 ca <= 27 rows (two more row blocks per wave, the full-resolution frame tile beside the others)
// and the bias row (the sum of the B operand) come from this kernel too -- the row-block kernel would stage all of dY a second time for them.
#pragma once
#ifndef GWF_OPAQUE
#define GWF_OPAQUE 0
#endif

struct GenWfParams {
    const void* a; int a_u8;              // NARROW only: A [n,hw,hw,ca] (ca <= 3: the frames), fp32 or uint8 (/255)
    const float* b; const float* dy;      // B [n,hw/2,hw/2,cb];  dY [n,hw,hw,co]
    float* slab;                          // [G][9 * ci_total * co + co]
    int n, hw, lw, ca, cb, co;
    int G, ncs, cw;                       // chunk shares; output-channel slices of cw channels
    int th, parts, units;                 // chunk = th rows of one image; parts = hw / th; units = chunks in the job
    int ps, ds, buf_floats;               // LDS pixel strides (floats) of the B tile / the dY tile; floats per buffer
};

template <int RB, int NCOB, bool NARROW>
__global__ void __launch_bounds__(512) gen_wgrad_fold_kernel(GenWfParams P) {
    if (CGS_KARG_PREFETCH) cgs_kernarg_prefetch<sizeof(GenWfParams)>();
    extern __shared__ __attribute__((aligned(16))) float4 gfsm[];
    float* const sm = (float*)gfsm;
    constexpr int CS = 4 * RB, Q4 = RB;                                  // B's channels; quads per pixel
    constexpr int KB = 4, KD = 5, KA = NARROW ? 2 : 0;                   // staged 16-byte items per thread (host: the chunk fits)
    constexpr int NAB = NARROW ? 2 : 0;                                  // A's row blocks (NARROW)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, kq = lane >> 4;
    const int cls = wave & 3, ph = wave >> 2, py = cls >> 1, px = cls & 1;
    const int W = P.hw, lw = P.lw, h = W >> 1, lh = lw - 1, PWB = h + 2, th = P.th, thb = (th >> 1) + 2;
    const int cos = blockIdx.x % P.ncs, g = blockIdx.x / P.ncs;
    const int cs0 = cos * P.cw, cwl = min(P.cw, P.co - cs0), qd = cwl >> 2;
    const int ps = P.ps, ds = P.ds, BUF = P.buf_floats, BTF = thb * PWB * ps;      // (the dY tile follows the B tile)
    const int PWA = W + 2, DTF = th * W * ds + 64;                                 // NARROW: the frame tile [(th + 2)][W + 2][4 floats] follows the dY tile and its slack
    const int ci_total = P.ca + P.cb;
    const int u0 = (int)((long)g * P.units / P.G), u1 = (int)((long)(g + 1) * P.units / P.G);

    // ---- this lane's A-operand offsets (bytes): row m = 16 rb + l15 = (fold f = 2 a + b, ci) ----
    int aoff[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int m = 16 * rb + l15, f = m / CS, ci = m - f * CS;
        aoff[rb] = 4 * ((((f >> 1) + py) * PWB + (f & 1) + px + kq) * ps + ci);
    }
    // NARROW: A's rows m = 16 fb + l15 = (tap, c), 27 at most; rows past 9 ca read a valid address and are dropped at the end
    [[maybe_unused]] int aoffa[NAB > 0 ? NAB : 1];
    if constexpr (NARROW) {
#pragma unroll
        for (int fb = 0; fb < NAB; ++fb) {
            const int m = 16 * fb + l15, mm = m < 9 * P.ca ? m : 0, tap = mm / P.ca, c = mm - tap * P.ca, ky = tap / 3, kx = tap - 3 * ky;
            aoffa[fb] = 4 * (((py + ky) * PWA + px + kx + 2 * kq) * 4 + c);
        }
    }
    const int boff = (py * W + px + 2 * kq) * ds + l15;                  // B operand: pixel (2 Yl + py, 2 (X0 + kq) + px), column 16 c + l15
    frag4 acc[RB][NCOB];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int c = 0; c < NCOB; ++c) acc[r][c] = frag4{0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] frag4 acca[NAB > 0 ? NAB : 1][NCOB];
    [[maybe_unused]] float bsum[NCOB];
#pragma unroll
    for (int c = 0; c < NCOB; ++c) {
        bsum[c] = 0.f;
#pragma unroll
        for (int r = 0; r < (NAB > 0 ? NAB : 1); ++r) acca[r][c] = frag4{0.f, 0.f, 0.f, 0.f};
    }

    // the halo columns of the B tile are zero for every chunk (both buffers); so is the slack behind a buffer's dY tile
    for (int e = tid; e < 2 * thb * 2 * Q4; e += 512) {
        const int q = e % Q4, side = (e / Q4) & 1, r = (e / (2 * Q4)) % thb, bf = e / (2 * Q4 * thb);
        *(float4*)(sm + bf * BUF + (r * PWB + (side ? h + 1 : 0)) * ps + 4 * q) = f4zero();
    }
    if (tid < 32) {
        *(float4*)(sm + BUF - 64 + 4 * (tid & 15) + (tid >> 4) * BUF) = f4zero();
    }
    if constexpr (NARROW) {
        for (int e = tid; e < 2 * (th + 2) * 2; e += 512) {
            const int side = e & 1, r = (e >> 1) % (th + 2), bf = e / (2 * (th + 2));
            *(float4*)(sm + bf * BUF + BTF + DTF + (r * PWA + (side ? W + 1 : 0)) * 4) = f4zero();
        }
    }

    // ---- staging: item = 16 bytes; B items (tile row r, low-resolution column, quad), dY items (tile row, column, quad) ----
    const int NBI = thb * h * Q4, NDI = th * W * qd;
    const uint32_t mqd = 65536u / (uint32_t)qd + 1u;                     // item / qd by a multiply (exact below 5461 items)
    const int NAI = (th + 2) * W;                                        // NARROW: frame-tile pixels
    float4 st[KB + KD + KA];
    [[maybe_unused]] uint8_t st8[KA > 0 ? KA : 1][3];                     // NARROW, uint8 frames: the raw bytes (converted in store(): v_cvt_f32_ubyte0 needs no mask)
    auto issue = [&](int u) __attribute__((always_inline)) {
        int ltid = tid;
        if (GWF_OPAQUE || (NARROW && NCOB == 3 && RB >= 8)) asm volatile("" : "+v"(ltid));
        const int img = u / P.parts, row0 = (u % P.parts) * th, yb0 = (row0 >> 1) - 1;
        // (no branch around a load and no use of a loaded value before the last load is issued)
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int e = ltid + 512 * k, pix = e / Q4, q = e - pix * Q4, xc = pix & (h - 1), r = pix >> lh, yb = yb0 + r;
            const bool in = e < NBI && yb >= 0 && yb < h;
            const uint32_t off = (uint32_t)((img * h + yb) * h + xc) * (uint32_t)P.cb + 4u * q;        // (< 2^31 floats: host)
            st[k] = *(const float4*)(P.b + (in ? off : 0u));
        }
#pragma unroll
        for (int k = 0; k < KD; ++k) {
            const int e = ltid + 512 * k, pix = (int)(((uint32_t)e * mqd) >> 16), q = e - pix * qd;
            const bool in = e < NDI;
            const uint32_t off = (uint32_t)((img * W + row0) * W + pix) * (uint32_t)P.co + cs0 + 4u * q;
            st[KB + k] = *(const float4*)(P.dy + (in ? off : 0u));
        }
        if constexpr (NARROW) {
#pragma unroll
            for (int k = 0; k < KA; ++k) {
                const int e = ltid + 512 * k, xa = e & (W - 1), r = e >> lw, y = row0 - 1 + r;
                const bool in = e < NAI && y >= 0 && y < W;
                const uint32_t off = in ? (uint32_t)((img * W + y) * W + xa) * (uint32_t)P.ca : 0u;
                const int c1 = P.ca > 1 ? 1 : 0, c2 = P.ca > 2 ? 2 : 0;
                // RAW bits only (converted in store()): a conversion here is a use of the loaded value -- the wait for it would also drain the B / dY
                // loads issued above, and the whole prefetch with them.  Both forms are loaded (the one that is not the source's from offset 0): two
                // exclusive branches loading into the same registers make the second one wait for everything in flight.
                const uint8_t* s8 = (const uint8_t*)P.a + (P.a_u8 ? off : 0u);
                const float* s32 = (const float*)P.a + (P.a_u8 ? 0u : off);
                st[KB + KD + k] = make_float4(s32[0], s32[c1], s32[c2], 0.f);
                st8[k][0] = s8[0]; st8[k][1] = s8[c1]; st8[k][2] = s8[c2];
            }
        }
    };
    auto store = [&](int bf, int u) __attribute__((always_inline)) {
        int ltid = tid;
        if (GWF_OPAQUE || (NARROW && NCOB == 3 && RB >= 8)) asm volatile("" : "+v"(ltid));
        const int row0 = (u % P.parts) * th, yb0 = (row0 >> 1) - 1;
        float* tb = sm + bf * BUF;
        float* td = tb + BTF;
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int e = ltid + 512 * k, pix = e / Q4, q = e - pix * Q4, xc = pix & (h - 1), r = pix >> lh, yb = yb0 + r;
            if (e < NBI) *(float4*)(tb + (r * PWB + 1 + xc) * ps + 4 * q) = (yb >= 0 && yb < h) ? st[k] : f4zero();
        }
#pragma unroll
        for (int k = 0; k < KD; ++k) {
            const int e = ltid + 512 * k, pix = (int)(((uint32_t)e * mqd) >> 16), q = e - pix * qd;
            // (unconditional store: items past the tile land in the buffer's slack -- behind a branch the compiler keeps std_ in scratch memory)
            *(float4*)(e < NDI ? td + pix * ds + 4 * q : tb + BUF - 16) = st[KB + k];
        }
        if constexpr (NARROW) {
            float* ta = td + DTF;
            const float sc = P.a_u8 ? 1.f / 255.f : 1.f;
#pragma unroll
            for (int k = 0; k < KA; ++k) {
                const int e = ltid + 512 * k, xa = e & (W - 1), r = e >> lw, y = row0 - 1 + r;
                const bool in = y >= 0 && y < W;
                float4 v = st[KB + KD + k];
                if (P.a_u8) v = make_float4((float)st8[k][0], (float)st8[k][1], (float)st8[k][2], 0.f);
                v = make_float4(in ? v.x * sc : 0.f, in && P.ca > 1 ? v.y * sc : 0.f, in && P.ca > 2 ? v.z * sc : 0.f, 0.f);
                *(float4*)(e < NAI ? ta + (r * PWA + 1 + xa) * 4 : tb + BUF - 16) = v;
            }
        }
    };

    // ---- matrix loop: k-step s of a chunk = cells (Yl, X0 .. X0 + 3) of this wave's class; the phases take alternate steps ----
    const int lgx = lh - 2, nst = ((th >> 1) << lgx) >> 1;              // cell groups per cell row (log2); steps per phase (even: host)
    auto load_ops = [&](const float* tb, int it, float (&a)[RB + NAB], float (&b)[NCOB]) __attribute__((always_inline)) {
        const int s = 2 * it + ph, yl = s >> lgx, x0 = (s & ((1 << lgx) - 1)) << 2;
        const char* ap = (const char*)(tb + (yl * PWB + x0) * ps);
        const float* bp = tb + BTF + (2 * yl * W + 2 * x0) * ds + boff;
#pragma unroll
        for (int r = 0; r < RB; ++r) a[r] = *(const float*)(ap + aoff[r]);
        if constexpr (NARROW) {
            const char* aa = (const char*)(tb + BTF + DTF + (2 * yl * PWA + 2 * x0) * 4);
#pragma unroll
            for (int r = 0; r < NAB; ++r) a[RB + r] = *(const float*)(aa + aoffa[r]);
        }
#pragma unroll
        for (int c = 0; c < NCOB; ++c) b[c] = bp[16 * c];
    };
    auto mfmas = [&](const float (&a)[RB + NAB], const float (&b)[NCOB]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < NCOB; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[c], acc[r][c], 0, 0, 0);
        if constexpr (NARROW) {
#pragma unroll
            for (int r = 0; r < NAB; ++r)
#pragma unroll
                for (int c = 0; c < NCOB; ++c) acca[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[RB + r], b[c], acca[r][c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NCOB; ++c) bsum[c] += b[c];
        }
    };

    if (u0 < u1) { issue(u0); store(0, u0); }
    __syncthreads();
    for (int u = u0; u < u1; ++u) {
        const int bf = (u - u0) & 1;
        const bool more = u + 1 < u1;
        if (more) issue(u + 1);
        const float* tb = sm + bf * BUF;
        float a0[RB + NAB], b0[NCOB], a1[RB + NAB], b1[NCOB];
        load_ops(tb, 0, a0, b0);
#pragma unroll 1
        for (int it = 0; it < nst; it += 2) {
            load_ops(tb, it + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            load_ops(tb, it + 2 < nst ? it + 2 : it, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) store(bf ^ 1, u + 1);
        __syncthreads();
    }

    // ---- F[class][m = (fold, ci)][col] in LDS: phase 1 writes, phase 0 adds its own (fixed order), then every thread unfolds ----
    constexpr int CWP = 16 * NCOB;
    float* F = sm;
    float* FA = F + 16 * CS * CWP;                       // NARROW: [class][32 rows][CWP], then the bias partial sums [8 waves][4 kq][CWP]
    float* FB = FA + 4 * 32 * CWP;
    auto fidx = [&](int rb, int j, int c) { return ((cls * 4 * CS) + 16 * rb + 4 * kq + j) * CWP + 16 * c + l15; };
    auto faidx = [&](int fb, int j, int c) { return (cls * 32 + 16 * fb + 4 * kq + j) * CWP + 16 * c + l15; };
    if constexpr (NARROW) {
#pragma unroll
        for (int c = 0; c < NCOB; ++c) FB[(wave * 4 + kq) * CWP + 16 * c + l15] = bsum[c];
        if (ph == 1) {
#pragma unroll
            for (int fb = 0; fb < NAB; ++fb)
#pragma unroll
                for (int c = 0; c < NCOB; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) FA[faidx(fb, j, c)] = acca[fb][c][j];
        }
    }
    if (ph == 1) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int c = 0; c < NCOB; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) F[fidx(rb, j, c)] = acc[rb][c][j];
    }
    __syncthreads();
    if (ph == 0) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int c = 0; c < NCOB; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int i = fidx(rb, j, c); F[i] = acc[rb][c][j] + F[i]; }
        if constexpr (NARROW) {
#pragma unroll
            for (int fb = 0; fb < NAB; ++fb)
#pragma unroll
                for (int c = 0; c < NCOB; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int i = faidx(fb, j, c); FA[i] = acca[fb][c][j] + FA[i]; }
        }
    }
    __syncthreads();
    float* row = P.slab + (size_t)g * (9 * ci_total * P.co + P.co);
    for (int e = tid; e < 9 * CS * cwl; e += 512) {
        const int rc = e / cwl, col = e - rc * cwl;
        const int tap = rc / CS, ci = rc - tap * CS, ky = tap / 3, kx = tap - 3 * ky;
        float v = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {                                  // classes in the order (0,0) (0,1) (1,0) (1,1)
            const int cy = c4 >> 1, cx = c4 & 1;
            const int a = cy ? (ky == 2 ? 1 : 0) : (ky == 0 ? 0 : 1), b = cx ? (kx == 2 ? 1 : 0) : (kx == 0 ? 0 : 1);
            v += F[((c4 * 4 + 2 * a + b) * CS + ci) * CWP + col];
        }
        row[((size_t)tap * ci_total + P.ca + ci) * P.co + cs0 + col] = v;
    }
    if constexpr (NARROW) {
        for (int e = tid; e < (9 * P.ca + 1) * cwl; e += 512) {
            const int m = e / cwl, col = e - m * cwl;
            float v = 0.f;
            if (m < 9 * P.ca) {                        // A's rows: the four classes in order
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) v += FA[(c4 * 32 + m) * CWP + col];
                const int tap = m / P.ca, c = m - tap * P.ca;
                row[((size_t)tap * ci_total + c) * P.co + cs0 + col] = v;
            } else {                                   // the bias row: 8 waves x 4 pixel lanes in order
                for (int i = 0; i < 32; ++i) v += FB[i * CWP + col];
                row[(size_t)9 * ci_total * P.co + cs0 + col] = v;
            }
        }
    }
}

// Body of the MFMA implicit-GEMM weight-gradient kernels + instance table.  Included by conv_wgrad.hip and
// fused_small.hip.
#pragma once
#include "conv_tile.h"

typedef float frag4 __attribute__((ext_vector_type(4)));

struct WgradParams {
    const void* src_a;
    const float* src_b;
    const float* dy;
    const uint32_t* amask;
    float* slab;
    int n, ntiles;
    cgs_dropout drop;
    // WSRC_MIX: source A = the replaced | injected mixes computed on the fly (see conv_tile.h load_a_mix)
    const uint8_t* mix_a; const uint8_t* mix_b; const float* mix_z;
    int mix_n_a;
};

template <int H_, int W_, int TH_, int IMGS_, int THREADS_>
struct WGeo {
    static constexpr int H = H_, W = W_, TH = TH_, IMGS = IMGS_, THREADS = THREADS_, LT = THREADS_;
    static constexpr int RQ = TH / 2, QW = W / 2, QH = H / 2;  // for load_poolexp
    static constexpr int TRA = TH + 2, PWA = W + 2, STRIPS = H / TH;
    static constexpr int NSTEP = IMGS * TH * W / 4, NW = THREADS / 64;
    static_assert(IMGS == 1 || TH == H, "multi-image tiles hold whole images");
    static_assert((TH * W) % 4 == 0 && H % TH == 0, "tile shape");
};

enum { WSRC_F32 = 0, WSRC_U8 = 1, WSRC_MIX = 2, WDY_F32 = 0, WDY_POOLEXP = 1 };

// C: G (WGeo), SRC (WSRC_*), CA, CB, UPS, CO, DY (WDY_*)
// X tile: [img][TRA][PWA][PCI] floats, PCI = 4*(SA+SB): source A occupies SA float4 slots per pixel
// (3-channel images are padded to one slot), source B the following SB slots.
// Body: processes tiles tile0, tile0+tstride, ... < tend and writes ONE slab (sum over the workgroup's waves).
template <class C>
__device__ __forceinline__ void wgrad_body(const WgradParams& P, const int tile0, const int tstride, const int tend,
                                           float* slab, float4* smem) {
    using G = typename C::G;
    constexpr int CI = C::CA + C::CB, CO = C::CO;
    constexpr int SA = (C::CA + 3) / 4, SB = C::CB / 4, S = SA + SB, PCI = 4 * S;
    constexpr int ROWS = 9 * CI + 1, NRB = (ROWS + 15) / 16;
    constexpr int NPIX = G::IMGS * G::TRA * G::PWA;      // tile pixels incl. halo
    constexpr int XT4 = NPIX * S;                          // float4 slots
    constexpr int YT = G::IMGS * G::TH * G::W * CO;
    static_assert(CO <= 16 && CO % 4 == 0, "one 16-wide column block");
    float* xt = (float*)smem;
    float* yt = xt + XT4 * 4;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int N = P.n;
    const DropCtx dc = drop_ctx(P.drop);

    // Work split over the workgroup's waves.  Default: every wave keeps all row blocks and the waves split the tile ROWS (k
    // dimension) -> cross-wave sum at the end.  Small maps with many row blocks (dec_model.2/3: 14-28 blocks, 4x4 / 8x8
    // pixels): the waves split the ROW BLOCKS and each walks all pixels -> no reduction (it cost more than the MFMAs
    // there), each wave writes its rows of the slab itself.  (On the 16x16 / 32x32 decoder layers the uneven block split
    // -- 10 blocks over 4 waves -- costs more than the reduction saves: measured 37 vs 30 us on dec_model.0.)
    constexpr bool MSPLIT = NRB >= 2 * G::NW && G::H <= 8;
    constexpr int NRBW = MSPLIT ? (NRB + G::NW - 1) / G::NW : NRB;     // row blocks held by one wave
    int rbase[NRBW];
#pragma unroll
    for (int rbi = 0; rbi < NRBW; ++rbi) {
        const int rb = MSPLIT ? wave + rbi * G::NW : rbi;
        int r = rb * 16 + l15;
        if (MSPLIT && rb >= NRB) r = ROWS + 16;      // a wave's surplus slot: padding rows
        if (r < 9 * CI) {
            int tap = r / CI, ci = r % CI;
            int lch = ci < C::CA ? ci : 4 * SA + (ci - C::CA);
            rbase[rbi] = ((tap / 3) * G::PWA + (tap % 3)) * PCI + lch;
        } else {
            rbase[rbi] = (r == 9 * CI) ? -1 : -2;
        }
    }
    frag4 acc[NRBW];
#pragma unroll
    for (int rb = 0; rb < NRBW; ++rb) acc[rb] = frag4{0.f, 0.f, 0.f, 0.f};

    // ------------------------------------------------------------------------------------------
    // Software pipeline over this workgroup's tiles: the global loads of tile t+1 are issued into
    // registers (fetch) right after tile t has been committed to LDS, so they are in flight during
    // tile t's MFMA phase; commit() writes them to LDS after the phase ends.
    // ------------------------------------------------------------------------------------------
    constexpr int NA = (C::CA % 4 == 0) ? NPIX * SA : NPIX;
    constexpr int NB = NPIX * SB;
    constexpr int PO = CO / 4;
    constexpr int NY = (C::DY == WDY_F32) ? YT / 4 : G::IMGS * G::RQ * (G::W / 2) * PO;
    constexpr int ITA = (NA + G::THREADS - 1) / G::THREADS, ITB = (NB + G::THREADS - 1) / G::THREADS;
    constexpr int ITY = (NY + G::THREADS - 1) / G::THREADS;
    float4 ra[ITA], rb[ITB > 0 ? ITB : 1], ry[ITY];
    uint32_t rn[ITY];
    [[maybe_unused]] float rz[ITA];     // WSRC_MIX: the mask value of the pixel

    auto tile_origin = [&](int tile, int& n0, int& row0) {
        n0 = (G::IMGS == 1) ? tile / G::STRIPS : tile * G::IMGS;
        row0 = (G::IMGS == 1) ? (tile % G::STRIPS) * G::TH : 0;
    };
    auto pix_decode = [&](int e, int n0, int row0, int& n, int& y, int& x) -> bool {
        int c = e % G::PWA, r = (e / G::PWA) % G::TRA, img = e / (G::PWA * G::TRA);
        n = n0 + img; y = row0 + r - 1; x = c - 1;
        return n < N && y >= 0 && y < G::H && x >= 0 && x < G::W;
    };

    auto fetch = [&](int tile) {
        int n0, row0;
        tile_origin(tile, n0, row0);
#pragma unroll
        for (int it = 0; it < ITA; ++it) {
            int e = tid + it * G::THREADS; e = e < NA ? e : NA - 1;
            int n, y, x;
            if constexpr (C::CA % 4 == 0) {
                bool in = pix_decode(e / SA, n0, row0, n, y, x);
                int gi = in ? ((n * G::H + y) * G::W + x) * SA + (e % SA) : 0;
                ra[it] = ((const float4*)P.src_a)[gi];
            } else {
                bool in = pix_decode(e, n0, row0, n, y, x);
                int pix = in ? (n * G::H + y) * G::W + x : 0;
                if constexpr (C::SRC == WSRC_MIX) {
                    // raw dwords only (decoded in commit): A (lo, hi), B (lo, hi) around the pixel's 3 bytes, and z
                    const bool inj = in && n >= P.mix_n_a;
                    const int spix = in ? ((inj ? n - P.mix_n_a : n) * G::H + y) * G::W + x : 0;
                    const uint32_t* a32 = (const uint32_t*)P.mix_a;
                    const uint32_t* b32 = (const uint32_t*)P.mix_b;
                    const int off = spix * 3, last = P.mix_n_a * G::H * G::W * 3 / 4 - 1, d = off >> 2, d1 = d + 1 <= last ? d + 1 : last;
                    ra[it] = make_float4(__uint_as_float(a32[d]), __uint_as_float(a32[d1]), __uint_as_float(b32[d]), __uint_as_float(b32[d1]));
                    rz[it] = P.mix_z[spix];
                } else if constexpr (C::SRC == WSRC_U8) {
                    const uint32_t* s32 = (const uint32_t*)P.src_a;
                    int off = pix * 3, last = N * G::H * G::W * 3 / 4 - 1;
                    int d = off >> 2;
                    uint32_t lo = s32[d], hi = s32[d + 1 <= last ? d + 1 : last];
                    uint64_t both = (((uint64_t)hi << 32) | lo) >> ((off & 3) * 8);
                    const float sc = 1.f / 255.f;
                    const uint32_t b3 = (uint32_t)both;      // (32-bit: a uint64_t -> float conversion is ~30 instructions, v_cvt_f32_ubyteN one)
                    ra[it] = make_float4((b3 & 255u) * sc, ((b3 >> 8) & 255u) * sc, ((b3 >> 16) & 255u) * sc, 0.f);
                } else {
                    const float* sf = (const float*)P.src_a;
                    ra[it] = make_float4(sf[pix * 3], sf[pix * 3 + 1], sf[pix * 3 + 2], 0.f);
                }
            }
        }
        if constexpr (SB > 0) {
#pragma unroll
            for (int it = 0; it < ITB; ++it) {
                int e = tid + it * G::THREADS; e = e < NB ? e : NB - 1;
                int n, y, x;
                bool in = pix_decode(e / SB, n0, row0, n, y, x);
                int gi;
                if constexpr (C::UPS == 2) gi = in ? ((n * G::QH + (y >> 1)) * G::QW + (x >> 1)) * SB + (e % SB) : 0;
                else gi = in ? n * SB + (e % SB) : 0;
                rb[it] = ((const float4*)P.src_b)[gi];
            }
        }
#pragma unroll
        for (int it = 0; it < ITY; ++it) {
            int e = tid + it * G::THREADS; e = e < NY ? e : NY - 1;
            if constexpr (C::DY == WDY_F32) {
                constexpr int PER = G::TH * G::W * CO / 4;
                int n = n0 + e / PER;
                ry[it] = ((const float4*)P.dy)[n < N ? ((n * G::H + row0) * G::W * CO) / 4 + e % PER : 0];
            } else {
                constexpr int HP = G::H / 2, WP = G::W / 2, AMW = (PO + 1) / 2;
                int p = e % PO, px = (e / PO) % WP, j = (e / (PO * WP)) % G::RQ, img = e / (PO * WP * G::RQ);
                int n = n0 + img, pr = row0 / 2 + j;
                bool in = n < N;
                int pi = in ? (n * HP + pr) * WP + px : 0;
                ry[it] = ((const float4*)P.dy)[pi * PO + p];
                rn[it] = in ? ((P.amask[pi * AMW + (p >> 1)] >> ((p & 1) * 16)) & 0xFFFFu) : 0xFFFFu;
            }
        }
    };

    auto commit = [&](int tile) {
        int n0, row0;
        tile_origin(tile, n0, row0);
#pragma unroll
        for (int it = 0; it < ITA; ++it) {
            int e = tid + it * G::THREADS; e = e < NA ? e : NA - 1;
            int n, y, x;
            if constexpr (C::CA % 4 == 0) {
                bool in = pix_decode(e / SA, n0, row0, n, y, x);
                float4 v = ra[it];
                if (dc.on) v = v * drop_mult4(dc, (uint32_t)(in ? ((n * G::H + y) * G::W + x) * SA + (e % SA) : 0));
                ((float4*)xt)[(e / SA) * S + (e % SA)] = in ? v : f4zero();
            } else {
                bool in = pix_decode(e, n0, row0, n, y, x);
                float4 v = ra[it];
                if constexpr (C::SRC == WSRC_MIX) {
                    const bool inj = in && n >= P.mix_n_a;
                    const int spix = in ? ((inj ? n - P.mix_n_a : n) * G::H + y) * G::W + x : 0;
                    const int sh = ((spix * 3) & 3) * 8;
                    uint64_t a6 = (((uint64_t)__float_as_uint(v.y) << 32) | __float_as_uint(v.x)) >> sh;
                    uint64_t b6 = (((uint64_t)__float_as_uint(v.w) << 32) | __float_as_uint(v.z)) >> sh;
                    if (inj) { uint64_t t = a6; a6 = b6; b6 = t; }
                    const float sc = 1.f / 255.f, zi = rz[it];
                    float m[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float av = (((uint32_t)a6 >> (8 * c)) & 255u) * sc, bv = (((uint32_t)b6 >> (8 * c)) & 255u) * sc;      // (32-bit conversions)
                        m[c] = av * (1.f - zi) + zi * bv;
                    }
                    v = make_float4(m[0], m[1], m[2], 0.f);
                }
                ((float4*)xt)[e * S] = in ? v : f4zero();
            }
        }
        if constexpr (SB > 0) {
#pragma unroll
            for (int it = 0; it < ITB; ++it) {
                int e = tid + it * G::THREADS; e = e < NB ? e : NB - 1;
                int n, y, x;
                bool in = pix_decode(e / SB, n0, row0, n, y, x);
                ((float4*)xt)[(e / SB) * S + SA + (e % SB)] = in ? rb[it] : f4zero();
            }
        }
#pragma unroll
        for (int it = 0; it < ITY; ++it) {
            int e = tid + it * G::THREADS; e = e < NY ? e : NY - 1;
            if constexpr (C::DY == WDY_F32) {
                constexpr int PER = G::TH * G::W * CO / 4;
                ((float4*)yt)[e] = (n0 + e / PER < N) ? ry[it] : f4zero();
            } else {
                constexpr int WP = G::W / 2;
                int p = e % PO, px = (e / PO) % WP, j = (e / (PO * WP)) % G::RQ, img = e / (PO * WP * G::RQ);
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                    int r = 2 * j + (pos >> 1), x = 2 * px + (pos & 1);
                    ((float4*)yt)[((img * G::TH + r) * G::W + x) * PO + p] = nib_select(ry[it], rn[it], pos);
                }
            }
        }
    };

    constexpr int U = NRBW >= 16 ? 1 : (NRBW >= 8 ? 2 : 4);
    int tile = tile0;
    if (tile < tend) fetch(tile);
    for (; tile < tend; tile += tstride) {
        commit(tile);
        __syncthreads();
        if (tile + tstride < tend) fetch(tile + tstride);   // in flight during the MFMA phase
        // Each wave owns whole tile rows; inside a row the k-steps advance by constant strides, so every LDS
        // address is (per-row base) + immediate offset: no per-step index arithmetic next to the MFMAs.
        constexpr int NROWS = G::IMGS * G::TH, SPR = G::W / 4;
        constexpr int UU = SPR < U ? SPR : U;
        for (int R = MSPLIT ? 0 : wave; R < NROWS; R += MSPLIT ? 1 : G::NW) {
            const int img = R / G::TH, yl = R % G::TH;
            const int xrow = ((img * G::TRA + yl) * G::PWA + kq) * PCI;
            const int yrow = ((img * G::TH + yl) * G::W + kq) * CO + (l15 < CO ? l15 : 0);
            int xa[NRBW];
#pragma unroll
            for (int rb = 0; rb < NRBW; ++rb) xa[rb] = xrow + (rbase[rb] >= 0 ? rbase[rb] : 0);
#pragma unroll
            for (int c = 0; c < SPR; c += UU) {
                float a[UU][NRBW], b[UU];
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    float bv = yt[yrow + (c + u) * 4 * CO];
                    b[u] = (l15 < CO) ? bv : 0.f;
#pragma unroll
                    for (int rb = 0; rb < NRBW; ++rb) {
                        float av = xt[xa[rb] + (c + u) * 4 * PCI];
                        a[u][rb] = (rbase[rb] >= 0) ? av : (rbase[rb] == -1 ? 1.f : 0.f);
                    }
                }
#pragma unroll
                for (int u = 0; u < UU; ++u)
#pragma unroll
                    for (int rb = 0; rb < NRBW; ++rb)
                        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][rb], b[u], acc[rb], 0, 0, 0);
                asm volatile("" ::: "memory");   // keep the next chunk's LDS reads behind this chunk's (bounded registers)
            }
        }
        __syncthreads();
    }

    // ---- sum the waves' accumulators through LDS (wave by wave), then one coalesced slab per workgroup ----
    // D layout: col = lane & 15 (= co), row = (lane >> 4) * 4 + reg (= r within the row block)
    if constexpr (MSPLIT) {
        // every wave owns its row blocks outright: straight to the slab (64-byte runs per row)
        if (l15 < CO) {
#pragma unroll
            for (int rbi = 0; rbi < NRBW; ++rbi)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = (wave + rbi * G::NW) * 16 + kq * 4 + j;
                    if (r < ROWS) slab[r * CO + l15] = acc[rbi][j];
                }
        }
        __syncthreads();   // the LDS region may be reused by a following stage
        return;
    }
    float* red = (float*)smem;
    static_assert(ROWS * CO <= XT4 * 4 + YT, "reduction buffer fits in the tile storage");
#pragma unroll 1
    for (int w = 0; w < G::NW; ++w) {
        if (wave == w && l15 < CO) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int r = rb * 16 + kq * 4 + j;
                    if (r < ROWS) red[r * CO + l15] = (w == 0 ? 0.f : red[r * CO + l15]) + acc[rb][j];
                }
        }
        __syncthreads();
    }
    for (int i = tid; i < ROWS * CO; i += G::THREADS) slab[i] = red[i];
    __syncthreads();   // the LDS region may be reused by a following stage
}

template <class C>
__global__ void __launch_bounds__(C::G::THREADS) wgrad_kernel(WgradParams P) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    constexpr int SLAB = (9 * (C::CA + C::CB) + 1) * C::CO;
    wgrad_body<C>(P, blockIdx.x, gridDim.x, P.ntiles, P.slab + (size_t)blockIdx.x * SLAB, smem);
}

template <class C>
static constexpr size_t wgrad_lds_bytes() {
    using G = typename C::G;
    constexpr int S = (C::CA + 3) / 4 + C::CB / 4;
    return ((size_t)G::IMGS * G::TRA * G::PWA * S * 4 + (size_t)G::IMGS * G::TH * G::W * C::CO + 4) * sizeof(float);
}

// masker.2 (16 -> 1): a single output channel would waste 15/16 of the MFMA columns, so the product is
// re-associated:  dW[tap][ci] = sum_{p'} X[p'][ci] * dY[p' - tapoffset]   (rows = ci, columns = tap).
template <class G>
__global__ void __launch_bounds__(G::THREADS) wgrad_co1_kernel(WgradParams P) {
    constexpr int CI = 16;
    constexpr int PW = 68;                       // haloed row (66) padded to a multiple of 4 positions
    constexpr int XT = G::TRA * PW * CI, YT = G::TH * G::W;
    constexpr int SPR = PW / 4;                  // k-steps per haloed row
    static_assert(G::IMGS == 1 && G::W == 64, "co1 tile");
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float* xt = (float*)smem;
    float* yt = xt + XT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int ky = l15 / 3, kx = l15 % 3;
    frag4 acc = frag4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    // register prefetch of the next tile (same pipeline as wgrad_body)
    constexpr int NX = XT / 4, NYY = YT / 4;
    constexpr int ITX = (NX + G::THREADS - 1) / G::THREADS, ITYY = (NYY + G::THREADS - 1) / G::THREADS;
    float4 rx[ITX], ryy[ITYY];
    auto fetch = [&](int tile) {
        const int n = tile / G::STRIPS, row0 = (tile % G::STRIPS) * G::TH;
#pragma unroll
        for (int it = 0; it < ITX; ++it) {
            int e = tid + it * G::THREADS; e = e < NX ? e : NX - 1;
            int q4 = e % (CI / 4), c = (e / (CI / 4)) % PW, r = e / ((CI / 4) * PW);
            int y = row0 + r - 1, x = c - 1;
            bool in = y >= 0 && y < G::H && x >= 0 && x < G::W;
            rx[it] = ((const float4*)P.src_a)[in ? ((n * G::H + y) * G::W + x) * (CI / 4) + q4 : 0];
        }
#pragma unroll
        for (int it = 0; it < ITYY; ++it) {
            int e = tid + it * G::THREADS; e = e < NYY ? e : NYY - 1;
            ryy[it] = ((const float4*)P.dy)[((n * G::H + row0) * G::W) / 4 + e];
        }
    };
    auto commit = [&](int tile) {
        const int row0 = (tile % G::STRIPS) * G::TH;
#pragma unroll
        for (int it = 0; it < ITX; ++it) {
            int e = tid + it * G::THREADS; e = e < NX ? e : NX - 1;
            int c = (e / (CI / 4)) % PW, r = e / ((CI / 4) * PW);
            int y = row0 + r - 1, x = c - 1;
            bool in = y >= 0 && y < G::H && x >= 0 && x < G::W;
            ((float4*)xt)[e] = in ? rx[it] : f4zero();
        }
#pragma unroll
        for (int it = 0; it < ITYY; ++it) {
            int e = tid + it * G::THREADS; e = e < NYY ? e : NYY - 1;
            ((float4*)yt)[e] = ryy[it];
        }
    };
    int tile = blockIdx.x;
    if (tile < P.ntiles) fetch(tile);
    for (; tile < P.ntiles; tile += gridDim.x) {
        commit(tile);
        __syncthreads();
        if (tile + (int)gridDim.x < P.ntiles) fetch(tile + gridDim.x);
        for (int r = wave; r < G::TRA; r += G::NW) {      // each wave owns whole haloed rows
            const int xrow = (r * PW + kq) * CI + l15;
            const int yl = r - ky;
            const bool rowok = l15 < 9 && yl >= 0 && yl < G::TH;
            const int yrow = (rowok ? yl : 0) * G::W + kq - kx;
#pragma unroll
            for (int c = 0; c < SPR; c += 4) {
                float a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (c + u < SPR) {
                        a[u] = xt[xrow + (c + u) * 4 * CI];
                        // only the first and the last k-step of a row can fall outside the image columns
                        constexpr bool edge_possible = true;
                        const bool interior = (c + u >= 1) && (c + u <= 15);
                        int x = 4 * (c + u) + kq - kx;
                        bool in = rowok && (interior || (x >= 0 && x < G::W));
                        float bv = yt[in ? yrow + 4 * (c + u) : 0];
                        b[u] = in ? bv : 0.f;
                        bsum += b[u];                 // only the centre-tap lanes' sums are used (dbias)
                        (void)edge_possible;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (c + u < SPR) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // cross-wave sum through LDS, one slab per workgroup: [tap][ci] (row = ci, col = tap) then dbias
    float* red = (float*)smem;
    bsum = (l15 == 4) ? bsum : 0.f;
    bsum = wave_sum(bsum);
#pragma unroll 1
    for (int w = 0; w < G::NW; ++w) {
        if (wave == w) {
            if (l15 < 9) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int i = l15 * CI + kq * 4 + j;
                    red[i] = (w == 0 ? 0.f : red[i]) + acc[j];
                }
            }
            if (lane == 0) red[9 * CI] = (w == 0 ? 0.f : red[9 * CI]) + bsum;
        }
        __syncthreads();
    }
    float* slab = P.slab + (size_t)blockIdx.x * (9 * CI + 1);
    for (int i = tid; i < 9 * CI + 1; i += G::THREADS) slab[i] = red[i];
}

#define CGS_WG_CFG(NAME, HW, TH_, IMGS_, SRC_, CA_, CB_, UPS_, CO_, DY_)      \
    struct NAME {                                                             \
        using G = WGeo<HW, HW, TH_, IMGS_, 256>;                              \
        static constexpr int SRC = SRC_, CA = CA_, CB = CB_, UPS = UPS_, CO = CO_, DY = DY_; \
    };

CGS_WG_CFG(WEnc0U8, 64, 8, 1, WSRC_U8, 3, 0, 2, 8, WDY_POOLEXP)
CGS_WG_CFG(WEnc0Mix, 64, 8, 1, WSRC_MIX, 3, 0, 2, 8, WDY_POOLEXP)
CGS_WG_CFG(WEnc0F32, 64, 8, 1, WSRC_F32, 3, 0, 2, 8, WDY_POOLEXP)
CGS_WG_CFG(WEnc1, 32, 16, 1, WSRC_F32, 8, 0, 2, 8, WDY_POOLEXP)
CGS_WG_CFG(WEnc2, 16, 16, 2, WSRC_F32, 8, 0, 2, 8, WDY_POOLEXP)
CGS_WG_CFG(WEnc3, 8, 8, 8, WSRC_F32, 8, 0, 2, 16, WDY_POOLEXP)
CGS_WG_CFG(WDec3, 4, 4, 4, WSRC_F32, 16, 32, 4, 16, WDY_F32)
CGS_WG_CFG(WDec2, 8, 8, 4, WSRC_F32, 8, 16, 2, 8, WDY_F32)
CGS_WG_CFG(WDec1, 16, 16, 1, WSRC_F32, 8, 8, 2, 8, WDY_F32)
CGS_WG_CFG(WDec0, 32, 8, 1, WSRC_F32, 8, 8, 2, 8, WDY_F32)
CGS_WG_CFG(WMask0U8, 64, 4, 1, WSRC_U8, 3, 8, 2, 16, WDY_F32)
CGS_WG_CFG(WMask0F32, 64, 4, 1, WSRC_F32, 3, 8, 2, 16, WDY_F32)
using WMask2G = WGeo<64, 64, 4, 1, 384>;   // 6 haloed rows, one per wave


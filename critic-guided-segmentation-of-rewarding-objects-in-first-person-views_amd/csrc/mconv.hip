// 3x3 convolution (forward / data gradient) as an implicit GEMM on the matrix cores, for the layers where the
// GEMM shape fills an MFMA tile: 16 output channels (masker.0 forward, dec_model.3 forward), also with many
// input channels at tiny spatial size (dec_model.3 forward: 48 -> 16 channels at 4x4).  (A data-gradient variant for
// dec_model.3 was measured slower than the shared-launch VALU kernel and is not kept.)
//
//   out[pixel][oc] = sum_k Xcol[pixel][k] * Wm[k][oc],  k = tap*PCI + channel   (v_mfma_f32_16x16x4_f32, exact fp32)
//
// A operand: 16 pixels x 4 k per instruction, read from an NHWC LDS tile with halo (concat of the skip input and
// the nearest-upsampled low-res input materialised, 3-channel images padded to 4): for a fixed k-step the address is
// (per-pixel base) + compile-time constant.  B operand: the weights, one register per (k-step, 16-channel block),
// loaded once per wave and kept for all tiles the (persistent) workgroup processes.
// The direct VALU kernels (conv_body.h) stay the choice for 8-channel layers, where half of an MFMA tile is padding.
#include "wgrad_body.h"   // WGeo, frag4, WSRC_*

struct MConvParams {
    const void* src_a;
    const float* src_b;
    const float* w;
    const float* bias;
    float* out;     // forward: output; dgrad: d_a
    float* out2;    // dgrad: d_b
    int n, ntiles;
};

enum { MEPI_PLAIN = 0 };

// C: G (WGeo), SRC, CA, CB, UPS, WT, WCI, WCO, NOUT, ACT, EPI, OUT_A
template <class C>
__global__ void __launch_bounds__(C::G::THREADS) mconv_kernel(MConvParams P) {
    using G = typename C::G;
    constexpr int SA = (C::CA + 3) / 4, SB = C::CB / 4, S = SA + SB, PCI = 4 * S;
    constexpr int NK = 9 * PCI / 4, NCB = (C::NOUT + 15) / 16;
    constexpr int NPIX = G::IMGS * G::TRA * G::PWA;
    constexpr int NT = G::IMGS * G::TH * G::W / 16;            // 16-pixel MFMA tiles per workgroup tile
    static_assert((G::TH * G::W) % 16 == 0 && (G::W >= 16 || (G::W == 4 && G::TH == 4)), "pixel tiling");
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float* xt = (float*)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int N = P.n;

    // ---- weights -> registers (B operand): lane (k = 4s+kq, oc = 16cb + l15) ----
    float wr[NK][NCB];
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        const int tap = (4 * s) / PCI, lch = (4 * s) % PCI + kq;     // PCI % 4 == 0: one tap per k-step
        int ci = lch < C::CA ? lch : ((lch >= 4 * SA) ? C::CA + (lch - 4 * SA) : -1);
        if (lch >= C::CA && lch < 4 * SA) ci = -1;                     // padding channel of a 3-channel image
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            int oc = cb * 16 + l15;
            float v = 0.f;
            if (ci >= 0 && oc < C::NOUT) {
                if constexpr (C::WT == 0) v = P.w[(tap * C::WCI + ci) * C::WCO + oc];
                else v = P.w[((8 - tap) * C::WCI + oc) * C::WCO + ci];
            }
            wr[s][cb] = v;
        }
    }

    for (int tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
        const int n0 = (G::IMGS == 1) ? tile / G::STRIPS : tile * G::IMGS;
        const int row0 = (G::IMGS == 1) ? (tile % G::STRIPS) * G::TH : 0;
        // ---- X tile (same layout as the weight-gradient kernels) ----
        if constexpr (C::CA % 4 == 0) {
            for_elems<NPIX * SA, G::THREADS>(tid, [&](int e) {
                int s = e % SA, c = (e / SA) % G::PWA, r = (e / (SA * G::PWA)) % G::TRA, img = e / (SA * G::PWA * G::TRA);
                int n = n0 + img, y = row0 + r - 1, x = c - 1;
                bool in = n < N && y >= 0 && y < G::H && x >= 0 && x < G::W;
                float4 v = ((const float4*)P.src_a)[in ? ((n * G::H + y) * G::W + x) * SA + s : 0];
                ((float4*)xt)[(e / SA) * S + s] = in ? v : f4zero();
            });
        } else {
            for_elems<NPIX, G::THREADS>(tid, [&](int e) {
                int c = e % G::PWA, r = (e / G::PWA) % G::TRA, img = e / (G::PWA * G::TRA);
                int n = n0 + img, y = row0 + r - 1, x = c - 1;
                bool in = n < N && y >= 0 && y < G::H && x >= 0 && x < G::W;
                int pix = in ? (n * G::H + y) * G::W + x : 0;
                float4 v;
                if constexpr (C::SRC == WSRC_U8) {
                    const uint32_t* s32 = (const uint32_t*)P.src_a;
                    int off = pix * 3, last = N * G::H * G::W * 3 / 4 - 1, d = off >> 2;
                    uint32_t lo = s32[d], hi = s32[d + 1 <= last ? d + 1 : last];
                    uint64_t both = (((uint64_t)hi << 32) | lo) >> ((off & 3) * 8);
                    const float sc = 1.f / 255.f;
                    v = make_float4((both & 255) * sc, ((both >> 8) & 255) * sc, ((both >> 16) & 255) * sc, 0.f);
                } else {
                    const float* sf = (const float*)P.src_a;
                    v = make_float4(sf[pix * 3], sf[pix * 3 + 1], sf[pix * 3 + 2], 0.f);
                }
                ((float4*)xt)[e * S] = in ? v : f4zero();
            });
        }
        if constexpr (SB > 0) {
            for_elems<NPIX * SB, G::THREADS>(tid, [&](int e) {
                int s = e % SB, c = (e / SB) % G::PWA, r = (e / (SB * G::PWA)) % G::TRA, img = e / (SB * G::PWA * G::TRA);
                int n = n0 + img, y = row0 + r - 1, x = c - 1;
                bool in = n < N && y >= 0 && y < G::H && x >= 0 && x < G::W;
                int gi;
                if constexpr (C::UPS == 2) gi = in ? ((n * G::QH + (y >> 1)) * G::QW + (x >> 1)) * SB + s : 0;
                else gi = in ? n * SB + s : 0;
                float4 v = ((const float4*)P.src_b)[gi];
                ((float4*)xt)[(e / SB) * S + SA + s] = in ? v : f4zero();
            });
        }
        __syncthreads();

        for (int t = wave; t < NT; t += G::NW) {
            // pixel of this lane's A row, and the 4 pixels of its D rows
            const int pa = 16 * t + l15;
            const int xa = pa % G::W, ya = (pa / G::W) % G::TH, ia = pa / (G::W * G::TH);
            const int abase = ((ia * G::TRA + ya) * G::PWA + xa) * PCI + kq;
            frag4 acc[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = frag4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NK; ++s) {
                constexpr int dummy = 0;
                const int tap = (4 * s) / PCI, lch0 = (4 * s) % PCI;
                float a = xt[abase + ((tap / 3) * G::PWA + (tap % 3)) * PCI + lch0];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wr[s][cb], acc[cb], 0, 0, 0);
                (void)dummy;
            }
            // ---- epilogue: D[row = 4kq + j][col = l15] ----
            {
                const float b = (l15 < C::NOUT) ? P.bias[l15] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int pd = 16 * t + 4 * kq + j;
                    int x = pd % G::W, yl = (pd / G::W) % G::TH, img = pd / (G::W * G::TH);
                    int n = n0 + img;
                    if (n < N && l15 < C::NOUT)
                        P.out[((size_t)(n * G::H + row0 + yl) * G::W + x) * C::NOUT + l15] = act_fwd<C::ACT>(acc[0][j] + b);
                }
            }
        }
        __syncthreads();
    }
}

struct MMask0U8 { using G = WGeo<64, 64, 8, 1, 256>; static constexpr int SRC = WSRC_U8, CA = 3, CB = 8, UPS = 2, WT = 0, WCI = 11, WCO = 16, NOUT = 16, ACT = CGS_ACT_LRELU, EPI = MEPI_PLAIN, OUT_A = 0; };
struct MMask0F32 { using G = WGeo<64, 64, 8, 1, 256>; static constexpr int SRC = WSRC_F32, CA = 3, CB = 8, UPS = 2, WT = 0, WCI = 11, WCO = 16, NOUT = 16, ACT = CGS_ACT_LRELU, EPI = MEPI_PLAIN, OUT_A = 0; };
struct MDec3F { using G = WGeo<4, 4, 4, 8, 256>; static constexpr int SRC = WSRC_F32, CA = 16, CB = 32, UPS = 4, WT = 0, WCI = 48, WCO = 16, NOUT = 16, ACT = CGS_ACT_NONE, EPI = MEPI_PLAIN, OUT_A = 0; };

template <class C>
static int launch_mconv(MConvParams P, hipStream_t st) {
    using G = typename C::G;
    constexpr int S = (C::CA + 3) / 4 + C::CB / 4;
    if (P.n <= 0) return CGS_OK;
    P.ntiles = (G::IMGS == 1) ? P.n * G::STRIPS : (P.n + G::IMGS - 1) / G::IMGS;
    int blocks = P.ntiles < 2048 ? P.ntiles : 2048;
    size_t lds = (size_t)G::IMGS * G::TRA * G::PWA * S * 4 * sizeof(float);
    hipLaunchKernelGGL(mconv_kernel<C>, dim3(blocks), dim3(G::THREADS), lds, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// Called from cgs_conv3x3_fwd / cgs_conv3x3_bwd_data (conv_fwd.hip) for the layers routed to the matrix cores.
int mconv_fwd_dispatch(int which, int n, const void* src_a, const float* src_b, const float* w, const float* bias, float* out,
                       hipStream_t st) {
    MConvParams P{};
    P.src_a = src_a; P.src_b = src_b; P.w = w; P.bias = bias; P.out = out; P.n = n;
    switch (which) {
        case 0: return launch_mconv<MMask0U8>(P, st);
        case 1: return launch_mconv<MMask0F32>(P, st);
        case 2: return launch_mconv<MDec3F>(P, st);
    }
    return CGS_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// Mask head backward on the matrix cores: d(masker.0 output) rebuilt from dzpre (as SRC_DH in conv_body.h), then
// the masker.0 data gradient w.r.t. its low-resolution input o0 with the nearest-upsample FOLDED into the weights:
//   d_o0[q][c] = sum over the 2x2 cell of q of the 3x3 data gradient
//              = sum_{u,v in 0..3} sum_oc dH[2q + (u-1, v-1)][oc] * W4[u][v][oc][c],
//   W4[u][v][oc][c] = sum_{a in {0,1}, ky = a+2-u in 0..2} sum_{b in {0,1}, kx = b+2-v in 0..2} W[ky][kx][3+c][oc]
// i.e. one stride-2 4x4 convolution (16 taps per output) instead of four 3x3 ones (36 taps): 2.25x fewer MACs.
// GEMM: M = 16 low-res pixels of one row, K = 16 window positions x 16 channels, N = 8 (half of the tile is padding).
// The dH tile is NHWC in LDS with a pixel stride of 17 floats so the stride-2 pixel reads of a wave hit 32 banks.
// The masker.2 WEIGHT gradient rides along: dW2[ky][kx][ch] = sum_p h[p][ch] * dzpre[p + (1-ky, 1-kx)] uses exactly the
// (h pixel, dzpre neighbour) pairs the dH rebuild already has in registers, so no second pass over h is needed.
// ------------------------------------------------------------------------------------------------
struct MHeadParams {
    const float* dzpre; const float* h; const float* w2; const float* w0;
    float* dh; float* d_o0;
    float* slab;     // optional: masker.2 weight-gradient partials, one [145] slab per workgroup
    int n, ntiles;
};

template <int TH>
struct MHeadGeo {
    static constexpr int H = 64, W = 64, TRA = TH + 2, PW = W + 2, PS = 17, DZW = W + 4, DZR = TH + 4, STRIPS = H / TH;
    static constexpr int XT = TRA * PW * PS, W4 = 16 * 16 * 8, DZ = DZR * DZW;
    static constexpr int FLOATS = 2 * XT + W4 + 2 * DZ;          // both tiles are double buffered
    static constexpr size_t LDS = (size_t)((FLOATS + 3) / 4) * 16;
};

// Workgroup = 8 waves with two roles, one tile (TH rows of one image) apart:
//   waves 0-3 ("builders", VALU + memory): rebuild dH of tile i into xt[i&1] (and the masker.2 weight-gradient partials),
//   waves 4-7 ("matrix" waves): run the folded 4x4 convolution of tile i-1 out of xt[(i-1)&1] on the matrix cores,
// with ONE workgroup barrier per tile.  Each SIMD hosts one wave of each role, so the VALU and MFMA pipes and the
// memory system work at the same time instead of one phase after the other.
template <int TH, bool WG>
__global__ void __launch_bounds__(512) mask_head_kernel(MHeadParams P) {
    using G = MHeadGeo<TH>;
    constexpr int H = G::H, W = G::W, TRA = G::TRA, PW = G::PW, PS = G::PS, DZW = G::DZW;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float* xt0 = (float*)smem;         // dH tiles [2][TRA][PW][PS]
    float* w4 = xt0 + 2 * G::XT;       // folded weights [u][v][oc][c]
    float* dz0 = w4 + G::W4;           // dzpre tiles [2] with a 2-pixel halo
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool builder = tid < 256;
    const int btid = tid & 255;

    for (int e = tid; e < G::W4; e += 512) {
        const int c = e & 7, oc = (e >> 3) & 15, v = (e >> 7) & 3, u = e >> 9;
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                int ky = a + 2 - u, kx = b + 2 - v;
                if (ky >= 0 && ky <= 2 && kx >= 0 && kx <= 2) s += P.w0[((ky * 3 + kx) * 11 + 3 + c) * 16 + oc];
            }
        w4[e] = s;
    }
    for (int e = tid; e < 2 * TRA * 2 * 16; e += 512) {      // zero halo columns of both tiles (never written again)
        int ch = e & 15, side = (e >> 4) & 1, r = (e >> 5) % TRA, bufi = e / (32 * TRA);
        xt0[bufi * G::XT + (r * PW + (side ? PW - 1 : 0)) * PS + ch] = 0.f;
    }

    const int grid = gridDim.x, bid = blockIdx.x;
    const int T = (P.ntiles - bid + grid - 1) / grid;      // tiles of this workgroup: bid, bid+grid, ...
    auto tile_of = [&](int i) { return bid + (i < T ? i : T - 1) * grid; };   // clamped: prefetches past the end re-read

    // ---------------- builder state ----------------
    const int pl = btid & 3;           // the 4-channel plane of dH this thread builds
    constexpr int IT = TRA * W * 4 / 256, DIT = (G::DZ + 255) / 256;
    static_assert((TRA * W * 4) % 256 == 0, "whole iterations: no element is visited twice");
    float w2r[9][4];
    float wacc[9][4], bacc = 0.f;      // masker.2 weight-gradient partials of this thread (plane pl)
    float4 hvs[IT];
    float dzr[DIT];
    auto load_h = [&](int tile) {
        const int n0 = tile / G::STRIPS, row0 = (tile % G::STRIPS) * TH;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            int e = btid + it * 256;
            int x = (e >> 2) % W, r = e / (4 * W), y = row0 + r - 1;
            bool in = y >= 0 && y < H;
            hvs[it] = ((const float4*)P.h)[in ? ((n0 * H + y) * W + x) * 4 + pl : 0];
        }
    };
    auto load_dz = [&](int tile) {
        const int n0 = tile / G::STRIPS, row0 = (tile % G::STRIPS) * TH;
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            int e = btid + it * 256;
            e = e < G::DZ ? e : G::DZ - 1;
            int c2 = e % DZW, r2 = e / DZW;
            int y = row0 + r2 - 2, x = c2 - 2;
            bool in = y >= 0 && y < H && x >= 0 && x < W;
            float v = P.dzpre[in ? (n0 * H + y) * W + x : 0];
            dzr[it] = in ? v : 0.f;
        }
    };
    auto store_dz = [&](float* dz) {
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            int e = btid + it * 256;
            dz[e < G::DZ ? e : G::DZ - 1] = dzr[it];
        }
    };
    if (builder) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) { w2r[t][c] = P.w2[t * 16 + 4 * pl + c]; wacc[t][c] = 0.f; }
        load_dz(tile_of(0));
        store_dz(dz0);                 // dz tile of tile i lives in dz0 + (i&1)*DZ
        load_dz(tile_of(1));
        load_h(tile_of(0));
    }
    __syncthreads();

    // ---------------- matrix-wave state ----------------
    const int l15 = lane & 15, kq = lane >> 4, mwave = wave & 3;
    // B operand of k-step s = (window position, channel plane): w4[(4s + kq)*8 + l15] for the 8 real columns.  The 8
    // padding columns just repeat them (their results are never stored).  Read from LDS per tile, base + constant.
    const float* wb = w4 + kq * 8 + (l15 & 7);

    for (int i = 0; i <= T; ++i) {
        if (builder) {
            if (i < T) {
                const int tile = tile_of(i);
                const int n0 = tile / G::STRIPS, row0 = (tile % G::STRIPS) * TH;
                float* xt = xt0 + (i & 1) * G::XT;
                const float* dz = dz0 + (i & 1) * G::DZ;
                store_dz(dz0 + ((i + 1) & 1) * G::DZ);           // dzpre of tile i+1 (read one barrier from now)
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int e = btid + it * 256;
                    const int x = (e >> 2) % W, r = e / (4 * W), y = row0 + r - 1;
                    const bool in = y >= 0 && y < H;
                    const int gi = in ? ((n0 * H + y) * W + x) * 4 + pl : 0;
                    const float4 hv = hvs[it];
                    const bool own = in && r >= 1 && r <= TH;   // rows owned by this strip (halo rows: the neighbours')
                    const float4 hw_ = own ? hv : f4zero();
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            float d = dz[(r + 2 - ky) * DZW + x + 3 - kx];
                            a0 = fmaf(d, w2r[ky * 3 + kx][0], a0); a1 = fmaf(d, w2r[ky * 3 + kx][1], a1);
                            a2 = fmaf(d, w2r[ky * 3 + kx][2], a2); a3 = fmaf(d, w2r[ky * 3 + kx][3], a3);
                            if (WG) {
                                wacc[ky * 3 + kx][0] = fmaf(d, hw_.x, wacc[ky * 3 + kx][0]);
                                wacc[ky * 3 + kx][1] = fmaf(d, hw_.y, wacc[ky * 3 + kx][1]);
                                wacc[ky * 3 + kx][2] = fmaf(d, hw_.z, wacc[ky * 3 + kx][2]);
                                wacc[ky * 3 + kx][3] = fmaf(d, hw_.w, wacc[ky * 3 + kx][3]);
                            }
                        }
                    if (WG) {
                        bacc += (own && pl == 0) ? dz[(r + 1) * DZW + x + 2] : 0.f;
                        // pin the accumulators here: otherwise their FMAs are sunk past the whole item loop and every
                        // dz / h value of all items stays live (hundreds of registers)
#pragma unroll
                        for (int t = 0; t < 9; ++t)
#pragma unroll
                            for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(wacc[t][c]));
                    }
                    float4 v = make_float4(a0 * (hv.x > 0.f ? 1.f : 0.01f), a1 * (hv.y > 0.f ? 1.f : 0.01f),
                                           a2 * (hv.z > 0.f ? 1.f : 0.01f), a3 * (hv.w > 0.f ? 1.f : 0.01f));
                    v = in ? v : f4zero();
                    float* d = xt + (r * PW + x + 1) * PS + 4 * pl;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                    if (own) ((float4*)P.dh)[gi] = v;
                    __builtin_amdgcn_sched_barrier(0);
                }
                load_dz(tile_of(i + 2));     // in flight across the next tile's rebuild
                load_h(tile_of(i + 1));
            }
        } else if (i > 0) {
            const int tile = tile_of(i - 1);
            const int n0 = tile / G::STRIPS, row0 = (tile % G::STRIPS) * TH;
            const float* xt = xt0 + ((i - 1) & 1) * G::XT;
            constexpr int NT = (TH / 2) * 2;           // 16-pixel tiles: 2 per low-res row
            for (int t = mwave; t < NT; t += 4) {
                const int qyl = t >> 1, qx0 = (t & 1) * 16;
                const int abase = ((2 * qyl) * PW + 2 * (qx0 + l15)) * PS + kq;
                frag4 acc = frag4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 64; ++s) {
                    const int pos = s >> 2, u = pos >> 2, v = pos & 3;
                    float a = xt[abase + (u * PW + v) * PS + 4 * (s & 3)];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wb[s * 32], acc, 0, 0, 0);
                    if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // at most 8 k-steps of operands in flight
                }
                if (l15 < 8) {
                    const int qy = row0 / 2 + qyl;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        P.d_o0[((size_t)(n0 * 32 + qy) * 32 + qx0 + 4 * kq + j) * 8 + l15] = acc[j];
                }
            }
        }
        __syncthreads();
    }

    if constexpr (WG) {
        // builder lanes with equal (lane & 3) hold the same channel plane: butterfly over the other 16 lanes, then over
        // the 4 builder waves through LDS
        float* red = xt0;                // [4 waves][4 planes][37]
        if (builder) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v = wacc[t][c];
                    v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                    if (lane < 4) red[(wave * 4 + lane) * 37 + t * 4 + c] = v;
                }
            float v = bacc;
            v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (lane < 4) red[(wave * 4 + lane) * 37 + 36] = v;
        }
        __syncthreads();
        if (tid < 145) {
            // slab layout = HWIO [9][16][1] weights then the bias
            int t = tid / 16, ch = tid % 16, p = ch >> 2, c = ch & 3;
            int idx = (tid < 144) ? t * 4 + c : 36;
            if (tid == 144) p = 0;
            float v = (red[(0 * 4 + p) * 37 + idx] + red[(1 * 4 + p) * 37 + idx]) + (red[(2 * 4 + p) * 37 + idx] + red[(3 * 4 + p) * 37 + idx]);
            P.slab[(size_t)blockIdx.x * 145 + tid] = v;
        }
    }
}

// one workgroup per CU (LDS), persistent over its tiles
static int mask_head_blocks(int n) { int t = n * MHeadGeo<8>::STRIPS; return t < 256 ? t : 256; }
int mconv_mask_head_slabs(int n) { return n <= 0 ? 0 : mask_head_blocks(n); }

int mconv_mask_head(int n, const float* dzpre, const float* h, const float* w2, const float* w0, float* dh, float* d_o0,
                    float* slab, hipStream_t st) {
    constexpr int TH = 8;
    using G = MHeadGeo<TH>;
    if (n <= 0) return CGS_OK;
    MHeadParams P{dzpre, h, w2, w0, dh, d_o0, slab, n, n * G::STRIPS};
    int blocks = mask_head_blocks(n);
    static hipError_t attr = [] {   // 105 KB of dynamic LDS: above the default limit
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mask_head_kernel<TH, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&mask_head_kernel<TH, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    }();
    if (attr != hipSuccess) return (int)attr;
    if (slab) hipLaunchKernelGGL((mask_head_kernel<TH, true>), dim3(blocks), dim3(512), G::LDS, st, P);
    else hipLaunchKernelGGL((mask_head_kernel<TH, false>), dim3(blocks), dim3(512), G::LDS, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

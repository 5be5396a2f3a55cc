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

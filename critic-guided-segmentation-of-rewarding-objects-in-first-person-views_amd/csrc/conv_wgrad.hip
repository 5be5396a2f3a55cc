// Weight/bias gradients of the 3x3 convolutions as an implicit GEMM on the matrix cores.
//
//   dW[r][co] = sum_pixels  Xcol[pixel][r] * dY[pixel][co],   r = tap*CI + ci  (+ one all-ones row = dbias)
//
// is a true GEMM whose reduction dimension is the pixel index (millions), so it runs on
// v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain): 16 rows of r x 16 output channels per
// instruction, 4 pixels per k-step, accumulators stay in 4 VGPRs per 16x16 block for the whole launch --
// no cross-lane reduction, no float atomics.  Each wave walks a disjoint set of pixels and finally writes
// its own partial slab; cgs_reduce_slabs() adds the slabs in a fixed order (bitwise reproducible).
//
// LDS tiles keep the NHWC order of global memory: X tile [img][TH+2][W+2][CI] with the concat of the
// skip input and the nearest-upsampled low-res input materialised (zero halo), dY tile [img][TH][W][CO].
// Lanes (r = lane&15, pixel = lane>>4) then read consecutive floats: bank-conflict free.
#include "wgrad_sparse.h"
#include "head_wgrad.h"
#include "wgrad_dec3.h"

// dec_model.0 (16 -> 8 channels at 32x32): outer products on v_mfma_f32_4x4x1 (wgrad_dec0.hip)
int wgrad_dec0_slabs(int n);
int wgrad_dec0_launch(int n, const float* e0, const float* o1, const float* dy, float* slab, hipStream_t st);

static constexpr int kMaxWgradBlocks = 1024;

template <class G>
static int wg_tiles(int n) { return (G::IMGS == 1) ? n * G::STRIPS : (n + G::IMGS - 1) / G::IMGS; }
template <class G>
static int wg_blocks(int n) {
    int t = wg_tiles<G>(n);
    return t < kMaxWgradBlocks ? t : kMaxWgradBlocks;
}

template <class C, bool SPARSE>
__global__ void __launch_bounds__(C::G::THREADS) wgrad_any_kernel(WgradParams P) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    constexpr int SLAB = (9 * (C::CA + C::CB) + 1) * C::CO;
    wgrad_dispatch<C, SPARSE>(P, blockIdx.x, gridDim.x, P.ntiles, P.slab + (size_t)blockIdx.x * SLAB, smem);
}

// the sparse form is latency-bound with small tiles: more workgroups in flight than the MFMA form wants
static constexpr int kMaxSparseBlocks = 512;
#ifndef CGS_CAP_W0U8
#define CGS_CAP_W0U8 896      // (round 5, next to the head's 192 workgroups at four per CU: 0.5483 ms; 1024: 0.5499, 832: 0.5494 -- r05af)
#endif
template <class C>
static int wg_blocks_any(int n) {
    using G = typename C::G;
    if (sparse_cfg<C>::ok && wgrad_sparse_enabled()) {
        const int cap = G::H >= 64 ? CGS_CAP_W0U8 : kMaxSparseBlocks;      // measured: 1024 / 512
        int t = wg_tiles<G>(n);
        return t < cap ? t : cap;
    }
    return wg_blocks<G>(n);
}

template <class C>
static int launch_wgrad(WgradParams P, hipStream_t st) {
    using G = typename C::G;
    P.ntiles = wg_tiles<G>(P.n);
    if (P.ntiles == 0) return CGS_OK;
    if constexpr (sparse_cfg<C>::ok) {
        if (wgrad_sparse_enabled()) {
            const size_t lds = wgrad_any_lds_bytes<C, true>();
            hipLaunchKernelGGL((wgrad_any_kernel<C, true>), dim3(wg_blocks_any<C>(P.n)), dim3(G::THREADS), lds, st, P);
            CGS_HIP_CHECK_LAUNCH();
            return CGS_OK;
        }
    }
    const size_t lds = wgrad_lds_bytes<C>();
    hipLaunchKernelGGL((wgrad_any_kernel<C, false>), dim3(wg_blocks<G>(P.n)), dim3(G::THREADS), lds, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// features.0's weight gradient on the uint8 frames (the A pass) and the critic head's weight gradients in ONE launch: the head GEMM
// is a latency-bound 8 us launch of < 200 workgroups that only the step's final reduction waits for; as extra workgroups of this
// launch it disappears behind the sparse gather (round 3: one dependent launch less on the critical path).
// (round 5) nbw1 > 0: features.3's sparse weight gradient of the same pass (P1) as nbw1 more workgroups between the two roles -- its data
// gradient runs inside the tail backward launch (cgs_tail_enc_bwd_enc1).
#ifndef CGS_R1_U8
#define CGS_R1_U8 1
#endif
// four waves per SIMD = 128 registers: what the kernel's own two roles need (114 + 4); without the cap the compiler parks the dec_model.3 rider's
// 28 accumulators in AGPRs ON TOP of them and the launch drops to three workgroups per CU (-Rpass-analysis=kernel-resource-usage)
#ifndef CGS_U8_WAVES
#define CGS_U8_WAVES 4
#endif
#if CGS_U8_WAVES
#define CGS_U8_OCC __attribute__((amdgpu_waves_per_eu(CGS_U8_WAVES, CGS_U8_WAVES)))
#else
#define CGS_U8_OCC
#endif
// (round 5) nb3 > 0: dec_model.3's weight gradient as a GEMM over the images (wgrad_dec3.h) in the launch's LAST nb3 workgroups.
// R1 / R3: which rider roles this instance carries -- a role's registers are the whole launch's (the general features.3 body took this
// kernel from 114 to 256 VGPRs), so a launch without a rider runs the instance without its code.
template <bool R1, bool R3>
__global__ void __launch_bounds__(256) CGS_U8_OCC wgrad_enc0u8_head_kernel(WgradParams P, HeadWgradParams H, int nbw, WgradParams P1, int nbw1,
                                                                           Dec3WgParams D3, int nb3) {
    if (CGS_KARG_PREFETCH) cgs_kernarg_prefetch<2 * sizeof(WgradParams) + sizeof(HeadWgradParams) + sizeof(Dec3WgParams) + 24>();
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    if constexpr (R3) {
        if ((int)blockIdx.x >= (int)gridDim.x - nb3) {
            dec3_wgrad_body(D3, (int)blockIdx.x - ((int)gridDim.x - nb3), nb3, (float*)smem);
            return;
        }
    }
    // block order (A/B: CGS_R1_U8): 0 = [riders | features.0 | head], 1 = [features.0 | riders | head], 2 = [features.0 | head | riders]
    const int gx = (int)gridDim.x - nb3, bx = (int)blockIdx.x;
    const int r_lo = CGS_R1_U8 == 0 ? 0 : (CGS_R1_U8 == 1 ? nbw : gx - nbw1);
    if constexpr (R1) {
        if (bx >= r_lo && bx < r_lo + nbw1) {
            constexpr int SLAB1 = (9 * 8 + 1) * 8;
            const int b1 = bx - r_lo;
            wgrad_dispatch<WEnc1, true>(P1, b1, nbw1, P1.ntiles, P1.slab + (size_t)b1 * SLAB1, smem);
            return;
        }
    }
    const int bm = bx - ((R1 && bx >= r_lo) ? nbw1 : 0);
    if (bm < nbw) {
        constexpr int SLAB = (9 * 3 + 1) * 8;
        wgrad_dispatch<WEnc0U8, true>(P, bm, nbw, P.ntiles, P.slab + (size_t)bm * SLAB, smem);
    } else {
        tail_head_wgrad_body(H, bm - nbw, (float*)smem);
    }
}

// e0_1 / dy1 / am1 / slab1 (all or none): features.3's weight gradient over n1w images ([n1w,32,32,8] input, [n1w,16,16,8] pooled output
// gradient, argmax nibbles, slab1 [nslab1][584] with nslab1 = cgs_enc1_wgrad_rider_slabs(n1w)) as extra workgroups of this launch.
extern "C" int cgs_dec3_wgrad_rider_slabs(int32_t n) { return n < 0 ? CGS_ERR_BADARG : (n < 64 ? n : 64); }

// ... + dec_model.3's weight gradient over n3 images (e3 [n3,4,4,16], o4 [n3,32], d o3 [n3,4,4,16] as cgs_dec0_tail_dec_bwd_do3 leaves it;
// slab3 [cgs_dec3_wgrad_rider_slabs(n3)][6928]; slab3 = NULL: none) as the launch's last workgroups (round 5).
extern "C" int cgs_enc0_wgrad_u8_with_head_riders(int32_t n, const uint8_t* x_u8, const float* dy, const uint32_t* amask, float* slab,
                                                  int32_t n0, const float* hvec0, const float* e4_0, const float* d_o4_0, int32_t n_o4_0,
                                                  int32_t n1, const float* hvec1, const float* e4_1, const float* d_o4_1, int32_t n_o4_1,
                                                  float* slab_head, float* slab_pw,
                                                  int32_t n1w, const float* e0_1, const float* dy1, const uint32_t* am1, float* slab1, int32_t nslab1,
                                                  int32_t n3, const float* e3, const float* o4, const float* do3, float* slab3,
                                                  cgs_stream_t stream) {
    if (slab3 && (n3 <= 0 || !e3 || !o4 || !do3)) return CGS_ERR_BADARG;
    if (n <= 0 || !x_u8 || !dy || !amask || !slab) return CGS_ERR_BADARG;
    if (n0 < 0 || n1 < 0 || n0 + n1 == 0 || !slab_head || (n0 > 0 && (!hvec0 || !e4_0)) || (n1 > 0 && (!hvec1 || !e4_1))) return CGS_ERR_BADARG;
    if ((d_o4_0 || d_o4_1) && !slab_pw) return CGS_ERR_BADARG;
    if (slab1 && (n1w <= 0 || !e0_1 || !dy1 || !am1 || nslab1 <= 0)) return CGS_ERR_BADARG;
    if (slab1 && !(sparse_cfg<WEnc1>::ok && wgrad_sparse_enabled())) return CGS_ERR_UNSUPPORTED;
    WgradParams P{};
    P.src_a = x_u8; P.dy = dy; P.amask = amask; P.slab = slab; P.n = n;
    P.ntiles = wg_tiles<WEnc0U8::G>(n);
    HeadWgradParams H{{{hvec0, e4_0, d_o4_0, n0, n_o4_0}, {hvec1, e4_1, d_o4_1, n1, n_o4_1}}, slab_head, slab_pw};
    const int nbw = wg_blocks_any<WEnc0U8>(n), nbh = (n0 + n1 + kHwIpb - 1) / kHwIpb;
    size_t lds = wgrad_any_lds_bytes<WEnc0U8, true>();
    if (sizeof(float) * (size_t)kHwLdsFloats > lds) lds = sizeof(float) * (size_t)kHwLdsFloats;
    WgradParams P1{};
    int nbw1 = 0;
    if (slab1) {
        P1.src_a = e0_1; P1.dy = dy1; P1.amask = am1; P1.slab = slab1; P1.n = n1w;
        P1.ntiles = wg_tiles<WEnc1::G>(n1w);
        nbw1 = nslab1;
        const size_t l1 = wgrad_any_lds_bytes<WEnc1, true>();
        if (l1 > lds) lds = l1;
    }
    Dec3WgParams D3{e3, o4, do3, slab3, n3};
    const int nb3 = slab3 ? cgs_dec3_wgrad_rider_slabs(n3) : 0;
    if (nb3 > 0 && sizeof(float) * (size_t)kWD3LdsFloats > lds) lds = sizeof(float) * (size_t)kWD3LdsFloats;
    const dim3 grid(nbw + nbh + nbw1 + nb3);
    if (nbw1 > 0 && nb3 > 0) hipLaunchKernelGGL((wgrad_enc0u8_head_kernel<true, true>), grid, dim3(256), lds, (hipStream_t)stream, P, H, nbw, P1, nbw1, D3, nb3);
    else if (nbw1 > 0) hipLaunchKernelGGL((wgrad_enc0u8_head_kernel<true, false>), grid, dim3(256), lds, (hipStream_t)stream, P, H, nbw, P1, nbw1, D3, nb3);
    else if (nb3 > 0) hipLaunchKernelGGL((wgrad_enc0u8_head_kernel<false, true>), grid, dim3(256), lds, (hipStream_t)stream, P, H, nbw, P1, nbw1, D3, nb3);
    else hipLaunchKernelGGL((wgrad_enc0u8_head_kernel<false, false>), grid, dim3(256), lds, (hipStream_t)stream, P, H, nbw, P1, nbw1, D3, nb3);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_enc0_wgrad_u8_with_head_enc1(int32_t n, const uint8_t* x_u8, const float* dy, const uint32_t* amask, float* slab,
                                                int32_t n0, const float* hvec0, const float* e4_0, const float* d_o4_0, int32_t n_o4_0,
                                                int32_t n1, const float* hvec1, const float* e4_1, const float* d_o4_1, int32_t n_o4_1,
                                                float* slab_head, float* slab_pw,
                                                int32_t n1w, const float* e0_1, const float* dy1, const uint32_t* am1, float* slab1, int32_t nslab1,
                                                cgs_stream_t stream) {
    return cgs_enc0_wgrad_u8_with_head_riders(n, x_u8, dy, amask, slab, n0, hvec0, e4_0, d_o4_0, n_o4_0, n1, hvec1, e4_1, d_o4_1, n_o4_1,
                                              slab_head, slab_pw, n1w, e0_1, dy1, am1, slab1, nslab1, 0, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int cgs_enc0_wgrad_u8_with_head(int32_t n, const uint8_t* x_u8, const float* dy, const uint32_t* amask, float* slab,
                                           int32_t n0, const float* hvec0, const float* e4_0, const float* d_o4_0, int32_t n_o4_0,
                                           int32_t n1, const float* hvec1, const float* e4_1, const float* d_o4_1, int32_t n_o4_1,
                                           float* slab_head, float* slab_pw, cgs_stream_t stream) {
    return cgs_enc0_wgrad_u8_with_head_enc1(n, x_u8, dy, amask, slab, n0, hvec0, e4_0, d_o4_0, n_o4_0, n1, hvec1, e4_1, d_o4_1, n_o4_1,
                                            slab_head, slab_pw, 0, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

static bool wdesc_is(const cgs_conv_desc* d, int hw, int ca, int cb, int co, int src, int ups, int pool) {
    return d->h == hw && d->w == hw && d->ca == ca && d->cb == cb && d->co == co && d->src_a == src &&
           (cb == 0 || d->ups == ups) && d->pool == pool;
}

// returns slabs for the descriptor, or an error code
extern "C" int cgs_conv3x3_bwd_weight_slabs(const cgs_conv_desc* d) {
    if (!d || d->n < 0) return CGS_ERR_BADARG;
    const int n = d->n;
#define SLABS(CFG) return wg_blocks_any<CFG>(n)
    if (wdesc_is(d, 64, 3, 0, 8, CGS_SRC_U8, 2, 1)) SLABS(WEnc0U8);
    if (wdesc_is(d, 64, 3, 0, 8, CGS_SRC_F32, 2, 1)) SLABS(WEnc0F32);
    if (wdesc_is(d, 32, 8, 0, 8, CGS_SRC_F32, 2, 1)) SLABS(WEnc1);
    if (wdesc_is(d, 16, 8, 0, 8, CGS_SRC_F32, 2, 1)) SLABS(WEnc2);
    if (wdesc_is(d, 8, 8, 0, 16, CGS_SRC_F32, 2, 1)) SLABS(WEnc3);
    if (wdesc_is(d, 4, 16, 32, 16, CGS_SRC_F32, 4, 0)) SLABS(WDec3);
    if (wdesc_is(d, 8, 8, 16, 8, CGS_SRC_F32, 2, 0)) SLABS(WDec2);
    if (wdesc_is(d, 16, 8, 8, 8, CGS_SRC_F32, 2, 0)) SLABS(WDec1);
    if (wdesc_is(d, 32, 8, 8, 8, CGS_SRC_F32, 2, 0)) return wgrad_dec0_slabs(n);
    if (wdesc_is(d, 64, 3, 8, 16, CGS_SRC_U8, 2, 0)) SLABS(WMask0U8);
    if (wdesc_is(d, 64, 3, 8, 16, CGS_SRC_F32, 2, 0)) SLABS(WMask0F32);
    if (wdesc_is(d, 64, 16, 0, 1, CGS_SRC_F32, 2, 0)) return wg_blocks<WMask2G>(n);
#undef SLABS
    return CGS_ERR_UNSUPPORTED;
}

extern "C" int cgs_conv3x3_bwd_weight(const cgs_conv_desc* d, const void* src_a, const float* src_b, const float* dy,
                                      const uint32_t* amask, float* slab, cgs_stream_t stream) {
    if (!d || !src_a || !dy || !slab || d->n < 0) return CGS_ERR_BADARG;
    if (d->cb > 0 && !src_b) return CGS_ERR_BADARG;
    if (d->pool && !amask) return CGS_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    WgradParams P{};
    P.src_a = src_a; P.src_b = src_b; P.dy = dy; P.amask = amask; P.slab = slab; P.n = d->n; P.drop = d->drop_a;
    if (d->drop_a.p > 0.f && !wdesc_is(d, 8, 8, 0, 16, CGS_SRC_F32, 2, 1)) return CGS_ERR_UNSUPPORTED;
    if (wdesc_is(d, 64, 3, 0, 8, CGS_SRC_U8, 2, 1)) return launch_wgrad<WEnc0U8>(P, st);
    if (wdesc_is(d, 64, 3, 0, 8, CGS_SRC_F32, 2, 1)) return launch_wgrad<WEnc0F32>(P, st);
    if (wdesc_is(d, 32, 8, 0, 8, CGS_SRC_F32, 2, 1)) return launch_wgrad<WEnc1>(P, st);
    if (wdesc_is(d, 16, 8, 0, 8, CGS_SRC_F32, 2, 1)) return launch_wgrad<WEnc2>(P, st);
    if (wdesc_is(d, 8, 8, 0, 16, CGS_SRC_F32, 2, 1)) return launch_wgrad<WEnc3>(P, st);
    if (wdesc_is(d, 4, 16, 32, 16, CGS_SRC_F32, 4, 0)) return launch_wgrad<WDec3>(P, st);
    if (wdesc_is(d, 8, 8, 16, 8, CGS_SRC_F32, 2, 0)) return launch_wgrad<WDec2>(P, st);
    if (wdesc_is(d, 16, 8, 8, 8, CGS_SRC_F32, 2, 0)) return launch_wgrad<WDec1>(P, st);
    if (wdesc_is(d, 32, 8, 8, 8, CGS_SRC_F32, 2, 0)) return wgrad_dec0_launch(d->n, (const float*)src_a, src_b, dy, slab, st);
    if (wdesc_is(d, 64, 3, 8, 16, CGS_SRC_U8, 2, 0)) return launch_wgrad<WMask0U8>(P, st);
    if (wdesc_is(d, 64, 3, 8, 16, CGS_SRC_F32, 2, 0)) return launch_wgrad<WMask0F32>(P, st);
    if (wdesc_is(d, 64, 16, 0, 1, CGS_SRC_F32, 2, 0)) {
        using G = WMask2G;
        size_t lds = ((size_t)G::TRA * 68 * 16 + (size_t)G::TH * G::W) * sizeof(float);
        P.ntiles = wg_tiles<G>(P.n);
        if (P.ntiles == 0) return CGS_OK;
        hipLaunchKernelGGL(wgrad_co1_kernel<G>, dim3(wg_blocks<G>(P.n)), dim3(G::THREADS), lds, st, P);
        CGS_HIP_CHECK_LAUNCH();
        return CGS_OK;
    }
    return CGS_ERR_UNSUPPORTED;
}

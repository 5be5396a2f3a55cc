// Weight gradient and data gradient of one small layer in ONE launch ("horizontal" fusion).
//
// Both consume the same dY and are independent of each other; for the 16x16-and-smaller layers each is a
// latency-bound launch that leaves most CUs idle.  Here the first `nbw` workgroups run the MFMA wgrad body and
// the rest run the VALU data-gradient body, so the two overlap on the chip without a second stream (cross-queue
// joins inside a HIP graph cost 6-10 us each on ROCm 7.2, more than they gain).
#include "conv_body.h"
#include "wgrad_sparse.h"

template <class CWG, class CDG, bool SPARSE>
__global__ void __launch_bounds__(CWG::G::THREADS) conv_bwd_both_kernel(WgradParams pw, ConvParams pd, int nbw) {
    if (CGS_KARG_PREFETCH) cgs_kernarg_prefetch<sizeof(WgradParams) + sizeof(ConvParams) + 8>();
    static_assert(CWG::G::THREADS == CDG::THREADS * CDG::CW, "both halves use the same workgroup size");
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    if ((int)blockIdx.x < nbw) {
        constexpr int SLAB = (9 * (CWG::CA + CWG::CB) + 1) * CWG::CO;
        wgrad_dispatch<CWG, SPARSE>(pw, blockIdx.x, nbw, pw.ntiles, pw.slab + (size_t)blockIdx.x * SLAB, smem);
    } else {
        if constexpr (tpw_of<CDG>::value > 1) conv3x3_body_pipe<CDG>(pd, (blockIdx.x - nbw) * tpw_of<CDG>::value, smem);
        else conv3x3_body<CDG, false>(pd, blockIdx.x - nbw, smem);
    }
}

// instances with matching workgroup sizes
CGS_DG_CFG(DEnc3x, 8, 128, SRC_POOLEXP, 16, 8, 16, 2, 0, 8, 4, 8, CGS_ACT_NONE, 2)       // 256 threads
struct WDec2x { using G = WGeo<8, 8, 8, 4, 192>; static constexpr int SRC = WSRC_F32, CA = 8, CB = 16, UPS = 2, CO = 8, DY = WDY_F32; };
struct WDec3x { using G = WGeo<4, 4, 4, 4, 384>; static constexpr int SRC = WSRC_F32, CA = 16, CB = 32, UPS = 4, CO = 16, DY = WDY_F32; };
CGS_DG_CFG(DDec3y, 4, 64, SRC_F32, 16, 48, 16, 4, 0, 48, 8, 16, CGS_ACT_NONE, 6)          // 384 threads: one 8-channel chunk per wave
struct WMask0U8x { using G = WGeo<64, 64, 4, 1, 128>; static constexpr int SRC = WSRC_U8, CA = 3, CB = 8, UPS = 2, CO = 16, DY = WDY_F32; };
struct WMask0F32x { using G = WGeo<64, 64, 4, 1, 128>; static constexpr int SRC = WSRC_F32, CA = 3, CB = 8, UPS = 2, CO = 16, DY = WDY_F32; };

#ifndef CGS_CAP_W0MIX
#define CGS_CAP_W0MIX 512
#endif
#ifndef CGS_CAP_W1R
#define CGS_CAP_W1R 256
#endif
static constexpr int kMaxBothWgradBlocks = 256;
static constexpr int kMaxBothWgradBlocksBig = 512;

static constexpr int kMaxBothSparseBlocks = 512;     // the sparse form is latency-bound: more, smaller workgroups
template <class CWG>
static constexpr bool both_sparse_ok = sparse_cfg<CWG>::ok;
template <class CWG>
static int both_slabs(int n) {
    using GW = typename CWG::G;
    int tiles = (GW::IMGS == 1) ? n * GW::STRIPS : (n + GW::IMGS - 1) / GW::IMGS;
    int cap = GW::H >= 32 ? kMaxBothWgradBlocksBig : kMaxBothWgradBlocks;
    if (both_sparse_ok<CWG> && wgrad_sparse_enabled()) {
        cap = GW::H >= 64 ? CGS_CAP_W0MIX : CGS_CAP_W1R;      // persistent sparse workgroups (swept on the step in round 3; A/B macros round 5)
    }
    return tiles < cap ? tiles : cap;
}

template <class CWG, class CDG>
static int launch_both(WgradParams pw, const ConvParams& pd, hipStream_t st) {
    using GW = typename CWG::G;
    using GD = Geo<CDG::H, CDG::W, CDG::THREADS, CDG::CW>;
    if (pd.n <= 0) return CGS_OK;
    int tiles = (GW::IMGS == 1) ? pw.n * GW::STRIPS : (pw.n + GW::IMGS - 1) / GW::IMGS;
    int nbw = both_slabs<CWG>(pw.n);
    int nbd = (GD::IMGS == 1) ? pd.n * GD::STRIPS : (pd.n + GD::IMGS - 1) / GD::IMGS;
    static_assert(tpw_of<CDG>::value == 1 || (GD::IMGS == 1 && GD::STRIPS % tpw_of<CDG>::value == 0), "pipelined strips of one image");
    nbd /= tpw_of<CDG>::value;
    pw.ntiles = tiles;
    const size_t ld = conv_lds_bytes<CDG>();
    if constexpr (sparse_cfg<CWG>::ok) {
        if (wgrad_sparse_enabled()) {
            const size_t lw = wgrad_any_lds_bytes<CWG, true>();
            hipLaunchKernelGGL((conv_bwd_both_kernel<CWG, CDG, true>), dim3(nbw + nbd), dim3(GW::THREADS), lw > ld ? lw : ld, st, pw, pd, nbw);
            CGS_HIP_CHECK_LAUNCH();
            return CGS_OK;
        }
    }
    const size_t lw = wgrad_lds_bytes<CWG>();
    hipLaunchKernelGGL((conv_bwd_both_kernel<CWG, CDG, false>), dim3(nbw + nbd), dim3(GW::THREADS), lw > ld ? lw : ld, st, pw, pd, nbw);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

static int which(const cgs_conv_desc* d) {
    auto is = [&](int hw, int ca, int cb, int co, int pool) {
        return d->h == hw && d->w == hw && d->ca == ca && d->cb == cb && d->co == co && d->src_a == CGS_SRC_F32 &&
               (cb == 0 || d->ups == 2) && d->pool == pool;
    };
    if (is(8, 8, 0, 16, 1) && d->act == CGS_ACT_RELU) return 1;    // features.10
    if (is(16, 8, 0, 8, 1) && d->act == CGS_ACT_RELU) return 2;    // features.6
    if (is(16, 8, 8, 8, 0) && d->act == CGS_ACT_NONE) return 3;    // dec_model.1
    if (is(8, 8, 16, 8, 0) && d->act == CGS_ACT_NONE) return 4;    // dec_model.2
    if (is(32, 8, 0, 8, 1) && d->act == CGS_ACT_RELU) return 5;    // features.3
    if (is(32, 8, 8, 8, 0) && d->act == CGS_ACT_NONE) return 6;    // dec_model.0
    if (d->h == 4 && d->w == 4 && d->ca == 16 && d->cb == 32 && d->co == 16 && d->ups == 4 && !d->pool) return 7;  // dec_model.3
    if (is(64, 3, 0, 8, 1) && d->act == CGS_ACT_RELU) return 8;    // features.0 on the fp32 mixes
    if (d->h == 64 && d->ca == 3 && d->cb == 8 && d->co == 16 && !d->pool) return d->src_a == CGS_SRC_U8 ? 9 : 10;  // masker.0
    return 0;
}

extern "C" int cgs_conv3x3_bwd_both_slabs(const cgs_conv_desc* d) {
    if (!d || d->n < 0) return CGS_ERR_BADARG;
    switch (which(d)) {
        case 1: return both_slabs<WEnc3>(d->n);
        case 2: return both_slabs<WEnc2>(d->n);
        case 3: return both_slabs<WDec1>(d->n);
        case 4: return both_slabs<WDec2x>(d->n);
        case 5: return both_slabs<WEnc1>(d->n);
        case 6: return both_slabs<WDec0>(d->n);
        case 7: return both_slabs<WDec3x>(d->n);
        case 8: return both_slabs<WEnc0F32>(d->n);
        case 9: return both_slabs<WMask0U8>(d->n);
        case 10: return both_slabs<WMask0F32>(d->n);
    }
    return CGS_ERR_UNSUPPORTED;
}

extern "C" int cgs_conv3x3_bwd_both(const cgs_conv_desc* d, const void* src_a, const float* src_b, const float* dy,
                                    const uint32_t* amask, const float* w, const float* addend, int32_t n_addend,
                                    float* d_a, float* d_b, float* slab, cgs_stream_t stream) {
    if (!d || !src_a || !dy || !w || !slab || (!d_a && !d_b) || d->n < 0) return CGS_ERR_BADARG;
    if (d->pool && !amask) return CGS_ERR_BADARG;
    if (d->cb > 0 && !src_b) return CGS_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    WgradParams pw{};
    pw.src_a = src_a; pw.src_b = src_b; pw.dy = dy; pw.amask = amask; pw.slab = slab; pw.n = d->n; pw.drop = d->drop_a;
    ConvParams pd{};
    pd.src_a = dy; pd.amask_in = amask; pd.w = w; pd.out = d_a; pd.out2 = d_b; pd.addend = addend; pd.n_addend = n_addend;
    pd.n = d->n; pd.drop = d->drop_a;
    switch (which(d)) {
        case 1: return launch_both<WEnc3, DEnc3x>(pw, pd, st);
        case 2: if (d->drop_a.p > 0.f) return CGS_ERR_UNSUPPORTED; return launch_both<WEnc2, DEnc2>(pw, pd, st);
        case 3: if (d->drop_a.p > 0.f) return CGS_ERR_UNSUPPORTED; return launch_both<WDec1, DDec1>(pw, pd, st);
        case 4: if (d->drop_a.p > 0.f) return CGS_ERR_UNSUPPORTED; return launch_both<WDec2x, DDec2>(pw, pd, st);
        case 5: if (d->drop_a.p > 0.f) return CGS_ERR_UNSUPPORTED; return CGS_CONV_PIPE ? launch_both<WEnc1, DEnc1P>(pw, pd, st) : launch_both<WEnc1, DEnc1>(pw, pd, st);
        case 6: if (d->drop_a.p > 0.f) return CGS_ERR_UNSUPPORTED; return launch_both<WDec0, DDec0>(pw, pd, st);
        case 7: if (d->drop_a.p > 0.f) return CGS_ERR_UNSUPPORTED; return launch_both<WDec3x, DDec3y>(pw, pd, st);
        case 8: if (d->drop_a.p > 0.f) return CGS_ERR_UNSUPPORTED; return launch_both<WEnc0F32, DEnc0>(pw, pd, st);
        case 9: if (d->drop_a.p > 0.f || d_a) return CGS_ERR_UNSUPPORTED; return launch_both<WMask0U8, DMask0>(pw, pd, st);
        case 10: if (d->drop_a.p > 0.f || d_a) return CGS_ERR_UNSUPPORTED; return launch_both<WMask0F32, DMask0>(pw, pd, st);
    }
    return CGS_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// features.0 backward of the replaced / injected passes TOGETHER with the mix backward (main.py:395,406 backward).
// The data gradient of a convolution is linear in dY and the mix backward only needs d_rep - d_inj, so a data-gradient
// workgroup builds ONE gradient tile = expand(dY_rep) - expand(dY_inj) of its strip (conv_tile.h load_poolexp_diff), runs one
// data-gradient pass and writes d(pre-sigmoid mask) from the epilogue:
//   dzpre = [ sum_c (B - A) (d_rep - d_inj) + l1s sign(z) + 2 l2s z ] z (1 - z)
// so the 2 x 25 MB image gradients are never stored, cgs_mix_bwd's pass over them disappears and half of the two passes'
// FMAs are never issued.  The weight-gradient workgroups of features.0 (over all mixes) share the launch as in
// conv_bwd_both_kernel (bit-identical to it); dzpre equals the two-launch form up to the order of the sums.
// ------------------------------------------------------------------------------------------------
struct DEnc0D : DEnc0 { static constexpr bool MIX_EPI = true; static constexpr int SRC = SRC_POOLEXP_DIFF; };
struct DEnc0DP : DEnc0D { static constexpr int TPW = 4; };
struct MixBwdArgs {
    const uint8_t* a; const uint8_t* b; const float* z; float* dzpre;
    int n_a, inject;
    float l1s, l2s;
    const float* vf_pred;
};

// (round 5) nbw1 > 0: features.3's sparse weight gradient (pw1; the role conv_bwd_both_kernel<WEnc1, ...> gave it) as nbw1 more workgroups
// behind features.0's -- its data gradient now runs inside the tail backward launch (cgs_tail_enc_bwd_enc1), and only the step's final
// reduction waits for this slab.
#ifndef CGS_R1_MIX
#define CGS_R1_MIX 1
#endif
#ifndef CGS_MIX_WAVES
#define CGS_MIX_WAVES 0
#endif
#if CGS_MIX_WAVES
#define CGS_MIX_OCC __attribute__((amdgpu_waves_per_eu(CGS_MIX_WAVES, CGS_MIX_WAVES)))
#else
#define CGS_MIX_OCC
#endif
template <class CWG, bool SPARSE, bool R1>       // R1: the instance that carries the features.3 rider role (a role's registers are the whole launch's)
__global__ void __launch_bounds__(256) CGS_MIX_OCC enc0_bwd_mix_kernel(WgradParams pw, ConvParams pd, MixBwdArgs M, int nbw, WgradParams pw1, int nbw1) {
    if (CGS_KARG_PREFETCH) cgs_kernarg_prefetch<2 * sizeof(WgradParams) + sizeof(ConvParams) + sizeof(MixBwdArgs) + 16>();
    using G = Geo<DEnc0::H, DEnc0::W, DEnc0::THREADS, DEnc0::CW>;
    static_assert(CWG::G::THREADS == 256 && DEnc0::THREADS * DEnc0::CW == 256 && G::IMGS == 1, "workgroup shape");
    static_assert(WEnc1::G::THREADS == 256, "workgroup shape");
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    // block order (A/B: CGS_R1_MIX): 0 = [riders | features.0 weight gradient | data gradient], 1 = [w | riders | d], 2 = [w | d | riders]
    const int gx = (int)gridDim.x, bx = (int)blockIdx.x;
    const int r_lo = CGS_R1_MIX == 0 ? 0 : (CGS_R1_MIX == 1 ? nbw : gx - nbw1);
    if constexpr (R1) {
        if (bx >= r_lo && bx < r_lo + nbw1) {
            constexpr int SLAB1 = (9 * 8 + 1) * 8;
            const int b1 = bx - r_lo;
            wgrad_dispatch<WEnc1, SPARSE>(pw1, b1, nbw1, pw1.ntiles, pw1.slab + (size_t)b1 * SLAB1, smem);
            return;
        }
    }
    const int bm = bx - ((R1 && bx >= r_lo) ? nbw1 : 0);       // index among the launch's own roles
    if (bm < nbw) {
        constexpr int SLAB = (9 * 3 + 1) * 8;
#ifdef CGS_WHATIF_MIX_NOW      // (timing experiments only: wrong results) one role of the launch compiled out
        if (pw.n >= 0) return;
#endif
        wgrad_dispatch<CWG, SPARSE>(pw, bm, nbw, pw.ntiles, pw.slab + (size_t)bm * SLAB, smem);
        return;
    }
#ifdef CGS_WHATIF_MIX_NOD
    if (pw.n >= 0) return;
#endif
    const int bid = bm - nbw;
    pd.mix_a = M.a; pd.mix_b = M.b; pd.mix_z = M.z; pd.mix_dz = M.dzpre; pd.mix_l1s = M.l1s; pd.mix_l2s = M.l2s; pd.mix_vf_pred = M.vf_pred;
    pd.mix_inject = M.inject;
    pd.mix_n_a = M.n_a;
    if constexpr (CGS_CONV_PIPE) conv3x3_body_pipe<DEnc0DP>(pd, bid * DEnc0DP::TPW, smem);
    else conv3x3_body<DEnc0D, true>(pd, bid, smem);
}

extern "C" int cgs_enc0_bwd_mix_slabs(int32_t n_mix) { return n_mix < 0 ? CGS_ERR_BADARG : both_slabs<WEnc0F32>(n_mix); }

// e0_1 / dy1 / am1 / slab1 (all or none): features.3's weight gradient over the same n_mix images (cgs_conv3x3_bwd_weight of the 8 -> 8
// layer at 32x32 with ReLU + pool: input e0_1 [n_mix,32,32,8], pooled output gradient dy1 [n_mix,16,16,8], argmax nibbles am1, slab1
// [cgs_enc1_wgrad_rider_slabs(n_mix)][584]) as extra workgroups of this launch (round 5).
extern "C" int cgs_enc1_wgrad_rider_slabs(int32_t n) { return n < 0 ? CGS_ERR_BADARG : both_slabs<WEnc1>(n); }

extern "C" int cgs_enc0_bwd_mix_enc1(int32_t n_a, int32_t inject, const float* mixed, const float* dy, const uint32_t* amask,
                                     const float* w, const uint8_t* a, const uint8_t* b, const float* z, float l1_scale,
                                     float l2_scale, const float* valuefak_pred, float* dzpre, float* slab,
                                     const float* e0_1, const float* dy1, const uint32_t* am1, float* slab1, cgs_stream_t stream) {
    if (n_a < 0 || !dy || !amask || !w || !a || !b || !z || !dzpre) return CGS_ERR_BADARG;
    if (mixed && !slab) return CGS_ERR_BADARG;           // `mixed` is only the weight gradient's input
    if (slab1 && (!e0_1 || !dy1 || !am1)) return CGS_ERR_BADARG;
    if (n_a == 0) return CGS_OK;
    using GW = WEnc0F32::G;
    using GD = Geo<DEnc0::H, DEnc0::W, DEnc0::THREADS, DEnc0::CW>;
    const int n_mix = inject ? 2 * n_a : n_a;
    WgradParams pw{};
    pw.src_a = mixed; pw.dy = dy; pw.amask = amask; pw.slab = slab; pw.n = n_mix;
    pw.ntiles = n_mix * GW::STRIPS;
    pw.mix_a = a; pw.mix_b = b; pw.mix_z = z; pw.mix_n_a = n_a;
    ConvParams pd{};
    pd.src_a = dy; pd.amask_in = amask; pd.w = w; pd.n = n_mix;
    MixBwdArgs M{a, b, z, dzpre, n_a, inject ? 1 : 0, l1_scale, l2_scale, valuefak_pred};
    const int nbw = slab ? both_slabs<WEnc0F32>(n_mix) : 0;
    const int nbd = CGS_CONV_PIPE ? n_a * GD::STRIPS / DEnc0DP::TPW : n_a * GD::STRIPS;
    const bool sp = wgrad_sparse_enabled() != 0;
    WgradParams pw1{};
    int nbw1 = 0;
    size_t lw1 = 0;
    if (slab1) {
        using GW1 = WEnc1::G;
        pw1.src_a = e0_1; pw1.dy = dy1; pw1.amask = am1; pw1.slab = slab1; pw1.n = n_mix;
        pw1.ntiles = (GW1::IMGS == 1) ? n_mix * GW1::STRIPS : (n_mix + GW1::IMGS - 1) / GW1::IMGS;
        nbw1 = both_slabs<WEnc1>(n_mix);
        lw1 = sp ? wgrad_any_lds_bytes<WEnc1, true>() : wgrad_lds_bytes<WEnc1>();
    }
    const size_t lw = sp ? wgrad_any_lds_bytes<WEnc0F32, true>() : wgrad_lds_bytes<WEnc0F32>(), ld = conv_lds_bytes<DEnc0>();
    size_t lds = lw > ld ? lw : ld;                    // 41 KB (data-gradient tile): three workgroups per CU
    if (lw1 > lds) lds = lw1;
    const dim3 grid(nbw + nbd + nbw1);
    const bool r1 = nbw1 > 0;
    auto k = (mixed || !slab)       // materialised mixes (or no weight gradient at all) / mixes recomputed in the tile loader
                 ? (sp ? (r1 ? enc0_bwd_mix_kernel<WEnc0F32, true, true> : enc0_bwd_mix_kernel<WEnc0F32, true, false>)
                       : (r1 ? enc0_bwd_mix_kernel<WEnc0F32, false, true> : enc0_bwd_mix_kernel<WEnc0F32, false, false>))
                 : (sp ? (r1 ? enc0_bwd_mix_kernel<WEnc0Mix, true, true> : enc0_bwd_mix_kernel<WEnc0Mix, true, false>)
                       : (r1 ? enc0_bwd_mix_kernel<WEnc0Mix, false, true> : enc0_bwd_mix_kernel<WEnc0Mix, false, false>));
    hipLaunchKernelGGL(k, grid, dim3(256), lds, (hipStream_t)stream, pw, pd, M, nbw, pw1, nbw1);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_enc0_bwd_mix(int32_t n_a, int32_t inject, const float* mixed, const float* dy, const uint32_t* amask,
                                const float* w, const uint8_t* a, const uint8_t* b, const float* z, float l1_scale,
                                float l2_scale, const float* valuefak_pred, float* dzpre, float* slab, cgs_stream_t stream) {
    return cgs_enc0_bwd_mix_enc1(n_a, inject, mixed, dy, amask, w, a, b, z, l1_scale, l2_scale, valuefak_pred, dzpre, slab,
                                 nullptr, nullptr, nullptr, nullptr, stream);
}

// bf16 TRAINING kernels of BASELINE config 5 (the build-defined 128x128 Hourglass, hourglass128.py): bf16 activations and activation
// gradients in HBM, fp32 accumulation, fp32 master weights.  The forward pass and the data gradients run on the 16-bit convolution of
// gen_f16.hip (the data gradient = the same kernel on the flipped / transposed operand, cgs_genbf16_pack_weights_t); this file holds
//
//   bf16_wgrad_kernel   : weight + bias gradient of conv3x3(cat(A, nearest-up(B))) as a GEMM over the PIXELS on
//                         v_mfma_f32_16x16x32_bf16 (gfx950's K = 32 form): D[ci][co] += sum_{32 pixels} X[pixel + tap][ci] dY[pixel][co].
//                         Both operands want 8 consecutive K (= pixels) of one channel per lane while the tiles are NHWC in LDS
//                         ([pixel][16 channels], 32 B per pixel): ds_read_b64_tr_b16 -- gfx950's transposing LDS read, a 4-pixel x
//                         16-channel block delivered channel-major to a 16-lane group -- supplies them without a transposed copy.
//                         One wave owns taps {w, w + 4, w + 8} (no cross-wave reduction), accumulators persist over a workgroup's
//                         tiles, one slab row per workgroup -> cgs_reduce_slabs.
//   element-wise steps  : pooled-gradient re-expansion (argmax bytes of the forward pass), LeakyReLU', the split of d cat(A, up(B)) into
//                         the skip gradient and the cell-summed low-resolution gradient, fp32 <-> bf16 copies with channel padding.
//
// No reference counterpart (the reference cannot run 128x128 frames): parity unpinned, see hourglass128.py.
#include "gen_common.h"

namespace {

typedef short short4_t __attribute__((ext_vector_type(4)));
typedef short short8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) short4_t lds_short4_t;

__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }      // round to nearest even
__device__ __forceinline__ float bf2f(uint16_t u) { return __uint_as_float((uint32_t)u << 16); }

// 4 pixels x 16 channels of an NHWC LDS tile, delivered channel-major: lane i of a 16-lane group gets channel i of the block's 4
// pixels.  `addr` = this lane's part of the block: pixel (lane & 15) >> 2, channels 4 (lane & 3) .. + 3.  EXEC must be all ones.
__device__ __forceinline__ short4_t tr_read(const uint16_t* addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)addr);
}
__device__ __forceinline__ bf16x8_t pack8(short4_t a, short4_t b) {
    const short8_t s = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return __builtin_bit_cast(bf16x8_t, s);
}

struct WgParams {
    const void* a; const uint16_t* b; const uint16_t* dy; float* slab;
    int a_kind, ca, cb, ups, n, hw, co;     // a_kind: 0 bf16 (ca % 4 == 0), 1 uint8 (/255), 2 fp32
    int th, ti, ntiles, nblocks;            // tile = ti images x th rows x hw columns (a multiple of 32 pixels)
    int dyc;                                // channels per pixel of dY in memory (>= co, a multiple of 4: zero-padded columns)
};

template <int NCI>
__global__ void __launch_bounds__(256) bf16_wgrad_kernel(WgParams P) {
    extern __shared__ __attribute__((aligned(16))) float4 wsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4, q = l15 >> 2, p = l15 & 3;
    const int H = P.hw, W = P.hw, TH = P.th, TI = P.ti, PW = W + 2, PH = TH + 2;
    const int pa4 = (P.ca + 3) & ~3, cp = pa4 + P.cb;
    constexpr int CIP = NCI * 16;
    const int TP = TI * TH * W;
    uint16_t* const xt = (uint16_t*)wsm;                          // [TI][PH][PW][CIP]
    uint16_t* const dt = xt + (size_t)TI * PH * PW * CIP;         // [TP][16]
    const int ush = P.ups == 4 ? 2 : (P.ups == 2 ? 1 : 0), HB = H >> ush, WB = W >> ush;
    const int lgW = __builtin_ctz(W), lgTW = __builtin_ctz(TH * W);
    const int strips = H / TH;

    frag4 acc[3][NCI];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int c = 0; c < NCI; ++c) acc[s][c] = frag4{0.f, 0.f, 0.f, 0.f};
    frag4 accb = frag4{0.f, 0.f, 0.f, 0.f};
    const short one = (short)0x3F80;                              // bf16 1.0: the bias gradient = column sums of dY (A = all ones)
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, short8_t{one, one, one, one, one, one, one, one});

    for (int t = blockIdx.x; t < P.ntiles; t += P.nblocks) {
        const int img0 = TI == 1 ? t / strips : t * TI, row0 = TI == 1 ? (t % strips) * TH : 0;
        // ---- X tile: the virtual cat(A, up(B)) with a one-pixel zero halo, channel quads beyond the layer's zero ----
        // (flat over (tile pixel, channel chunk): divisions by the run-time tile width through an exact float reciprocal -- a half-integer
        //  over PW is never within 0.5 / PW of an integer, far outside the rounding error at these magnitudes; the row-by-row form
        //  measured slower: 520 items per 128-pixel row leave most of the third round idle)
        const float inv_pw = 1.f / (float)PW, inv_ph = 1.f / (float)PH;
        if (P.a_kind == 0 && !(P.ca & 7) && !(P.cb & 7)) {
            // bf16 sources in 8-channel chunks: one 16-byte load and one 16-byte LDS store per item
            constexpr int NO = CIP / 8;
            for (int e = tid; e < TI * PH * PW * NO; e += 256) {
                const int g = e % NO, pix = e / NO;
                const int prow = (int)(((float)pix + 0.5f) * inv_pw), xx = pix - prow * PW;
                const int ii = TI == 1 ? 0 : (int)(((float)prow + 0.5f) * inv_ph), rr = prow - ii * PH;
                const int img = img0 + ii, y = row0 + rr - 1, x = xx - 1, k0 = 8 * g;
                short8_t v = short8_t{0, 0, 0, 0, 0, 0, 0, 0};
                if (img < P.n && y >= 0 && y < H && x >= 0 && x < W && k0 < cp) {
                    if (k0 < pa4) v = *(const short8_t*)((const uint16_t*)P.a + (((size_t)img * H + y) * W + x) * P.ca + k0);
                    else v = *(const short8_t*)(P.b + (((size_t)img * HB + (y >> ush)) * WB + (x >> ush)) * P.cb + (k0 - pa4));
                }
                *(short8_t*)(xt + (size_t)pix * CIP + k0) = v;
            }
        } else {
        constexpr int NQ = CIP / 4;
        for (int e = tid; e < TI * PH * PW * NQ; e += 256) {
            const int g = e % NQ, pix = e / NQ;
            const int prow = (int)(((float)pix + 0.5f) * inv_pw), xx = pix - prow * PW;
            const int ii = TI == 1 ? 0 : (int)(((float)prow + 0.5f) * inv_ph), rr = prow - ii * PH;
            const int img = img0 + ii, y = row0 + rr - 1, x = xx - 1, k0 = 4 * g;
            short4_t v = short4_t{0, 0, 0, 0};
            if (img < P.n && y >= 0 && y < H && x >= 0 && x < W && k0 < cp) {
                if (k0 < pa4) {
                    const size_t pg = ((size_t)img * H + y) * W + x;
                    if (P.a_kind == 0) {
                        v = *(const short4_t*)((const uint16_t*)P.a + pg * P.ca + k0);
                    } else {
                        float f[4] = {0.f, 0.f, 0.f, 0.f};
                        for (int c = 0; c < 4; ++c)
                            if (k0 + c < P.ca)
                                f[c] = P.a_kind == 1 ? (float)((const uint8_t*)P.a)[pg * P.ca + k0 + c] * (1.f / 255.f) : ((const float*)P.a)[pg * P.ca + k0 + c];
                        v = short4_t{(short)f2bf(f[0]), (short)f2bf(f[1]), (short)f2bf(f[2]), (short)f2bf(f[3])};
                    }
                } else {
                    const size_t pb = ((size_t)img * HB + (y >> ush)) * WB + (x >> ush);
                    v = *(const short4_t*)(P.b + pb * P.cb + (k0 - pa4));
                }
            }
            *(short4_t*)(xt + (size_t)pix * CIP + k0) = v;
        }
        }
        // ---- dY tile: [pixel][16 output channels] ----
        if (!(P.dyc & 7)) {                 // 8-channel chunks: 16-byte accesses
            for (int e = tid; e < TP * 2; e += 256) {
                const int g = e & 1, pl = e >> 1, ii = pl >> lgTW, rem = pl & (TH * W - 1), y = rem >> lgW, x = rem & (W - 1);
                const int img = img0 + ii;
                short8_t v = short8_t{0, 0, 0, 0, 0, 0, 0, 0};
                if (img < P.n && 8 * g < P.dyc) v = *(const short8_t*)(P.dy + (((size_t)img * H + row0 + y) * W + x) * P.dyc + 8 * g);
                *(short8_t*)(dt + (size_t)pl * 16 + 8 * g) = v;
            }
        } else {
            for (int e = tid; e < TP * 4; e += 256) {
                const int g = e & 3, pl = e >> 2, ii = pl >> lgTW, rem = pl & (TH * W - 1), y = rem >> lgW, x = rem & (W - 1);
                const int img = img0 + ii;
                short4_t v = short4_t{0, 0, 0, 0};
                if (img < P.n && 4 * g < P.dyc) v = *(const short4_t*)(P.dy + (((size_t)img * H + row0 + y) * W + x) * P.dyc + 4 * g);
                *(short4_t*)(dt + (size_t)pl * 16 + 4 * g) = v;
            }
        }
        __syncthreads();
        // ---- K blocks of 32 pixels: lane group kq supplies pixels 8 kq .. 8 kq + 7 (two transposed 4-pixel reads) ----
        for (int blk = 0; blk < TP / 32; ++blk) {
            int pa[2];
            short4_t bv[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int p0 = 32 * blk + 8 * kq + 4 * h, ii = p0 >> lgTW, rem = p0 & (TH * W - 1), y = rem >> lgW, x0 = rem & (W - 1);
                bv[h] = tr_read(dt + (size_t)(p0 + q) * 16 + 4 * p);
                pa[h] = (ii * PH + y) * PW + x0 + q;
            }
            const bf16x8_t B = pack8(bv[0], bv[1]);
            if (wave == 3) accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, B, accb, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int tap = wave + 4 * s;
                if (tap < 9) {                                     // wave-uniform
                    const int toff = (tap / 3) * PW + tap % 3;
#pragma unroll
                    for (int c = 0; c < NCI; ++c) {
                        const short4_t a0 = tr_read(xt + (size_t)(pa[0] + toff) * CIP + 16 * c + 4 * p);
                        const short4_t a1 = tr_read(xt + (size_t)(pa[1] + toff) * CIP + 16 * c + 4 * p);
                        acc[s][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pack8(a0, a1), B, acc[s][c], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    // ---- slab row [9][ca + cb][co] + [co]: D[m = 4 kq + r][n = l15], m = padded input channel, n = output channel ----
    const int ci_total = P.ca + P.cb;
    float* const sl = P.slab + (size_t)blockIdx.x * ((size_t)9 * ci_total * P.co + P.co);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int tap = wave + 4 * s;
        if (tap >= 9) continue;
#pragma unroll
        for (int c = 0; c < NCI; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * c + 4 * kq + r;
                const int ci = k < pa4 ? (k < P.ca ? k : -1) : (k < cp ? P.ca + (k - pa4) : -1);
                if (ci >= 0 && l15 < P.co) sl[((size_t)tap * ci_total + ci) * P.co + l15] = acc[s][c][r];
            }
    }
    if (wave == 3 && kq == 0 && l15 < P.co) sl[(size_t)9 * ci_total * P.co + l15] = accb[0];
}

// dY_full [n,2h,2w,c] = (dP [n,h,w,c] (+ addend)) where the forward argmax byte == 2 (y & 1) + (x & 1), else 0.  One thread = 4 channels
// of one pooled pixel (c % 4 == 0).
__global__ void __launch_bounds__(256) bf16_pool_expand_kernel(const uint16_t* __restrict__ dp, const uint16_t* __restrict__ addend,
                                                               const uint8_t* __restrict__ codes, int64_t quads, int hp, int c,
                                                               uint16_t* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= quads) return;
    const int cq = c / 4;
    const int g = (int)(e % cq);
    const int64_t pix = e / cq;                 // (n, y, x) of the pooled map
    const int x = (int)(pix % hp), y = (int)((pix / hp) % hp);
    const int64_t img = pix / ((int64_t)hp * hp);
    const short4_t d = *(const short4_t*)(dp + pix * c + 4 * g);
    float f[4] = {bf2f((uint16_t)d.x), bf2f((uint16_t)d.y), bf2f((uint16_t)d.z), bf2f((uint16_t)d.w)};
    if (addend) {
        const short4_t a = *(const short4_t*)(addend + pix * c + 4 * g);
        f[0] += bf2f((uint16_t)a.x); f[1] += bf2f((uint16_t)a.y); f[2] += bf2f((uint16_t)a.z); f[3] += bf2f((uint16_t)a.w);
    }
    const uint32_t cd = *(const uint32_t*)(codes + pix * c + 4 * g);
    const int W = 2 * hp;
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
        short4_t v;
        v.x = ((cd & 255u) == (uint32_t)pos) ? (short)f2bf(f[0]) : (short)0;
        v.y = (((cd >> 8) & 255u) == (uint32_t)pos) ? (short)f2bf(f[1]) : (short)0;
        v.z = (((cd >> 16) & 255u) == (uint32_t)pos) ? (short)f2bf(f[2]) : (short)0;
        v.w = ((cd >> 24) == (uint32_t)pos) ? (short)f2bf(f[3]) : (short)0;
        const int64_t o = ((img * W + 2 * y + (pos >> 1)) * W + 2 * x + (pos & 1)) * c + 4 * g;
        *(short4_t*)(out + o) = v;
    }
}

// d cat(A [ca], up_ups(B [cb])) [n,hw,hw,ca+cb] -> d_skip [n,hw,hw,ca] (optional) and d_low [n,hw/ups,hw/ups,cb] = the sum over each
// ups x ups cell (bf16, or fp32 when low_f32: the bottleneck's gradient feeds the fp32 head GEMMs)
__global__ void __launch_bounds__(256) bf16_cat_split_kernel(const uint16_t* __restrict__ dcat, int n, int hw, int ca, int cb, int ups,
                                                             uint16_t* __restrict__ dskip, void* __restrict__ dlow, int low_f32) {
    const int ct = ca + cb, hl = hw / ups;
    const int64_t n_skip = dskip ? (int64_t)n * hw * hw * ca : 0, n_low = (int64_t)n * hl * hl * cb;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_skip + n_low; e += (int64_t)gridDim.x * 256) {
        if (e < n_skip) {
            const int c = (int)(e % ca);
            const int64_t pix = e / ca;
            dskip[e] = dcat[pix * ct + c];
        } else {
            const int64_t r = e - n_skip;
            const int c = (int)(r % cb);
            const int64_t lp = r / cb;
            const int lx = (int)(lp % hl), ly = (int)((lp / hl) % hl);
            const int64_t img = lp / ((int64_t)hl * hl);
            float s = 0.f;
            for (int dy = 0; dy < ups; ++dy)
                for (int dx = 0; dx < ups; ++dx)
                    s += bf2f(dcat[((img * hw + ly * ups + dy) * hw + lx * ups + dx) * ct + ca + c]);
            if (low_f32) ((float*)dlow)[r] = s; else ((uint16_t*)dlow)[r] = f2bf(s);
        }
    }
}

// d [i] *= h[i] > 0 ? 1 : slope   (LeakyReLU', from the saved OUTPUT: its sign is the pre-activation's)
__global__ void __launch_bounds__(256) bf16_lrelu_bwd_kernel(uint16_t* __restrict__ d, const uint16_t* __restrict__ h, int64_t count, float slope) {
    const int64_t quads = count / 4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < quads; e += (int64_t)gridDim.x * 256) {
        short4_t dv = ((const short4_t*)d)[e];
        const short4_t hv = ((const short4_t*)h)[e];
        if (!(bf2f((uint16_t)hv.x) > 0.f)) dv.x = (short)f2bf(bf2f((uint16_t)dv.x) * slope);
        if (!(bf2f((uint16_t)hv.y) > 0.f)) dv.y = (short)f2bf(bf2f((uint16_t)dv.y) * slope);
        if (!(bf2f((uint16_t)hv.z) > 0.f)) dv.z = (short)f2bf(bf2f((uint16_t)dv.z) * slope);
        if (!(bf2f((uint16_t)hv.w) > 0.f)) dv.w = (short)f2bf(bf2f((uint16_t)dv.w) * slope);
        ((short4_t*)d)[e] = dv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
        const int64_t e = 4 * quads + threadIdx.x;
        if (!(bf2f(h[e]) > 0.f)) d[e] = f2bf(bf2f(d[e]) * slope);
    }
}

// rows of c_src channels -> rows of c_dst channels (zero padded / truncated); dir 0: fp32 -> bf16, 1: bf16 -> fp32
__global__ void __launch_bounds__(256) bf16_convert_kernel(const void* __restrict__ src, void* __restrict__ dst, int64_t rows, int c_src, int c_dst, int dir) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < rows * c_dst; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % c_dst);
        const int64_t r = e / c_dst;
        if (dir == 0) ((uint16_t*)dst)[e] = c < c_src ? f2bf(((const float*)src)[r * c_src + c]) : (uint16_t)0;
        else ((float*)dst)[e] = c < c_src ? bf2f(((const uint16_t*)src)[r * c_src + c]) : 0.f;
    }
}

struct WgGeom { int th, ti, ntiles, nblocks; size_t lds; int nci; };

WgGeom wg_geom(int n, int hw, int ca, int cb) {
    WgGeom g{};
    const int cp = ((ca + 3) & ~3) + cb;
    g.nci = (cp + 15) / 16;
    if (hw >= 32) { g.ti = 1; g.th = (g.nci == 1 ? 1024 : 512) / hw; }      // 1024-pixel tiles (less halo re-staging) while the tile fits
    else if (hw == 16) { g.ti = 2; g.th = 16; }
    else if (hw == 8) { g.ti = 8; g.th = 8; }
    else { g.ti = 16; g.th = 4; }
    g.ntiles = g.ti == 1 ? n * (hw / g.th) : (n + g.ti - 1) / g.ti;
    g.nblocks = g.ntiles < 512 ? g.ntiles : 512;
    g.lds = (size_t)g.ti * (g.th + 2) * (hw + 2) * g.nci * 16 * 2 + (size_t)g.ti * g.th * hw * 32;
    return g;
}

bool bf16_hw_ok(int hw) { return hw == 4 || hw == 8 || hw == 16 || hw == 32 || hw == 64 || hw == 128; }

}  // namespace

extern "C" int cgs_bf16_conv3x3_bwd_weight_slabs(int32_t n, int32_t hw, int32_t ca, int32_t cb) {
    if (n < 0 || !bf16_hw_ok(hw) || ca <= 0 || cb < 0) return CGS_ERR_BADARG;
    return wg_geom(n, hw, ca, cb).nblocks;
}

extern "C" int cgs_bf16_conv3x3_bwd_weight(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t dy_channels, int32_t a_kind,
                                           int32_t ups, const void* src_a, const void* src_b, const void* dy, float* slab, cgs_stream_t stream) {
    if (n < 0 || !src_a || !dy || !slab || ca <= 0 || cb < 0 || co <= 0 || co > 16 || a_kind < 0 || a_kind > 2) return CGS_ERR_BADARG;
    if (dy_channels < co || dy_channels > 16 || (dy_channels & 3)) return CGS_ERR_BADARG;
    if (!bf16_hw_ok(hw) || (a_kind == 0 && (ca & 3)) || (cb > 0 && (!src_b || (cb & 3) || (ups != 2 && ups != 4)))) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    const WgGeom g = wg_geom(n, hw, ca, cb);
    if (g.nci > 3) return CGS_ERR_UNSUPPORTED;            // chfak 1 shapes: at most 48 (padded) input channels
    WgParams P{src_a, (const uint16_t*)src_b, (const uint16_t*)dy, slab, a_kind, ca, cb, cb > 0 ? ups : 1, n, hw, co, g.th, g.ti, g.ntiles, g.nblocks, dy_channels};
    auto k = g.nci == 1 ? bf16_wgrad_kernel<1> : (g.nci == 2 ? bf16_wgrad_kernel<2> : bf16_wgrad_kernel<3>);
    if (g.lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k, dim3(g.nblocks), dim3(256), g.lds, (hipStream_t)stream, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_bf16_pool_expand(int32_t n, int32_t hp, int32_t c, const void* dp, const void* addend, const uint8_t* codes, void* out,
                                    cgs_stream_t stream) {
    if (n < 0 || hp <= 0 || c <= 0 || (c & 3) || !dp || !codes || !out) return CGS_ERR_BADARG;
    const int64_t quads = (int64_t)n * hp * hp * (c / 4);
    if (quads == 0) return CGS_OK;
    hipLaunchKernelGGL(bf16_pool_expand_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)dp, (const uint16_t*)addend, codes, quads, hp, c, (uint16_t*)out);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_bf16_cat_split(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t ups, const void* dcat, void* dskip, void* dlow,
                                  int32_t low_is_f32, cgs_stream_t stream) {
    if (n < 0 || hw <= 0 || ca < 0 || cb <= 0 || (ups != 2 && ups != 4) || hw % ups || !dcat || !dlow) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    const int64_t total = (dskip ? (int64_t)n * hw * hw * ca : 0) + (int64_t)n * (hw / ups) * (hw / ups) * cb;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(bf16_cat_split_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)dcat, n, hw, ca, cb, ups, (uint16_t*)dskip, dlow, low_is_f32);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_bf16_lrelu_bwd(int64_t count, void* d, const void* h, float slope, cgs_stream_t stream) {
    if (count < 0 || !d || !h) return CGS_ERR_BADARG;
    if (count == 0) return CGS_OK;
    const int64_t blocks = (count / 4 + 255) / 256 + 1;
    hipLaunchKernelGGL(bf16_lrelu_bwd_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream,
                       (uint16_t*)d, (const uint16_t*)h, count, slope);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_bf16_convert(int64_t rows, int32_t c_src, int32_t c_dst, int32_t to_f32, const void* src, void* dst, cgs_stream_t stream) {
    if (rows < 0 || c_src <= 0 || c_dst <= 0 || !src || !dst) return CGS_ERR_BADARG;
    if (rows == 0) return CGS_OK;
    const int64_t blocks = (rows * c_dst + 255) / 256;
    hipLaunchKernelGGL(bf16_convert_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, src, dst, rows,
                       c_src, c_dst, to_f32);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// Shape-generic kernels (runtime channel counts and map sizes) for the model sizes the hand-specialised kernels do not
// cover: NewCritic / UnetDecoder with chfak != 1 (the paper's model is chfak = 5: 40/40/40/80/160 channels, docs/index.html:151,
// nets.py:166,184,190) and the legacy single-module hourglass `Unet` with its ConvTranspose2d(4,2,1) decoder and
// LeakyReLU(0.2) (nets.py:356-449).  Forward passes only for the two big families (inference: `-process`, `-eval`, the module
// API in eval mode); the transposed convolution also has its data- and weight-gradient kernels.
//
//   gen_conv3x3_fwd : Conv2d(3x3, s1, p1) over cat(A, nearest-up(B)) + bias + {none, ReLU, LeakyReLU(slope), sigmoid}
//                     (+ MaxPool2d(2) with a 2-bit argmax) as an implicit GEMM on v_mfma_f32_16x16x4_f32: NHWC LDS tile of 16
//                     input channels at a time, weights from memory (HWIO), accumulators persistent over the channel chunks.
//   gen_gemm        : out[n][N] = act(X[n][K] W[K][N] + b): the 4x4 valid convolution at the bottleneck, the Linear layers,
//                     the 1x1 convolution, ConvTranspose2d(4,1,0) on a 1x1 map.
//   gen_convt4s2_*  : ConvTranspose2d(4, 2, 1) over cat(A, B): forward, data gradient, weight gradient (direct form).
#include "gen_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// out[m][n] = act(sum_k X[m][k] W[k][n] + bias[n]);  one wave per 16 x 16 output tile (v_mfma_f32_16x16x4_f32)
// ------------------------------------------------------------------------------------------------
struct GenGemmParams {
    const float* x; const float* w; const float* bias; float* out;
    int m, k, n, act;
    float slope;
};

__global__ void __launch_bounds__(64) gen_gemm_kernel(GenGemmParams P) {
    const int lane = threadIdx.x, l15 = lane & 15, kq = lane >> 4;
    const int ntn = (P.n + 15) / 16;
    const int m0 = (blockIdx.x / ntn) * 16, n0 = (blockIdx.x % ntn) * 16;
    const int row = m0 + l15, col = n0 + l15;
    const float* xr = P.x + (size_t)(row < P.m ? row : 0) * P.k;
    frag4 acc = frag4{0.f, 0.f, 0.f, 0.f};
    const bool rok = row < P.m, cok = col < P.n;
    const float* wc = P.w + (cok ? col : 0);
    for (int k0 = 0; k0 < P.k; k0 += 32) {          // 8 k-steps per round: 16 independent loads in flight, then 8 MFMAs
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + kq, kc = k < P.k ? k : P.k - 1;
            a[u] = xr[kc];
            b[u] = wc[(size_t)kc * P.n];
            a[u] = (rok && k < P.k) ? a[u] : 0.f;
            b[u] = (cok && k < P.k) ? b[u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
    }
    if (col < P.n) {
        const float bias = P.bias ? P.bias[col] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = m0 + 4 * kq + j;
            if (r < P.m) P.out[(size_t)r * P.n + col] = gen_act(acc[j] + bias, P.act, P.slope);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose2d(kernel 4, stride 2, padding 1) over cat(A [ca], B [cb]) (both [n,h,w,*] NHWC fp32): out [n,2h,2w,co]
//   out[oy][ox][co] = b[co] + sum_{ky,kx,ci : oy = 2 iy - 1 + ky, ox = 2 ix - 1 + kx} x[iy][ix][ci] W[ky][kx][ci][co]
// (kernel layout of the weight: [ky][kx][ci][co]; PyTorch stores [ci][co][ky][kx]).  Direct form, one thread per output element;
// the legacy model is small (<= 32 channels), this family is about exact semantics, not speed.
// ------------------------------------------------------------------------------------------------
struct GenConvTParams {
    const float* a; const float* b; const float* w; const float* bias; const float* dy;
    float* out; float* da; float* db; float* dw; float* dbias;
    int n, h, ca, cb, co, act;
    float slope;
};

__device__ __forceinline__ float convt_in(const GenConvTParams& P, int img, int iy, int ix, int ci) {
    const size_t pix = ((size_t)img * P.h + iy) * P.h + ix;
    return ci < P.ca ? P.a[pix * P.ca + ci] : P.b[pix * P.cb + (ci - P.ca)];
}

__global__ void __launch_bounds__(256) gen_convt_fwd_kernel(GenConvTParams P) {
    const int OH = 2 * P.h, ci_total = P.ca + P.cb;
    const size_t total = (size_t)P.n * OH * OH * P.co;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int co = e % P.co, ox = (e / P.co) % OH, oy = (e / ((size_t)P.co * OH)) % OH, img = e / ((size_t)P.co * OH * OH);
        float s = P.bias[co];
        for (int ky = (oy + 1) & 1; ky < 4; ky += 2) {
            const int iy = (oy + 1 - ky) >> 1;
            if (iy < 0 || iy >= P.h) continue;
            for (int kx = (ox + 1) & 1; kx < 4; kx += 2) {
                const int ix = (ox + 1 - kx) >> 1;
                if (ix < 0 || ix >= P.h) continue;
                const float* wp = P.w + ((size_t)(ky * 4 + kx) * ci_total) * P.co + co;
                for (int ci = 0; ci < ci_total; ++ci) s = fmaf(convt_in(P, img, iy, ix, ci), wp[(size_t)ci * P.co], s);
            }
        }
        P.out[e] = gen_act(s, P.act, P.slope);
    }
}

// dy: gradient w.r.t. the PRE-activation output [n,2h,2w,co].  da / db: gradients w.r.t. A and B.
__global__ void __launch_bounds__(256) gen_convt_bwd_data_kernel(GenConvTParams P) {
    const int OH = 2 * P.h, ci_total = P.ca + P.cb;
    const size_t total = (size_t)P.n * P.h * P.h * ci_total;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int ci = e % ci_total, ix = (e / ci_total) % P.h, iy = (e / ((size_t)ci_total * P.h)) % P.h,
                  img = e / ((size_t)ci_total * P.h * P.h);
        float s = 0.f;
        for (int ky = 0; ky < 4; ++ky) {
            const int oy = 2 * iy - 1 + ky;
            if (oy < 0 || oy >= OH) continue;
            for (int kx = 0; kx < 4; ++kx) {
                const int ox = 2 * ix - 1 + kx;
                if (ox < 0 || ox >= OH) continue;
                const float* dp = P.dy + (((size_t)img * OH + oy) * OH + ox) * P.co;
                const float* wp = P.w + ((size_t)(ky * 4 + kx) * ci_total + ci) * P.co;
                for (int co = 0; co < P.co; ++co) s = fmaf(dp[co], wp[co], s);
            }
        }
        const size_t pix = ((size_t)img * P.h + iy) * P.h + ix;
        if (ci < P.ca) { if (P.da) P.da[pix * P.ca + ci] = s; }
        else if (P.db) P.db[pix * P.cb + (ci - P.ca)] = s;
    }
}

// one workgroup per weight element (ky, kx, ci, co) (+ one per bias element): fixed-order tree sum over all pixels
__global__ void __launch_bounds__(256) gen_convt_bwd_weight_kernel(GenConvTParams P) {
    __shared__ float red[256];
    const int OH = 2 * P.h, ci_total = P.ca + P.cb, nw = 16 * ci_total * P.co;
    const int e = blockIdx.x;
    float s = 0.f;
    if (e < nw) {
        const int co = e % P.co, ci = (e / P.co) % ci_total, kx = (e / (P.co * ci_total)) % 4, ky = e / (P.co * ci_total * 4);
        const int npix = P.n * P.h * P.h;
        for (int p = threadIdx.x; p < npix; p += 256) {
            const int ix = p % P.h, iy = (p / P.h) % P.h, img = p / (P.h * P.h);
            const int oy = 2 * iy - 1 + ky, ox = 2 * ix - 1 + kx;
            if (oy < 0 || oy >= OH || ox < 0 || ox >= OH) continue;
            s = fmaf(convt_in(P, img, iy, ix, ci), P.dy[(((size_t)img * OH + oy) * OH + ox) * P.co + co], s);
        }
    } else {
        const int co = e - nw, npix = P.n * OH * OH;
        for (int p = threadIdx.x; p < npix; p += 256) s += P.dy[(size_t)p * P.co + co];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) { if (e < nw) P.dw[e] = red[0]; else P.dbias[e - nw] = red[0]; }
}

}  // namespace

// the 3x3 convolution: lane = pixel on v_mfma_f32_4x4x1 (gen4.hip, round 3)
struct Gen4Launch {
    GenSrc src; const float* wp; const float* bias; float* out; uint8_t* argmax; const float* addend;
    int n_addend, n, hw, co, act, pool;
    float slope;
    float* out2; int split_ca, split_ups;
    int fold;
};
int gen4_conv_launch(const Gen4Launch& L, hipStream_t st);
long gen4_packed_floats(int ca, int cb, int co, int fold);
int gen4_pack_launch(int ca, int cb, int co, int transposed, const float* w, float* wp, int ci_layer, int ci_off, hipStream_t st);
struct Gen4PackJob { const float* w; float* wp; int ca, cb, co, transposed, ci_layer, ci_off; };
int gen4_pack_batch_launch(const Gen4PackJob* jobs, int njobs, hipStream_t st);

extern "C" int64_t cgs_gen_conv_packed_floats(int32_t ca, int32_t cb, int32_t co) {
    if (ca <= 0 || cb < 0 || co <= 0 || (cb & 3)) return CGS_ERR_BADARG;
    return gen4_packed_floats(ca, cb, co, 0);
}

// the FOLDED forward operand of a layer over cat(A, nearest-up_2(B)) (cgs_gen_conv_pack_weights with transposed = 2, cgs_gen_conv3x3_fwd_folded)
extern "C" int64_t cgs_gen_conv_packed_floats_folded(int32_t ca, int32_t cb, int32_t co) {
    if (ca <= 0 || cb <= 0 || co <= 0 || (cb & 3)) return CGS_ERR_BADARG;
    return gen4_packed_floats(ca, cb, co, 1);
}

extern "C" int cgs_gen_conv_pack_weights(int32_t ca, int32_t cb, int32_t co, int32_t transposed, const float* w, float* wp,
                                         cgs_stream_t stream) {
    if (ca <= 0 || cb < 0 || co <= 0 || (cb & 3) || !w || !wp || transposed < 0 || transposed > 2) return CGS_ERR_BADARG;
    if ((transposed == 1 && cb) || (transposed == 2 && !cb)) return CGS_ERR_BADARG;
    return gen4_pack_launch(ca, cb, co, transposed, w, wp, co, 0, (hipStream_t)stream);
}

// The data gradient's operand for a WINDOW of the layer's input channels: w = HWIO [9][ci_layer][co_layer]; the packed operand maps dY
// (co_layer channels) to d(input channels [ci_off, ci_off + ci_n)) -- e.g. only the decoder channels of cat(image, up(o0)), whose image
// part needs no gradient.
extern "C" int cgs_gen_conv_pack_weights_window(int32_t co_layer, int32_t ci_layer, int32_t ci_off, int32_t ci_n, const float* w, float* wp,
                                                cgs_stream_t stream) {
    if (co_layer <= 0 || ci_layer <= 0 || ci_off < 0 || ci_n <= 0 || ci_off + ci_n > ci_layer || !w || !wp) return CGS_ERR_BADARG;
    return gen4_pack_launch(co_layer, 0, ci_n, 1, w, wp, ci_layer, ci_off, (hipStream_t)stream);
}

// The data gradient towards the nearest-upsampled (x2) source of a layer, computed at the source's resolution (gen4_conv3x3_kernel<NG, 2>):
// w = HWIO [9][ci_layer][co_layer]; the operand maps dY [n,hw,hw,co_layer] to d B [n,hw/2,hw/2,ci_n] = the gradient of input channels
// [ci_off, ci_off + ci_n) summed over each 2 x 2 cell -- what cgs_gen_conv3x3_bwd_data_split writes to d_b, with 16 instead of 36 steps per cell.
extern "C" int64_t cgs_gen_conv_packed_floats_up2(int32_t co_layer, int32_t ci_n) {
    if (co_layer <= 0 || ci_n <= 0 || (co_layer & 3)) return CGS_ERR_BADARG;
    return gen4_packed_floats(co_layer, 0, ci_n, 3);
}
extern "C" int cgs_gen_conv_pack_weights_up2(int32_t co_layer, int32_t ci_layer, int32_t ci_off, int32_t ci_n, const float* w, float* wp,
                                             cgs_stream_t stream) {
    if (co_layer <= 0 || (co_layer & 3) || ci_layer <= 0 || ci_off < 0 || ci_n <= 0 || ci_off + ci_n > ci_layer || !w || !wp) return CGS_ERR_BADARG;
    return gen4_pack_launch(co_layer, 0, ci_n, 3, w, wp, ci_layer, ci_off, (hipStream_t)stream);
}
extern "C" int cgs_gen_conv3x3_bwd_data_up2(int32_t n, int32_t hw, int32_t co_layer, int32_t ci_n, const float* dy, const float* wp, float* d_b,
                                            cgs_stream_t stream) {
    if (n < 0 || !dy || !wp || !d_b || co_layer <= 0 || ci_n <= 0) return CGS_ERR_BADARG;
    if ((hw != 16 && hw != 32 && hw != 64) || (co_layer & 3)) return CGS_ERR_UNSUPPORTED;
    if (n == 0) return CGS_OK;
    const int pbw = (co_layer + 15) / 16 * 16;
    Gen4Launch L{GenSrc{dy, nullptr, nullptr, GEN_SRC_F32, co_layer, pbw, 1}, wp, nullptr, d_b, nullptr, nullptr, 0, n, hw / 2, ci_n, CGS_ACT_NONE, 0, 0.f,
                 nullptr, 0, 0, 2};
    return gen4_conv_launch(L, (hipStream_t)stream);
}

// Every 3x3 layer's operand of one step in one launch: job i packs like cgs_gen_conv_pack_weights (ci_layer = 0) or
// cgs_gen_conv_pack_weights_window (ci_layer > 0: ca = the layer's output channels, co = the window's width).
extern "C" int cgs_gen_conv_pack_batch(const cgs_gen_pack_job* jobs, int32_t njobs, cgs_stream_t stream) {
    static_assert(sizeof(cgs_gen_pack_job) == sizeof(Gen4PackJob), "job layout");
    if (njobs < 0 || (njobs > 0 && !jobs)) return CGS_ERR_BADARG;
    if (njobs == 0) return CGS_OK;
    Gen4PackJob tmp[64];
    for (int j0 = 0; j0 < njobs; j0 += 64) {
        const int nb = njobs - j0 < 64 ? njobs - j0 : 64;
        for (int j = 0; j < nb; ++j) {
            const cgs_gen_pack_job& J = jobs[j0 + j];
            if (!J.w || !J.wp || J.ca <= 0 || J.cb < 0 || J.co <= 0 || (J.cb & 3) || J.transposed < 0 || J.transposed > 3) return CGS_ERR_BADARG;
            if (((J.transposed == 1 || J.transposed == 3) && J.cb) || (J.transposed == 2 && !J.cb)) return CGS_ERR_BADARG;
            if (J.transposed == 3 && (J.ci_layer <= 0 || (J.ca & 3))) return CGS_ERR_BADARG;      // (the up2 data-gradient operand: the window form)
            if (J.ci_layer > 0 && ((J.transposed != 1 && J.transposed != 3) || J.ci_off < 0 || J.ci_off + J.co > J.ci_layer)) return CGS_ERR_BADARG;
            // (a whole-layer operand: the forward form reads w as [9][ca + cb][co]; the transposed one as [9][co][ca])
            tmp[j] = Gen4PackJob{J.w, J.wp, J.ca, J.cb, J.co, J.transposed, J.ci_layer > 0 ? J.ci_layer : J.co, J.ci_layer > 0 ? J.ci_off : 0};
        }
        const int rc = gen4_pack_batch_launch(tmp, nb, (hipStream_t)stream);
        if (rc != CGS_OK) return rc;
    }
    return CGS_OK;
}

static int gen_conv3x3_fwd(int fold, int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_is_u8, int32_t ups,
                           int32_t act, float slope, int32_t pool, const void* src_a, const float* src_b, const float* wp,
                           const float* bias, float* out, uint8_t* argmax, cgs_stream_t stream) {
    if (n < 0 || !src_a || !wp || !bias || !out || ca <= 0 || cb < 0 || co <= 0) return CGS_ERR_BADARG;
    if (cb > 0 && (!src_b || (cb & 3) || (ups != 1 && ups != 2 && ups != 4))) return CGS_ERR_BADARG;
    if (!gen_hw_ok(hw)) return CGS_ERR_UNSUPPORTED;
    if (act < CGS_ACT_NONE || act > CGS_ACT_SIGMOID) return CGS_ERR_BADARG;
    if (fold && (cb <= 0 || ups != 2 || hw < 16 || pool)) return CGS_ERR_UNSUPPORTED;
    if (n == 0) return CGS_OK;
    Gen4Launch L{GenSrc{src_a, src_b, nullptr, a_is_u8 ? GEN_SRC_U8 : GEN_SRC_F32, ca, cb, cb > 0 ? ups : 1}, wp, bias, out, argmax,
                 nullptr, 0, n, hw, co, act, pool, slope, nullptr, 0, 0, fold};
    return gen4_conv_launch(L, (hipStream_t)stream);
}
extern "C" int cgs_gen_conv3x3_fwd(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_is_u8, int32_t ups,
                                   int32_t act, float slope, int32_t pool, const void* src_a, const float* src_b, const float* wp,
                                   const float* bias, float* out, uint8_t* argmax, cgs_stream_t stream) {
    return gen_conv3x3_fwd(0, n, hw, ca, cb, co, a_is_u8, ups, act, slope, pool, src_a, src_b, wp, bias, out, argmax, stream);
}
// the same layer with the FOLDED operand (cgs_gen_conv_pack_weights, transposed = 2): cb > 0, ups = 2, hw >= 16, pool = 0
extern "C" int cgs_gen_conv3x3_fwd_folded(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_is_u8, int32_t act,
                                          float slope, const void* src_a, const float* src_b, const float* wp, const float* bias,
                                          float* out, cgs_stream_t stream) {
    return gen_conv3x3_fwd(1, n, hw, ca, cb, co, a_is_u8, 2, act, slope, 0, src_a, src_b, wp, bias, out, nullptr, stream);
}

// Data gradient of a 3x3 layer = the same convolution over the output gradient with the flipped, transposed kernel:
// d_cat [n,hw,hw,ci] = conv3x3(dY [co channels], wp) (+ addend for images < n_addend), wp = cgs_gen_conv_pack_weights(co, 0, ci,
// transposed = 1, the layer's HWIO weights).
extern "C" int cgs_gen_conv3x3_bwd_data(int32_t n, int32_t hw, int32_t co, int32_t ci, const float* dy, const uint8_t* dy_argmax,
                                        const float* wp, const float* addend, int32_t n_addend, float* d_cat,
                                        cgs_stream_t stream) {
    if (n < 0 || !dy || !wp || !d_cat || co <= 0 || ci <= 0 || n_addend < 0) return CGS_ERR_BADARG;
    if (dy_argmax && (co & 3)) return CGS_ERR_BADARG;
    if (!gen_hw_ok(hw)) return CGS_ERR_UNSUPPORTED;
    if (n == 0) return CGS_OK;
    Gen4Launch L{GenSrc{dy, nullptr, dy_argmax, dy_argmax ? GEN_SRC_POOLEXP : GEN_SRC_F32, co, 0, 1}, wp, nullptr, d_cat, nullptr,
                 addend, n_addend, n, hw, ci, CGS_ACT_NONE, 0, 0.f, nullptr, 0, 0};
    return gen4_conv_launch(L, (hipStream_t)stream);
}

// The data gradient of a layer over cat(A [ca], nearest-up_ups(B [cb])) written straight as d_a [n,hw,hw,ca] (may be NULL) and d_b
// [n,hw/ups,hw/ups,cb] (the sum over each ups x ups cell) -- cgs_gen_conv3x3_bwd_data + cgs_gen_cat_split without the d_cat tensor.
// CGS_ERR_UNSUPPORTED when the kernel's output passes do not fall on one side of the split each (the caller then takes the two-step form).
extern "C" int cgs_gen_conv3x3_bwd_data_split(int32_t n, int32_t hw, int32_t co, int32_t ca, int32_t cb, int32_t ups, const float* dy,
                                              const float* wp, float* d_a, float* d_b, cgs_stream_t stream) {
    if (n < 0 || !dy || !wp || !d_b || co <= 0 || ca < 0 || cb <= 0 || (ca & 3) || (cb & 3) || (ca == 0 && d_a)) return CGS_ERR_BADARG;
    if (!gen_hw_ok(hw)) return CGS_ERR_UNSUPPORTED;
    if (n == 0) return CGS_OK;
    Gen4Launch L{GenSrc{dy, nullptr, nullptr, GEN_SRC_F32, co, 0, 1}, wp, nullptr, d_a, nullptr, nullptr, 0, n, hw, ca + cb, CGS_ACT_NONE, 0, 0.f,
                 d_b, ca, ups};
    return gen4_conv_launch(L, (hipStream_t)stream);
}

extern "C" int cgs_gen_gemm(int32_t m, int32_t k, int32_t n, int32_t act, float slope, const float* x, const float* w,
                            const float* bias, float* out, cgs_stream_t stream) {
    if (m < 0 || k <= 0 || n <= 0 || !x || !w || !out) return CGS_ERR_BADARG;
    if (m == 0) return CGS_OK;
    GenGemmParams P{x, w, bias, out, m, k, n, act, slope};
    hipLaunchKernelGGL(gen_gemm_kernel, dim3(((m + 15) / 16) * ((n + 15) / 16)), dim3(64), 0, (hipStream_t)stream, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

static bool convt_ok(int n, int h, int ca, int cb, int co) { return n >= 0 && h > 0 && h <= 64 && ca > 0 && cb >= 0 && co > 0; }

extern "C" int cgs_gen_convt4s2_fwd(int32_t n, int32_t h, int32_t ca, int32_t cb, int32_t co, int32_t act, float slope,
                                    const float* a, const float* b, const float* w, const float* bias, float* out,
                                    cgs_stream_t stream) {
    if (!convt_ok(n, h, ca, cb, co) || !a || (cb > 0 && !b) || !w || !bias || !out) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    GenConvTParams P{a, b, w, bias, nullptr, out, nullptr, nullptr, nullptr, nullptr, n, h, ca, cb, co, act, slope};
    const size_t total = (size_t)n * 4 * h * h * co;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(gen_convt_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_gen_convt4s2_bwd_data(int32_t n, int32_t h, int32_t ca, int32_t cb, int32_t co, const float* dy,
                                         const float* w, float* da, float* db, cgs_stream_t stream) {
    if (!convt_ok(n, h, ca, cb, co) || !dy || !w || (!da && !db)) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    GenConvTParams P{nullptr, nullptr, w, nullptr, dy, nullptr, da, db, nullptr, nullptr, n, h, ca, cb, co, 0, 0.f};
    const size_t total = (size_t)n * h * h * (ca + cb);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(gen_convt_bwd_data_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_gen_convt4s2_bwd_weight(int32_t n, int32_t h, int32_t ca, int32_t cb, int32_t co, const float* a,
                                           const float* b, const float* dy, float* dw, float* dbias, cgs_stream_t stream) {
    if (!convt_ok(n, h, ca, cb, co) || !a || (cb > 0 && !b) || !dy || !dw || !dbias) return CGS_ERR_BADARG;
    GenConvTParams P{a, b, nullptr, nullptr, dy, nullptr, nullptr, nullptr, dw, dbias, n, h, ca, cb, co, 0, 0.f};
    hipLaunchKernelGGL(gen_convt_bwd_weight_kernel, dim3(16 * (ca + cb) * co + co), dim3(256), 0, (hipStream_t)stream, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

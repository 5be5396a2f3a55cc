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

struct GenConvParams {
    GenSrc src; const float* w; const float* bias;      // bias may be NULL (data gradient)
    float* out; uint8_t* argmax;
    const float* addend; int n_addend;                  // pool = 0: out += addend for images < n_addend (same shape as out)
    int n, hw, co, act, pool, th;
    float slope;
};

constexpr int GEN_MAX_TPW = 4;        // pixel tiles (16 pixels) per wave: strips hold <= 256 pixels

// grid: ((image * strips + strip) * column-block groups + group); 256 threads.  A workgroup computes NCB (<= 3) blocks of 16 output
// channels from one staged input tile: an A operand read from LDS feeds NCB MFMAs, a weight operand the wave's pixel tiles.
template <int NCB, bool WLDS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) gen_conv3x3_fwd_kernel(GenConvParams P) {
    extern __shared__ __attribute__((aligned(16))) float4 gsm[];
    float* tile = (float*)gsm;                      // [(th + 2)][(hw + 2)][GEN_KC]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int H = P.hw, W = P.hw, TH = P.th, PW = W + 2;
    const int strips = H / TH, ncg = ((P.co + 15) / 16 + NCB - 1) / NCB;
    const int cg = blockIdx.x % ncg, strip = (blockIdx.x / ncg) % strips, img = blockIdx.x / (ncg * strips);
    const int row0 = strip * TH;
    const GenSrc& S = P.src;
    const int cp = gen_pa4(S) + S.cb, ci_total = S.ca + S.cb;
    const int nchunk = (cp + GEN_KC - 1) / GEN_KC;
    const int col0 = cg * NCB * 16 + l15;           // this lane's column in the group's first block (+16 per block)
    const int ntiles = TH * W / 16, QW = W / 2;

    frag4 acc[GEN_MAX_TPW][NCB];
    int abase[GEN_MAX_TPW];
#pragma unroll
    for (int i = 0; i < GEN_MAX_TPW; ++i) {
#pragma unroll
        for (int c = 0; c < NCB; ++c) acc[i][c] = frag4{0.f, 0.f, 0.f, 0.f};
        const int t = wave + 4 * i;
        const int q = 4 * t + (l15 >> 2), qy = q / QW, qx = q % QW;
        const int y = 2 * qy + ((l15 >> 1) & 1), x = 2 * qx + (l15 & 1);     // strip-local
        abase[i] = (y * PW + x) * GEN_KC + kq;
    }

    constexpr int NC = NCB == 2 ? 48 : 16 * NCB;                 // row stride of the weight tile (48: the 4 k-rows of a read hit 64 banks)
    float* wl = tile + (TH + 2) * PW * GEN_KC;                   // [9 taps][16 channels][NC]
    gen_zero_halo_cols(tile, W, TH, 1, tid);                      // (the staging writes the interior columns only)
    for (int ch = 0; ch < nchunk; ++ch) {
        int ltid = tid;                                           // opaque per chunk: keeps the staging addresses of all
        asm volatile("" : "+v"(ltid));                            // iterations from being hoisted out of this loop (registers)
        gen_stage<2>(tile, S, img, H, W, row0, TH, 1, ch, ltid);     // 16 channels of the strip (with halo)
        if constexpr (WLDS) {
        // the chunk's weights: [tap][channel][NCB x 16 columns], zero for padding channels / columns (9 loads in flight at a time)
#pragma unroll 1
        for (int bt = 0; bt < NCB; ++bt) {
            float wv[9];
#pragma unroll
            for (int it = 0; it < 9; ++it) {
                const int e = ltid + 256 * (9 * bt + it), c = e % (16 * NCB), k = (e / (16 * NCB)) & 15, tap = e / (256 * NCB);
                const int ci = gen_real_channel(S, ch * GEN_KC + k), col = cg * NCB * 16 + c;
                wv[it] = (ci >= 0 && col < P.co) ? P.w[((size_t)tap * ci_total + ci) * P.co + col] : 0.f;
            }
#pragma unroll
            for (int it = 0; it < 9; ++it) {
                const int e = ltid + 256 * (9 * bt + it), c = e % (16 * NCB), k = (e / (16 * NCB)) & 15, tap = e / (256 * NCB);
                wl[(tap * 16 + k) * NC + c] = wv[it];
            }
        }
        }
        __syncthreads();
        const int rem = cp - ch * GEN_KC, ksteps = rem >= GEN_KC ? 4 : (rem + 3) >> 2;      // (a partial last chunk: fewer k-steps)
        // ---- 9 taps x k-steps: operands from LDS only; a weight operand is shared by the wave's pixel tiles ----
        // (the weight operand of a k-step comes from L1/L2: it is requested ONE k-step ahead, so its round trip overlaps the
        //  previous step's MFMAs instead of standing in front of its own -- 36 serial round trips per chunk otherwise)
        auto load_b = [&](float (&bv)[NCB], int tap, int s) {
            if constexpr (WLDS) {
#pragma unroll
                for (int c = 0; c < NCB; ++c) bv[c] = wl[(tap * 16 + 4 * s + kq) * NC + 16 * c + l15];
            } else {
                const int ci = gen_real_channel(S, ch * GEN_KC + 4 * s + kq);
#pragma unroll
                for (int c = 0; c < NCB; ++c)
                    bv[c] = (ci >= 0 && col0 + 16 * c < P.co) ? P.w[((size_t)tap * ci_total + ci) * P.co + col0 + 16 * c] : 0.f;
            }
        };
        float bn[NCB];
        load_b(bn, 0, 0);
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = ((tap / 3) * PW + tap % 3) * GEN_KC;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < ksteps) {
                    float b[NCB];
#pragma unroll
                    for (int c = 0; c < NCB; ++c) b[c] = bn[c];
                    if (s + 1 < ksteps) load_b(bn, tap, s + 1);
                    else if (tap < 8) load_b(bn, tap + 1, 0);
#pragma unroll
                    for (int i = 0; i < GEN_MAX_TPW; ++i) {
                        if (wave + 4 * i < ntiles) {
                            const float a = tile[abase[i] + toff + 4 * s];
#pragma unroll
                            for (int c = 0; c < NCB; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[c], acc[i][c], 0, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // ---- epilogue ----
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        const int col = col0 + 16 * c;
        if (col >= P.co) continue;
        const float bias = P.bias ? P.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < GEN_MAX_TPW; ++i) {
            const int t = wave + 4 * i;
            if (t >= ntiles) continue;
            const int q = 4 * t + kq, qy = q / QW, qx = q % QW;
            if (P.pool) {
                float m = gen_act(acc[i][c][0] + bias, P.act, P.slope);
                int idx = 0;
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    const float v = gen_act(acc[i][c][j] + bias, P.act, P.slope);
                    if (v > m) { m = v; idx = j; }
                }
                const size_t pp = (((size_t)img * (H / 2) + row0 / 2 + qy) * (W / 2) + qx) * P.co + col;
                P.out[pp] = m;
                // (ReLU: a pooled value <= 0 passes no gradient -- marked in bit 2 for the backward loaders)
                if (P.argmax) P.argmax[pp] = (uint8_t)(idx | ((P.act == CGS_ACT_RELU && !(m > 0.f)) ? 4 : 0));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = row0 + 2 * qy + (j >> 1), x = 2 * qx + (j & 1);
                    const size_t o = (((size_t)img * H + y) * W + x) * P.co + col;
                    float v = gen_act(acc[i][c][j] + bias, P.act, P.slope);
                    if (P.addend && img < P.n_addend) v += P.addend[o];
                    P.out[o] = v;
                }
            }
        }
    }
}

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

static int gen_conv_launch(GenConvParams P, cgs_stream_t stream) {
    P.th = gen_strip_rows(P.hw);
    const int ncb = (P.co + 15) / 16, strips = P.hw / P.th;
    // column blocks per workgroup: 3 (2 when that covers the layer exactly); one staged tile then feeds all of them
    const int per = ncb == 1 ? 1 : ((ncb == 2 || ncb == 4) ? 2 : 3);
    constexpr bool wlds = false;      // weights through LDS: measured slower (occupancy 6 -> 4 workgroups per CU), DESIGN.md section 7
    const size_t lds = ((size_t)(P.th + 2) * (P.hw + 2) * GEN_KC + (wlds ? (size_t)9 * 16 * (per == 2 ? 48 : 16 * per) : 0)) * sizeof(float);
    const dim3 grid(P.n * strips * ((ncb + per - 1) / per));
    auto k = wlds ? (per == 1 ? gen_conv3x3_fwd_kernel<1, true> : per == 2 ? gen_conv3x3_fwd_kernel<2, true> : gen_conv3x3_fwd_kernel<3, true>)
                  : (per == 1 ? gen_conv3x3_fwd_kernel<1, false> : per == 2 ? gen_conv3x3_fwd_kernel<2, false> : gen_conv3x3_fwd_kernel<3, false>);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, (hipStream_t)stream, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_gen_conv3x3_fwd(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_is_u8, int32_t ups,
                                   int32_t act, float slope, int32_t pool, const void* src_a, const float* src_b, const float* w,
                                   const float* bias, float* out, uint8_t* argmax, cgs_stream_t stream) {
    if (n < 0 || !src_a || !w || !bias || !out || ca <= 0 || cb < 0 || co <= 0) return CGS_ERR_BADARG;
    if (cb > 0 && (!src_b || (cb & 3) || (ups != 1 && ups != 2 && ups != 4))) return CGS_ERR_BADARG;
    if (!gen_hw_ok(hw)) return CGS_ERR_UNSUPPORTED;
    if (act < CGS_ACT_NONE || act > CGS_ACT_SIGMOID) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    GenConvParams P{};
    P.src = GenSrc{src_a, src_b, nullptr, a_is_u8 ? GEN_SRC_U8 : GEN_SRC_F32, ca, cb, cb > 0 ? ups : 1};
    P.w = w; P.bias = bias; P.out = out; P.argmax = argmax;
    P.n = n; P.hw = hw; P.co = co; P.act = act; P.pool = pool; P.slope = slope;
    return gen_conv_launch(P, stream);
}

// Data gradient of a 3x3 layer = the same convolution over the output gradient with the flipped, transposed kernel
// (cgs_gen_flip_weights): d_cat [n,hw,hw,ci] = conv3x3(dY, wflip [9][co][ci]) (+ addend for images < n_addend).
extern "C" int cgs_gen_conv3x3_bwd_data(int32_t n, int32_t hw, int32_t co, int32_t ci, const float* dy, const uint8_t* dy_argmax,
                                        const float* wflip, const float* addend, int32_t n_addend, float* d_cat,
                                        cgs_stream_t stream) {
    if (n < 0 || !dy || !wflip || !d_cat || co <= 0 || ci <= 0 || n_addend < 0) return CGS_ERR_BADARG;
    if (dy_argmax && (co & 3)) return CGS_ERR_BADARG;
    if (!gen_hw_ok(hw)) return CGS_ERR_UNSUPPORTED;
    if (n == 0) return CGS_OK;
    GenConvParams P{};
    P.src = GenSrc{dy, nullptr, dy_argmax, dy_argmax ? GEN_SRC_POOLEXP : GEN_SRC_F32, co, 0, 1};
    P.w = wflip; P.out = d_cat; P.addend = addend; P.n_addend = n_addend;
    P.n = n; P.hw = hw; P.co = ci; P.act = CGS_ACT_NONE;
    return gen_conv_launch(P, stream);
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

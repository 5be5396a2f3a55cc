// Training pass of the shape-generic family (chfak != 1: the paper's model is chfak = 5, docs/index.html:151, nets.py:166,184,190):
// what the backward of NewCritic.forward / UnetDecoder.forward (nets.py:197-212, 494-523) needs besides the forward kernel of
// gen.hip -- the data gradient of a 3x3 layer IS that kernel over the output gradient with flipped weights (cgs_gen_conv3x3_bwd_data).
//
//   gen_conv3x3_wgrad : dW[tap][ci][co] = sum over images, pixels of in[p + tap][ci] dY[p][co] (+ dbias) as an MFMA GEMM with the
//                       pixels as the K dimension: persistent workgroups own a (16 input channels x 16 output channels) block for
//                       all nine taps and a share of the images; one slab row per image share, summed by cgs_reduce_slabs.
//   gen_flip_weights  : HWIO [tap][ci][co] -> [8 - tap][co][ci] (the data-gradient kernel's weight operand).
//   gen_cat_split     : gradient of cat(A, nearest-up(B)): channels [0,ca) -> d_a, channels [ca,ca+cb) summed over the cells -> d_b.
//   gen_grad_fix      : d = (d * dropout mask + addend) * act'(saved output): the element-wise steps between two layers.
//   gen_dropout_fwd   : out = x * keep-mask / (1 - p)  (nets.py:179,183,192).
//   gen_gemm_ex       : out[m][n] (+)= act(sum_k A(m,k) B(k,n) + bias[n]) with free strides (Linear layers, their data and weight
//                       gradients: X W^T, X^T dY, column sums through a ones vector).
//   gen_u8_to_f32     : frames / 255 (the weight gradient of features.0 reads A and the mixes from one fp32 buffer).
#include "gen_common.h"

namespace {

struct GenWgradParams {
    GenSrc in;          // the layer's input cat(A, up(B))
    GenSrc dy;          // gradient at the pre-activation output: GEN_SRC_F32 [n,hw,hw,co] or GEN_SRC_POOLEXP (dE + argmax); ca = co
    float* slab;        // [G][9 * ci_total * co + co]
    int n, hw, th, G, ncib, ncob;
};

// grid: (image share g, input-channel block, group of NCOB output-channel blocks); 256 threads = 4 waves that split the pixel
// groups.  One staged input chunk meets NCOB (<= 3) staged chunks of dY: an A operand read from LDS feeds NCOB MFMAs.
// SMALL (9 (ca + cb) <= 32, i.e. the image layer): the GEMM's rows are the (tap, channel) pairs -- 2 row blocks instead of 9 taps
// x one mostly empty block of 16 channels.
template <int NCOB, bool SMALL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) gen_conv3x3_wgrad_kernel(GenWgradParams P) {
    extern __shared__ __attribute__((aligned(16))) float4 gsm[];
    constexpr int NT = SMALL ? 2 : 9, NV = 4 * NT + 1;          // row blocks; floats per lane and column block in the reduction
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int H = P.hw, W = P.hw, TH = P.th, PW = W + 2;
    float* tin = (float*)gsm;                                  // [(TH + 2)][(W + 2)][16]
    float* tdy = tin + (TH + 2) * PW * GEN_KC;                 // [NCOB][TH][W][16]
    const int DYT = TH * W * GEN_KC;
    const int cog = blockIdx.x % P.ncob, cib = (blockIdx.x / P.ncob) % P.ncib, g = blockIdx.x / (P.ncob * P.ncib);      // (ncob = groups)
    const int co = P.dy.ca, ci_total = P.in.ca + P.in.cb;
    const int ngroups = TH * W / 4, strips = H / TH;
    const int nblk = (co + 15) / 16;
    int aoff[NT];                                              // this lane's A-operand offset inside the tile, per row block
    bool aok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if constexpr (SMALL) {
            const int m = 16 * t + l15, tap = m / ci_total;
            aok[t] = m < 9 * ci_total;
            aoff[t] = aok[t] ? ((tap / 3) * PW + tap % 3) * GEN_KC + m % ci_total : 0;
        } else {
            aok[t] = true;
            aoff[t] = ((t / 3) * PW + t % 3) * GEN_KC + l15;
        }
    }

    frag4 acc[NT][NCOB];
    float bsum[NCOB];
#pragma unroll
    for (int c = 0; c < NCOB; ++c) {
        bsum[c] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t][c] = frag4{0.f, 0.f, 0.f, 0.f};
    }

    gen_zero_halo_cols(tin, W, TH, 1, tid);                    // (the staging writes the interior columns only)
    for (int img = g; img < P.n; img += P.G) {
        for (int strip = 0; strip < strips; ++strip) {
            int ltid = tid;                                    // opaque per strip: the staging addresses of all iterations are
            asm volatile("" : "+v"(ltid));                     // recomputed here instead of living in registers across the loops
            gen_stage<7>(tin, P.in, img, H, W, strip * TH, TH, 1, cib, ltid);
#pragma unroll 1
            for (int c = 0; c < NCOB; ++c)
                if (cog * NCOB + c < nblk) gen_stage<7>(tdy + c * DYT, P.dy, img, H, W, strip * TH, TH, 0, cog * NCOB + c, ltid);
            __syncthreads();
            for (int grp = wave; grp < ngroups; grp += 4) {
                const int p0 = 4 * grp, y = p0 / W, x = p0 % W;          // 4 consecutive pixels of one row = the K slice
                float b[NCOB];
#pragma unroll
                for (int c = 0; c < NCOB; ++c) {
                    b[c] = (cog * NCOB + c < nblk) ? tdy[c * DYT + (p0 + kq) * GEN_KC + l15] : 0.f;
                    bsum[c] += b[c];
                }
                const float* ap = tin + ((size_t)(y * PW + x + kq)) * GEN_KC;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float a = ap[aoff[t]];
                    if constexpr (SMALL) a = aok[t] ? a : 0.f;
#pragma unroll
                    for (int c = 0; c < NCOB; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[c], acc[t][c], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }
    // ---- the four waves' partial blocks summed through LDS (fixed order), one column block at a time; then the slab row ----
    float* red = (float*)gsm;                                  // [3 waves][NV][64]
    float* row = P.slab + (size_t)g * (9 * ci_total * co + co);
#pragma unroll
    for (int c = 0; c < NCOB; ++c) {
        if (wave > 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) red[((wave - 1) * NV + 4 * t + j) * 64 + lane] = acc[t][c][j];
            red[((wave - 1) * NV + 4 * NT) * 64 + lane] = bsum[c];
        }
        __syncthreads();
        if (wave == 0) {
            float bs = bsum[c];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][c][j] += red[(w * NV + 4 * t + j) * 64 + lane];
                bs += red[(w * NV + 4 * NT) * 64 + lane];
            }
            const int col = (cog * NCOB + c) * 16 + l15;
            if (col < co) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (SMALL) {      // row m = tap * ci_total + ci: the slab's own order
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const int m = 16 * t + 4 * kq + j;
                            if (m < 9 * ci_total) row[(size_t)m * co + col] = acc[t][c][j];
                        }
                    } else {
                        const int ci = gen_real_channel(P.in, cib * GEN_KC + 4 * kq + j);
                        if (ci < 0) continue;
#pragma unroll
                        for (int t = 0; t < 9; ++t) row[((size_t)t * ci_total + ci) * co + col] = acc[t][c][j];
                    }
                }
            }
            bs += __shfl_xor(bs, 16, 64);
            bs += __shfl_xor(bs, 32, 64);
            if (cib == 0 && kq == 0 && col < co) row[(size_t)9 * ci_total * co + col] = bs;
        }
        __syncthreads();
    }
}

#include "gen_wgrad_rows.h"
#include "gen_wgrad_fold.h"

__global__ void __launch_bounds__(256) gen_flip_weights_kernel(const float* __restrict__ w, int ci, int co, float* __restrict__ out) {
    const int total = 9 * ci * co;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int i = e % ci, c = (e / ci) % co, tap = e / (ci * co);
        out[e] = w[((size_t)(8 - tap) * ci + i) * co + c];
    }
}

struct GenSplitParams {
    const float* dcat; float* d_a; float* d_b;
    int n, hw, ca, cb, ups;
};

__global__ void __launch_bounds__(256) gen_cat_split_kernel(GenSplitParams P) {
    const int ct = P.ca + P.cb, hb = P.hw / P.ups;
    const size_t na = P.d_a ? (size_t)P.n * P.hw * P.hw * P.ca : 0, nb = P.d_b ? (size_t)P.n * hb * hb * P.cb : 0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < na + nb; e += (size_t)gridDim.x * 256) {
        if (e < na) {
            const size_t pix = e / P.ca;
            P.d_a[e] = P.dcat[pix * ct + e % P.ca];
        } else {
            const size_t f = e - na;
            const int c = f % P.cb, xb = (f / P.cb) % hb, yb = (f / ((size_t)P.cb * hb)) % hb, img = f / ((size_t)P.cb * hb * hb);
            float s = 0.f;
            for (int dy = 0; dy < P.ups; ++dy)
                for (int dx = 0; dx < P.ups; ++dx)
                    s += P.dcat[(((size_t)img * P.hw + yb * P.ups + dy) * P.hw + xb * P.ups + dx) * ct + P.ca + c];
            P.d_b[f] = s;
        }
    }
}

struct GenFixParams {
    float* d; const float* saved; const float* addend;
    long count, addend_count;
    int act; float slope;
    cgs_dropout drop;
};

__device__ __forceinline__ float gen_act_grad(float out, int act, float slope) {
    if (act == CGS_ACT_RELU) return out > 0.f ? 1.f : 0.f;
    if (act == CGS_ACT_LRELU) return out > 0.f ? 1.f : slope;
    if (act == CGS_ACT_SIGMOID) return out * (1.f - out);
    return 1.f;
}

// one thread per 4 consecutive floats (the dropout stream's unit)
__global__ void __launch_bounds__(256) gen_grad_fix_kernel(GenFixParams P) {
    const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
    if (4 * i4 >= P.count) return;
    const DropCtx dc = drop_ctx(P.drop);
    const float4 m = dc.on ? drop_mult4(dc, (uint32_t)i4) : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long i = 4 * i4 + k;
        if (i >= P.count) break;
        float v = P.d[i] * f4get(m, k);
        if (P.addend && i < P.addend_count) v += P.addend[i];
        if (P.saved) v *= gen_act_grad(P.saved[i], P.act, P.slope);
        P.d[i] = v;
    }
}

__global__ void __launch_bounds__(256) gen_dropout_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ out, long count4,
                                                              cgs_dropout d) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count4) return;
    const DropCtx dc = drop_ctx(d);
    out[i] = dc.on ? x[i] * drop_mult4(dc, (uint32_t)i) : x[i];
}

__global__ void __launch_bounds__(256) gen_u8_to_f32_kernel(const uint8_t* __restrict__ x, float* __restrict__ out, long count) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) out[i] = x[i] * (1.f / 255.f);
}

struct GenGemmExParams {
    const float* x; const float* w; const float* bias; float* out;
    int m, k, n, act, accumulate;
    long sxm, sxk, swk, swn;
    float slope;
    int kper;           // > 0: split K -- blockIdx.y owns k in [y kper, (y + 1) kper) and writes its partial product to out + y m n
};

// one workgroup per 32 x 16 output tile (x K share): two row blocks share every B fragment; its 4 waves split K in groups of 16 (summed
// in wave order through LDS);  A(m,k) = x[m sxm + k sxk], B(k,n) = w[k swk + n swn].
// A lane (l15, kq) supplies k = group + 4 kq + i to matrix step i of a group (any bijection of k shared by A and B is a valid order of
// the sum), so an operand that is contiguous along k (AV / BV: stride 1, rows 16-byte aligned, K share a multiple of 4) is ONE 16-byte
// load per group instead of four dwords 4 * stride apart.  Four groups' loads (up to 12 x 16 bytes per lane) are issued before their
// 32 matrix instructions: the kernel's life is its dependent memory round trips (round 4: 16 x 16 tiles, dword loads: 35 us for the
// 1024 x 1280 x 160 product of features.14 at chfak 5).
template <bool V>
__device__ __forceinline__ float4 gemm_ld4(const float* base, long stride, int k4, int kb, int ke) {      // raw: k past the share reads k = kb
    if constexpr (V) return *(const float4*)(base + (k4 < ke ? k4 : kb));
    else {
        float t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = base[(size_t)(k4 + i < ke ? k4 + i : kb) * stride];
        return make_float4(t[0], t[1], t[2], t[3]);
    }
}
__device__ __forceinline__ float4 gemm_mask4(float4 v, int k4, int ke) {
    return make_float4(k4 < ke ? v.x : 0.f, k4 + 1 < ke ? v.y : 0.f, k4 + 2 < ke ? v.z : 0.f, k4 + 3 < ke ? v.w : 0.f);
}

template <bool AV, bool BV>
__global__ void __launch_bounds__(256) gen_gemm_ex_kernel(GenGemmExParams P) {
    __shared__ float red[3][2][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, kq = lane >> 4;
    const int ntn = (P.n + 15) / 16;
    const int m0 = (blockIdx.x / ntn) * 32, n0 = (blockIdx.x % ntn) * 16;
    const int row0 = m0 + l15, row1 = row0 + 16, col = n0 + l15;
    const float* xr0 = P.x + (size_t)(row0 < P.m ? row0 : P.m - 1) * P.sxm;
    const float* xr1 = P.x + (size_t)(row1 < P.m ? row1 : P.m - 1) * P.sxm;
    const float* wc = P.w + (size_t)(col < P.n ? col : P.n - 1) * P.swn;
    frag4 acc0 = frag4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const int kb = P.kper > 0 ? (int)blockIdx.y * P.kper : 0, ke = P.kper > 0 ? min(P.k, kb + P.kper) : P.k;
    constexpr int U = 4;
    for (int kg = kb + 16 * wave; kg < ke; kg += 64 * U) {
        // (no branch and no use of a loaded value between the loads: either one makes the compiler wait for the loads issued so far)
        float4 a0[U], a1[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k4 = kg + 64 * u + 4 * kq;
            a0[u] = gemm_ld4<AV>(xr0, P.sxk, k4, kb, ke);
            a1[u] = gemm_ld4<AV>(xr1, P.sxk, k4, kb, ke);
            b[u] = gemm_ld4<BV>(wc, P.swk, k4, kb, ke);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (kg + 64 * u < ke) {                // (wave-uniform)
                const int k4 = kg + 64 * u + 4 * kq;
                const float4 x0 = gemm_mask4(a0[u], k4, ke), x1 = gemm_mask4(a1[u], k4, ke), y = gemm_mask4(b[u], k4, ke);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(x0, i), f4get(y, i), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(x1, i), f4get(y, i), acc1, 0, 0, 0);
                }
            }
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[wave - 1][0][j][lane] = acc0[j]; red[wave - 1][1][j][lane] = acc1[j]; }
    }
    __syncthreads();
    if (wave == 0 && col < P.n) {
        const float bias = P.bias ? P.bias[col] : 0.f;
        float* const obase = P.out + (size_t)blockIdx.y * (P.kper > 0 ? (size_t)P.m * P.n : 0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = m0 + 16 * h + 4 * kq + j;
                if (r < P.m) {
                    float* o = obase + (size_t)r * P.n + col;
                    const float own = h ? acc1[j] : acc0[j];
                    const float sum = ((own + red[0][h][j][lane]) + red[1][h][j][lane]) + red[2][h][j][lane];
                    const float v = gen_act(sum + bias, P.act, P.slope);
                    *o = P.accumulate ? *o + v : v;
                }
            }
        }
    }
}

// an operand may be read 16 bytes along k: unit stride, 16-byte aligned rows, K shares that are multiples of 4
static bool gemm_vec_ok(const float* p, long sk, long srow, int k, int kper) {
    return sk == 1 && (srow & 3) == 0 && ((uintptr_t)p & 15) == 0 && (k & 3) == 0 && (kper & 3) == 0;
}
static void gemm_ex_launch(const GenGemmExParams& P, int nsplit, hipStream_t st) {
    const dim3 grid(((P.m + 31) / 32) * ((P.n + 15) / 16), nsplit);
    const bool av = gemm_vec_ok(P.x, P.sxk, P.sxm, P.k, P.kper), bv = gemm_vec_ok(P.w, P.swk, P.swn, P.k, P.kper);
    if (av && bv) hipLaunchKernelGGL((gen_gemm_ex_kernel<true, true>), grid, dim3(256), 0, st, P);
    else if (av) hipLaunchKernelGGL((gen_gemm_ex_kernel<true, false>), grid, dim3(256), 0, st, P);
    else if (bv) hipLaunchKernelGGL((gen_gemm_ex_kernel<false, true>), grid, dim3(256), 0, st, P);
    else hipLaunchKernelGGL((gen_gemm_ex_kernel<false, false>), grid, dim3(256), 0, st, P);
}

int ew_blocks(size_t items) { size_t b = (items + 255) / 256; return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }

}  // namespace

static int wgrad_per(int co) {      // blocks of 16 output channels per workgroup
    const int nblk = (co + 15) / 16;
    return nblk == 1 ? 1 : ((nblk == 2 || nblk == 4) ? 2 : 3);
}
static int wgrad_groups(int n, int ncib, int ncog) {      // image shares: about 1536 workgroups (three rounds of two per CU; swept on the chfak-5 step, tools/sweep_genwgrad.sh: 512: 9.33, 768: 8.74, 1024: 8.82, 1536: 8.51, 2048: 8.66, 3072: 8.59 ms), at most one per image
    int g = (1536 + ncib * ncog - 1) / (ncib * ncog);
    return g < n ? g : n;
}

extern "C" int cgs_gen_conv3x3_bwd_weight_slabs(int32_t n, int32_t ca, int32_t cb, int32_t co) {
    if (n < 0 || ca <= 0 || cb < 0 || co <= 0 || (cb & 3)) return CGS_ERR_BADARG;
    if (n == 0) return 0;
    const GenWrPlan wr = gen_wr_plan(n, ca, cb, co);
    if (wr.ok) return wr.G;
    const int cp = ((ca + 3) & ~3) + cb, per = wgrad_per(co);
    return wgrad_groups(n, (cp + 15) / 16, ((co + 15) / 16 + per - 1) / per);
}

// the row-block form (gen_wgrad_rows.h): chunk geometry from the map size, LDS for two buffers; false: does not fit
template <int NCOB, int RBW, bool POOLED>
static int launch_wgrad_rows(const GenWrParams& P, size_t lds, int grid, hipStream_t st) {
    static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gen_wgrad_rows_kernel<NCOB, RBW, POOLED>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL((gen_wgrad_rows_kernel<NCOB, RBW, POOLED>), dim3(grid), dim3(512), lds, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}
template <int NCOB, bool POOLED>
static int launch_wgrad_rows_r(int rbw, const GenWrParams& P, size_t lds, int grid, hipStream_t st) {
    if (rbw == 2) return launch_wgrad_rows<NCOB, 2, POOLED>(P, lds, grid, st);
    if (rbw == 3) return launch_wgrad_rows<NCOB, 3, POOLED>(P, lds, grid, st);
    if (rbw == 6) return launch_wgrad_rows<NCOB, 6, POOLED>(P, lds, grid, st);
    return launch_wgrad_rows<NCOB, 7, POOLED>(P, lds, grid, st);
}
static int wgrad_rows(const GenWrPlan& wr, int n, int hw, int ca, int cb, int co, int a_is_u8, int ups, const void* a, const float* b,
                      const float* dy, const uint8_t* am, float* slab, hipStream_t st, int ci_stride = 0) {
    GenWrParams P{};
    P.ci_stride = ci_stride;
    P.a = a; P.b = b; P.dy = dy; P.am = am; P.slab = slab;
    P.n = n; P.hw = hw; P.lw = __builtin_ctz(hw); P.ca = ca; P.cb = cb; P.ush = ups == 4 ? 2 : (ups == 2 ? 1 : 0); P.co = co;
    P.a_u8 = a_is_u8; P.nrg = wr.nrg;
    P.G = wr.G; P.nsl = wr.nsl; P.ncs = wr.ncs; P.cs = wr.cs; P.cw = wr.cw;
    // chunk: the most pixels (512 ... 8 per pixel phase; th rows of one image, whole images below 16 x 16) whose two buffers fit
    // the LDS and whose quads fit the staging registers (7 or 9 + 3 per thread)
    const int nph = 8 / wr.nrg;
    int th = 0, imgs = 0;
    size_t lds = 0;
    for (int px = 512; px >= 8 * nph; px >>= 1) {
        if (hw * hw >= px) { imgs = 1; th = px / hw; if (th < 2) { th = 0; continue; } }
        else { imgs = px / (hw * hw); th = hw; }
        const size_t inf = ((size_t)imgs * (th + 2) * (hw + 2) * wr.cs + 3) & ~(size_t)3;
        const size_t dyf = (size_t)imgs * (am ? (th / 2) * (hw / 2) : th * hw) * wr.cw;
        const size_t buf = (inf + dyf + (am ? dyf / 4 : 0) + 64 + 3) & ~(size_t)3;
        lds = 2 * buf * sizeof(float);
        P.buf_floats = (int)buf;
        // staging registers: a thread holds KI pixels of one main quad column, KO odd quads (a uint8 / odd-width A), KD dY pixels
        const bool odd_a = (ca & 3) || a_is_u8;
        const bool narrow = cb == 0 && ca < 4;
        const int q4 = narrow ? 1 : wr.cs / 4, qo = odd_a ? (((ca + 3) / 4) < q4 ? (ca + 3) / 4 : q4) : 0, qm = q4 - qo, qd = wr.cw / 4;
        const int npx = imgs * (th + 2) * hw, npd = imgs * (am ? (th / 2) * (hw / 2) : th * hw);
        const int ki = wr.ncob == 1 ? 9 : 7;
        const bool fits = (qm == 0 || npx <= ki * (512 / qm)) && npx * qo <= 2 * 512 && npd <= 3 * (512 / qd);
        if (lds <= 158 * 1024 && fits) break;
        th = 0;
    }
    if (!th) return CGS_ERR_UNSUPPORTED;
    const size_t red = (size_t)7 * (4 * wr.rbw + 1) * 64 * sizeof(float);      // the final sum over the pixel phases, per column block
    if (lds < red) lds = red;
    P.imgs = imgs; P.th = th; P.parts = hw / th;
    P.units = ((n + imgs - 1) / imgs) * P.parts;
    const int grid = P.G * P.nsl * P.ncs;
    if (am) return wr.ncob == 1 ? launch_wgrad_rows_r<1, true>(wr.rbw, P, lds, grid, st) : launch_wgrad_rows_r<3, true>(wr.rbw, P, lds, grid, st);
    return wr.ncob == 1 ? launch_wgrad_rows_r<1, false>(wr.rbw, P, lds, grid, st) : launch_wgrad_rows_r<3, false>(wr.rbw, P, lds, grid, st);
}

extern "C" int cgs_gen_conv3x3_bwd_weight(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_is_u8, int32_t ups,
                                          const void* src_a, const float* src_b, const float* dy, const uint8_t* dy_argmax,
                                          float* slab, cgs_stream_t stream) {
    if (n < 0 || !src_a || !dy || !slab || ca <= 0 || cb < 0 || co <= 0) return CGS_ERR_BADARG;
    if (cb > 0 && (!src_b || (cb & 3) || (ups != 1 && ups != 2 && ups != 4))) return CGS_ERR_BADARG;
    if (dy_argmax && (co & 3)) return CGS_ERR_BADARG;
    if (!gen_hw_ok(hw)) return CGS_ERR_UNSUPPORTED;
    if (n == 0) return CGS_OK;
    const GenWrPlan wr = gen_wr_plan(n, ca, cb, co);
    if (wr.ok) {
        const int rc = wgrad_rows(wr, n, hw, ca, cb, co, a_is_u8, cb > 0 ? ups : 1, src_a, src_b, dy, dy_argmax, slab, (hipStream_t)stream);
        if (rc != CGS_ERR_UNSUPPORTED) return rc;
    }
    GenWgradParams P{};
    P.in = GenSrc{src_a, src_b, nullptr, a_is_u8 ? GEN_SRC_U8 : GEN_SRC_F32, ca, cb, cb > 0 ? ups : 1};
    P.dy = GenSrc{dy, nullptr, dy_argmax, dy_argmax ? GEN_SRC_POOLEXP : GEN_SRC_F32, co, 0, 1};
    P.slab = slab; P.n = n; P.hw = hw; P.th = gen_strip_rows(hw);
    const int cp = ((ca + 3) & ~3) + cb, per = wgrad_per(co);
    P.ncib = (cp + 15) / 16; P.ncob = ((co + 15) / 16 + per - 1) / per;
    P.G = wr.ok ? wr.G : wgrad_groups(n, P.ncib, P.ncob);       // (the slab count is a function of the channel counts alone)
    size_t lds = ((size_t)(P.th + 2) * (hw + 2) + (size_t)per * P.th * hw) * GEN_KC * sizeof(float);
    const size_t red = (size_t)3 * 37 * 64 * sizeof(float);
    if (lds < red) lds = red;
    const dim3 grid(P.G * P.ncib * P.ncob);
    const bool small = cb == 0 && 9 * ca <= 32;       // the image layer: (tap, channel) pairs as the GEMM's rows
    auto k = small ? (per == 1 ? gen_conv3x3_wgrad_kernel<1, true> : per == 2 ? gen_conv3x3_wgrad_kernel<2, true> : gen_conv3x3_wgrad_kernel<3, true>)
                   : (per == 1 ? gen_conv3x3_wgrad_kernel<1, false> : per == 2 ? gen_conv3x3_wgrad_kernel<2, false> : gen_conv3x3_wgrad_kernel<3, false>);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, (hipStream_t)stream, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// ---- a layer over cat(A, nearest-up_2(B)): A's rows + the bias row by the row-block kernel (slab strides of the whole layer), B's rows
//      by gen_wgrad_fold_kernel on B at its own resolution; both write the same G slab rows ----
static bool wgrad_fold_ok(int hw, int ca, int cb, int co) {
    return (hw == 16 || hw == 32 || hw == 64) && ca > 0 && (cb == 16 || cb == 24 || cb == 32 || cb == 40) && co > 0 && !(co & 3);
}
extern "C" int cgs_gen_conv3x3_bwd_weight_folded_slabs(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co) {
    if (n < 0 || !wgrad_fold_ok(hw, ca, cb, co)) return CGS_ERR_UNSUPPORTED;
    if (n == 0) return 0;
    const GenWrPlan wr = gen_wr_plan(n, ca, 0, co);
    return wr.ok ? wr.G : CGS_ERR_UNSUPPORTED;
}
template <int RB, int NCOB, bool NARROW>
static int launch_wgrad_fold1(const GenWfParams& P, size_t lds, int grid, hipStream_t st) {
    static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gen_wgrad_fold_kernel<RB, NCOB, NARROW>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL((gen_wgrad_fold_kernel<RB, NCOB, NARROW>), dim3(grid), dim3(512), lds, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}
template <int RB>
static int launch_wgrad_fold(int ncob, bool narrow, const GenWfParams& P, size_t lds, int grid, hipStream_t st) {
    if (narrow) return ncob == 1 ? launch_wgrad_fold1<RB, 1, true>(P, lds, grid, st) : launch_wgrad_fold1<RB, 3, true>(P, lds, grid, st);
    return ncob == 1 ? launch_wgrad_fold1<RB, 1, false>(P, lds, grid, st) : launch_wgrad_fold1<RB, 3, false>(P, lds, grid, st);
}
extern "C" int cgs_gen_conv3x3_bwd_weight_folded(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_is_u8,
                                                 const void* src_a, const float* src_b, const float* dy, float* slab, cgs_stream_t stream) {
    if (n < 0 || !src_a || !src_b || !dy || !slab) return CGS_ERR_BADARG;
    if (!wgrad_fold_ok(hw, ca, cb, co)) return CGS_ERR_UNSUPPORTED;
    if (n == 0) return CGS_OK;
    const GenWrPlan wr = gen_wr_plan(n, ca, 0, co);
    if (!wr.ok) return CGS_ERR_UNSUPPORTED;
    GenWfParams P{};
    P.b = src_b; P.dy = dy; P.slab = slab; P.n = n; P.hw = hw; P.lw = __builtin_ctz(hw); P.ca = ca; P.cb = cb; P.co = co;
    P.G = wr.G; P.ncs = wr.ncs; P.cw = wr.cw;
    const int h = hw / 2, ncob = wr.ncob, cwp = 16 * ncob;
    const bool narrow = ca <= 3;                                    // the frames: their rows and the bias row ride in the folded kernel
    P.a = src_a; P.a_u8 = a_is_u8;
    P.ds = wr.cw;
    // the final [class][fold, ci][column] block (+ narrow: [class][32][column] and the bias partial sums)
    const size_t fin = ((size_t)16 * cb * cwp + (narrow ? (size_t)(4 * 32 + 32) * cwp : 0)) * sizeof(float);
    size_t lds = 0;
    P.th = 0;
    // pixel stride of the B tile: 16 (mod 32) floats -> conflict-free A-operand reads; the channel count itself when only that lets a larger chunk fit
    const int ps_pad = (cb & 31) == 16 ? cb : ((cb + 15) / 32) * 32 + 16;
    for (int th = hw < 32 ? hw : 32; th >= 2 && !P.th; th >>= 1) {
        for (int pass = 0; pass < 2 && !P.th; ++pass) {
            const int ps = pass == 0 ? ps_pad : cb;
            const int thb = th / 2 + 2, steps = (th / 2) * (h / 4);
            const size_t buf = ((size_t)thb * (h + 2) * ps + (size_t)th * hw * P.ds + 64 + (narrow ? (size_t)(th + 2) * (hw + 2) * 4 + 16 : 0) + 3) & ~(size_t)3;      // (+ 16: the staging's dummy slot behind the frame tile)
            const bool fits = thb * h * (cb / 4) <= 4 * 512 && th * hw * (wr.cw / 4) <= 5 * 512 && !(steps & 3) && (!narrow || (th + 2) * hw <= 2 * 512);
            if (fits && 2 * buf * sizeof(float) <= 158 * 1024) { P.th = th; P.ps = ps; P.buf_floats = (int)buf; lds = 2 * buf * sizeof(float); }
        }
    }
    if (!P.th || fin > 158 * 1024) return CGS_ERR_UNSUPPORTED;
    if (lds < fin) lds = fin;
    P.parts = hw / P.th; P.units = n * P.parts;
    if (!narrow) {
        // A's rows and the bias row (ci_stride: the slab row is the whole layer's)
        const int rc = wgrad_rows(wr, n, hw, ca, 0, co, a_is_u8, 1, src_a, nullptr, dy, nullptr, slab, (hipStream_t)stream, ca + cb);
        if (rc != CGS_OK) return rc;
    }
    const int grid = P.G * P.ncs;
    switch (cb) {
        case 16: return launch_wgrad_fold<4>(ncob, narrow, P, lds, grid, (hipStream_t)stream);
        case 24: return launch_wgrad_fold<6>(ncob, narrow, P, lds, grid, (hipStream_t)stream);
        case 32: return launch_wgrad_fold<8>(ncob, narrow, P, lds, grid, (hipStream_t)stream);
        default: return launch_wgrad_fold<10>(ncob, narrow, P, lds, grid, (hipStream_t)stream);
    }
}

extern "C" int cgs_gen_flip_weights(int32_t ci, int32_t co, const float* w, float* wflip, cgs_stream_t stream) {
    if (ci <= 0 || co <= 0 || !w || !wflip) return CGS_ERR_BADARG;
    hipLaunchKernelGGL(gen_flip_weights_kernel, dim3(ew_blocks((size_t)9 * ci * co)), dim3(256), 0, (hipStream_t)stream, w, ci, co, wflip);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_gen_cat_split(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t ups, const float* d_cat, float* d_a,
                                 float* d_b, cgs_stream_t stream) {
    if (n < 0 || hw <= 0 || ca < 0 || cb < 0 || !d_cat || (ups != 1 && ups != 2 && ups != 4) || hw % ups) return CGS_ERR_BADARG;
    if ((d_a && ca == 0) || (d_b && cb == 0) || (!d_a && !d_b)) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    GenSplitParams P{d_cat, d_a, d_b, n, hw, ca, cb, ups};
    const size_t items = (d_a ? (size_t)n * hw * hw * ca : 0) + (d_b ? (size_t)n * (hw / ups) * (hw / ups) * cb : 0);
    hipLaunchKernelGGL(gen_cat_split_kernel, dim3(ew_blocks(items)), dim3(256), 0, (hipStream_t)stream, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_gen_grad_fix(int64_t count, float* d, const float* saved, int32_t act, float slope, const float* addend,
                                int64_t addend_count, cgs_dropout drop, cgs_stream_t stream) {
    if (count < 0 || !d || addend_count < 0 || act < CGS_ACT_NONE || act > CGS_ACT_SIGMOID) return CGS_ERR_BADARG;
    if (count == 0) return CGS_OK;
    GenFixParams P{d, saved, addend, (long)count, (long)addend_count, act, slope, drop};
    const long c4 = (count + 3) / 4;
    hipLaunchKernelGGL(gen_grad_fix_kernel, dim3((unsigned)((c4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_gen_dropout_fwd(int64_t count, const float* x, float* out, cgs_dropout drop, cgs_stream_t stream) {
    if (count < 0 || (count & 3) || !x || !out) return CGS_ERR_BADARG;
    if (count == 0) return CGS_OK;
    const long c4 = count / 4;
    hipLaunchKernelGGL(gen_dropout_fwd_kernel, dim3((unsigned)((c4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x, (float4*)out, c4, drop);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_gen_u8_to_f32(int64_t count, const uint8_t* x, float* out, cgs_stream_t stream) {
    if (count < 0 || !x || !out) return CGS_ERR_BADARG;
    if (count == 0) return CGS_OK;
    hipLaunchKernelGGL(gen_u8_to_f32_kernel, dim3(ew_blocks((size_t)count)), dim3(256), 0, (hipStream_t)stream, x, out, (long)count);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_gen_gemm_ex(int32_t m, int32_t k, int32_t n, const float* x, int64_t sxm, int64_t sxk, const float* w,
                               int64_t swk, int64_t swn, const float* bias, int32_t act, float slope, int32_t accumulate, float* out,
                               cgs_stream_t stream) {
    if (m < 0 || k <= 0 || n <= 0 || !x || !w || !out || act < CGS_ACT_NONE || act > CGS_ACT_SIGMOID) return CGS_ERR_BADARG;
    if (m == 0) return CGS_OK;
    GenGemmExParams P{x, w, bias, out, m, k, n, act, accumulate, (long)sxm, (long)sxk, (long)swk, (long)swn, slope, 0};
    gemm_ex_launch(P, 1, (hipStream_t)stream);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_gen_gemm_ex_splitk(int32_t m, int32_t k, int32_t n, const float* x, int64_t sxm, int64_t sxk, const float* w,
                                      int64_t swk, int64_t swn, int32_t nsplit, float* slab, cgs_stream_t stream) {
    if (m < 0 || k <= 0 || n <= 0 || !x || !w || !slab || nsplit < 1 || nsplit > 65535) return CGS_ERR_BADARG;
    if (m == 0) return CGS_OK;
    GenGemmExParams P{x, w, nullptr, slab, m, k, n, CGS_ACT_NONE, 0, (long)sxm, (long)sxk, (long)swk, (long)swn, 0.f, (k + nsplit - 1) / nsplit};
    gemm_ex_launch(P, nsplit, (hipStream_t)stream);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

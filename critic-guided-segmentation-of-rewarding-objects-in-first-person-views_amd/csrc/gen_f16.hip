// fp16 inference family (BASELINE.json config 4: "-process inference-only mask path ... fp16 conv kernels"): every layer of
// NewCritic.forward / UnetDecoder.forward (nets.py:197-212, 494-523; main.py:1130-1151) with fp16 activations in HBM and LDS, fp16
// weights, fp32 accumulation on v_mfma_f32_16x16x32_f16 (round 4: gfx950's K = 32 form -- one instruction per PAIR of taps and 16-channel
// chunk, five per chunk instead of nine), for any
// channel factor (runtime channel counts, like gen.hip).  An opt-in precision mode: the fp32 path stays the default everywhere.
//
//   gen16_pack_weights : HWIO fp32 [9][ci][co] -> fp16 [tap pair][chunk][kq][padded co][8]: the B operand of a lane (8 consecutive
//                        input channels of one output channel at one of the pair's taps) is one 16-byte load, a wave's loads are contiguous.
//   gen16_conv3x3      : conv3x3(cat(A, nearest-up(B))) + bias + act (+ MaxPool2d(2)); A uint8 (frames, /255 fused) or fp16 NHWC,
//                        B fp16 NHWC; output fp16 NHWC, or fp32 for the mask layer.  LDS tile [rows][cols][4 quads of 4 halves]
//                        with the quad index XOR-swizzled by the row parity: the 16-byte A-operand reads (two adjacent quads) of a
//                        wave's two pixel rows land in different bank halves.
//   gen16_gemm         : Linear layers / the 4x4 valid convolution / the 1x1 convolution on fp16 or fp32 rows, fp32 weights.
#include "gen_common.h"

namespace {

// 16-bit element types: IEEE half (config 4) and bfloat16 (config 5: the build-defined 128x128 variant, hourglass128.py).  Both are
// stored as raw 16-bit words; E16<BF> converts and picks the MFMA (v_mfma_f32_16x16x32_f16 / _bf16 for the convolutions: K = 32 = two taps
// of a 16-channel chunk; the K = 16 forms remain for other users of E16).
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half_t;
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8v_t __attribute__((ext_vector_type(8)));
typedef short short8v_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8v_t __attribute__((ext_vector_type(8)));
template <bool BF> struct E16;
template <> struct E16<false> {
    using T = half_t; using V4 = half4_t;
    __device__ static __forceinline__ T cvt(float f) { return (half_t)f; }
    __device__ static __forceinline__ float up(T v) { return (float)v; }
    __device__ static __forceinline__ V4 zero() { return V4{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f}; }
    __device__ static __forceinline__ frag4 mfma(V4 a, V4 b, frag4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
    using V8 = half8v_t;           // gfx950's K = 32 form: 8 elements per lane
    __device__ static __forceinline__ frag4 mfma32(V8 a, V8 b, frag4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct E16<true> {
    using T = __bf16; using V4 = short4_t;
    __device__ static __forceinline__ T cvt(float f) { return (__bf16)f; }          // v_cvt_pk_bf16_f32: round to nearest even
    __device__ static __forceinline__ float up(T v) { return (float)v; }
    __device__ static __forceinline__ V4 zero() { return V4{0, 0, 0, 0}; }
    __device__ static __forceinline__ frag4 mfma(V4 a, V4 b, frag4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
    using V8 = short8v_t;
    __device__ static __forceinline__ frag4 mfma32(V8 a, V8 b, frag4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8v_t, a), __builtin_bit_cast(bf16x8v_t, b), c, 0, 0, 0);
    }
};

struct Gen16ConvParams {
    const void* a; const void* b; const void* w16; const float* bias;
    void* out;                 // fp16 NHWC (out_f32 = 0) or fp32
    int a_u8, ca, cb, ups, n, hw, co, act, pool, th, out_f32;
    float slope;
    unsigned long long* dbg;   // debug hook (dbg_gen16_stamps): s_memtime at the stage boundaries of the first 4096 workgroups
    uint8_t* codes;            // training (pool = 1): argmax position 0..3 of every pooled element, 4 = pooled value <= 0 (no gradient)
    int a_f32;                 // source A is fp32 NHWC with ca channels (the replaced / injected mixes of the config-5 training step)
};

unsigned long long* g_gen16_stamps = nullptr;
#define G16_STAMP(k) do { if (CGS_STAMP_PTR(P.dbg) && tid == 0 && blockIdx.x < 4096) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)

__device__ __forceinline__ int g16_pa4(const Gen16ConvParams& P) { return (P.ca + 3) & ~3; }

// Operand layout of the K = 32 instruction (round 4: v_mfma_f32_16x16x32_{f16,bf16}; the K = 16 form spent nine instructions and nine
// 8-byte LDS reads per 16 pixels and 16-channel chunk, this one five and five 16-byte reads): [tap pair tp][chunk][kq][padded co][8],
// lane group kq = tap 2 tp + (kq >> 1), channels 8 (kq & 1) .. + 7 of the 16-channel chunk (the tenth tap is zero).
template <bool BF>
__global__ void __launch_bounds__(256) gen16_pack_weights_kernel(const float* __restrict__ w, int ca, int cb, int co, typename E16<BF>::T* __restrict__ out) {
    const int pa4 = (ca + 3) & ~3, cp = pa4 + cb, nchunk = (cp + 15) / 16, ncol = (co + 15) / 16 * 16, ci_total = ca + cb;
    const int total = 5 * nchunk * 4 * ncol * 8;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int j = e & 7, col = (e >> 3) % ncol, kq = ((e >> 3) / ncol) & 3, ch = ((e >> 3) / (ncol * 4)) % nchunk, tp = (e >> 3) / (ncol * 4 * nchunk);
        const int tap = 2 * tp + (kq >> 1), k = ch * 16 + 8 * (kq & 1) + j;
        const int ci = k < pa4 ? (k < ca ? k : -1) : (k < cp ? ca + (k - pa4) : -1);
        out[e] = E16<BF>::cvt((tap < 9 && ci >= 0 && col < co) ? w[((size_t)tap * ci_total + ci) * co + col] : 0.f);
    }
}

// The data gradient's operand: conv3x3(dY [co_layer channels]) with w'[tap][k = layer output channel][col = layer input channel] =
// w[8 - tap][col][k] (flipped taps, transposed channels) -- the forward kernel then computes d cat(A, up(B)) at full resolution.
template <bool BF>
__global__ void __launch_bounds__(256) gen16_pack_weights_T_kernel(const float* __restrict__ w, int ci_layer, int co_layer, typename E16<BF>::T* __restrict__ out) {
    const int cp = (co_layer + 3) & ~3, nchunk = (cp + 15) / 16, ncol = (ci_layer + 15) / 16 * 16;
    const int total = 5 * nchunk * 4 * ncol * 8;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int j = e & 7, col = (e >> 3) % ncol, kq = ((e >> 3) / ncol) & 3, ch = ((e >> 3) / (ncol * 4)) % nchunk, tp = (e >> 3) / (ncol * 4 * nchunk);
        const int tap = 2 * tp + (kq >> 1), k = ch * 16 + 8 * (kq & 1) + j;
        out[e] = E16<BF>::cvt((tap < 9 && k < co_layer && col < ci_layer) ? w[((size_t)(8 - tap) * ci_layer + col) * co_layer + k] : 0.f);
    }
}

constexpr int G16_MAX_TPW = 4;

// grid: ((image * strips + strip) * column-block groups + group); 256 threads; NCB blocks of 16 output channels per workgroup
template <int NCB, bool A_U8, bool HASB, bool BF>
__global__ void __launch_bounds__(256) gen16_conv3x3_kernel(Gen16ConvParams P) {
    using EL = E16<BF>;
    using half_t = typename EL::T;          // (the names below say "half": either 16-bit type)
    using half4_t = typename EL::V4;
    extern __shared__ __attribute__((aligned(16))) float4 gsm[];
    half_t* tile = (half_t*)gsm;                    // [(th + 2)][(hw + 2)][4 quads (swizzled)][4 halves]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int H = P.hw, W = P.hw, TH = P.th, PW = W + 2;
    const int ncolb = (P.co + 15) / 16, ncg = (ncolb + NCB - 1) / NCB, ncol = ncolb * 16, strips = H / TH;
    const int cg = blockIdx.x % ncg, strip = (blockIdx.x / ncg) % strips, img = blockIdx.x / (ncg * strips);
    const int row0 = strip * TH;
    const int pa4 = g16_pa4(P), cp = pa4 + P.cb, nchunk = (cp + 15) / 16;
    const int ntiles = TH * W / 16, QW = W / 2;
    const int ush = P.ups == 4 ? 2 : (P.ups == 2 ? 1 : 0), HB = H >> ush, WB = W >> ush;
    const int lw = __builtin_ctz(W) + 2;                 // log2 of the interior elements per tile row (W pixels x 4 quads)

    G16_STAMP(0);
    frag4 acc[G16_MAX_TPW][NCB];
    int apix[G16_MAX_TPW], apar[G16_MAX_TPW];       // the lane's pixel (tile coordinates, before the tap offset) and its row parity
#pragma unroll
    for (int i = 0; i < G16_MAX_TPW; ++i) {
#pragma unroll
        for (int c = 0; c < NCB; ++c) acc[i][c] = frag4{0.f, 0.f, 0.f, 0.f};
        const int t = wave + 4 * i;
        const int q = 4 * t + (l15 >> 2), qy = q / QW, qx = q % QW;
        const int y = 2 * qy + ((l15 >> 1) & 1), x = 2 * qx + (l15 & 1);     // strip-local
        apix[i] = y * PW + x;
        apar[i] = y & 1;
    }
    using half8_t = typename EL::V8;
    const half_t* wl = (const half_t*)P.w16 + ((size_t)kq * ncol + cg * NCB * 16 + l15) * 8;      // + ((tp * nchunk + ch) * 4) * ncol * 8

    for (int e = tid; e < (TH + 2) * 2 * 4; e += 256) {      // the two halo columns: zero for every chunk, written once
        const int g = e & 3, side = (e >> 2) & 1, r = e >> 3;
        *(half4_t*)(tile + ((size_t)(r * PW + (side ? W + 1 : 0)) * 4 + g) * 4) = EL::zero();
    }
    for (int ch = 0; ch < nchunk; ++ch) {
        // ---- stage 16 channels of the strip (halo included) as halves: one 8-byte quad per (pixel, quad) ----
        // (a rolled loop on purpose: this kernel lives on occupancy -- 28 VGPRs; batching the rounds' loads or preloading the
        //  nine taps' weights raised the register count and made it 1.1x / 2x slower)
        // Row-wise, shift-only index math (the staging is VALU-bound: s_memtime stamps put it at half of a workgroup's life when
        // the element index was decomposed by divisions): a tile row is W x 4 quads = a power of two of elements, a thread keeps
        // ONE (column, quad) for all rows; only the interior columns are written (the halo columns were zeroed once above).
        {
            const int rows = TH + 2, rpi = lw >= 8 ? 1 : (256 >> lw);
            const int g = tid & 3, rsub = lw >= 8 ? 0 : tid >> lw;
            const int k0 = ch * 16 + 4 * g;
            const bool kok = k0 < cp, isa = k0 < pa4;
            // (W = 128, the build-defined config-5 variant: a tile row is 512 (column, quad) elements: two column passes per thread)
            for (int x = (tid & ((1 << (lw < 8 ? lw : 8)) - 1)) >> 2; x < W; x += 64) {
            half_t* const dst0 = tile + ((size_t)(1 + x) * 4) * 4;
            for (int rb = 0; rb < rows; rb += rpi) {
                const int r = rb + rsub, y = row0 + r - 1;
                if (r < rows) {
                    half4_t v = EL::zero();
                    if (kok && y >= 0 && y < H) {
                        if (isa) {
                            const uint32_t pix = (uint32_t)((img * H + y) * W + x);
                            if constexpr (A_U8) {
                                const uint8_t* s8 = (const uint8_t*)P.a + pix * (uint32_t)P.ca + k0;
                                const float sc = 1.f / 255.f;
                                half_t t4[4] = {EL::cvt(s8[0] * sc), EL::cvt(0.f), EL::cvt(0.f), EL::cvt(0.f)};
                                if (k0 + 1 < P.ca) t4[1] = EL::cvt(s8[1] * sc);
                                if (k0 + 2 < P.ca) t4[2] = EL::cvt(s8[2] * sc);
                                if (k0 + 3 < P.ca) t4[3] = EL::cvt(s8[3] * sc);
                                v = *(const half4_t*)t4;
                            } else if (P.a_f32) {          // fp32 NHWC, any ca (uniform branch: training of the 128x128 variant only)
                                const float* sf = (const float*)P.a + pix * (uint32_t)P.ca + k0;
                                half_t t4[4] = {EL::cvt(sf[0]), EL::cvt(0.f), EL::cvt(0.f), EL::cvt(0.f)};
                                if (k0 + 1 < P.ca) t4[1] = EL::cvt(sf[1]);
                                if (k0 + 2 < P.ca) t4[2] = EL::cvt(sf[2]);
                                if (k0 + 3 < P.ca) t4[3] = EL::cvt(sf[3]);
                                v = *(const half4_t*)t4;
                            } else {
                                v = *(const half4_t*)((const half_t*)P.a + (pix * (uint32_t)P.ca + k0));       // (ca % 4 == 0)
                            }
                        } else if constexpr (HASB) {
                            const uint32_t pixb = (uint32_t)((img * HB + (y >> ush)) * WB + (x >> ush));
                            v = *(const half4_t*)((const half_t*)P.b + (pixb * (uint32_t)P.cb + (k0 - pa4)));
                        }
                    }
                    *(half4_t*)(dst0 + ((size_t)r * PW * 4 + (g ^ ((r & 1) << 1))) * 4) = v;
                }
            }
            if (W <= 64) break;
            }
        }
        G16_STAMP(1);
        __syncthreads();
        G16_STAMP(2);
#pragma unroll 1
        for (int tp = 0; tp < 5; ++tp) {              // tap pairs: lane group kq takes tap 2 tp + (kq >> 1) (the tenth tap: zero weights)
            int tap = 2 * tp + (kq >> 1);
            tap = tap < 9 ? tap : 8;
            const int ty = tap / 3, toff = ty * PW + tap % 3;
            half8_t b[NCB];
            const half_t* wp = wl + (size_t)((tp * nchunk + ch) * 4) * ncol * 8;
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                b[c] = *(const half8_t*)(wp + 128 * c);
                if (!(cg * NCB * 16 + 16 * c + l15 < ncol)) b[c] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int i = 0; i < G16_MAX_TPW; ++i) {
                if (wave + 4 * i < ntiles) {
                    // channels 8 (kq & 1) .. + 7 = quads 2 (kq & 1), + 1; the row-parity swizzle (quad ^ 2 par) moves the pair as a whole
                    const int opos = (kq & 1) ^ ((apar[i] + ty) & 1);
                    const half8_t a = *(const half8_t*)(tile + ((size_t)(apix[i] + toff) * 4 + 2 * opos) * 4);
#pragma unroll
                    for (int c = 0; c < NCB; ++c) acc[i][c] = EL::mfma32(a, b[c], acc[i][c]);
                }
            }
        }
        G16_STAMP(3);
        __syncthreads();
    }
    G16_STAMP(4);
    // ---- epilogue: D[m = 4 kq + j][n = l15] ----
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        const int col = cg * NCB * 16 + 16 * c + l15;
        if (col >= P.co) continue;
        const float bias = P.bias[col];
#pragma unroll
        for (int i = 0; i < G16_MAX_TPW; ++i) {
            const int t = wave + 4 * i;
            if (t >= ntiles) continue;
            const int q = 4 * t + kq, qy = q / QW, qx = q % QW;
            if (P.pool) {
                float m = gen_act(acc[i][c][0] + bias, P.act, P.slope);
                uint32_t code = 0;
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    const float vj = gen_act(acc[i][c][j] + bias, P.act, P.slope);
                    if (vj > m) { m = vj; code = j; }          // first maximum wins, as max_pool2d
                }
                const size_t pp = (((size_t)img * (H / 2) + row0 / 2 + qy) * (W / 2) + qx) * P.co + col;
                if (P.out_f32) ((float*)P.out)[pp] = m; else ((half_t*)P.out)[pp] = EL::cvt(m);
                if (P.codes) P.codes[pp] = (uint8_t)(m > 0.f ? code : 4u);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = row0 + 2 * qy + (j >> 1), x = 2 * qx + (j & 1);
                    const size_t o = (((size_t)img * H + y) * W + x) * P.co + col;
                    const float v = gen_act(acc[i][c][j] + bias, P.act, P.slope);
                    if (P.out_f32) ((float*)P.out)[o] = v; else ((half_t*)P.out)[o] = EL::cvt(v);
                }
            }
        }
    }
    G16_STAMP(5);
}

struct Gen16GemmParams {
    const void* x; const float* w; const float* bias; void* out;
    int m, k, n, act, x_f16, out_f16;
    float slope;
};

// out[m][n] = act(sum_k x[m][k] w[k][n] + bias[n]); one wave per 16 x 16 tile, fp32 MFMA on converted rows (tiny layers)
template <bool BF>
__global__ void __launch_bounds__(64) gen16_gemm_kernel(Gen16GemmParams P) {
    using EL = E16<BF>;
    using half_t = typename EL::T;
    const int lane = threadIdx.x, l15 = lane & 15, kq = lane >> 4;
    const int ntn = (P.n + 15) / 16;
    const int m0 = (blockIdx.x / ntn) * 16, n0 = (blockIdx.x % ntn) * 16;
    const int row = m0 + l15, col = n0 + l15;
    const bool rok = row < P.m, cok = col < P.n;
    const size_t xoff = (size_t)(rok ? row : 0) * P.k;
    const float* wc = P.w + (cok ? col : 0);
    frag4 acc = frag4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < P.k; k0 += 32) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + kq, kc = k < P.k ? k : P.k - 1;
            a[u] = P.x_f16 ? EL::up(((const half_t*)P.x)[xoff + kc]) : ((const float*)P.x)[xoff + kc];
            b[u] = wc[(size_t)kc * P.n];
            a[u] = (rok && k < P.k) ? a[u] : 0.f;
            b[u] = (cok && k < P.k) ? b[u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
    }
    if (cok) {
        const float bias = P.bias ? P.bias[col] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = m0 + 4 * kq + j;
            if (r < P.m) {
                const float v = gen_act(acc[j] + bias, P.act, P.slope);
                if (P.out_f16) ((half_t*)P.out)[(size_t)r * P.n + col] = EL::cvt(v); else ((float*)P.out)[(size_t)r * P.n + col] = v;
            }
        }
    }
}

}  // namespace

#ifdef CGS_DEBUG_STAMPS
extern "C" int dbg_gen16_stamps(unsigned long long* p) { g_gen16_stamps = p; return CGS_OK; }
#endif

extern "C" int64_t cgs_gen16_packed_weight_halves(int32_t ca, int32_t cb, int32_t co) {
    if (ca <= 0 || cb < 0 || co <= 0 || (cb & 3)) return CGS_ERR_BADARG;
    const int cp = ((ca + 3) & ~3) + cb;
    return (int64_t)5 * ((cp + 15) / 16) * 4 * ((co + 15) / 16 * 16) * 8;
}

static int gen16_pack(bool bf, int32_t ca, int32_t cb, int32_t co, const float* w, void* w16, cgs_stream_t stream) {
    if (ca <= 0 || cb < 0 || co <= 0 || (cb & 3) || !w || !w16) return CGS_ERR_BADARG;
    const int64_t total = cgs_gen16_packed_weight_halves(ca, cb, co);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (bf) hipLaunchKernelGGL(gen16_pack_weights_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, ca, cb, co, (__bf16*)w16);
    else hipLaunchKernelGGL(gen16_pack_weights_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, ca, cb, co, (half_t*)w16);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}
extern "C" int cgs_gen16_pack_weights(int32_t ca, int32_t cb, int32_t co, const float* w, void* w16, cgs_stream_t stream) {
    return gen16_pack(false, ca, cb, co, w, w16, stream);
}
extern "C" int cgs_genbf16_pack_weights(int32_t ca, int32_t cb, int32_t co, const float* w, void* w16, cgs_stream_t stream) {
    return gen16_pack(true, ca, cb, co, w, w16, stream);
}

template <bool BF>
static int gen16_conv(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_is_u8, int32_t ups, int32_t act, float slope,
                      int32_t pool, int32_t out_is_f32, const void* src_a, const void* src_b, const void* w16, const float* bias, void* out,
                      cgs_stream_t stream, uint8_t* codes = nullptr) {
    const int a_f32 = a_is_u8 == 2;                                   // a_is_u8: 0 = 16-bit, 1 = uint8, 2 = fp32 (bf16 training path)
    a_is_u8 = a_is_u8 == 1;
    if (n < 0 || !src_a || !w16 || !bias || !out || ca <= 0 || cb < 0 || co <= 0) return CGS_ERR_BADARG;
    if (!a_is_u8 && !a_f32 && (ca & 3)) return CGS_ERR_BADARG;        // 16-bit sources are read 4 channels at a time
    if (codes && !pool) return CGS_ERR_BADARG;
    if (cb > 0 && (!src_b || (cb & 3) || (ups != 1 && ups != 2 && ups != 4))) return CGS_ERR_BADARG;
    if (!(gen_hw_ok(hw) || hw == 128)) return CGS_ERR_UNSUPPORTED;    // 128: the build-defined config-5 variant
    if (act < CGS_ACT_NONE || act > CGS_ACT_SIGMOID) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    Gen16ConvParams P{src_a, src_b, w16, bias, out, a_is_u8, ca, cb, cb > 0 ? ups : 1, n, hw, co, act, pool,
                      gen_strip_rows(hw), out_is_f32, slope, g_gen16_stamps, codes, a_f32};
    const int ncb = (co + 15) / 16, strips = hw / P.th;
    const int per = ncb == 1 ? 1 : ((ncb == 2 || ncb == 4) ? 2 : 3);
    const size_t lds = (size_t)(P.th + 2) * (hw + 2) * 16 * 2;
    const dim3 grid(n * strips * ((ncb + per - 1) / per));
#define G16K(NCB_) (a_is_u8 ? (cb > 0 ? gen16_conv3x3_kernel<NCB_, true, true, BF> : gen16_conv3x3_kernel<NCB_, true, false, BF>) \
                            : (cb > 0 ? gen16_conv3x3_kernel<NCB_, false, true, BF> : gen16_conv3x3_kernel<NCB_, false, false, BF>))
    auto k = per == 1 ? G16K(1) : (per == 2 ? G16K(2) : G16K(3));
#undef G16K
    hipLaunchKernelGGL(k, grid, dim3(256), lds, (hipStream_t)stream, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}
extern "C" int cgs_gen16_conv3x3_fwd(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_is_u8, int32_t ups,
                                     int32_t act, float slope, int32_t pool, int32_t out_is_f32, const void* src_a, const void* src_b,
                                     const void* w16, const float* bias, void* out, cgs_stream_t stream) {
    if (hw == 128 || (a_is_u8 != 0 && a_is_u8 != 1)) return CGS_ERR_UNSUPPORTED;          // fp16 = the reference's 64x64 model (config 4)
    return gen16_conv<false>(n, hw, ca, cb, co, a_is_u8, ups, act, slope, pool, out_is_f32, src_a, src_b, w16, bias, out, stream);
}
extern "C" int cgs_genbf16_conv3x3_fwd(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_is_u8, int32_t ups,
                                       int32_t act, float slope, int32_t pool, int32_t out_is_f32, const void* src_a, const void* src_b,
                                       const void* w16, const float* bias, void* out, cgs_stream_t stream) {
    return gen16_conv<true>(n, hw, ca, cb, co, a_is_u8, ups, act, slope, pool, out_is_f32, src_a, src_b, w16, bias, out, stream);
}

// training form of the bf16 convolution: a_kind 0 = bf16, 1 = uint8, 2 = fp32 source A; codes (pool = 1) = the argmax bytes the backward
// pass re-expands the pooled gradient with (cgs_bf16_pool_expand)
extern "C" int cgs_genbf16_conv3x3_fwd_train(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_kind, int32_t ups,
                                             int32_t act, float slope, int32_t pool, int32_t out_is_f32, const void* src_a, const void* src_b,
                                             const void* w16, const float* bias, void* out, uint8_t* codes, cgs_stream_t stream) {
    if (a_kind < 0 || a_kind > 2) return CGS_ERR_BADARG;
    return gen16_conv<true>(n, hw, ca, cb, co, a_kind, ups, act, slope, pool, out_is_f32, src_a, src_b, w16, bias, out, stream, codes);
}

extern "C" int cgs_genbf16_pack_weights_t(int32_t ci_layer, int32_t co_layer, const float* w, void* w16, cgs_stream_t stream) {
    if (ci_layer <= 0 || co_layer <= 0 || !w || !w16) return CGS_ERR_BADARG;
    const int64_t total = cgs_gen16_packed_weight_halves(co_layer, 0, ci_layer);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(gen16_pack_weights_T_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, ci_layer, co_layer, (__bf16*)w16);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// every layer's operands of a training step in ONE launch (the 128x128 variant repacks its bf16 copies after each Adam step: ~30 small
// launches before): jobs = DEVICE array; transposed = the data gradient's operand (ca = the layer's input channels, co = its outputs)
template <bool BF>
__global__ void __launch_bounds__(256) gen16_pack_batch_kernel(const cgs_gen16_pack_job* __restrict__ jobs) {
    const cgs_gen16_pack_job J = jobs[blockIdx.y];
    typename E16<BF>::T* out = (typename E16<BF>::T*)J.out;
    if (!J.transposed) {
        const int pa4 = (J.ca + 3) & ~3, cp = pa4 + J.cb, nchunk = (cp + 15) / 16, ncol = (J.co + 15) / 16 * 16, ci_total = J.ca + J.cb;
        const int total = 5 * nchunk * 4 * ncol * 8;
        for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
            const int j = e & 7, col = (e >> 3) % ncol, kq = ((e >> 3) / ncol) & 3, ch = ((e >> 3) / (ncol * 4)) % nchunk, tp = (e >> 3) / (ncol * 4 * nchunk);
            const int tap = 2 * tp + (kq >> 1), k = ch * 16 + 8 * (kq & 1) + j;
            const int ci = k < pa4 ? (k < J.ca ? k : -1) : (k < cp ? J.ca + (k - pa4) : -1);
            out[e] = E16<BF>::cvt((tap < 9 && ci >= 0 && col < J.co) ? J.w[((size_t)tap * ci_total + ci) * J.co + col] : 0.f);
        }
    } else {
        const int ci_layer = J.ca + J.cb, co_layer = J.co;
        const int cp = (co_layer + 3) & ~3, nchunk = (cp + 15) / 16, ncol = (ci_layer + 15) / 16 * 16;
        const int total = 5 * nchunk * 4 * ncol * 8;
        for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
            const int j = e & 7, col = (e >> 3) % ncol, kq = ((e >> 3) / ncol) & 3, ch = ((e >> 3) / (ncol * 4)) % nchunk, tp = (e >> 3) / (ncol * 4 * nchunk);
            const int tap = 2 * tp + (kq >> 1), k = ch * 16 + 8 * (kq & 1) + j;
            out[e] = E16<BF>::cvt((tap < 9 && k < co_layer && col < ci_layer) ? J.w[((size_t)(8 - tap) * ci_layer + col) * co_layer + k] : 0.f);
        }
    }
}

extern "C" int cgs_genbf16_pack_batch(const cgs_gen16_pack_job* jobs, int32_t njobs, cgs_stream_t stream) {
    if (!jobs || njobs < 0) return CGS_ERR_BADARG;
    if (njobs == 0) return CGS_OK;
    hipLaunchKernelGGL(gen16_pack_batch_kernel<true>, dim3(16, njobs), dim3(256), 0, (hipStream_t)stream, jobs);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

static int gen16_gemm(bool bf, int32_t m, int32_t k, int32_t n, int32_t act, float slope, int32_t x_is_16, int32_t out_is_16, const void* x,
                      const float* w, const float* bias, void* out, cgs_stream_t stream) {
    if (m < 0 || k <= 0 || n <= 0 || !x || !w || !out || act < CGS_ACT_NONE || act > CGS_ACT_SIGMOID) return CGS_ERR_BADARG;
    if (m == 0) return CGS_OK;
    Gen16GemmParams P{x, w, bias, out, m, k, n, act, x_is_16, out_is_16, slope};
    const dim3 grid(((m + 15) / 16) * ((n + 15) / 16));
    if (bf) hipLaunchKernelGGL(gen16_gemm_kernel<true>, grid, dim3(64), 0, (hipStream_t)stream, P);
    else hipLaunchKernelGGL(gen16_gemm_kernel<false>, grid, dim3(64), 0, (hipStream_t)stream, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}
extern "C" int cgs_gen16_gemm(int32_t m, int32_t k, int32_t n, int32_t act, float slope, int32_t x_is_f16, int32_t out_is_f16,
                              const void* x, const float* w, const float* bias, void* out, cgs_stream_t stream) {
    return gen16_gemm(false, m, k, n, act, slope, x_is_f16, out_is_f16, x, w, bias, out, stream);
}
extern "C" int cgs_genbf16_gemm(int32_t m, int32_t k, int32_t n, int32_t act, float slope, int32_t x_is_bf16, int32_t out_is_bf16,
                                const void* x, const float* w, const float* bias, void* out, cgs_stream_t stream) {
    return gen16_gemm(true, m, k, n, act, slope, x_is_bf16, out_is_bf16, x, w, bias, out, stream);
}

// features.0 of the shape-generic family (NewCritic at chfak != 1, nets.py:170-172: Conv2d(3, 8 chfak, 3, 1, 1) + ReLU + MaxPool2d(2) on 64x64 frames),
// forward -- a kernel of its own (round 4): on gen4_conv3x3_kernel the layer ran at 22 % of the fp32 matrix peak (258 us per 1024 frames at
// chfak 5): with 27 k-steps per output group a tap is 30 matrix instructions, too few to hide the per-tap weight-register loads, and the
// lane = pixel epilogue pools with 11 vector instructions per channel and pixel.  Here
//   * lane = one 2x2 POOL CELL (as the fixed-shape chfak-1 kernels, conv_tile.h): its 4x4 input patch sits in registers, the four pixels of
//     the cell are four accumulator sets, ReLU + MaxPool2d(2) + the argmax byte are in-lane and cost a quarter per pixel;
//   * ALL weights of the layer live in registers for the lifetime of a persistent workgroup: v_mfma_f32_4x4x1_16B_f32 with the A operand
//     broadcast from one of its 16 blocks (cbsz = 4, abid) -- a register holds 16 (k-step, group) combinations, 27 NG / 16 registers in all
//     (17 at 40 output channels); nothing is loaded inside the matrix loop but the patch (16 ds_read_b128 per 256 pixels);
//   * output groups in passes of GP groups (4 x GP accumulators: 80 registers at GP = 5), the patch is reused from registers;
//   * a wave = 64 cells = a 64 x 4 pixel band of one image; a workgroup = a 16-row strip, persistent over the strips of the batch; pooled
//     results leave through a per-wave LDS block in full lines.
// Same results as the generic kernel up to the summation order (fp32 FMA chains either way); parity through the chfak != 1 captures.
#include <type_traits>
#include <utility>
#include "gen_common.h"

namespace {

struct GEnc0FwdParams {
    const void* a;            // uint8 or fp32 frames [n,64,64,3]
    const float* w;           // HWIO [9][3][co]
    const float* bias;        // [co]
    float* out;               // [n,32,32,co]
    uint8_t* am;              // [n,32,32,co] argmax bytes (position 0..3, bit 2: pooled value <= 0) or NULL
    int n, a_is_u8, nstrips;
};

template <int N, class F>
__device__ __forceinline__ void ge_static_for(F&& f) {
    if constexpr (N > 0) {
        ge_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__device__ __forceinline__ float ge_f4get(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// NG: output groups of 4 channels (co = 4 NG); GP: groups per pass (NG % GP == 0).
// Every WAVE works on its own: it stages the 6 input rows of its 64 x 4 pixel band into a private LDS block (the halo rows are loaded twice per
// image: 3 bytes per pixel), multiplies, pools, and copies its two pooled rows out through the same block -- no workgroup barrier anywhere in the
// loop.  (r4 A/B: a 16-row strip per workgroup with two barriers per strip, with and without the next strip's loads in flight under the matrix
// loop, measures the same 127 us per 1024 frames at 40 channels: the kernel is bound by instruction issue -- 1080 matrix + ~930 vector
// instructions per 256 pixels, 0.45-0.5 of the fp32 matrix peak -- not by barriers or staging.)
template <int NG, int GP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) genc0_fwd_kernel(GEnc0FwdParams P) {
    constexpr int CO = 4 * NG, NK = 27 * NG, NV = (NK + 15) / 16, NPASS = NG / GP;
    constexpr int TW = 66, TROWS = 6;                       // a wave's input tile: 4 rows + halo, 64 columns + halo, float4 (r, g, b, 0) per pixel
    constexpr int PITCH = CO + 4;                           // floats per cell in the output block (pitch = 4 mod 8 quads: conflict-free b128)
    constexpr int OBF = 64 * PITCH + 64 * NG, TF = TROWS * TW * 4;
    constexpr int WBLK = OBF > TF ? OBF : TF;               // floats of a wave's block: values [64][PITCH] + codes [64][NG]; the tile aliases its head
    static_assert(NG % GP == 0, "passes of equal size");
    extern __shared__ __attribute__((aligned(16))) float4 gsm[];
    float* const bias_s = (float*)gsm;                      // [CO]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* const ob = bias_s + CO + wave * WBLK;
    uint32_t* const cb = (uint32_t*)(ob + 64 * PITCH);
    float4* const tile = (float4*)ob;                       // [TROWS][TW]: dead once the patch is in registers

    // ---- weights -> registers: register v, lane 4 b + i = combination q = 16 v + b = (k-step ks = q / NG, group g = q % NG), output channel 4 g + i ----
    float wreg[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int q = 16 * v + (lane >> 2), ks = q / NG, g = q % NG;
        wreg[v] = q < NK ? P.w[ks * CO + 4 * g + (lane & 3)] : 0.f;
    }
    for (int e = tid; e < CO; e += 256) bias_s[e] = P.bias[e];
    __syncthreads();                                        // (the only workgroup barrier: the bias table)

    // staging: the band's 6 rows x 16 items of 4 pixels (12 bytes / 12 floats) = 96 items, two per lane at most: fetch (all loads back to back),
    // then commit (conversion + LDS stores)
    float4 rx[2][3];
    auto fetch = [&](int band) {
        const int img = band >> 4, row0 = (band & 15) * 4;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = lane + 64 * k, g4 = e & 15, r = e >> 4, y = row0 + r - 1;
            const bool in = r < TROWS && y >= 0 && y < 64;
            const size_t gi = in ? (((size_t)img * 64 + y) * 64 + g4 * 4) * 3 / 4 : 0;
            if (P.a_is_u8) {
                const uint32_t* su = (const uint32_t*)P.a;
                rx[k][0] = make_float4(__uint_as_float(su[gi]), __uint_as_float(su[gi + 1]), __uint_as_float(su[gi + 2]), 0.f);
            } else {
                const float4* sf = (const float4*)P.a;
                rx[k][0] = sf[gi]; rx[k][1] = sf[gi + 1]; rx[k][2] = sf[gi + 2];
            }
        }
    };
    auto commit = [&](int band) {
        const int row0 = (band & 15) * 4;
        if (lane < TROWS * 2) tile[(lane >> 1) * TW + ((lane & 1) ? TW - 1 : 0)] = f4zero();       // halo columns (the block was the output block)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = lane + 64 * k, g4 = e & 15, r = e >> 4, y = row0 + r - 1;
            if (r >= TROWS) continue;
            const bool in = y >= 0 && y < 64;
            float f[12];
            if (P.a_is_u8) {
                const uint32_t d[3] = {__float_as_uint(rx[k][0].x), __float_as_uint(rx[k][0].y), __float_as_uint(rx[k][0].z)};
#pragma unroll
                for (int j = 0; j < 12; ++j) f[j] = (float)((d[j >> 2] >> (8 * (j & 3))) & 255u) * (1.f / 255.f);
            } else {
                f[0] = rx[k][0].x; f[1] = rx[k][0].y; f[2] = rx[k][0].z; f[3] = rx[k][0].w; f[4] = rx[k][1].x; f[5] = rx[k][1].y;
                f[6] = rx[k][1].z; f[7] = rx[k][1].w; f[8] = rx[k][2].x; f[9] = rx[k][2].y; f[10] = rx[k][2].z; f[11] = rx[k][2].w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                tile[r * TW + 1 + 4 * g4 + j] = in ? make_float4(f[3 * j], f[3 * j + 1], f[3 * j + 2], 0.f) : f4zero();
        }
    };
    const int nwaves = gridDim.x * 4, band0 = blockIdx.x * 4 + wave;
    for (int band = band0; band < P.nstrips; band += nwaves) {
        const int img = band >> 4, row0 = (band & 15) * 4;
        fetch(band);
        commit(band);
        __builtin_amdgcn_wave_barrier();
        // ---- lane = cell (cy = lane >> 5, cx = lane & 31) of the band ----
        const int cy = lane >> 5, cx = lane & 31;
        float4 pt[4][4];                                    // input patch rows 2 cy - 1 .. + 2, columns 2 cx - 1 .. + 2 (tile coordinates: + 1)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) pt[r][c] = tile[(2 * cy + r) * TW + 2 * cx + c];
        __builtin_amdgcn_wave_barrier();                    // (every lane's patch is read before any lane writes results into the same block)
        ge_static_for<NPASS>([&](auto PASS) {
            constexpr int pass = decltype(PASS)::value;
            frag4 acc[4][GP];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int g = 0; g < GP; ++g) acc[k][g] = frag4{0.f, 0.f, 0.f, 0.f};
            ge_static_for<27>([&](auto S) {
                constexpr int s = decltype(S)::value, tap = s / 3, c = s % 3, ky = tap / 3, kx = tap % 3;
#pragma unroll
                for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                    for (int ox = 0; ox < 2; ++ox) {
                        const float x = ge_f4get(pt[oy + ky][ox + kx], c);
                        ge_static_for<GP>([&](auto G) {
                            constexpr int g = decltype(G)::value, q = s * NG + pass * GP + g;
                            acc[oy * 2 + ox][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[q >> 4], x, acc[oy * 2 + ox][g], 4, q & 15, 0);
                        });
                    }
            });
            // ---- ReLU + MaxPool2d(2) + argmax byte, in the lane ----
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                const float4 b4 = *(const float4*)(bias_s + 4 * (pass * GP + g));
                const float ba[4] = {b4.x, b4.y, b4.z, b4.w};
                float m[4];
                uint32_t word = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // max over the cell of ReLU(v) = ReLU(max v); where that is positive the first position holding max v is the first one holding
                    // the maximum of the ReLU'd values (where it is not, bit 2 says "no gradient" and the position is not read)
                    const float v0 = acc[0][g][i] + ba[i], v1 = acc[1][g][i] + ba[i], v2 = acc[2][g][i] + ba[i], v3 = acc[3][g][i] + ba[i];
                    const float mv = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
                    const uint32_t cd = v0 == mv ? 0u : (v1 == mv ? 1u : (v2 == mv ? 2u : 3u));
                    word |= (cd | (mv > 0.f ? 0u : 4u)) << (8 * i);
                    m[i] = fmaxf(mv, 0.f);
                }
                *(float4*)(ob + lane * PITCH + 4 * (pass * GP + g)) = make_float4(m[0], m[1], m[2], m[3]);
                cb[lane * NG + pass * GP + g] = word;
            }
        });
        // ---- the wave's 2 pooled rows x 32 cells x CO channels leave in full lines ----
        __builtin_amdgcn_wave_barrier();                    // (LDS operations of one wave execute in order: the reads below see the writes above)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const size_t rowbase = (((size_t)img * 32 + row0 / 2 + r) * 32) * CO;
            for (int f = lane; f < 32 * NG; f += 64) {
                const int cell = f / NG, g = f % NG;
                *(float4*)(P.out + rowbase + 4 * f) = *(const float4*)(ob + (r * 32 + cell) * PITCH + 4 * g);
                if (P.am) *(uint32_t*)(P.am + rowbase + 4 * f) = cb[(r * 32 + cell) * NG + g];
            }
        }
        __builtin_amdgcn_wave_barrier();                    // (results read out before the next band's tile overwrites the block)
    }
}

template <int NG, int GP>
int genc0_fwd_launch(GEnc0FwdParams P, hipStream_t st) {
    constexpr int CO = 4 * NG;
    constexpr size_t obf = (size_t)64 * (CO + 4) + 64 * NG, tf = (size_t)6 * 66 * 4;
    constexpr size_t lds = (size_t)CO * 4 + 4 * (obf > tf ? obf : tf) * 4;
    P.nstrips = P.n * 16;                                   // bands of 4 rows, one per wave and trip
    auto k = genc0_fwd_kernel<NG, GP>;
    if (lds > 64 * 1024) {
        static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (attr != hipSuccess) return (int)attr;
    }
    const int per_cu = (int)((160 * 1024) / lds) < 2 ? 1 : 2;      // (226+ registers: two waves per SIMD)
    const int cap = 256 * per_cu, wgs = (P.nstrips + 3) / 4, rounds = (wgs + cap - 1) / cap, blocks = (wgs + rounds - 1) / rounds;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

}  // namespace

// features.0 of NewCritic at chfak 2 / 3 / 4 / 5 (co = 16 / 24 / 32 / 40): ReLU(conv3x3(frames) + bias) -> MaxPool2d(2), out [n,32,32,co] + the
// argmax bytes of cgs_gen_conv3x3_fwd (am may be NULL).  x: uint8 (x_is_u8) or fp32 frames [n,64,64,3]; w: the layer's HWIO weights [9][3][co].
// CGS_ERR_UNSUPPORTED for other channel counts (the caller takes cgs_gen_conv3x3_fwd).
extern "C" int cgs_gen_enc0_fwd(int32_t n, int32_t co, int32_t x_is_u8, const void* x, const float* w_hwio, const float* bias, float* out,
                                uint8_t* am, cgs_stream_t stream) {
    if (n < 0 || !x || !w_hwio || !bias || !out) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    const GEnc0FwdParams P{x, w_hwio, bias, out, am, n, x_is_u8 ? 1 : 0, 0};
    switch (co) {
        case 16: return genc0_fwd_launch<4, 4>(P, (hipStream_t)stream);
        case 24: return genc0_fwd_launch<6, 3>(P, (hipStream_t)stream);
        case 32: return genc0_fwd_launch<8, 4>(P, (hipStream_t)stream);
        case 40: return genc0_fwd_launch<10, 5>(P, (hipStream_t)stream);
    }
    return CGS_ERR_UNSUPPORTED;
}

namespace {

// ---------------------------------------------------------------------------------------------------------------------------------------
// features.0, weight + bias gradient: dW[(tap, c)][oc] = sum over pixels of X[pixel + tap][c] * dY[pixel][oc], with dY given as the POOLED
// gradient d e0 [n,32,32,co] + the forward pass's argmax bytes (the gradient sits at the pixel that held the cell's maximum).
// v_mfma_f32_4x4x1_16B_f32 WITHOUT broadcast = 16 independent 4x4 outer products: block b = pixel b of a 16-pixel step, A = four of the
// 28 rows ((tap, c) pairs + the bias row of ones) of that pixel, B = four output channels of that pixel; the 16 blocks are 16 partial
// sums that are added once, at the end of a workgroup's life.  7 row quads x NG column quads = 70 instructions per 16 pixels at 40 channels;
// wave w owns the column quads w, w + 4, w + 8 (<= 21 accumulators) and walks every pixel of the strip: A is one 4-byte LDS read per row quad
// (per-lane tap offsets precomputed), B one value + one argmax byte per column quad, selected on the fly.  The shape-generic row-block
// kernel (gen_wgrad_rows.h) ran this layer at 14 % of the fp32 matrix peak (27 rows = 2 padded row blocks of 16x16x4 tiles, 8 waves sharing
// one staged chunk of 3 channels).
struct GEnc0WgParams {
    const void* a;            // uint8 or fp32 frames [n,64,64,3]
    const float* de;          // pooled gradient [n,32,32,co]
    const uint8_t* am;        // argmax bytes [n,32,32,co]
    float* slab;              // [blocks][27 co + co]
    int n, a_is_u8, nstrips;
};

#ifndef GC0W_UNROLL
#define GC0W_UNROLL 1
#endif
template <class F, int... Is>
__device__ __forceinline__ void genc0_static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void genc0_static_for(F&& f) { genc0_static_for_impl(f, std::make_integer_sequence<int, N>{}); }
template <int NG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) genc0_wgrad_kernel(GEnc0WgParams P) {
    constexpr int CO = 4 * NG, TW = 66, TH = 8, TROWS = TH + 2, NQW = (NG + 3) / 4;      // NQW: column quads per wave (at most)
    extern __shared__ __attribute__((aligned(16))) float4 gsm[];
    float4* const tile = gsm;                                       // [TROWS][TW] float4 (r, g, b, 1): the 1 is the bias row's operand
    float* const des = (float*)(gsm + TROWS * TW);                  // [TH / 2][32][CO] pooled gradient
    uint8_t* const ams = (uint8_t*)(des + (TH / 2) * 32 * CO);      // [TH / 2][32][CO] argmax bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = lane >> 2, i = lane & 3;

    // A operand of row quad rq, lane 4 b + i: row r = 4 rq + i = (tap, c) -> float offset inside the tile relative to the pixel; row 27 = the ones
    int aoff[7];
#pragma unroll
    for (int rq = 0; rq < 7; ++rq) {
        const int r = 4 * rq + i, tap = r < 27 ? r / 3 : 4, c = r < 27 ? r % 3 : 3;
        aoff[rq] = ((tap / 3) * TW + tap % 3) * 4 + c;
    }
    frag4 acc[NQW][7];
#pragma unroll
    for (int q = 0; q < NQW; ++q)
#pragma unroll
        for (int rq = 0; rq < 7; ++rq) acc[q][rq] = frag4{0.f, 0.f, 0.f, 0.f};

    for (int e = tid; e < TROWS * 2; e += 256) tile[(e >> 1) * TW + ((e & 1) ? TW - 1 : 0)] = f4zero();      // halo columns
    __syncthreads();
#if GC0W_UNROLL
    // per-lane operand bases of step 0 (pixel b of tile row 0): the steps add compile-time offsets
    const float* apx[7];
    const float* dpx[NQW];
    const uint8_t* mpx[NQW];
#pragma unroll
    for (int rq = 0; rq < 7; ++rq) apx[rq] = (const float*)(tile + b) + aoff[rq];
#pragma unroll
    for (int q = 0; q < NQW; ++q) {
        const int cq = wave + 4 * q, col0 = (b >> 1) * CO + 4 * (cq < NG ? cq : 0) + i;      // (past NG: a valid address, never multiplied)
        dpx[q] = des + col0; mpx[q] = ams + col0;
    }
    const uint32_t pos_even = (uint32_t)(b & 1), pos_odd = 2u + (uint32_t)(b & 1);
#endif

    for (int strip = blockIdx.x; strip < P.nstrips; strip += gridDim.x) {
        const int img = strip >> 3, row0 = (strip & 7) * TH;
        // ---- staging: every global load of the strip issued back to back (ONE memory round trip per strip), then converted / stored ----
        constexpr int NDE = (TH / 2) * 32 * CO / 4, NIT = (NDE + 255) / 256;        // float4 / uint32 items of the pooled gradient / argmax bytes
        float4 rx[3], rd[NIT];
        uint32_t ra[NIT];
        const bool xitem = tid < TROWS * 16;                        // frames: items of 4 pixels, one per thread
        const int g4 = tid & 15, xr = tid >> 4, xy = row0 + xr - 1;
        const bool xin = xitem && xy >= 0 && xy < 64;
        {
            const size_t gi = xin ? (((size_t)img * 64 + xy) * 64 + g4 * 4) * 3 / 4 : 0;
            if (P.a_is_u8) {
                const uint32_t* su = (const uint32_t*)P.a;
                rx[0] = make_float4(__uint_as_float(su[gi]), __uint_as_float(su[gi + 1]), __uint_as_float(su[gi + 2]), 0.f);
            } else {
                const float4* sf = (const float4*)P.a;
                rx[0] = sf[gi]; rx[1] = sf[gi + 1]; rx[2] = sf[gi + 2];
            }
            const size_t base = ((size_t)img * 32 + row0 / 2) * 32 * CO;
            const float4* sd = (const float4*)(P.de + base);
            const uint32_t* sa = (const uint32_t*)(P.am + base);
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int e = tid + 256 * k, ee = e < NDE ? e : NDE - 1;
                rd[k] = sd[ee]; ra[k] = sa[ee];
            }
        }
        if (xitem) {
            float f[12];
            if (P.a_is_u8) {
                const uint32_t d[3] = {__float_as_uint(rx[0].x), __float_as_uint(rx[0].y), __float_as_uint(rx[0].z)};
#pragma unroll
                for (int j = 0; j < 12; ++j) f[j] = (float)((d[j >> 2] >> (8 * (j & 3))) & 255u) * (1.f / 255.f);
            } else {
                f[0] = rx[0].x; f[1] = rx[0].y; f[2] = rx[0].z; f[3] = rx[0].w; f[4] = rx[1].x; f[5] = rx[1].y; f[6] = rx[1].z; f[7] = rx[1].w;
                f[8] = rx[2].x; f[9] = rx[2].y; f[10] = rx[2].z; f[11] = rx[2].w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                tile[xr * TW + 1 + 4 * g4 + j] = xin ? make_float4(f[3 * j], f[3 * j + 1], f[3 * j + 2], 1.f) : make_float4(0.f, 0.f, 0.f, 1.f);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int e = tid + 256 * k;
            if (e < NDE) { ((float4*)des)[e] = rd[k]; ((uint32_t*)ams)[e] = ra[k]; }
        }
        __syncthreads();
        // ---- 16-pixel steps: 16 consecutive pixels of a row; block b = pixel x0 + b.  Two operand sets: the LDS reads of step s + 1 are in flight
        //      behind the matrix instructions of step s (left to the compiler the reads of a step were waited for in front of its own
        //      instructions: matrix pipe busy 0.33, waves waiting 0.54 of their cycles) ----
        auto mfmas = [&](const float (&av)[7], const float (&bv)[NQW], const uint32_t (&bm)[NQW + 1]) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < NQW; ++q) {
                if (wave + 4 * q < NG) {                            // (wave-uniform)
                    const float v = bm[q] == bm[NQW] ? bv[q] : 0.f;
#pragma unroll
                    for (int rq = 0; rq < 7; ++rq) acc[q][rq] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[rq], v, acc[q][rq], 0, 0, 0);
                }
            }
        };
#if GC0W_UNROLL
        // (round 6) The 32 steps of a strip as straight-line code: a step's place in the strip is a compile-time constant, so every LDS read is
        // a per-lane base register (set once per workgroup) + an IMMEDIATE offset and the select's position code one of two per-lane registers --
        // 6 vector instructions per step beside its <= 21 matrix instructions where the loop form spent 30 on addresses (fp32 matrix and vector
        // instructions of a SIMD do not overlap: the kernel ran at 0.35 matrix-pipe busy).  Same operands in the same order: bitwise the same sums.
        auto load_at = [&](auto SC, float (&av)[7], float (&bv)[NQW], uint32_t (&bm)[NQW + 1]) __attribute__((always_inline)) {
            constexpr int s = decltype(SC)::value, y = s >> 2, xs = (s & 3) * 16;
#pragma unroll
            for (int rq = 0; rq < 7; ++rq) av[rq] = apx[rq][(y * TW + xs) * 4];
            constexpr int cellc = (y >> 1) * 32 + (xs >> 1);
            bm[NQW] = (y & 1) ? pos_odd : pos_even;
#pragma unroll
            for (int q = 0; q < NQW; ++q) { bv[q] = dpx[q][cellc * CO]; bm[q] = mpx[q][cellc * CO]; }
        };
        {
            float a0[7], a1[7], b0[NQW], b1[NQW];
            uint32_t m0[NQW + 1], m1[NQW + 1];
            load_at(std::integral_constant<int, 0>{}, a0, b0, m0);
            genc0_static_for<TH * 2>([&](auto KC) {
                constexpr int s = 2 * decltype(KC)::value;
                load_at(std::integral_constant<int, s + 1>{}, a1, b1, m1);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(a0, b0, m0);
                __builtin_amdgcn_sched_barrier(0);
                load_at(std::integral_constant<int, (s + 2 < TH * 4 ? s + 2 : s)>{}, a0, b0, m0);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(a1, b1, m1);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
#else
        auto load_ops = [&](int s, float (&av)[7], float (&bv)[NQW], uint32_t (&bm)[NQW + 1]) __attribute__((always_inline)) {
            const int y = s >> 2, x = (s & 3) * 16 + b;
            const float* px = (const float*)(tile + y * TW + x);    // tap (0,0) of the pixel's 3x3 window (tile row 0 = image row row0 - 1)
#pragma unroll
            for (int rq = 0; rq < 7; ++rq) av[rq] = px[aoff[rq]];
            const int cell = (y >> 1) * 32 + (x >> 1);
            bm[NQW] = 2 * (y & 1) + (x & 1);
#pragma unroll
            for (int q = 0; q < NQW; ++q) {
                const int cq = wave + 4 * q;                        // this wave's column quad (past NG: a valid address, never multiplied)
                const int col = cell * CO + 4 * (cq < NG ? cq : 0) + i;
                bv[q] = des[col]; bm[q] = ams[col];
            }
        };
        {
            float a0[7], a1[7], b0[NQW], b1[NQW];
            uint32_t m0[NQW + 1], m1[NQW + 1];
            load_ops(0, a0, b0, m0);
#pragma unroll 1
            for (int s = 0; s < TH * 4; s += 2) {
                load_ops(s + 1, a1, b1, m1);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(a0, b0, m0);
                __builtin_amdgcn_sched_barrier(0);
                load_ops(s + 2 < TH * 4 ? s + 2 : s, a0, b0, m0);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(a1, b1, m1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
        __syncthreads();
    }
    // ---- add the 16 blocks (lanes with equal lane & 3), one slab row per workgroup: D_b[r][j] -> row 4 rq + r, column 4 cq + j (j = lane & 3) ----
    float* const sl = P.slab + (size_t)blockIdx.x * (28 * CO);
#pragma unroll
    for (int q = 0; q < NQW; ++q) {
        const int cq = wave + 4 * q;
        if (cq >= NG) continue;
#pragma unroll
        for (int rq = 0; rq < 7; ++rq)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[q][rq][r];
                v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                const int row = 4 * rq + r;                         // rows 0 .. 26: (tap, c); row 27: bias
                if (lane < 4) sl[(row < 27 ? row * CO : 27 * CO) + 4 * cq + lane] = v;
            }
    }
}

template <int NG>
struct GEnc0Wg {
    static constexpr int CO = 4 * NG;
    static constexpr size_t lds = (size_t)10 * 66 * 16 + (size_t)4 * 32 * CO * 4 + (size_t)4 * 32 * CO;
    static int blocks(int n) {
        const int nstrips = n * 8;
        if (nstrips <= 0) return 0;
        int per_cu = (int)((160 * 1024) / lds);
        per_cu = per_cu < 3 ? per_cu : 3;
        const int cap = 256 * per_cu, rounds = (nstrips + cap - 1) / cap;
        return (nstrips + rounds - 1) / rounds;
    }
    static int launch(GEnc0WgParams P, hipStream_t st) {
        P.nstrips = P.n * 8;
        hipLaunchKernelGGL(genc0_wgrad_kernel<NG>, dim3(blocks(P.n)), dim3(256), lds, st, P);
        CGS_HIP_CHECK_LAUNCH();
        return CGS_OK;
    }
};

}  // namespace

// Slab rows cgs_gen_enc0_bwd_weight writes for n images (0: unsupported channel count).
extern "C" int cgs_gen_enc0_bwd_weight_slabs(int32_t n, int32_t co) {
    if (n < 0) return CGS_ERR_BADARG;
    switch (co) {
        case 16: return GEnc0Wg<4>::blocks(n);
        case 24: return GEnc0Wg<6>::blocks(n);
        case 32: return GEnc0Wg<8>::blocks(n);
        case 40: return GEnc0Wg<10>::blocks(n);
    }
    return 0;
}

// dW / db slabs [rows][27 co + co] of features.0 (co = 16 / 24 / 32 / 40) from the frames x (uint8 / fp32 [n,64,64,3]), the pooled gradient
// de [n,32,32,co] and the forward pass's argmax bytes am (same tensors as cgs_gen_conv3x3_bwd_weight(hw 64, ca 3, cb 0)).
extern "C" int cgs_gen_enc0_bwd_weight(int32_t n, int32_t co, int32_t x_is_u8, const void* x, const float* de, const uint8_t* am, float* slab,
                                       cgs_stream_t stream) {
    if (n < 0 || !x || !de || !am || !slab) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    const GEnc0WgParams P{x, de, am, slab, n, x_is_u8 ? 1 : 0, 0};
    switch (co) {
        case 16: return GEnc0Wg<4>::launch(P, (hipStream_t)stream);
        case 24: return GEnc0Wg<6>::launch(P, (hipStream_t)stream);
        case 32: return GEnc0Wg<8>::launch(P, (hipStream_t)stream);
        case 40: return GEnc0Wg<10>::launch(P, (hipStream_t)stream);
    }
    return CGS_ERR_UNSUPPORTED;
}

namespace {

// ---------------------------------------------------------------------------------------------------------------------------------------
// features.0, data gradient (the image gradient of the two mixes): d x [n,64,64,3] = conv3x3^T(dY) with dY given as the pooled gradient d e0
// [n,32,32,co] + argmax bytes.  Three outputs and K = 9 co: on gen4 (lane = pixel, one output group) every input element costs one LDS read per tap
// and the layer ran at 0.20 of the matrix peak.  Here lane = one 2x2 cell of d x (four accumulator sets of ONE group: 16 registers), ALL
// 9 co weight steps in 9 co / 16 registers (23 at 40 channels: v_mfma_f32_4x4x1 with A broadcast, rows = the three colour channels), and the
// cell's 4x4 patch of dY is built in registers channel quad by channel quad from the 3x3 pooled cells around it (9 value + 9 argmax-word LDS
// reads per quad and lane, each value kept where its byte says the maximum was): 144 matrix instructions per quad and 256 pixels.
struct GEnc0DgParams {
    const float* de;          // pooled gradient [n,32,32,co]
    const uint8_t* am;        // argmax bytes [n,32,32,co]
    const float* w;           // HWIO [9][3][co]
    float* dx;                // [n,64,64,3]
    int n, nstrips;
};

template <int NG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) genc0_dgrad_kernel(GEnc0DgParams P) {
    constexpr int CO = 4 * NG, NK = 9 * CO, NV = (NK + 15) / 16;
    constexpr int CW = 34, CROWS = 10;                      // pooled tile: 8 cell rows + halo, 32 cells + halo; [CROWS][CW][CO] values, then the argmax bytes
    extern __shared__ __attribute__((aligned(16))) float4 gsm[];
    float* const des = (float*)gsm;
    uint8_t* const ams = (uint8_t*)(des + CROWS * CW * CO);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // weights: step q = tap' * CO + oc of the TRANSPOSED convolution reads W[8 - tap'][c = i][oc]; register q >> 4, lanes 4 (q & 15) + i
    float wreg[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int q = 16 * v + (lane >> 2), tp = q / CO, oc = q % CO, i = lane & 3;
        wreg[v] = (q < NK && i < 3) ? P.w[((8 - tp) * 3 + i) * CO + oc] : 0.f;
    }
    // halo cells of the tile: value 0 with "no gradient" bytes, for every strip (columns 0 and CW - 1)
    for (int e = tid; e < CROWS * 2 * CO; e += 256) {
        const int c = e % CO, side = (e / CO) & 1, r = e / (2 * CO), cell = r * CW + (side ? CW - 1 : 0);
        des[cell * CO + c] = 0.f; ams[cell * CO + c] = 4;
    }
    __syncthreads();

    for (int strip = blockIdx.x; strip < P.nstrips; strip += gridDim.x) {
        const int img = strip >> 2, crow0 = (strip & 3) * 8;            // 16 pixel rows = 8 cell rows of the image
        // ---- stage the pooled rows crow0 - 1 .. crow0 + 8: all loads first (float4 values, uint32 argmax words), then the LDS stores ----
        constexpr int NI = CROWS * 32 * NG, NIT = (NI + 255) / 256;
        float4 rv[NIT];
        uint32_t ra[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int e = tid + 256 * k, ee = e < NI ? e : 0, g = ee % NG, cx = (ee / NG) & 31, r = ee / (32 * NG), cy = crow0 + r - 1;
            const bool in = cy >= 0 && cy < 32;
            const size_t gi = in ? (((size_t)img * 32 + cy) * 32 + cx) * NG + g : 0;
            rv[k] = ((const float4*)P.de)[gi];
            ra[k] = ((const uint32_t*)P.am)[gi];
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int e = tid + 256 * k;
            if (e >= NI) continue;
            const int g = e % NG, cx = (e / NG) & 31, r = e / (32 * NG), cy = crow0 + r - 1;
            const bool in = cy >= 0 && cy < 32;
            const int cell = r * CW + 1 + cx;
            *(float4*)(des + cell * CO + 4 * g) = in ? rv[k] : f4zero();
            *(uint32_t*)(ams + cell * CO + 4 * g) = in ? ra[k] : 0x04040404u;
        }
        __syncthreads();
        // ---- this wave's band: cell rows 2 wave, 2 wave + 1 of the strip; lane = cell (cy = lane >> 5, cx = lane & 31) ----
        const int cy = 2 * wave + (lane >> 5), cx = lane & 31;
        frag4 acc[4] = {frag4{0.f, 0.f, 0.f, 0.f}, frag4{0.f, 0.f, 0.f, 0.f}, frag4{0.f, 0.f, 0.f, 0.f}, frag4{0.f, 0.f, 0.f, 0.f}};
        ge_static_for<NG>([&](auto G) {
            constexpr int g = decltype(G)::value;
            // the 3 x 3 pooled cells around the lane's cell, channel quad g (tile coordinates: + 1 row / column for the halo)
            float4 cv[3][3];
            uint32_t cw[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const int cell = (cy + a) * CW + cx + b;
                    cv[a][b] = *(const float4*)(des + cell * CO + 4 * g);
                    cw[a][b] = *(const uint32_t*)(ams + cell * CO + 4 * g);
                }
            // patch pixel (r, c), r, c = 0 .. 3 = image pixel (2 cy' - 1 + r, 2 cx' - 1 + c): cell (r + 1) >> 1, (c + 1) >> 1 of the 3 x 3, position
            // 2 ((r + 1) & 1) + ((c + 1) & 1) inside it
            // the 16 patch pixels of this quad, each kept where its argmax byte says the cell's maximum was
            float pv[4][4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int a = (r + 1) >> 1, b = (c + 1) >> 1;
                    const uint32_t pos = 2 * ((r + 1) & 1) + ((c + 1) & 1), w = cw[a][b];
                    const float4 v4 = cv[a][b];
                    pv[r][c][0] = (w & 255u) == pos ? v4.x : 0.f; pv[r][c][1] = ((w >> 8) & 255u) == pos ? v4.y : 0.f;
                    pv[r][c][2] = ((w >> 16) & 255u) == pos ? v4.z : 0.f; pv[r][c][3] = (w >> 24) == pos ? v4.w : 0.f;
                }
            ge_static_for<9>([&](auto T) {
                constexpr int tp = decltype(T)::value, ky = tp / 3, kx = tp % 3, q0 = tp * CO + 4 * g;
                ge_static_for<4>([&](auto J) {              // consecutive instructions go to the four different accumulators
                    constexpr int j = decltype(J)::value;
#pragma unroll
                    for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                        for (int ox = 0; ox < 2; ++ox)
                            acc[oy * 2 + ox] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[(q0 + j) >> 4], pv[oy + ky][ox + kx][j], acc[oy * 2 + ox], 4,
                                                                                  (q0 + j) & 15, 0);
                });
            });
        });
        // ---- d x: the cell's 2 x 2 pixels x 3 channels; a row of the band = 64 pixels x 3 floats contiguous ----
        const int py = 2 * (crow0 + cy), px = 2 * cx;
#pragma unroll
        for (int oy = 0; oy < 2; ++oy) {
            float* o = P.dx + (((size_t)img * 64 + py + oy) * 64 + px) * 3;
            const frag4 a0 = acc[oy * 2], a1 = acc[oy * 2 + 1];
            *(float2*)o = make_float2(a0[0], a0[1]);
            *(float2*)(o + 2) = make_float2(a0[2], a1[0]);
            *(float2*)(o + 4) = make_float2(a1[1], a1[2]);
        }
        __syncthreads();
    }
}

template <int NG>
int genc0_dgrad_launch(GEnc0DgParams P, hipStream_t st) {
    constexpr int CO = 4 * NG;
    constexpr size_t lds = (size_t)10 * 34 * CO * 4 + (size_t)10 * 34 * CO;
    P.nstrips = P.n * 4;
    auto k = genc0_dgrad_kernel<NG>;
    if (lds > 64 * 1024) {
        static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (attr != hipSuccess) return (int)attr;
    }
    const int per_cu = (int)((160 * 1024) / lds) < 2 ? 1 : 2;
    const int cap = 256 * per_cu, rounds = (P.nstrips + cap - 1) / cap, blocks = (P.nstrips + rounds - 1) / rounds;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

}  // namespace

// d x [n,64,64,3] of features.0 (co = 16 / 24 / 32 / 40) from the pooled gradient de [n,32,32,co], the forward pass's argmax bytes and the layer's
// HWIO weights [9][3][co] (the tensors of cgs_gen_conv3x3_bwd_data(hw 64, co, ci 3, dy_argmax) with the raw weights instead of the packed operand).
extern "C" int cgs_gen_enc0_bwd_data(int32_t n, int32_t co, const float* de, const uint8_t* am, const float* w_hwio, float* dx, cgs_stream_t stream) {
    if (n < 0 || !de || !am || !w_hwio || !dx) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    const GEnc0DgParams P{de, am, w_hwio, dx, n, 0};
    switch (co) {
        case 16: return genc0_dgrad_launch<4>(P, (hipStream_t)stream);
        case 24: return genc0_dgrad_launch<6>(P, (hipStream_t)stream);
        case 32: return genc0_dgrad_launch<8>(P, (hipStream_t)stream);
        case 40: return genc0_dgrad_launch<10>(P, (hipStream_t)stream);
    }
    return CGS_ERR_UNSUPPORTED;
}

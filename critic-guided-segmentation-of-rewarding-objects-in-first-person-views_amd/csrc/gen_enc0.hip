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

// NG: output groups of 4 channels (co = 4 NG); GP: groups per pass (NG % GP == 0)
template <int NG, int GP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) genc0_fwd_kernel(GEnc0FwdParams P) {
    constexpr int CO = 4 * NG, NK = 27 * NG, NV = (NK + 15) / 16, NPASS = NG / GP;
    constexpr int TW = 66, TROWS = 18;                      // input tile: 16 rows + halo, 64 columns + halo, float4 (r, g, b, 0) per pixel
    constexpr int PITCH = CO + 4;                           // floats per cell in the output block (pitch = 4 mod 8 quads: conflict-free b128)
    static_assert(NG % GP == 0, "passes of equal size");
    extern __shared__ __attribute__((aligned(16))) float4 gsm[];
    float4* const tile = gsm;                               // [TROWS][TW]
    float* const bias_s = (float*)(gsm + TROWS * TW);       // [CO]
    float* const ob = bias_s + CO + (threadIdx.x >> 6) * (64 * PITCH + 64 * NG);      // this wave's output block: values [64][PITCH], then codes [64][NG]
    uint32_t* const cb = (uint32_t*)(ob + 64 * PITCH);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- weights -> registers: register v, lane 4 b + i = combination q = 16 v + b = (k-step ks = q / NG, group g = q % NG), output channel 4 g + i ----
    float wreg[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int q = 16 * v + (lane >> 2), ks = q / NG, g = q % NG;
        wreg[v] = q < NK ? P.w[ks * CO + 4 * g + (lane & 3)] : 0.f;
    }
    for (int e = tid; e < CO; e += 256) bias_s[e] = P.bias[e];
    for (int e = tid; e < TROWS * 2; e += 256) tile[(e >> 1) * TW + ((e & 1) ? TW - 1 : 0)] = f4zero();      // halo columns: zero for every strip
    __syncthreads();

    for (int strip = blockIdx.x; strip < P.nstrips; strip += gridDim.x) {
        const int img = strip >> 2, row0 = (strip & 3) * 16;
        // ---- stage the tile: items of 4 pixels (12 bytes / 12 floats) ----
        for (int e = tid; e < TROWS * 16; e += 256) {
            const int g4 = e & 15, r = e >> 4, y = row0 + r - 1;
            const bool in = y >= 0 && y < 64;
            const size_t gi = in ? (((size_t)img * 64 + y) * 64 + g4 * 4) * 3 / 4 : 0;
            float f[12];
            if (P.a_is_u8) {
                const uint32_t* su = (const uint32_t*)P.a;
                const uint32_t d[3] = {su[gi], su[gi + 1], su[gi + 2]};
#pragma unroll
                for (int j = 0; j < 12; ++j) f[j] = (float)((d[j >> 2] >> (8 * (j & 3))) & 255u) * (1.f / 255.f);
            } else {
                const float4* sf = (const float4*)P.a;
                const float4 f0 = sf[gi], f1 = sf[gi + 1], f2 = sf[gi + 2];
                f[0] = f0.x; f[1] = f0.y; f[2] = f0.z; f[3] = f0.w; f[4] = f1.x; f[5] = f1.y; f[6] = f1.z; f[7] = f1.w;
                f[8] = f2.x; f[9] = f2.y; f[10] = f2.z; f[11] = f2.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                tile[r * TW + 1 + 4 * g4 + j] = in ? make_float4(f[3 * j], f[3 * j + 1], f[3 * j + 2], 0.f) : f4zero();
        }
        __syncthreads();

        // ---- this wave's band: rows 4 wave .. + 3 of the strip; lane = cell (cy = lane >> 5, cx = lane & 31) ----
        const int cy = lane >> 5, cx = lane & 31;
        float4 pt[4][4];                                    // input patch rows 2 cy - 1 .. + 2, columns 2 cx - 1 .. + 2 (tile coordinates: + 1)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) pt[r][c] = tile[(4 * wave + 2 * cy + r) * TW + 2 * cx + c];
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
                    const float v0 = fmaxf(acc[0][g][i] + ba[i], 0.f), v1 = fmaxf(acc[1][g][i] + ba[i], 0.f), v2 = fmaxf(acc[2][g][i] + ba[i], 0.f),
                                v3 = fmaxf(acc[3][g][i] + ba[i], 0.f);
                    const float mm = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
                    const uint32_t cd = v0 == mm ? 0u : (v1 == mm ? 1u : (v2 == mm ? 2u : 3u));       // first position holding the maximum
                    word |= (cd | (mm > 0.f ? 0u : 4u)) << (8 * i);
                    m[i] = mm;
                }
                *(float4*)(ob + lane * PITCH + 4 * (pass * GP + g)) = make_float4(m[0], m[1], m[2], m[3]);
                cb[lane * NG + pass * GP + g] = word;
            }
        });
        // ---- the wave's 2 pooled rows x 32 cells x CO channels leave in full lines (same wave wrote them: no workgroup barrier) ----
        __builtin_amdgcn_wave_barrier();                    // (LDS operations of one wave execute in order: the reads below see the writes above)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const size_t rowbase = (((size_t)img * 32 + row0 / 2 + 2 * wave + r) * 32) * CO;
            for (int f = lane; f < 32 * NG; f += 64) {
                const int cell = f / NG, g = f % NG;
                *(float4*)(P.out + rowbase + 4 * f) = *(const float4*)(ob + (r * 32 + cell) * PITCH + 4 * g);
                if (P.am) *(uint32_t*)(P.am + rowbase + 4 * f) = cb[(r * 32 + cell) * NG + g];
            }
        }
        __syncthreads();            // every wave is done with the tile (and its output block) before the next strip
    }
}

template <int NG, int GP>
int genc0_fwd_launch(GEnc0FwdParams P, hipStream_t st) {
    constexpr int CO = 4 * NG;
    constexpr size_t lds = (size_t)18 * 66 * 16 + (size_t)CO * 4 + 4 * ((size_t)64 * (CO + 4) + 64 * NG) * 4;
    P.nstrips = P.n * 4;
    auto k = genc0_fwd_kernel<NG, GP>;
    static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return (int)attr;
    const int per_cu = (int)((160 * 1024) / lds) < 2 ? 1 : 2;
    const int cap = 256 * per_cu, rounds = (P.nstrips + cap - 1) / cap, blocks = (P.nstrips + rounds - 1) / rounds;
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

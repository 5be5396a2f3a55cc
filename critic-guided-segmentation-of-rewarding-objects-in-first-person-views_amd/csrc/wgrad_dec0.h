// Weight gradient of dec_model.0 (cat(e0, Upsample(o1)): 16 -> 8 channels at 32x32, nets.py:480,516-517) as OUTER PRODUCTS on
// v_mfma_f32_4x4x1_16B_f32 (round 3).  Without broadcast the instruction is 16 independent 4x4 outer products: block b = pixel b
// of a 16-pixel row segment, A = four input channels of X at a tap, B = four channels of dY:
//     acc[tap][cog] (lane 4b + j, register r) += X[p_b + tap][4w + r] * dY[p_b][4 cog + j]
// Every multiply is useful: the 16x16x4 implicit GEMM this replaces (wgrad_body.h) runs with half of each 16-wide tile empty at 8
// output channels -- 80 MFMA cycles per pixel against 36 here.  Wave w owns input channels 4w .. 4w+3 (18 accumulators), all four
// waves walk every pixel of the tile; the 16 block sums are added once per persistent workgroup (DPP rotates + two shuffles).
// LDS: X tile with a 20-float pixel slot and dY tile with a 12-float slot: the 8 pixels x 4 dwords a 32-lane group reads fall on
// 32 different banks.  The next tile's global loads are issued right after the commit and fly during the MFMAs.
#pragma once
#include "conv_tile.h"

struct WDec0Params {
    const float* e0; const float* o1; const float* dy;
    float* slab;
    int n, ntiles;
};

namespace {
constexpr int kTH = 8, kH = 32, kW = 32, kStrips = kH / kTH, kTRA = kTH + 2, kPW = kW + 2;
constexpr int kXS = 20, kXRow = kPW * kXS, kYS = 12, kYRow = kW * kYS;
constexpr int kXFloats = kTRA * kXRow, kYFloats = kTH * kYRow;
constexpr int kSlab = 9 * 16 * 8 + 8;
constexpr int kWD0LdsFloats = kXFloats + kYFloats + 256 * 4;      // X tile | dY tile | bias reduction

__device__ __forceinline__ float block_sum16(float v) {      // sum over the 16 blocks (lanes with equal lane & 3)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false));     // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));     // row_ror:8
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
}  // namespace

// bid / nblocks: this workgroup's index among the persistent weight-gradient workgroups (256 threads); smem: kWD0LdsFloats floats, 16-byte
// aligned.  Runs as a kernel of its own (wgrad_dec0.hip) or as spare workgroups of a latency-bound launch (tail.hip).
__device__ __forceinline__ void wgrad_dec0_body(const WDec0Params& P, int bid, int nblocks, float* smem) {
    float* xt = smem;
    float* yt = smem + kXFloats;
    float* bred = yt + kYFloats;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = lane >> 2, li = lane & 3;

    for (int e = tid; e < kTRA * 2 * 5; e += 256) {              // zero halo columns (slots 0 and W + 1), written once
        const int q = e % 5, side = (e / 5) & 1, r = e / 10;
        *(float4*)(xt + r * kXRow + (side ? kW + 1 : 0) * kXS + 4 * q) = f4zero();
    }

    // ---- fetch (global -> registers) / commit (registers -> LDS) ----
    constexpr int NE = kTRA * kW * 2, ITE = (NE + 255) / 256;   // e0: float4 (plane p of pixel (r, x))
    constexpr int NO = 6 * 16 * 2, ITO = (NO + 255) / 256;      // o1: the 6 low-resolution rows under the tile
    constexpr int ND = kTH * kW * 2, ITD = ND / 256;            // dY
    float4 re[ITE], ro[ITO], rd[ITD];
    auto fetch = [&](int tile) {
        const int n = tile / kStrips, row0 = (tile % kStrips) * kTH;
#pragma unroll
        for (int it = 0; it < ITE; ++it) {
            int e = tid + 256 * it; e = e < NE ? e : NE - 1;
            const int p = e & 1, x = (e >> 1) % kW, r = e / (2 * kW), y = row0 + r - 1;
            const bool in = y >= 0 && y < kH;
            re[it] = ((const float4*)P.e0)[in ? ((size_t)(n * kH + y) * kW + x) * 2 + p : 0];
        }
#pragma unroll
        for (int it = 0; it < ITO; ++it) {
            int e = tid + 256 * it; e = e < NO ? e : NO - 1;
            const int p = e & 1, sx = (e >> 1) % 16, sr = e / 32, sy = row0 / 2 - 1 + sr;
            const bool in = sy >= 0 && sy < 16;
            ro[it] = ((const float4*)P.o1)[in ? ((size_t)(n * 16 + sy) * 16 + sx) * 2 + p : 0];
        }
#pragma unroll
        for (int it = 0; it < ITD; ++it) {
            const int e = tid + 256 * it, p = e & 1, x = (e >> 1) % kW, r = e / (2 * kW);
            rd[it] = ((const float4*)P.dy)[((size_t)(n * kH + row0 + r) * kW + x) * 2 + p];
        }
    };
    float bs[4] = {0.f, 0.f, 0.f, 0.f};       // bias gradient: this thread's dY elements (channels 4 * (tid & 1) ..)
    auto commit = [&](int tile) {
        const int row0 = (tile % kStrips) * kTH;
#pragma unroll
        for (int it = 0; it < ITE; ++it) {
            const int e = tid + 256 * it;
            if (e < NE) {
                const int p = e & 1, x = (e >> 1) % kW, r = e / (2 * kW), y = row0 + r - 1;
                *(float4*)(xt + r * kXRow + (x + 1) * kXS + 4 * p) = (y >= 0 && y < kH) ? re[it] : f4zero();
            }
        }
#pragma unroll
        for (int it = 0; it < ITO; ++it) {
            const int e = tid + 256 * it;
            if (e < NO) {
                const int p = e & 1, sx = (e >> 1) % 16, sr = e / 32, sy = row0 / 2 - 1 + sr;
                const float4 v = (sy >= 0 && sy < 16) ? ro[it] : f4zero();
#pragma unroll
                for (int d = 0; d < 4; ++d) {                    // the 2x2 pixels of the upsampled cell that lie inside the tile
                    const int r = 2 * sr - 1 + (d >> 1), x = 2 * sx + (d & 1);      // tile row of high-resolution row 2 sy + dy
                    if (r >= 0 && r < kTRA) *(float4*)(xt + r * kXRow + (x + 1) * kXS + 8 + 4 * p) = v;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < ITD; ++it) {
            const int e = tid + 256 * it, p = e & 1, x = (e >> 1) % kW, r = e / (2 * kW);
            *(float4*)(yt + r * kYRow + x * kYS + 4 * p) = rd[it];
            bs[0] += rd[it].x; bs[1] += rd[it].y; bs[2] += rd[it].z; bs[3] += rd[it].w;
        }
    };

    frag4 acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t) { acc[t][0] = frag4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = frag4{0.f, 0.f, 0.f, 0.f}; }
    const float* ap0 = xt + blk * kXS + 4 * wave + li;          // tap (0, 0) of pixel (row 0, column blk): tile row 0, slot blk
    const float* bp0 = yt + blk * kYS + li;

    int tile = bid;
    if (tile < P.ntiles) fetch(tile);
    while (tile < P.ntiles) {
        commit(tile);
        __syncthreads();
        if (tile + nblocks < P.ntiles) fetch(tile + nblocks);
        // 16 steps: rows 0..7 x the two 16-pixel halves; the operands of step s + 1 are read while the MFMAs of step s issue
        float av[2][9], bv[2][2];
        auto ld = [&](auto S, int buf) {
            constexpr int s = decltype(S)::value, y = s >> 1, x0 = 16 * (s & 1);
#pragma unroll
            for (int t = 0; t < 9; ++t) av[buf][t] = ap0[(y + t / 3) * kXRow + (x0 + t % 3) * kXS];
            bv[buf][0] = bp0[y * kYRow + x0 * kYS];
            bv[buf][1] = bp0[y * kYRow + x0 * kYS + 4];
        };
        ld(std::integral_constant<int, 0>{}, 0);
        static_for<16>([&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (s + 1 < 16) ld(std::integral_constant<int, s + 1>{}, (s + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                acc[t][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[s & 1][t], bv[s & 1][0], acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[s & 1][t], bv[s & 1][1], acc[t][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        __syncthreads();
        tile += nblocks;
    }

    // ---- slab [9*16*8 | 8]: wave w holds rows tap * 16 + 4w + r ----
    float* slab = P.slab + (size_t)bid * kSlab;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = block_sum16(acc[t][g][r]);
                if (lane < 4) slab[(t * 16 + 4 * wave + r) * 8 + 4 * g + lane] = v;
            }
#pragma unroll
    for (int c = 0; c < 4; ++c) bred[tid * 4 + c] = bs[c];
    __syncthreads();
    if (tid < 8) {                                               // channel tid: the threads with (tid & 1) == tid / 4, in a fixed order
        const int p = tid >> 2, c = tid & 3;
        float v = 0.f;
        for (int k = 0; k < 128; ++k) v += bred[(2 * k + p) * 4 + c];
        slab[9 * 16 * 8 + tid] = v;
    }
}


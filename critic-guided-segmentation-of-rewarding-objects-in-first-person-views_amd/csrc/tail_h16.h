// (round 6, BASELINE config 4: "-process inference-only, fp16 conv kernels") The tail kernels' 3x3 layers with fp16 OPERANDS on
// v_mfma_f32_16x16x16_f16, fp32 accumulation: at batch 2048 the stand-alone decoder tail is bound by the fp32 matrix pipe (SQ counters,
// profiles/r06_config4_sq_table_before_f16_tails.txt: 35.9 us at 0.60 busy -- dependent chains of v_mfma_f32_16x16x4_f32, 32 cycles each, on
// half-empty tiles), not by latency as at the training batch.  The fp32 LDS tiles, the epilogues and every output tensor stay as they are;
// only the multiply changes: a lane reads FOUR consecutive channels of its pixel at a tap (one 16-byte LDS read), converts them to halves
// (two v_cvt_pk) and one instruction covers 16 values of the flattened (tap, channel) index -- 4 x fewer matrix instructions, each a quarter
// of the fp32 instruction's cycles.  The weights are converted once per workgroup and held in registers across its images.
// Eval-mode inference only (engine.infer(fp16=True)); the training path and the fp32 inference path never instantiate these forms.
#pragma once
#include "tail4.h"

typedef _Float16 th4_t __attribute__((ext_vector_type(4)));

// B operands of a 3x3 layer over KCH tile channels per tap: instruction i covers k = 16 i .. 16 i + 15 of k = tap * KCH + c, lane (col = lane & 15,
// kq = lane >> 4) holds k = 16 i + 4 kq + j.  wf(tap, c) = the layer's weight of this lane's output column (0 when the column is padding);
// i runs over I0, I0 + ISTRIDE, ... (NI of them: a wave's share when the K range is split over the waves).
template <int KCH, int NI, int I0S, class WF>
__device__ __forceinline__ void h16_fill_w(th4_t (&w)[NI], int lane, int i0, bool col_ok, WF wf) {
    const int kq = lane >> 4;
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        const int i = i0 + n * I0S;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 16 * i + 4 * kq + j;
            const bool ok = col_ok && k < 9 * KCH;
            const int kk = ok ? k : 0;
            const float x = wf(kk / KCH, kk % KCH);      // (unconditional load of a clamped index, selected afterwards: no branch around the load)
            v[j] = ok ? x : 0.f;
        }
        w[n] = th4_t{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    }
}

// ONE such operand (instruction i, this lane): for tables built by a loop over (instruction, lane) entries
template <int KCH, class WF>
__device__ __forceinline__ th4_t h16_w1(int i, int kq, bool col_ok, WF wf) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = 16 * i + 4 * kq + j;
        const bool ok = col_ok && k < 9 * KCH;
        const int kk = ok ? k : 0;
        const float x = wf(kk / KCH, kk % KCH);
        v[j] = ok ? x : 0.f;
    }
    return th4_t{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
}

// float offset of the A read of instruction i (lane group kq) relative to tap (0, 0) of the lane's pixel (PS floats per pixel slot, PITCH per tile
// row; the four channels of a read lie in one pixel slot since KCH % 4 == 0); past the last tap: offset 0 (its weights are zero).  With KCH a
// multiple of 16 the tap does not depend on kq and everything but "+ 4 kq" folds at compile time.
template <class T, int KCH>
__device__ __forceinline__ int h16_off(int i, int kq) {
    static_assert(KCH % 4 == 0, "whole 4-channel groups");
    const int k = 16 * i + 4 * kq;
    const int kk = k < 9 * KCH ? k : 0, tap = kk / KCH, c = kk % KCH;
    return (tap / 3) * T::PITCH + (tap % 3) * T::PS + c;
}
template <class T, int KCH, int NI, int I0S>
__device__ __forceinline__ void h16_fill_off(int (&off)[NI], int lane, int i0) {
#pragma unroll
    for (int n = 0; n < NI; ++n) off[n] = h16_off<T, KCH>(i0 + n * I0S, lane >> 4);
}

// D[pixel = row 4 kq + r][col = lane & 15] += sum over this wave's NI instructions; `px` = tile address of tap (0, 0) of the pixel of A row
// lane & 15 (= pixel (y - 1, x - 1)) + the tile's first channel; offf(n) / wf(n) = the A offset / B operand of the wave's n-th instruction
// (register arrays, values folded at compile time, or an LDS table).  Two accumulation chains (even / odd instructions), added at the end.
template <int NI, class OFFF, class WFN>
__device__ __forceinline__ frag4 h16_conv_f(const float* px, OFFF offf, WFN wf, frag4 acc) {
    frag4 acc1 = frag4{0.f, 0.f, 0.f, 0.f};
    // reads in groups of four, the next group requested before this group's instructions (a wave issues in order); the scheduling barriers keep
    // the compiler from hoisting every read of the tile to the top (56 registers at NI = 14: spills)
    constexpr int GS = 4, NGR = (NI + GS - 1) / GS;
    float4 v[2][GS];
    th4_t b[2][GS];
    auto ld = [&](int g, int buf) {
#pragma unroll
        for (int j = 0; j < GS; ++j)
            if (g * GS + j < NI) { v[buf][j] = *(const float4*)(px + offf(g * GS + j)); b[buf][j] = wf(g * GS + j); }
    };
    ld(0, 0);
#pragma unroll
    for (int g = 0; g < NGR; ++g) {
        if (g + 1 < NGR) ld(g + 1, (g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            const int n = g * GS + j;
            if (n < NI) {
                const float4 f = v[g & 1][j];
                const th4_t a = th4_t{(_Float16)f.x, (_Float16)f.y, (_Float16)f.z, (_Float16)f.w};
                if (n & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b[g & 1][j], acc1, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b[g & 1][j], acc, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc + acc1;
}
template <int NI>
__device__ __forceinline__ frag4 h16_conv(const float* px, const int (&off)[NI], const th4_t (&w)[NI], frag4 acc) {
    return h16_conv_f<NI>(px, [&](int n) { return off[n]; }, [&](int n) { return w[n]; }, acc);
}

// pixel (y, x) of A row l15 of 16-pixel tile t of a W-wide map: four 2x2 quads per tile (i = 4 * quad + 2 * dy + dx), as conv_tiles
template <int W>
__device__ __forceinline__ void h16_tile_px(int t, int l15, int& y, int& x) {
    const int q = 4 * t + (l15 >> 2), qy = q / (W / 2), qx = q % (W / 2);
    y = 2 * qy + ((l15 >> 1) & 1);
    x = 2 * qx + (l15 & 1);
}

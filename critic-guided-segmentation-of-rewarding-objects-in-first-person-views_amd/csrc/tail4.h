// Building blocks of the tail kernels on v_mfma_f32_4x4x1_16B_f32 (round 3; tail_common.h holds the 16x16x4 forms they replace).
//
// Convolutions: the instruction with its A operand broadcast from one block (cbsz = 4, abid = block),
//     D[lane][r] += A[4*abid + r] * B[lane],
// with lane = PIXEL, B = the pixel's input value of one (tap, channel) step and A = four output-channel weights: no padding at
// 8 output channels (the 16x16x4 tiles of tail_common.h are half empty there), a weight REGISTER holds 16 steps (the 8 -> 8
// layers are 5 registers per 4 output channels, loaded once per workgroup), and one 16-byte LDS read feeds four steps -- the
// 16x16x4 form reads two dwords from LDS per instruction.  Consecutive groups of four lanes are the 2x2 quads of the map
// (lane = 4 * quad + 2 * dy + dx), so max-pooling and the upsample-backward sum are two DPP quad permutes.
// Weight gradients: the same instruction WITHOUT broadcast is 16 independent 4x4 outer products; block = pixel, A = 4 input
// channels at a tap, B = 4 output-gradient channels: all 256 multiplies are useful and the 16 per-block sums are added once
// per workgroup.
#pragma once
#include "tail_common.h"
#include <type_traits>
#include <utility>

// NHWC LDS tile with a one-pixel zero halo: PS floats per pixel slot, PITCH floats per row.  The pads are chosen so that the
// 16 lanes a ds_read_b128 serves together (8 neighbouring columns x 2 rows of one tap) hit 16 different 4-bank groups.
template <int H_, int W_, int C_, int PS_, int PITCH_>
struct TileP {
    static constexpr int H = H_, W = W_, C = C_, PS = PS_, PITCH = PITCH_, FLOATS = (H_ + 2) * PITCH_;
    static_assert(PS_ % 4 == 0 && PITCH_ % 4 == 0 && PITCH_ >= (W_ + 2) * PS_ && PS_ >= C_, "tile shape");
    __device__ static __forceinline__ int at(int y, int x) { return (y + 1) * PITCH + (x + 1) * PS; }
    static constexpr bool conflict_free() {
        bool used[16] = {};
        for (int dy = 0; dy < 2; ++dy)
            for (int x = 0; x < 8; ++x) {
                const int grp = ((dy * PITCH + x * PS) % 64) / 4;
                if (used[grp]) return false;
                used[grp] = true;
            }
        return true;
    }
    static_assert(conflict_free(), "b128 reads of a 16-lane group (8 columns x 2 rows) must not share banks");
};

template <class T>
__device__ __forceinline__ void tilep_zero(float* t, int tid) {
    for (int e = tid; e < T::FLOATS / 4; e += 256) ((float4*)t)[e] = f4zero();
}

template <int N, class F, int... Is>
__device__ __forceinline__ void t4_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void t4_static_for(F&& f) { t4_static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

// lane -> pixel.  16x16 map: wave w owns quad rows 2w, 2w+1 (64 pixels); 8x8 map: the 64 pixels are one wave's lanes (every
// wave of the workgroup uses the same mapping and takes a share of the output channels instead).
struct PxPos { int q, qy, qx, pos, y, x; };
__device__ __forceinline__ PxPos px16(int wave, int lane) {
    PxPos p;
    p.q = 16 * wave + (lane >> 2); p.qy = p.q >> 3; p.qx = p.q & 7; p.pos = lane & 3;
    p.y = 2 * p.qy + (p.pos >> 1); p.x = 2 * p.qx + (p.pos & 1);
    return p;
}
__device__ __forceinline__ PxPos px8(int lane) {
    PxPos p;
    p.q = lane >> 2; p.qy = p.q >> 2; p.qx = p.q & 3; p.pos = lane & 3;
    p.y = 2 * p.qy + (p.pos >> 1); p.x = 2 * p.qx + (p.pos & 1);
    return p;
}

// Weight registers: lane 4*b + i of register k of group g holds f(step 16*k + b, channel 4*g + i) (0 past NSTEP).
template <int NG, int NREG, int NSTEP, class F>
__device__ __forceinline__ void fill_wreg(float (&wreg)[NG][NREG], int lane, F f) {
    const int b = lane >> 2, i = lane & 3;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int k = 0; k < NREG; ++k) {
            const int step = 16 * k + b;
            const float v = f(step < NSTEP ? step : 0, 4 * g + i);
            wreg[g][k] = step < NSTEP ? v : 0.f;
        }
}

// acc[g] += sum over the 9 taps and the KCH channels CH0.. of the tile pixels around (y, x); the weight step of (tap, channel c
// of the tile) is tap * CIN + COFF + c  (CIN = all input channels of the layer, COFF = where this tile's channel 0 sits among
// them).  FLIP: tap t multiplies the pixel at offset -(t - center) (data gradient: flipped kernel).
template <class T, int CH0, int KCH, int CIN, int COFF, int NG, int NREG, bool FLIP = false>
__device__ __forceinline__ void conv_px(frag4 (&acc)[NG], const float* tile, int y, int x, const float (&wreg)[NG][NREG]) {
    static_assert(KCH % 4 == 0 && CH0 % 4 == 0, "whole float4 planes");
    const float* base = tile + T::at(y, x) + CH0;
    t4_static_for<9>([&](auto TAP) {
        constexpr int tap = decltype(TAP)::value, ky = tap / 3 - 1, kx = tap % 3 - 1;
        constexpr int off = FLIP ? (-ky * T::PITCH - kx * T::PS) : (ky * T::PITCH + kx * T::PS);
        t4_static_for<KCH / 4>([&](auto PL) {
            constexpr int p = decltype(PL)::value;
            const float4 v = *(const float4*)(base + off + 4 * p);
            t4_static_for<4>([&](auto CC) {
                constexpr int c = decltype(CC)::value, step = tap * CIN + COFF + CH0 + 4 * p + c, reg = step / 16, abid = step % 16;
                const float xv = f4get(v, c);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[g][reg], xv, acc[g], 4, abid, 0);
            });
        });
    });
}

// ---- quad (4 consecutive lanes) helpers on DPP quad permutes ----
__device__ __forceinline__ float dppf_xor1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float dppf_xor2(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float quad_sum(float v) {     // ((p0 + p1) + (p2 + p3)) in every lane
    v += dppf_xor1(v);
    v += dppf_xor2(v);
    return v;
}
// max over the quad of v (>= 0: post-ReLU) and the nibble of MaxPool2d's argmax: first position holding the maximum, 0xF when
// the maximum is not positive (ReLU' = 0).  Every lane of the quad gets both.
__device__ __forceinline__ float quad_pool(float v, int pos, uint32_t& nib) {
    float m = fmaxf(v, dppf_xor1(v));
    m = fmaxf(m, dppf_xor2(m));
    uint32_t code = (v == m) ? (uint32_t)pos : 4u;
    code = min(code, dpp_xor1(code));
    code = min(code, dpp_xor2(code));
    nib = m > 0.f ? code : 15u;
    return m;
}

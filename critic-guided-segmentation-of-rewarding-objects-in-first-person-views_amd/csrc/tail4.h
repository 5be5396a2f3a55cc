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
template <class T, int CH0, int KCH, int CIN, int COFF, int NG, int NREG, bool FLIP = false, bool ALLOW_2WAY = false>
__device__ __forceinline__ void conv_px(frag4 (&acc)[NG], const float* tile, int y, int x, const float (&wreg)[NG][NREG]) {
    static_assert(KCH % 4 == 0 && CH0 % 4 == 0, "whole float4 planes");
    // (16-channel tiles have no conflict-free pitch at a 16-float slot; a 20-float slot would cost 10 KB of LDS: 2-way is accepted there)
    static_assert(ALLOW_2WAY || T::conflict_free(), "b128 reads of a 16-lane group (8 columns x 2 rows) must not share banks");
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

// ---- weight gradient as outer products (v_mfma_f32_4x4x1 without broadcast): block b of a step = pixel (row, column b) of a
// 16-wide map (one row per step), A = 4 input channels CH0.. of the X tile at a tap, B = 4 output-gradient channels of the dY tile.
// acc[tap][cog] (lane 4b + j, register r) = sum over block b's pixels of X[p + tap][CH0 + r] dY[p][4 cog + j]; the 16 blocks are
// added by wg_block_sum() when the workgroup stores its slab.  The X tile's pixel slot is padded (PS = 20 for 16 channels) so the
// 8 pixels x 4 dwords a 32-lane group reads fall on 32 different banks.
template <class TX, class TY, int CH0, int NCOG>
__device__ __forceinline__ void wgrad_outer16(frag4 (&acc)[9][NCOG], const float* xt, const float* dyt, int lane) {
    static_assert(TX::W == 16 && TY::W == 16 && TX::H == TY::H, "one map row = the 16 blocks of a step");
    const int b = lane >> 2, i = lane & 3;
    const float* ap0 = xt + TX::at(-1, b - 1) + CH0 + i;        // tap (0, 0) of row 0
    const float* bp0 = dyt + TY::at(0, b) + i;
    float av[2][9], bv[2][NCOG];
    auto ld = [&](auto ROW, int buf) {
        constexpr int row = decltype(ROW)::value;
#pragma unroll
        for (int t = 0; t < 9; ++t) av[buf][t] = ap0[(row + t / 3) * TX::PITCH + (t % 3) * TX::PS];
#pragma unroll
        for (int g = 0; g < NCOG; ++g) bv[buf][g] = bp0[row * TY::PITCH + 4 * g];
    };
    ld(std::integral_constant<int, 0>{}, 0);
    t4_static_for<TX::H>([&](auto ROW) {
        constexpr int row = decltype(ROW)::value;
        if constexpr (row + 1 < TX::H) ld(std::integral_constant<int, row + 1>{}, (row + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);       // the next row's operands are read while this row's MFMAs issue
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int g = 0; g < NCOG; ++g)
                acc[t][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[row & 1][t], bv[row & 1][g], acc[t][g], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    });
}
// sum of the 16 blocks of a wave (lanes with equal lane & 3): rotates inside a row of 16 lanes, then across the 4 rows
__device__ __forceinline__ float wg_block_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false));     // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));     // row_ror:8
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// ---- the same weight gradient with the BLOCKS as (tap, input-channel group) combinations and one pixel per step: block c of
// register G = combination 16 G + c = (tap, channels 4 cig ..), A = X[p + tap][4 cig + i] (a per-lane constant offset from the
// pixel), B = dY[p][4 cog + j] (the same in every block).  A block accumulates ITS weights over the pixels, so nothing is summed
// across blocks afterwards and the whole 16 -> 8 layer is 6 accumulators (24 registers) instead of 72; 36 of the 48 block slots
// are used (a quarter more instructions than the pixel-block form).  Pixel slot 16 floats, row pitch = 16 (mod 32): the 8 blocks a
// 32-lane group reads (2 taps x 4 channel groups) cover all 32 banks.  Wave w walks the 64 pixels of rows 4w .. 4w+3.
template <class TX, class TY, int NCIG, int NCOG>
struct WgradTapBlk {
    static constexpr int NCOMB = 9 * NCIG, NG = (NCOMB + 15) / 16;
    static_assert(TX::W == 16 && TY::W == 16 && TX::PS % 32 == 16 && TX::PITCH % 32 == 16, "bank layout");
    frag4 acc[NG][NCOG];
    int offa[NG];
    __device__ __forceinline__ void init(int lane) {
        const int b = lane >> 2, i = lane & 3;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int c = 16 * g + b, cc = c < NCOMB ? c : 0, tap = cc / NCIG, cig = cc % NCIG;
            offa[g] = (tap / 3) * TX::PITCH + (tap % 3) * TX::PS + 4 * cig + i;
#pragma unroll
            for (int k = 0; k < NCOG; ++k) acc[g][k] = frag4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // xt0 / dy0: tile addresses of pixel (row 4w - 1, column -1) of X and of pixel (row 4w, column 0) of dY
    __device__ __forceinline__ void accumulate(const float* xt0, const float* dy0, int lane) {
        const int j = lane & 3;
#pragma unroll 1
        for (int yy = 0; yy < 4; ++yy) {            // (rows as a real loop: 64 unrolled steps made the register allocator give up)
            const float* xr = xt0 + yy * TX::PITCH;
            const float* dr = dy0 + yy * TY::PITCH + j;
            float av[2][NG], bv[2][NCOG];
            auto ld = [&](auto S, int buf) {
                constexpr int x = decltype(S)::value;
#pragma unroll
                for (int g = 0; g < NG; ++g) av[buf][g] = xr[x * TX::PS + offa[g]];
#pragma unroll
                for (int k = 0; k < NCOG; ++k) bv[buf][k] = dr[x * TY::PS + 4 * k];
            };
            ld(std::integral_constant<int, 0>{}, 0);
            t4_static_for<16>([&](auto S) {
                constexpr int x = decltype(S)::value;
                if constexpr (x + 1 < 16) ld(std::integral_constant<int, x + 1>{}, (x + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int k = 0; k < NCOG; ++k)
                        acc[g][k] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[x & 1][g], bv[x & 1][k], acc[g][k], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    }
    // sum of the 4 waves through `scratch` (>= 4 * NG * NCOG * 256 floats of LDS), rows tap * CI + ci of a [9 CI][CO] slab
    __device__ __forceinline__ void reduce_store(float* slab, float* scratch, int wave, int lane, int tid) const {
        constexpr int CI = 4 * NCIG, CO = 4 * NCOG;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int k = 0; k < NCOG; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) scratch[(((wave * NG + g) * NCOG + k) * 4 + r) * 64 + lane] = acc[g][k][r];
        __syncthreads();
        for (int e = tid; e < NG * NCOG * 256; e += 256) {
            const int ln = e & 63, r = (e >> 6) & 3, k = (e >> 8) % NCOG, g = (e >> 8) / NCOG;
            const int c = 16 * g + (ln >> 2);
            if (c < NCOMB && slab) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += scratch[(((w * NG + g) * NCOG + k) * 4 + r) * 64 + ln];
                slab[((c / NCIG) * CI + 4 * (c % NCIG) + r) * CO + 4 * k + (ln & 3)] = v;
            }
        }
        __syncthreads();
    }
};

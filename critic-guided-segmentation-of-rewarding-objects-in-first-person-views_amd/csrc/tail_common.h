// Building blocks of the "tail" kernels (tail.hip): the 16x16-and-smaller layers of the Hourglass run image by image inside
// one workgroup, every intermediate in LDS, instead of one latency-bound launch per layer over the whole batch.
//
// All convolutions here are implicit GEMMs on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains):
//   forward / data gradient:  D[pixel][col] = sum_k A[pixel][k] B[k][col]   A from an NHWC LDS tile with a zero halo
//   weight gradient:          D[row][co]    = sum_pixel X[pixel][row] dY[pixel][co]
// An MFMA pixel tile is FOUR 2x2 QUADS (i = 4*quad + 2*dy + dx), so a lane's four D values (rows 4*kq .. 4*kq+3) are one quad:
// max-pooling, its argmax nibble and the nearest-upsample backward sum never leave the lane.
#pragma once
#include "cgs_common.h"

typedef float frag4 __attribute__((ext_vector_type(4)));

// NHWC LDS tile of an H x W map with a one-pixel zero halo and PCI floats per pixel
template <int H_, int W_, int PCI_>
struct Tile {
    static constexpr int H = H_, W = W_, PCI = PCI_, PW = W_ + 2, PH = H_ + 2, FLOATS = PH * PW * PCI_;
    static constexpr int PS = PCI_, PITCH = (W_ + 2) * PCI_;      // (the names TileP uses: floats per pixel slot / per tile row)
    // interior pixel (y, x) -> float offset of its channel 0
    __device__ static __forceinline__ int at(int y, int x) { return ((y + 1) * PW + (x + 1)) * PCI; }
};

template <class T>
__device__ __forceinline__ void tile_zero(float* t, int tid) {
    for (int e = tid; e < T::FLOATS / 4; e += 256) ((float4*)t)[e] = f4zero();
}

// Forward / data-gradient convolution over a tile: KCH channels (multiple of 4) starting at channel CH0 of every tile pixel,
// NCB blocks of 16 output columns.  bf(tap, c, cb) = B[k = (tap, c)][col = 16*cb + (lane & 15)] (c = this lane's channel of the
// k-step, already including lane >> 4).  epi(quad, acc) receives the lane's quad index and its NCB frag4 (positions 0..3).
// The 16-pixel tiles are dealt round-robin to the workgroup's 4 waves.
template <class T, int CH0, int KCH, int NCB, class BF, class EPI>
__device__ __forceinline__ void conv_tiles(const float* xt, BF bf, EPI epi, int wave, int lane) {
    constexpr int NT = T::H * T::W / 16, QW = T::W / 2, NS = KCH / 4;
    const int l15 = lane & 15, kq = lane >> 4;
    for (int t = wave; t < NT; t += 4) {
        const int q = 4 * t + (l15 >> 2), qy = q / QW, qx = q % QW;
        const int y = 2 * qy + ((l15 >> 1) & 1), x = 2 * qx + (l15 & 1);
        const int abase = (y * T::PW + x) * T::PCI + CH0 + kq;     // tap (0,0) = pixel (y-1, x-1) = halo coordinates (y, x)
        frag4 acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = frag4{0.f, 0.f, 0.f, 0.f};
        // operands of tap t+1 are read while the MFMAs of tap t issue (a wave issues in order); the scheduling barriers keep
        // the compiler from hoisting every read of the tile to the top (register pressure -> scratch spills)
        float a[2][NS], b[2][NS][NCB];
        auto ld = [&](int tap, int buf) {
            const int toff = ((tap / 3) * T::PW + tap % 3) * T::PCI;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                a[buf][s] = xt[abase + toff + 4 * s];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) b[buf][s][cb] = bf(tap, 4 * s + kq, cb);
            }
        };
        ld(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) ld(tap + 1, (tap + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tap & 1][s], b[tap & 1][s][cb], acc[cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        epi(4 * t + kq, acc);
    }
}

// Weight gradient of a 3x3 layer over ONE image held in LDS: X tile TX (all its PCI channels are the layer's input channels,
// CI = TX::PCI), dY tile TY (first CO channels).  Rows r = tap*CI + ci, r = 9*CI is the bias row (X = 1).  The row blocks are
// dealt to the 4 waves (block rb = wave + 4*i), each wave walks all pixels: no cross-wave reduction; the accumulators stay in
// registers across the images of a (persistent) workgroup.
template <class TX, class TY, int CO, int NRBW>
struct WgradAcc {
    static constexpr int CI = TX::PCI, ROWS = 9 * CI + 1, NRB = (ROWS + 15) / 16;
    static_assert(NRBW * 4 >= NRB, "row blocks per wave");
    frag4 acc[NRBW];
    int roff[NRBW];     // per-lane A offset of row 16*rb + (lane & 15); -1: bias row, -2: padding row
    int nblk;           // row blocks this wave really owns (wave-uniform)

    __device__ __forceinline__ void init(int wave, int lane) {
        const int l15 = lane & 15;
        nblk = (NRB - wave + 3) / 4;
#pragma unroll
        for (int i = 0; i < NRBW; ++i) {
            acc[i] = frag4{0.f, 0.f, 0.f, 0.f};
            const int rb = wave + 4 * i, r = rb * 16 + l15;
            if (rb < NRB && r < 9 * CI) {
                const int tap = r / CI, ci = r % CI;
                roff[i] = ((tap / 3) * TX::PW + tap % 3) * TX::PCI + ci;
            } else {
                roff[i] = (rb < NRB && r == 9 * CI) ? -1 : -2;
            }
        }
    }

    __device__ __forceinline__ void accumulate(const float* xt, const float* dyt, int lane) {
        static_assert(TX::H == TY::H && TX::W == TY::W, "same map");
        const int l15 = lane & 15, kq = lane >> 4;
        const int co = l15 % CO;
        // BATCH pixel steps per trip: all their LDS reads are issued before the first MFMA (a wave issues in order: a read placed
        // behind an MFMA that waits for its operands is not in flight), the barriers keep the compiler from serialising them again
        constexpr int NSTEP = TX::H * TX::W / 4, BATCH = NSTEP % 4 == 0 ? 4 : 1;
#pragma unroll 1
        for (int s0 = 0; s0 < NSTEP; s0 += BATCH) {
            float a[BATCH][NRBW], b[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int p = 4 * (s0 + u) + kq, y = p / TX::W, x = p % TX::W;
                const int pa = (y * TX::PW + x) * TX::PCI;
                b[u] = dyt[TY::at(y, x) + co];
#pragma unroll
                for (int i = 0; i < NRBW; ++i) a[u][i] = xt[pa + (roff[i] >= 0 ? roff[i] : 0)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
#pragma unroll
                for (int i = 0; i < NRBW; ++i) {
                    if (i < nblk) {
                        const float av = roff[i] >= 0 ? a[u][i] : (roff[i] == -1 ? 1.f : 0.f);
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u], acc[i], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // slab: [9*CI*CO weights (HWIO) | CO bias]
    __device__ __forceinline__ void store(float* slab, int wave, int lane) const {
        const int l15 = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int i = 0; i < NRBW; ++i) {
            const int rb = wave + 4 * i;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = rb * 16 + 4 * kq + j;
                if (rb < NRB && r < ROWS && l15 < CO) slab[r * CO + l15] = acc[i][j];
            }
        }
    }
};

// ReLU + 2x2 max-pool of a lane's quad (first maximum wins, as max_pool2d; nibble 0xF when the pooled value is <= 0)
__device__ __forceinline__ float pool_quad(const frag4& acc, float bias, uint32_t& idx) {
    float m = fmaxf(acc[0] + bias, 0.f);
    idx = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const float v = fmaxf(acc[i] + bias, 0.f);
        if (v > m) { m = v; idx = i; }
    }
    if (!(m > 0.f)) idx = 15u;
    return m;
}

// Cross-lane moves on the vector ALU (DPP), no LDS round trip as ds_bpermute / __shfl have:
//   dpp_xor1 / dpp_xor2: lane ^ 1, lane ^ 2 (quad permutes);  dpp_mirror8: lane i <-> 7 - i inside every group of 8 lanes;
//   dpp_ror8: lane ^ 8 inside every row of 16 lanes.
__device__ __forceinline__ uint32_t dpp_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t dpp_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t dpp_mirror8(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true); }
__device__ __forceinline__ float dpp_ror8(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xF, 0xF, true));
}

// OR of the 8 nibbles of channels 8*g .. 8*g+7 (lanes with consecutive lane & 7): every lane gets the packed word
__device__ __forceinline__ uint32_t pack_nibbles(uint32_t idx, int ch) {
    uint32_t w = idx << (4 * (ch & 7));
    w |= dpp_mirror8(w);
    w |= dpp_xor1(w);
    w |= dpp_xor2(w);
    return w;
}

// K-split form of the weight gradient: every wave holds ALL NRB row blocks and walks the pixel steps s = wave, wave + 4, ...
// (balanced: each wave issues steps/4 * NRB MFMAs); the four partial sums are added once, when the workgroup stores its slab.
template <class TX, class TY, int CO>
struct WgradAccK {
    static constexpr int CI = TX::PCI, ROWS = 9 * CI + 1, NRB = (ROWS + 15) / 16, NSTEP = TX::H * TX::W / 4;
    static_assert(TX::H == TY::H && TX::W == TY::W && NSTEP % 4 == 0, "same map, steps split over 4 waves");
    frag4 acc[NRB];
    int roff[NRB];

    __device__ __forceinline__ void init(int lane) {
        const int l15 = lane & 15;
#pragma unroll
        for (int i = 0; i < NRB; ++i) {
            acc[i] = frag4{0.f, 0.f, 0.f, 0.f};
            const int r = i * 16 + l15;
            if (r < 9 * CI) {
                const int tap = r / CI, ci = r % CI;
                roff[i] = ((tap / 3) * TX::PW + tap % 3) * TX::PCI + ci;
            } else {
                roff[i] = (r == 9 * CI) ? -1 : -2;
            }
        }
    }

    // NW: the pixel steps are dealt to NW waves (wave = 0 .. NW-1 of them); the others may do something else meanwhile
    template <int NW = 4>
    __device__ __forceinline__ void accumulate(const float* xt, const float* dyt, int wave, int lane) {
        static_assert(NSTEP % NW == 0, "steps split over the waves");
        const int l15 = lane & 15, kq = lane >> 4;
        const int co = l15 % CO;
        constexpr int PERW = NSTEP / NW, BATCH = PERW % 4 == 0 ? 4 : (PERW % 2 == 0 ? 2 : 1);     // see WgradAcc::accumulate
#pragma unroll 1
        for (int k0 = 0; k0 < PERW; k0 += BATCH) {
            float a[BATCH][NRB], b[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int p = 4 * (wave + NW * (k0 + u)) + kq, y = p / TX::W, x = p % TX::W;
                const int pa = (y * TX::PW + x) * TX::PCI;
                b[u] = dyt[TY::at(y, x) + co];
#pragma unroll
                for (int i = 0; i < NRB; ++i) a[u][i] = xt[pa + (roff[i] >= 0 ? roff[i] : 0)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
#pragma unroll
                for (int i = 0; i < NRB; ++i) {
                    const float av = roff[i] >= 0 ? a[u][i] : (roff[i] == -1 ? 1.f : 0.f);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u], acc[i], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // sum of the four waves' partials through `scratch` (LDS, >= 4 * NRB * 256 floats), then the slab [9*CI*CO | CO]
    __device__ __forceinline__ void reduce_store(float* slab, float* scratch, int wave, int lane, int tid) const {
#pragma unroll
        for (int i = 0; i < NRB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) scratch[((wave * NRB + i) * 4 + j) * 64 + lane] = acc[i][j];
        __syncthreads();
        for (int e = tid; e < NRB * 256; e += 256) {
            const int i = e >> 8, j = (e >> 6) & 3, ln = e & 63;
            const int r = i * 16 + 4 * (ln >> 4) + j, co = ln & 15;
            if (r < ROWS && co < CO && slab) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += scratch[((w * NRB + i) * 4 + j) * 64 + ln];
                slab[r * CO + co] = v;
            }
        }
        __syncthreads();
    }
};

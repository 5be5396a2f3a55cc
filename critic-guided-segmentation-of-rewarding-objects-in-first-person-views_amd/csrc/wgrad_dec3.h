// dec_model.3's weight gradient (Conv2d(48 -> 16, 3x3) over cat(e3, Upsample(x4)(o4)) at 4x4, nets.py:483,503-505) as a GEMM OVER THE IMAGES in a few
// persistent workgroups (round 5).  Inside tail_dec_bwd every workgroup (= image at N = 512) kept the layer's 27 row blocks in MFMA
// registers and wrote a 27.7 KB slab row of its own: 14.2 of the 40.8 MB the step's final reduction reads, 3.7 us of the decoder
// backward's per-image chain (timing experiment r05y).  Here a workgroup walks n / nblocks images with the SAME accumulator class
// (tail_common.h WgradAcc: rows = (tap, input channel) + the bias row dealt to the four waves, K = the image's 16 pixels) and writes
// one slab row at the end: 64 rows instead of 512.  Inputs per image: e3 [4,4,16], o4 [32] (saved by the forward pass) and
// d o3 [4,4,16] (written by tail_dec_bwd_kernel<., false>).  Runs as extra workgroups of a latency-bound launch (conv_wgrad.hip).
#pragma once
#include "tail_common.h"

struct Dec3WgParams {
    const float* e3; const float* o4; const float* do3;
    float* slab;          // [nblocks][9 * 48 * 16 + 16]
    int n;
};

namespace {
using WD3X = Tile<4, 4, 48>;      // cat(e3, up4(o4)) with a zero halo
using WD3Y = Tile<4, 4, 16>;      // d o3
constexpr int kWD3Slab = 9 * 48 * 16 + 16;
constexpr int kWD3LdsFloats = WD3X::FLOATS + WD3Y::FLOATS;
}  // namespace

// bid / nblocks: this workgroup's index among the rider workgroups (256 threads); smem: kWD3LdsFloats floats, 16-byte aligned.
__device__ __forceinline__ void dec3_wgrad_body(const Dec3WgParams& P, int bid, int nblocks, float* smem) {
    float* t3 = smem;
    float* dy3 = smem + WD3X::FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (tail_common.h's WgradAcc with ONE pixel step per trip instead of four: the batched form holds 4 x 7 operand registers more -- as a role of
    //  another launch it took the host from 114 to 172 registers and from four workgroups per CU to two; nothing here waits for this chain)
    constexpr int CI = WD3X::PCI, ROWS = 9 * CI + 1, NRB = (ROWS + 15) / 16, NRBW = 7;
    static_assert(NRBW * 4 >= NRB, "row blocks per wave");
    const int l15 = lane & 15, kq = lane >> 4;
    frag4 acc[NRBW];
    int roff[NRBW];
    const int nblk = (NRB - wave + 3) / 4;
#pragma unroll
    for (int i = 0; i < NRBW; ++i) {
        acc[i] = frag4{0.f, 0.f, 0.f, 0.f};
        const int rb = wave + 4 * i, r = rb * 16 + l15;
        if (rb < NRB && r < 9 * CI) roff[i] = (((r / CI) / 3) * WD3X::PW + (r / CI) % 3) * WD3X::PCI + r % CI;
        else roff[i] = (rb < NRB && r == 9 * CI) ? -1 : -2;
    }
    auto accumulate = [&]() {
#pragma unroll 1
        for (int s0 = 0; s0 < 4; ++s0) {                       // 16 pixels = 4 steps of K = 4
            const int p = 4 * s0 + kq, y = p >> 2, x = p & 3;
            const int pa = (y * WD3X::PW + x) * WD3X::PCI;
            const float b = dy3[WD3Y::at(y, x) + l15];
#pragma unroll
            for (int i = 0; i < NRBW; ++i) {
                if (i < nblk) {
                    const float av = roff[i] >= 0 ? t3[pa + roff[i]] : (roff[i] == -1 ? 1.f : 0.f);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[i], 0, 0, 0);
                }
            }
        }
    };
    tile_zero<WD3X>(t3, tid);
    tile_zero<WD3Y>(dy3, tid);
    // one float4 per thread and image: threads 0..63 = e3 (pixel, channel quad), 64..127 = d o3, 128..255 = o4's eight quads once per pixel
    // (the nearest-upsample x4 of the 1x1 map: 16 copies, read 16 times from L2 rather than broadcast through LDS)
    const int e = tid & 63, p4 = e & 3, px = (e >> 2) & 3, py = e >> 4;          // e3 / d o3 element
    const int up = tid - 128, upart = up & 7, upix = up >> 3;                     // o4 copy
    auto fetch = [&](int img) {
        if (tid < 64) return ((const float4*)P.e3)[(size_t)img * 64 + e];
        if (tid < 128) return ((const float4*)P.do3)[(size_t)img * 64 + e];
        return ((const float4*)P.o4)[(size_t)img * 8 + upart];
    };
    int img = bid;
    float4 cur = img < P.n ? fetch(img) : f4zero();
    __syncthreads();                                   // the zero halos have landed
    for (; img < P.n; img += nblocks) {
        if (tid < 64) *(float4*)(t3 + WD3X::at(py, px) + 4 * p4) = cur;
        else if (tid < 128) *(float4*)(dy3 + WD3Y::at(py, px) + 4 * p4) = cur;
        else *(float4*)(t3 + WD3X::at(upix >> 2, upix & 3) + 16 + 4 * upart) = cur;
        __syncthreads();
        const int nxt = img + nblocks;
        if (nxt < P.n) cur = fetch(nxt);               // in flight during the multiply
        accumulate();
        __syncthreads();                               // every wave is done with the tiles before the next image is committed
    }
    float* slab = P.slab + (size_t)bid * kWD3Slab;                 // [9 * 48 * 16 weights (HWIO) | 16 bias]: WgradAcc::store's layout
#pragma unroll
    for (int i = 0; i < NRBW; ++i) {
        const int rb = wave + 4 * i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = rb * 16 + 4 * kq + j;
            if (rb < NRB && r < ROWS) slab[r * 16 + l15] = acc[i][j];
        }
    }
}

// Shared pieces of the shape-generic kernels (gen.hip: forward / data gradient; gen_train.hip: weight gradient, GEMMs and the
// element-wise steps of the training pass at chfak != 1).
#pragma once
#include "tail_common.h"

constexpr int GEN_KC = 16;            // channels staged per LDS chunk
enum { GEN_SRC_F32 = 0, GEN_SRC_U8 = 1, GEN_SRC_POOLEXP = 2 };

// cat(A [ca], nearest-up_ups(B [cb])) of one layer.  A: fp32 NHWC, uint8 NHWC (/255 in the loader), or -- GEN_SRC_POOLEXP -- the
// gradient of a max-pooled, ReLU'd layer: a = dE [n,hw/2,hw/2,ca] at the pooled resolution, am [same] = argmax position 0..3 of
// the forward pass (>= 4: pooled value <= 0, no gradient); element (y,x) is dE[y/2][x/2] where am == 2 (y&1) + (x&1), else 0.
struct GenSrc {
    const void* a; const float* b; const uint8_t* am;
    int mode, ca, cb, ups;
};

__device__ __forceinline__ float gen_act(float v, int act, float slope) {
    if (act == CGS_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == CGS_ACT_LRELU) return v > 0.f ? v : slope * v;
    if (act == CGS_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

// padded channel index space of the cat: A's channels rounded up to a multiple of 4, then B's (a multiple of 4)
__device__ __forceinline__ int gen_pa4(const GenSrc& S) { return (S.ca + 3) & ~3; }
__device__ __forceinline__ int gen_real_channel(const GenSrc& S, int k) {      // -1: padding
    const int pa4 = gen_pa4(S);
    return k < pa4 ? (k < S.ca ? k : -1) : (k < pa4 + S.cb ? S.ca + (k - pa4) : -1);
}

// Stages channels [16 ch, 16 ch + 16) of rows row0 - halo .. row0 + th - 1 + halo of image img into
// tile[(th + 2 halo)][(W + 2 halo)][16].  ONLY the interior columns are written: the `halo` columns either side are zero for
// every tile -- gen_zero_halo_cols() writes them once per workgroup.  256 threads.
// The loop is VALU-bound, not latency-bound (s_memtime stamps: half of a workgroup's life, ~170 instructions per element when the
// element index was decomposed by divisions): so a tile row is W x 4 quads = a power of two of elements, a thread keeps ONE
// (column, quad) for all rows (256 / (4 W) rows per round), all index math is shifts, element offsets are 32-bit, and each source
// kind has its own branch-free body (clamped addresses, selects) so the loads of a round's BATCH rows are in flight together.
enum { GEN_K_F32V4 = 0, GEN_K_F32S = 1, GEN_K_U8 = 2, GEN_K_POOLEXP = 3 };      // A source: float4-able fp32, odd-width fp32, uint8, pooled

__device__ __forceinline__ void gen_zero_halo_cols(float* tile, int W, int th, int halo, int tid) {
    const int PW = W + 2 * halo;
    for (int e = tid; e < (th + 2 * halo) * 2 * halo * 4; e += 256) {
        const int g = e & 3, side = (e >> 2) % (2 * halo), r = (e >> 2) / (2 * halo);
        const int c = side < halo ? side : W + side;
        *(float4*)(tile + ((size_t)(r * PW + c) * 4 + g) * 4) = f4zero();
    }
}

template <int KIND, bool HASB, int BATCH>
__device__ __forceinline__ void gen_stage_impl(float* tile, const GenSrc& S, int img, int H, int W, int row0, int th, int halo,
                                               int ch, int tid) {
    const int PW = W + 2 * halo, pa4 = gen_pa4(S), cp = pa4 + S.cb, rows = th + 2 * halo;
    const int lw = __builtin_ctz(W) + 2;                         // log2 of the elements per tile row (W pixels x 4 quads)
    const int ush = S.ups == 4 ? 2 : (S.ups == 2 ? 1 : 0), HB = H >> ush, WB = W >> ush;
    const int rpi = lw >= 8 ? 1 : (256 >> lw);                   // tile rows per round of 256 threads
    const int g = tid & 3, x = (tid & ((1 << lw) - 1)) >> 2, rsub = lw >= 8 ? 0 : tid >> lw;
    const int k0 = ch * GEN_KC + 4 * g;
    const bool kok = k0 < cp, isa = k0 < pa4;
    // this thread's channel offsets (the same for every row)
    const int ka = k0 > S.ca - 4 ? S.ca - 4 : k0;               // F32V4 / POOLEXP
    int kb = k0 - pa4;
    kb = kb < 0 ? 0 : (kb > S.cb - 4 ? S.cb - 4 : kb);
    const int c0 = k0 < S.ca ? k0 : S.ca - 1, c1 = k0 + 1 < S.ca ? k0 + 1 : S.ca - 1, c2 = k0 + 2 < S.ca ? k0 + 2 : S.ca - 1,
              c3 = k0 + 3 < S.ca ? k0 + 3 : S.ca - 1;
    float* const dst0 = tile + ((size_t)(halo + x) * 4 + g) * 4;      // + r * PW * 16
#pragma unroll 1
    for (int rb = 0; rb < rows; rb += rpi * BATCH) {
        float4 raw[BATCH];                    // F32V4 / POOLEXP / B part: the float4; F32S / U8: up to 4 scalars
        [[maybe_unused]] float4 rawb[HASB && KIND != GEN_K_F32V4 ? BATCH : 1];
        [[maybe_unused]] uint32_t am[KIND == GEN_K_POOLEXP ? BATCH : 1];
#pragma unroll
        for (int it = 0; it < BATCH; ++it) {
            const int r = rb + it * rpi + rsub, y = row0 + r - halo;
            const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
            const uint32_t pix = (uint32_t)((img * H + yc) * W + x);
            const float* pb = nullptr;
            if constexpr (HASB) pb = S.b + ((uint32_t)((img * HB + (yc >> ush)) * WB + (x >> ush)) * (uint32_t)S.cb + kb);
            if constexpr (KIND == GEN_K_F32V4) {
                const float* pa = (const float*)S.a + (pix * (uint32_t)S.ca + ka);
                if constexpr (HASB) pa = isa ? pa : pb;
                raw[it] = *(const float4*)pa;
            } else if constexpr (KIND == GEN_K_POOLEXP) {
                const uint32_t pp = (uint32_t)((img * (H >> 1) + (yc >> 1)) * (W >> 1) + (x >> 1)) * (uint32_t)S.ca + ka;
                raw[it] = *(const float4*)((const float*)S.a + pp);
                am[it] = *(const uint32_t*)(S.am + pp);
            } else {
                if constexpr (KIND == GEN_K_U8) {
                    const uint8_t* s = (const uint8_t*)S.a + pix * (uint32_t)S.ca;
                    raw[it] = make_float4((float)s[c0], (float)s[c1], (float)s[c2], (float)s[c3]);
                } else {
                    const float* s = (const float*)S.a + pix * (uint32_t)S.ca;
                    raw[it] = make_float4(s[c0], s[c1], s[c2], s[c3]);
                }
                if constexpr (HASB) rawb[it] = *(const float4*)pb;
            }
        }
#pragma unroll
        for (int it = 0; it < BATCH; ++it) {
            const int r = rb + it * rpi + rsub, y = row0 + r - halo;
            if (r < rows) {
                const bool inb = kok && y >= 0 && y < H;
                float4 v = raw[it];
                if constexpr (KIND == GEN_K_POOLEXP) {
                    const uint32_t pos = (uint32_t)(((y & 1) << 1) | (x & 1));
                    v.x = (am[it] & 255u) == pos ? v.x : 0.f;
                    v.y = ((am[it] >> 8) & 255u) == pos ? v.y : 0.f;
                    v.z = ((am[it] >> 16) & 255u) == pos ? v.z : 0.f;
                    v.w = (am[it] >> 24) == pos ? v.w : 0.f;
                } else if constexpr (KIND == GEN_K_F32S || KIND == GEN_K_U8) {
                    const float sc = KIND == GEN_K_U8 ? 1.f / 255.f : 1.f;
                    v.x = k0 < S.ca ? v.x * sc : 0.f;
                    v.y = k0 + 1 < S.ca ? v.y * sc : 0.f;
                    v.z = k0 + 2 < S.ca ? v.z * sc : 0.f;
                    v.w = k0 + 3 < S.ca ? v.w * sc : 0.f;
                    if constexpr (HASB) {      // (component-wise: a select between the two ARRAYS would index them dynamically)
                        const float4 rb4 = rawb[it];
                        v.x = isa ? v.x : rb4.x; v.y = isa ? v.y : rb4.y; v.z = isa ? v.z : rb4.z; v.w = isa ? v.w : rb4.w;
                    }
                }
                *(float4*)(dst0 + (size_t)r * PW * GEN_KC) = inb ? v : f4zero();
            }
        }
    }
}

template <int BATCH>
__device__ __forceinline__ void gen_stage(float* tile, const GenSrc& S, int img, int H, int W, int row0, int th, int halo,
                                          int ch, int tid) {
    // (uniform dispatch: one specialised, branch-free body per source kind)
    if (S.mode == GEN_SRC_POOLEXP) return gen_stage_impl<GEN_K_POOLEXP, false, BATCH>(tile, S, img, H, W, row0, th, halo, ch, tid);
    if (S.cb > 0) {
        if (S.mode == GEN_SRC_U8) return gen_stage_impl<GEN_K_U8, true, BATCH>(tile, S, img, H, W, row0, th, halo, ch, tid);
        if (S.ca & 3) return gen_stage_impl<GEN_K_F32S, true, BATCH>(tile, S, img, H, W, row0, th, halo, ch, tid);
        return gen_stage_impl<GEN_K_F32V4, true, BATCH>(tile, S, img, H, W, row0, th, halo, ch, tid);
    }
    if (S.mode == GEN_SRC_U8) return gen_stage_impl<GEN_K_U8, false, BATCH>(tile, S, img, H, W, row0, th, halo, ch, tid);
    if (S.ca & 3) return gen_stage_impl<GEN_K_F32S, false, BATCH>(tile, S, img, H, W, row0, th, halo, ch, tid);
    return gen_stage_impl<GEN_K_F32V4, false, BATCH>(tile, S, img, H, W, row0, th, halo, ch, tid);
}

static inline int gen_strip_rows(int hw) {      // strips of <= 256 pixels, at least 2 rows
    int th = 256 / hw;
    if (th > hw) th = hw;
    if (th < 2) th = 2;
    return th;
}
static inline bool gen_hw_ok(int hw) { return hw == 4 || hw == 8 || hw == 16 || hw == 32 || hw == 64; }

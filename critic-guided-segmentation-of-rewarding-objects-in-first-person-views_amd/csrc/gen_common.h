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

// Stages channels [16 ch, 16 ch + 16) of rows row0 - halo .. row0 + th - 1 + halo (all columns, `halo` zero columns either side)
// of image img into tile[(th + 2 halo)][(W + 2 halo)][16].  256 threads.
__device__ __forceinline__ void gen_stage(float* tile, const GenSrc& S, int img, int H, int W, int row0, int th, int halo,
                                          int ch, int tid) {
    const int PW = W + 2 * halo, pa4 = gen_pa4(S), cp = pa4 + S.cb;
    const int ngrp = (th + 2 * halo) * PW * (GEN_KC / 4);
    const int HB = H / S.ups, WB = W / S.ups;
    for (int e = tid; e < ngrp; e += 256) {
        const int g = e & 3, px = e >> 2, c = px % PW, r = px / PW;
        const int y = row0 + r - halo, x = c - halo, k0 = ch * GEN_KC + 4 * g;
        float4 v = f4zero();
        if (y >= 0 && y < H && x >= 0 && x < W && k0 < cp) {
            if (k0 < pa4) {
                const size_t pix = ((size_t)img * H + y) * W + x;
                if (S.mode == GEN_SRC_U8) {
                    const uint8_t* s = (const uint8_t*)S.a + pix * S.ca + k0;
                    const float sc = 1.f / 255.f;
                    v.x = s[0] * sc;
                    if (k0 + 1 < S.ca) v.y = s[1] * sc;
                    if (k0 + 2 < S.ca) v.z = s[2] * sc;
                    if (k0 + 3 < S.ca) v.w = s[3] * sc;
                } else if (S.mode == GEN_SRC_POOLEXP) {        // (ca % 4 == 0)
                    const size_t pp = (((size_t)img * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1)) * S.ca + k0;
                    const float4 d = *(const float4*)((const float*)S.a + pp);
                    const uint32_t am = *(const uint32_t*)(S.am + pp);
                    const uint32_t pos = (uint32_t)(((y & 1) << 1) | (x & 1));
                    v.x = (am & 255u) == pos ? d.x : 0.f;
                    v.y = ((am >> 8) & 255u) == pos ? d.y : 0.f;
                    v.z = ((am >> 16) & 255u) == pos ? d.z : 0.f;
                    v.w = (am >> 24) == pos ? d.w : 0.f;
                } else if ((S.ca & 3) == 0) {
                    v = *(const float4*)((const float*)S.a + pix * S.ca + k0);
                } else {
                    const float* s = (const float*)S.a + pix * S.ca + k0;
                    v.x = s[0];
                    if (k0 + 1 < S.ca) v.y = s[1];
                    if (k0 + 2 < S.ca) v.z = s[2];
                    if (k0 + 3 < S.ca) v.w = s[3];
                }
            } else {
                const size_t pixb = ((size_t)img * HB + y / S.ups) * WB + x / S.ups;
                v = *(const float4*)(S.b + pixb * S.cb + (k0 - pa4));
            }
        }
        *(float4*)(tile + (size_t)px * GEN_KC + 4 * g) = v;
    }
}

static inline int gen_strip_rows(int hw) {      // strips of <= 256 pixels, at least 2 rows
    int th = 256 / hw;
    if (th > hw) th = hw;
    if (th < 2) th = 2;
    return th;
}
static inline bool gen_hw_ok(int hw) { return hw == 4 || hw == 8 || hw == 16 || hw == 32 || hw == 64; }

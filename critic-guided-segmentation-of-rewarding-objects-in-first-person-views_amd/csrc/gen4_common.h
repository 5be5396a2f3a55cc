// Tile staging shared by the lane = pixel kernels of the shape-generic family (gen4.hip: convolution; gen4w.hip: weight gradient).
#pragma once
#include "gen_common.h"
#include "tail4.h"

// geometry of a 256-pixel tile: imgs images x th rows x hw columns (hw >= 16: one image part; hw = 8: 4 images; hw = 4: 16)
struct G4Geo {
    int n, hw, lw, imgs, th;
};
// where a staged float4 (4 channels of one pixel) goes: base[g * gstride + r * rstride + (x + xoff) * xstride] (float4 units),
// g = plane (4-channel group) inside the staged window, r = tile row (imgs * (th + 2 halo) of them), x = column
struct G4Dst {
    float4* base;
    int gstride, rstride, xstride, xoff;
};

// Stages channels [kbase, kbase + 4 np) (padded cat space of S, np <= npmax <= 4 planes) of the tile's rows (with `halo` rows above and
// below each image part; the halo COLUMNS are never written: the caller zeroes them once).
// Element e = tid + 256 * round -> (plane g, column x) are the THREAD's for every round (a tile row is W << lp <= 256 elements),
// only the tile row moves: everything that depends on (g, x) alone is computed once per chunk, a round costs one row clamp and
// one multiply-add per load.  (The staging runs beside other waves' MFMAs, which own the SIMD's issue slots: its cost is its
// instruction count.)
template <int KIND, bool HASB, bool SUM = false, int BATCH = 6>
__device__ __forceinline__ void gen4_stage(const G4Dst& D, const GenSrc& S, const G4Geo& P, int halo, int img0, int row0, int kbase, int npmax,
                                           int tid, float4* bs = nullptr) {
    // BATCH loads in flight per thread (6: every load of a 16-channel chunk; a tile is <= 6 rounds of 256 elements)
    const int H = P.hw, W = P.hw, lw = P.lw;
    const int pa4 = gen_pa4(S), cp = pa4 + S.cb;
    const int rem = cp - kbase;
    int np = rem >= GEN_KC ? 4 : (rem + 3) >> 2;
    np = np < npmax ? np : npmax;
    const int lp = np == 1 ? 0 : (np == 2 ? 1 : 2);
    const int rpi = P.th + 2 * halo, rows = P.imgs * rpi;
    // tile row -> image slot: r / rpi by a multiply (exact for r < 2048, rpi <= 66); a single-image tile never divides.  (As a plain
    // `r / rpi` selected against 0 the division ran for every element of every tile: ~300 of a staging call's ~530 vector instructions.)
    const bool multi = P.imgs > 1;
    const uint32_t mrpi = 65536u / (uint32_t)rpi + 1u;
    const int rpr = 256 >> (lp + lw);                          // tile rows per round of 256 threads (>= 1)
    const int g = tid & ((1 << lp) - 1), x = (tid >> lp) & (W - 1), rsub = tid >> (lp + lw);
    const int ush = S.ups == 4 ? 2 : (S.ups == 2 ? 1 : 0), HB = H >> ush, WB = W >> ush;
    const int k0 = kbase + 4 * g;
    const bool isa = k0 < pa4, kok = k0 < cp && g < np;
    // per-thread source offsets (floats / bytes) of row 0 of image 0
    const int ka = k0 > S.ca - 4 ? S.ca - 4 : k0;
    int kb = k0 - pa4;
    kb = kb < 0 ? 0 : (kb > S.cb - 4 ? S.cb - 4 : kb);
    const uint32_t offa = (uint32_t)x * (uint32_t)S.ca, offb = (uint32_t)(x >> ush) * (uint32_t)S.cb + kb;
    const uint32_t offp = (uint32_t)(x >> 1) * (uint32_t)S.ca + ka;                       // POOLEXP: pooled map
    const uint32_t rsa = (uint32_t)W * (uint32_t)S.ca, rsb = (uint32_t)WB * (uint32_t)S.cb, rsp = (uint32_t)(W >> 1) * (uint32_t)S.ca;
    const int c0 = k0 < S.ca ? k0 : S.ca - 1, c1 = k0 + 1 < S.ca ? k0 + 1 : S.ca - 1, c2 = k0 + 2 < S.ca ? k0 + 2 : S.ca - 1,
              c3 = k0 + 3 < S.ca ? k0 + 3 : S.ca - 1;
    float4* const dst0 = D.base + g * D.gstride + (x + D.xoff) * D.xstride;
#pragma unroll 1
    for (int rb = rsub; rb < rows; rb += rpr * BATCH) {
        float4 raw[BATCH];
        [[maybe_unused]] float4 rawb[HASB && KIND != GEN_K_F32V4 ? BATCH : 1];
        [[maybe_unused]] uint32_t am[KIND == GEN_K_POOLEXP ? BATCH : 1];
#pragma unroll
        for (int it = 0; it < BATCH; ++it) {
            int r = rb + it * rpr;
            r = r < rows ? r : rows - 1;
            const int il = multi ? (int)(((uint32_t)r * mrpi) >> 16) : 0, rr = r - il * rpi;
            const int y = row0 + rr - halo, yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
            int img = img0 + il;
            img = img < P.n ? img : P.n - 1;
            const float* pb = nullptr;
            if constexpr (HASB) pb = S.b + ((uint32_t)(img * HB + (yc >> ush)) * rsb + offb);
            if constexpr (KIND == GEN_K_F32V4) {
                const float* pa = (const float*)S.a + ((uint32_t)(img * H + yc) * rsa + offa + ka);
                if constexpr (HASB) pa = isa ? pa : pb;
                raw[it] = *(const float4*)pa;
            } else if constexpr (KIND == GEN_K_POOLEXP) {
                const uint32_t pp = (uint32_t)(img * (H >> 1) + (yc >> 1)) * rsp + offp;
                raw[it] = *(const float4*)((const float*)S.a + pp);
                am[it] = *(const uint32_t*)(S.am + pp);
            } else {
                const uint32_t po = (uint32_t)(img * H + yc) * rsa + offa;
                if constexpr (KIND == GEN_K_U8) {
                    const uint8_t* sp = (const uint8_t*)S.a + po;
                    raw[it] = make_float4((float)sp[c0], (float)sp[c1], (float)sp[c2], (float)sp[c3]);
                } else {
                    const float* sp = (const float*)S.a + po;
                    raw[it] = make_float4(sp[c0], sp[c1], sp[c2], sp[c3]);
                }
                if constexpr (HASB) rawb[it] = *(const float4*)pb;
            }
        }
#pragma unroll
        for (int it = 0; it < BATCH; ++it) {
            const int r = rb + it * rpr;
            if (r < rows) {
                const int il = multi ? (int)(((uint32_t)r * mrpi) >> 16) : 0, rr = r - il * rpi;
                const int y = row0 + rr - halo;
                const bool inb = kok && y >= 0 && y < H && img0 + il < P.n;
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
                    if constexpr (HASB) {
                        const float4 rb4 = rawb[it];
                        v.x = isa ? v.x : rb4.x; v.y = isa ? v.y : rb4.y; v.z = isa ? v.z : rb4.z; v.w = isa ? v.w : rb4.w;
                    }
                }
                v = inb ? v : f4zero();
                dst0[r * D.rstride] = v;
                if constexpr (SUM) { bs->x += v.x; bs->y += v.y; bs->z += v.z; bs->w += v.w; }      // (the weight gradient's bias row)
            }
        }
    }
}

// Space-to-depth view of a full-resolution tensor src [n, 2H, 2W, cs] as a map at HALF resolution with 4 parity blocks of pbw >= cs channels each
// (pbw a multiple of 16: a 16-channel chunk lies in one block): channel k = (2 py + px) * pbw + kk is src[2 y + py][2 x + px][kk] (zero for kk >= cs).
// The loader of the data gradient towards a nearest-upsampled source (gen4_conv3x3_kernel<NG, 2>).  P = the HALF-resolution geometry.  cs % 4 == 0.
template <int BATCH = 6>
__device__ __forceinline__ void gen4_stage_s2d(const G4Dst& D, const float* src, int cs, int pbw, const G4Geo& P, int img0, int row0, int kbase, int tid) {
    const int H = P.hw, W = P.hw, lw = P.lw;
    constexpr int lp = 2;                                       // four planes: a chunk is 16 channels of one parity block
    const int rpi = P.th + 2, rows = P.imgs * rpi;
    const bool multi = P.imgs > 1;
    const uint32_t mrpi = 65536u / (uint32_t)rpi + 1u;
    const int rpr = 256 >> (lp + lw);
    const int g = tid & 3, x = (tid >> lp) & (W - 1), rsub = tid >> (lp + lw);
    const int k0 = kbase + 4 * g, par = k0 / pbw, kk = k0 - par * pbw, py = par >> 1, px = par & 1;
    const bool kok = kk < cs;
    const uint32_t offa = (uint32_t)(2 * x + px) * (uint32_t)cs + (uint32_t)(kok ? kk : 0), rs = 2u * (uint32_t)W * (uint32_t)cs;
    float4* const dst0 = D.base + g * D.gstride + (x + D.xoff) * D.xstride;
#pragma unroll 1
    for (int rb = rsub; rb < rows; rb += rpr * BATCH) {
        float4 raw[BATCH];
#pragma unroll
        for (int it = 0; it < BATCH; ++it) {
            int r = rb + it * rpr;
            r = r < rows ? r : rows - 1;
            const int il = multi ? (int)(((uint32_t)r * mrpi) >> 16) : 0, rr = r - il * rpi;
            const int y = row0 + rr - 1, yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
            int img = img0 + il;
            img = img < P.n ? img : P.n - 1;
            raw[it] = *(const float4*)(src + ((uint32_t)(img * 2 * H + 2 * yc + py) * rs + offa));
        }
#pragma unroll
        for (int it = 0; it < BATCH; ++it) {
            const int r = rb + it * rpr;
            if (r < rows) {
                const int il = multi ? (int)(((uint32_t)r * mrpi) >> 16) : 0, rr = r - il * rpi;
                const int y = row0 + rr - 1;
                const bool inb = kok && y >= 0 && y < H && img0 + il < P.n;
                dst0[r * D.rstride] = inb ? raw[it] : f4zero();
            }
        }
    }
}

template <int BATCH = 6>
__device__ __forceinline__ void gen4_stage_any(const G4Dst& D, const GenSrc& S, const G4Geo& P, int halo, int img0, int row0, int kbase, int npmax, int tid) {
    if (S.mode == GEN_SRC_POOLEXP) return gen4_stage<GEN_K_POOLEXP, false, false, BATCH>(D, S, P, halo, img0, row0, kbase, npmax, tid);
    if (S.cb > 0) {
        if (S.mode == GEN_SRC_U8) return gen4_stage<GEN_K_U8, true, false, BATCH>(D, S, P, halo, img0, row0, kbase, npmax, tid);
        if (S.ca & 3) return gen4_stage<GEN_K_F32S, true, false, BATCH>(D, S, P, halo, img0, row0, kbase, npmax, tid);
        return gen4_stage<GEN_K_F32V4, true, false, BATCH>(D, S, P, halo, img0, row0, kbase, npmax, tid);
    }
    if (S.mode == GEN_SRC_U8) return gen4_stage<GEN_K_U8, false, false, BATCH>(D, S, P, halo, img0, row0, kbase, npmax, tid);
    if (S.ca & 3) return gen4_stage<GEN_K_F32S, false, false, BATCH>(D, S, P, halo, img0, row0, kbase, npmax, tid);
    return gen4_stage<GEN_K_F32V4, false, false, BATCH>(D, S, P, halo, img0, row0, kbase, npmax, tid);
}


// The same with the staged values summed per thread (a thread stages ONE plane for all rows): dY tiles of the weight gradient.
template <int BATCH = 6>
__device__ __forceinline__ void gen4_stage_sum(const G4Dst& D, const GenSrc& S, const G4Geo& P, int halo, int img0, int row0, int kbase, int npmax, int tid,
                                               float4& bs) {
    if (S.mode == GEN_SRC_POOLEXP) return gen4_stage<GEN_K_POOLEXP, false, true, BATCH>(D, S, P, halo, img0, row0, kbase, npmax, tid, &bs);
    if (S.ca & 3) return gen4_stage<GEN_K_F32S, false, true, BATCH>(D, S, P, halo, img0, row0, kbase, npmax, tid, &bs);
    return gen4_stage<GEN_K_F32V4, false, true, BATCH>(D, S, P, halo, img0, row0, kbase, npmax, tid, &bs);
}

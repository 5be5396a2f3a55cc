// (round 6, BASELINE config 4: the -process path with fp16 conv kernels, main.py:1130-1151)  Everything between features.3 and dec_model.0 of ONE
// image in one workgroup, eval mode, fp16 operands / fp32 accumulation:
//     e1 -> features.6 (+ReLU+pool) -> features.10 (+ReLU+pool) -> features.14 -> ReLU -> crit.1 -> ReLU -> crit.4 -> Sigmoid = pred    nets.py:176-194
//     e4 -> dec_model.4 -> dec_model.3 -> dec_model.2 -> dec_model.1 = o1                                                                nets.py:501-513
// It replaces the two launches cgs_tail_enc_fwd_h16 + cgs_tail_dec_fwd_h16 of the fused fp16 inference path.  What differs from them:
//   * the three LDS tiles cat(e1, up(o2)) / cat(e2, up(o3)) / cat(e3, up4(o4)) hold HALVES and serve both halves of the hourglass -- features.6 /
//     features.10 read the skip channels of the tiles the decoder layers read later, so e2 / e3 / o4 / o3 / o2 never leave the workgroup (the two-launch
//     form wrote and re-read them) and an A operand is one 8-byte LDS read with no conversion;
//   * features.14 (256 -> 32) runs on the matrix cores too (the 16 rows of its tile are the same vector: row 0 is taken);
//   * nothing the -process path does not consume is stored: only pred [n] and o1 [n,16,16,8] (fp32, what cgs_f16_dec0_fwd reads);
//   * 31 KB of LDS per workgroup instead of 49 KB.
// v_mfma_f32_16x16x16_f16 with D[pixel = row 4 kq + r][col = lane & 15] as in tail_h16.h; weights converted once per workgroup (registers, or LDS tables
// [instruction][64 lanes] for the layers a wave multiplies once per image).
#include "tail_h16.h"

#ifndef CGS_TAIL_INFER_GS
#define CGS_TAIL_INFER_GS 2      // LDS operand reads in flight per group of hh_conv (two groups: 16 GS registers)
#endif

namespace {

// NHWC tile of halves with a one-pixel zero halo: C halves per pixel
template <int H_, int W_, int C_>
struct HTile {
    static constexpr int H = H_, W = W_, C = C_, PS = C_, PITCH = (W_ + 2) * C_, HALVES = (H_ + 2) * PITCH;
    static_assert(C_ % 4 == 0 && (HALVES * 2) % 16 == 0, "8-byte operand reads, 16-byte zeroing");
    __device__ static __forceinline__ int at(int y, int x) { return (y + 1) * PITCH + (x + 1) * PS; }
};
using HT1 = HTile<16, 16, 16>;      // cat(e1, up(o2))
using HT2 = HTile<8, 8, 24>;        // cat(e2, up(o3))
using HT3 = HTile<4, 4, 48>;        // cat(e3, up4(o4))
using HT0 = HTile<16, 32, 8>;        // ENC1: a 16-row strip of e0 (18 x 34 pixels with the halo)

template <class T>
__device__ __forceinline__ void htile_zero(_Float16* t, int tid) {
    for (int e = tid; e < T::HALVES / 8; e += 256) ((float4*)t)[e] = f4zero();
}

__device__ __forceinline__ th4_t to_h4(const float4& f) { return th4_t{(_Float16)f.x, (_Float16)f.y, (_Float16)f.z, (_Float16)f.w}; }

// D += sum over the wave's NI instructions; px = the lane's pixel at tap (0, 0) (+ its channel group where the offsets are lane-independent);
// offf(n) in halves, wf(n) = B operand.  Reads in groups of four ahead of their instructions, two accumulation chains.
template <int NI, class OFFF, class WFN>
__device__ __forceinline__ frag4 hh_conv(const _Float16* px, OFFF offf, WFN wf, frag4 acc) {
    frag4 acc1 = frag4{0.f, 0.f, 0.f, 0.f};
    constexpr int GS = CGS_TAIL_INFER_GS, NGR = (NI + GS - 1) / GS;      // (operand registers: 8 GS)
    th4_t a[2][GS], b[2][GS];
    auto ld = [&](int g, int buf) {
#pragma unroll
        for (int j = 0; j < GS; ++j)
            if (g * GS + j < NI) { a[buf][j] = *(const th4_t*)(px + offf(g * GS + j)); b[buf][j] = wf(g * GS + j); }
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
                if (n & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a[g & 1][j], b[g & 1][j], acc1, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a[g & 1][j], b[g & 1][j], acc, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc + acc1;
}

struct TailInferParams {
    cgs_tail_enc_weights we;
    cgs_tail_dec_weights wd;
    const float* e1; float* pred; float* o1;
    int n, nblocks;
    const float4* e0; const float* w3e; const float* b3e;      // ENC1: features.3 in front (e0 fp16 [n,32,32,8], one float4 per pixel)
};

#ifndef CGS_TAIL_INFER_W6_LDS
#define CGS_TAIL_INFER_W6_LDS 0   // features.6's operands in an LDS table instead of 10 registers (for the 128-register build)
#endif
#ifndef CGS_TAIL_INFER_OCC
#define CGS_TAIL_INFER_OCC 3      // workgroups per CU (waves per SIMD) the kernel is compiled for.  Measured (r06_ti*.ab.txt, config 4 at batch 2048): 3 (168
                                  // registers, no spills) 0.1578 ms; 4 (128 registers: 13 .. 28 spilled, with the operands of features.6 in LDS, the constants in
                                  // LDS and the next image's loads late) 0.1621 .. 0.1711 ms
#endif

// DEC = false: the critic alone (infer(want_mask = False)).
// ENC1: features.3 (8 -> 8 at 32x32 + ReLU + MaxPool2d(2), nets.py:173-175) of the image in front, from the fp16 e0 that cgs_f16_enc0_fwd wrote: two 16-row
// strips through a 9.6 KB LDS tile, the pooled map straight into tile 1 -- e1 is never stored and the cgs_f16_enc1_fwd launch disappears.
template <bool DEC, bool ENC1>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CGS_TAIL_INFER_OCC, CGS_TAIL_INFER_OCC))) tail_infer_h16_kernel(TailInferParams P) {
    if (CGS_KARG_PREFETCH) cgs_kernarg_prefetch<sizeof(TailInferParams)>();
    __shared__ __attribute__((aligned(16))) _Float16 t1[HT1::HALVES];
    __shared__ __attribute__((aligned(16))) _Float16 t2[HT2::HALVES];
    __shared__ __attribute__((aligned(16))) _Float16 t3[HT3::HALVES];
    __shared__ __attribute__((aligned(16))) _Float16 t0[ENC1 ? HT0::HALVES : 8];
    __shared__ __attribute__((aligned(16))) th4_t w3et[ENC1 ? 5 * 64 : 1];       // features.3's operands
    __shared__ __attribute__((aligned(16))) th4_t w10t[5 * 64];                 // features.10's operands (one tile per wave and image)
    __shared__ __attribute__((aligned(16))) th4_t w6t[CGS_TAIL_INFER_W6_LDS ? 5 * 64 : 1];
    __shared__ __attribute__((aligned(16))) th4_t w2t[DEC ? 14 * 64 : 1];       // dec_model.2's (one tile per wave and image)
    __shared__ __attribute__((aligned(16))) th4_t w1t[DEC ? 9 * 64 : 1];        // dec_model.1's (four tiles per wave and image: 18 registers otherwise)
    __shared__ float part[DEC ? 4 : 1][16][16];                                 // dec_model.3: the waves split K, partial [pixel][co]
    __shared__ float red[4][32], red1[8][32], redp[DEC ? 8 : 1][32];            // features.14's K-split partials; crit.1's / dec_model.4's
    __shared__ float b14s[32], o4s[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int o = tid & 31, kg = tid >> 5;
    int img = blockIdx.x;
    float4 pe[ENC1 ? 3 : 2] = {f4zero(), f4zero()};
    // ENC1: strip s of image im = e0 rows 16 s - 1 .. 16 s + 16 (576 pixels of 16 bytes: 2.25 per thread); rows outside the image are zeroed at the commit
    auto fetch_strip = [&](int im, int sidx) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int e = tid + 256 * r, rr = e >> 5, x = e & 31, y = 16 * sidx - 1 + rr;
            const bool in = e < 576 && y >= 0 && y < 32;
            pe[ENC1 ? r : 0] = P.e0[in ? ((size_t)im * 32 + y) * 32 + x : 0];
        }
    };
    auto commit_strip = [&](int sidx) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int e = tid + 256 * r, rr = e >> 5, x = e & 31, y = 16 * sidx - 1 + rr;
            if (e < 576) *(float4*)(t0 + rr * HT0::PITCH + (x + 1) * HT0::PS) = (y >= 0 && y < 32) ? pe[ENC1 ? r : 0] : f4zero();
        }
    };
    if constexpr (ENC1) {
        if (img < P.n) fetch_strip(img, 0);
    } else {
        if (img < P.n) { pe[0] = ((const float4*)P.e1)[(size_t)img * 512 + tid]; pe[1] = ((const float4*)P.e1)[(size_t)img * 512 + tid + 256]; }
    }

    // ---- once per workgroup, in TWO batches of loads (all at once they are 116 floats in flight per lane: spills at 128 registers): first the
    //      operand tables (36 loads -> LDS) and the tiles' zeroes (the halos stay zero: every interior element is rewritten per image), then the
    //      register operands ----
    const bool tap_hi = (kq & 2) != 0;
    const int cq8 = 4 * (kq & 1);
    // A offsets of the two 8-channel layers: instruction n covers taps 2 n (lanes kq = 0, 1) and 2 n + 1 (kq = 2, 3), channels 4 (kq & 1) ..: a select
    // between two compile-time offsets instead of ten registers
    auto off8 = [&](auto tile, int n) {
        using T = decltype(tile);
        const int ta = 2 * n, tb = 2 * n + 1;
        const int oa = (ta / 3) * T::PITCH + (ta % 3) * T::PS, ob = tb < 9 ? (tb / 3) * T::PITCH + (tb % 3) * T::PS : 0;
        return tap_hi ? ob : oa;
    };
    {
        th4_t tab10[2], tab6[2], tab3e[2], tab2[DEC ? 4 : 1], tab1[DEC ? 3 : 1];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int e = tid + 256 * r, i = e >> 6, ln = e & 63;
            if constexpr (ENC1) tab3e[r] = h16_w1<8>(i, ln >> 4, (ln & 15) < 8, [&](int tap, int c) { return P.w3e[(tap * 8 + c) * 8 + (ln & 7)]; });
            tab10[r] = h16_w1<8>(i, ln >> 4, true, [&](int tap, int c) { return P.we.w10[(tap * 8 + c) * 16 + (ln & 15)]; });
            if constexpr (CGS_TAIL_INFER_W6_LDS)
                tab6[r] = h16_w1<8>(i, ln >> 4, (ln & 15) < 8, [&](int tap, int c) { return P.we.w6[(tap * 8 + c) * 8 + (ln & 7)]; });
        }
        if constexpr (DEC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = tid + 256 * r, i = e >> 6, ln = e & 63;
                tab2[r] = h16_w1<24>(i, ln >> 4, (ln & 15) < 8, [&](int tap, int c) { return P.wd.w2[(tap * 24 + c) * 8 + (ln & 7)]; });
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int e = tid + 256 * r, i = e >> 6, ln = e & 63;
                tab1[r] = h16_w1<16>(i, ln >> 4, (ln & 15) < 8, [&](int tap, int c) { return P.wd.w1[(tap * 16 + c) * 8 + (ln & 7)]; });
            }
        }
        htile_zero<HT1>(t1, tid);
        htile_zero<HT2>(t2, tid);
        htile_zero<HT3>(t3, tid);
        if constexpr (ENC1) htile_zero<HT0>(t0, tid);
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (tid + 256 * r < 5 * 64) {
                w10t[tid + 256 * r] = tab10[r];
                if constexpr (ENC1) w3et[tid + 256 * r] = tab3e[r];
                if constexpr (CGS_TAIL_INFER_W6_LDS) w6t[tid + 256 * r] = tab6[r];
            }
        if constexpr (DEC) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (tid + 256 * r < 14 * 64) w2t[tid + 256 * r] = tab2[r];
#pragma unroll
            for (int r = 0; r < 3; ++r)
                if (tid + 256 * r < 9 * 64) w1t[tid + 256 * r] = tab1[r];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    [[maybe_unused]] th4_t w6h[CGS_TAIL_INFER_W6_LDS ? 1 : 5];
    th4_t w14h[4][2];
    [[maybe_unused]] th4_t w3h[7];
    if constexpr (!CGS_TAIL_INFER_W6_LDS)
        h16_fill_w<8, 5, 1>(w6h, lane, 0, l15 < 8, [&](int tap, int c) { return P.we.w6[(tap * 8 + c) * 8 + (l15 & 7)]; });
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            // features.14: k = 16 (4 wave + ii) + 4 kq + j of the flat NHWC index (pooled pixel, channel), column 16 cb + l15
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = P.we.w14[(size_t)(16 * (4 * wave + ii) + 4 * kq + j) * 32 + 16 * cb + l15];
            w14h[ii][cb] = th4_t{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        }
    __builtin_amdgcn_sched_barrier(0);       // (a third batch: 32 + 28 floats in flight at once spill at 128 registers)
    if constexpr (DEC) h16_fill_w<48, 7, 4>(w3h, lane, wave, true, [&](int tap, int c) { return P.wd.w3[(tap * 48 + c) * 16 + l15]; });
    float w1r[4], wpr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w1r[j] = P.we.wl1[(kg * 4 + j) * 32 + o];
        wpr[j] = DEC ? P.we.wpw[(kg * 4 + j) * 32 + o] : 0.f;
    }
    const float b6v = P.we.b6[l15 & 7], b10v = P.we.b10[l15];
    [[maybe_unused]] const float b3ev = ENC1 ? P.b3e[l15 & 7] : 0.f;
    if (tid < 32) b14s[o] = P.we.b14[o];
    const float bl1 = P.we.bl1[o], wl2 = P.we.wl2[o], bl2 = P.we.bl2[0], bpw = DEC ? P.we.bpw[o] : 0.f;
    [[maybe_unused]] const float b2v = DEC ? P.wd.b2[l15 & 7] : 0.f, b1v = DEC ? P.wd.b1[l15 & 7] : 0.f;
    [[maybe_unused]] const float b3c = DEC ? P.wd.b3[tid & 15] : 0.f;      // (dec_model.3's bias of the sum stage's thread: a load per image there sat on the chain)
    __syncthreads();

    for (; img < P.n; img += P.nblocks) {
        int lz = 0;                         // opaque zero: keeps the lane-only LDS addresses from being hoisted out of the image loop
        asm volatile("" : "+v"(lz));
        const int lane_i = lane + lz;
        if constexpr (ENC1) {
            // ---- features.3 + ReLU + pool, two 16-row strips: commit | barrier | (the other strip's loads) | 8 tiles per wave | barrier ----
#pragma unroll 1
            for (int sidx = 0; sidx < 2; ++sidx) {
                commit_strip(sidx);
                __syncthreads();
                if (sidx == 0) fetch_strip(img, 1);
                else if (img + P.nblocks < P.n) fetch_strip(img + P.nblocks, 0);
#pragma unroll 1
                for (int u = 0; u < 8; ++u) {
                    const int t = wave + 4 * u;
                    int y, x;
                    h16_tile_px<32>(t, l15, y, x);
                    const frag4 acc = hh_conv<5>(t0 + HT0::at(y - 1, x - 1) + cq8, [&](int n) { return off8(HT0{}, n); },
                                                 [&](int n) { return w3et[n * 64 + lane_i]; }, frag4{0.f, 0.f, 0.f, 0.f});
                    uint32_t idx;
                    const float m = pool_quad(acc, b3ev, idx);
                    const int q = 4 * t + kq;                  // pool window of the strip: 8 rows of 16
                    if (l15 < 8) t1[HT1::at(8 * sidx + (q >> 4), q & 15) + l15] = (_Float16)m;
                }
                __syncthreads();
            }
        } else {
        // ---- e1 -> the skip channels of tile 1 (512 float4 -> 512 x 4 halves); then the next image's loads ----
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int e = tid + 256 * it, p = e & 1, x = (e >> 1) & 15, y = e >> 5;
            *(th4_t*)(t1 + HT1::at(y, x) + 4 * p) = to_h4(pe[it]);
        }
        __syncthreads();
        auto prefetch_next = [&]() {
            if (img + P.nblocks < P.n) {
                pe[0] = ((const float4*)P.e1)[(size_t)(img + P.nblocks) * 512 + tid];
                pe[1] = ((const float4*)P.e1)[(size_t)(img + P.nblocks) * 512 + tid + 256];
            }
        };
        prefetch_next();
        }
        // ---- features.6 + ReLU + pool: four 16-pixel tiles per wave; a lane's four values are one 2x2 quad = one pooled pixel -> tile 2 ----
#pragma unroll 1
        for (int u = 0; u < 4; ++u) {
            const int t = wave + 4 * u;
            int y, x;
            h16_tile_px<16>(t, l15, y, x);
            const frag4 acc = hh_conv<5>(t1 + HT1::at(y - 1, x - 1) + cq8, [&](int n) { return off8(HT1{}, n); },
                                         [&](int n) { if constexpr (CGS_TAIL_INFER_W6_LDS) return w6t[n * 64 + lane_i]; else return w6h[n]; }, frag4{0.f, 0.f, 0.f, 0.f});
            uint32_t idx;
            const float m = pool_quad(acc, b6v, idx);
            const int q = 4 * t + kq;
            if (l15 < 8) t2[HT2::at(q >> 3, q & 7) + l15] = (_Float16)m;
        }
        __syncthreads();
        // ---- features.10 + ReLU + pool: one tile per wave, 16 output channels -> tile 3 ----
        {
            int y, x;
            h16_tile_px<8>(wave, l15, y, x);
            const frag4 acc = hh_conv<5>(t2 + HT2::at(y - 1, x - 1) + cq8, [&](int n) { return off8(HT2{}, n); }, [&](int n) { return w10t[n * 64 + lane_i]; },
                                         frag4{0.f, 0.f, 0.f, 0.f});
            uint32_t idx;
            const float m = pool_quad(acc, b10v, idx);
            const int q = 4 * wave + kq;
            t3[HT3::at(q >> 2, q & 3) + l15] = (_Float16)m;
        }
        __syncthreads();
        // ---- features.14 (256 -> 32) + ReLU: wave w takes k = 64 w .. 64 w + 63 (pooled pixels 4 w .. 4 w + 3); all 16 rows of the tile read the same
        //      vector, row 0 (lanes kq = 0, register 0) is taken ----
        {
            frag4 a0 = frag4{0.f, 0.f, 0.f, 0.f}, a1 = frag4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int pq = 4 * wave + ii;
                const th4_t a = *(const th4_t*)(t3 + HT3::at(pq >> 2, pq & 3) + 4 * kq + lz);
                a0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, w14h[ii][0], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, w14h[ii][1], a1, 0, 0, 0);
            }
            if (kq == 0) { red[wave][l15] = a0[0]; red[wave][16 + l15] = a1[0]; }
        }
        __syncthreads();
        // ---- crit.1 (32 -> 32) and the decoder's 1x1 conv of e4: thread (o, kg) sums k = 4 kg .. +3; e4[k] = ReLU(b14[k] + the four waves' partials) is
        //      formed where it is used (no barrier for an e4 vector), both layers' partial sums go out together ----
        {
            float s1 = 0.f, sp = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kg * 4 + j;
                const float ev = fmaxf(b14s[k] + ((red[0][k] + red[1][k]) + (red[2][k] + red[3][k])), 0.f);
                s1 = fmaf(ev, w1r[j], s1);
                sp = fmaf(ev, wpr[j], sp);
            }
            red1[kg][o] = s1;
            if constexpr (DEC) redp[kg][o] = sp;
            __syncthreads();
            if (tid < 32) {
                float s = bl1;
#pragma unroll
                for (int g = 0; g < 8; ++g) s += red1[g][o];
                const float h = fmaxf(s, 0.f);
                if constexpr (DEC) {
                    float so = bpw;
#pragma unroll
                    for (int g = 0; g < 8; ++g) so += redp[g][o];
                    o4s[o] = so;
                }
                float t = h * wl2;                          // crit.4 (32 -> 1) -> Sigmoid (eval mode: no Dropout)
#pragma unroll
                for (int m = 16; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
                if (o == 0) P.pred[img] = 1.f / (1.f + expf(-(t + bl2)));
            }
        }
        __syncthreads();
        if constexpr (DEC) {
            if (tid < 128) {         // up4(o4): every pixel of the 4x4 map sees the bottleneck vector (channels 16 .. 47 of tile 3)
                const int p = tid & 7, pix = tid >> 3;
                *(th4_t*)(t3 + HT3::at(pix >> 2, pix & 3) + 16 + 4 * p) = to_h4(*(const float4*)(o4s + 4 * p));
            }
            __syncthreads();
            // ---- dec_model.3: one tile, K = 9 x 48 = 27 instructions dealt to the waves (i = wave, wave + 4, ...) ----
            {
                int y, x;
                h16_tile_px<4>(0, l15, y, x);
                const frag4 acc = hh_conv<7>(t3 + HT3::at(y - 1, x - 1) + 4 * kq, [&](int n) { return h16_off<HT3, 48>(wave + 4 * n, 0); },
                                             [&](int n) { return w3h[n]; }, frag4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int j = 0; j < 4; ++j) part[wave][4 * kq + j][l15] = acc[j];
            }
            __syncthreads();
            {
                const int i = tid >> 4, co = tid & 15;                       // tile pixel i = 4 * quad + 2 * dy + dx
                const int q = i >> 2, y = 2 * (q >> 1) + ((i >> 1) & 1), x = 2 * (q & 1) + (i & 1);
                const _Float16 v = (_Float16)(((part[0][i][co] + part[1][i][co]) + (part[2][i][co] + part[3][i][co])) + b3c);
#pragma unroll
                for (int d = 0; d < 4; ++d) t2[HT2::at(2 * y + (d >> 1), 2 * x + (d & 1)) + 8 + co] = v;
            }
            __syncthreads();
            // ---- dec_model.2: one tile per wave -> the upsampled channels of tile 1 ----
            {
                int y, x;
                h16_tile_px<8>(wave, l15, y, x);
                const frag4 acc = hh_conv<14>(t2 + HT2::at(y - 1, x - 1), [&](int n) { return h16_off<HT2, 24>(n, kq); },
                                              [&](int n) { return w2t[n * 64 + lane_i]; }, frag4{0.f, 0.f, 0.f, 0.f});
                if (l15 < 8) {
                    const int q = 4 * wave + kq, qy = q >> 2, qx = q & 3;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int yy = 2 * qy + (j >> 1), xx = 2 * qx + (j & 1);
                        const _Float16 v = (_Float16)(acc[j] + b2v);
#pragma unroll
                        for (int d = 0; d < 4; ++d) t1[HT1::at(2 * yy + (d >> 1), 2 * xx + (d & 1)) + 8 + l15] = v;
                    }
                }
            }
            __syncthreads();
            // ---- dec_model.1: four tiles per wave -> o1 (fp32, memory) ----
#pragma unroll 1
            for (int u = 0; u < 4; ++u) {
                const int t = wave + 4 * u;
                int y, x;
                h16_tile_px<16>(t, l15, y, x);
                const frag4 acc = hh_conv<9>(t1 + HT1::at(y - 1, x - 1) + 4 * kq, [&](int n) { return h16_off<HT1, 16>(n, 0); },
                                             [&](int n) { return w1t[n * 64 + lane_i]; }, frag4{b1v, b1v, b1v, b1v});
                if (l15 < 8) {
                    const int q = 4 * t + kq, qy = q >> 3, qx = q & 7;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        P.o1[((size_t)img * 256 + (2 * qy + (j >> 1)) * 16 + 2 * qx + (j & 1)) * 8 + l15] = acc[j];
                }
            }
            __syncthreads();     // (tile 1's skip channels are rewritten at the top of the next image)
        }
    }
}

}  // namespace

static int tail_infer_launch(int32_t n, const cgs_tail_enc_weights* we, const cgs_tail_dec_weights* wd, const float* e1, const void* e0_f16,
                             const float* w3e, const float* b3e, float* pred, float* o1, cgs_stream_t stream) {
    if (n < 0 || !we || !pred || ((wd != nullptr) != (o1 != nullptr))) return CGS_ERR_BADARG;
    if (!we->w6 || !we->b6 || !we->w10 || !we->b10 || !we->w14 || !we->b14 || !we->wl1 || !we->bl1 || !we->wl2 || !we->bl2) return CGS_ERR_BADARG;
    if (wd && (!we->wpw || !we->bpw || !wd->w3 || !wd->b3 || !wd->w2 || !wd->b2 || !wd->w1 || !wd->b1)) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    const int cap = 256 * CGS_TAIL_INFER_OCC, blocks = n < cap ? n : cap;
    TailInferParams P{*we, wd ? *wd : cgs_tail_dec_weights{}, e1, pred, o1, n, blocks, (const float4*)e0_f16, w3e, b3e};
    const hipStream_t st = (hipStream_t)stream;
    if (e0_f16) {
        if (wd) hipLaunchKernelGGL((tail_infer_h16_kernel<true, true>), dim3(blocks), dim3(256), 0, st, P);
        else hipLaunchKernelGGL((tail_infer_h16_kernel<false, true>), dim3(blocks), dim3(256), 0, st, P);
    } else {
        if (wd) hipLaunchKernelGGL((tail_infer_h16_kernel<true, false>), dim3(blocks), dim3(256), 0, st, P);
        else hipLaunchKernelGGL((tail_infer_h16_kernel<false, false>), dim3(blocks), dim3(256), 0, st, P);
    }
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// e1 [n,16,16,8] fp32 -> pred [n] and (wd / o1 given) o1 [n,16,16,8] fp32: the launches cgs_tail_enc_fwd_h16 + cgs_tail_dec_fwd_h16 in one, nothing
// else stored.  wd = NULL / o1 = NULL: the critic alone.
extern "C" int cgs_tail_infer_h16(int32_t n, const cgs_tail_enc_weights* we, const cgs_tail_dec_weights* wd, const float* e1, float* pred,
                                  float* o1, cgs_stream_t stream) {
    if (!e1) return CGS_ERR_BADARG;
    return tail_infer_launch(n, we, wd, e1, nullptr, nullptr, nullptr, pred, o1, stream);
}

// ... with features.3 in front (cgs_f16_enc1_fwd + cgs_tail_infer_h16 in one launch): e0 fp16 [n,32,32,8] (what cgs_f16_enc0_fwd writes), w3 / b3 =
// features.3's HWIO weights [3][3][8][8] and bias (nets.py:173-175 in eval mode, fp16 operands)
extern "C" int cgs_f16_enc1_tail_infer(int32_t n, const void* e0_f16, const float* w3, const float* b3, const cgs_tail_enc_weights* we,
                                       const cgs_tail_dec_weights* wd, float* pred, float* o1, cgs_stream_t stream) {
    if (!e0_f16 || !w3 || !b3) return CGS_ERR_BADARG;
    return tail_infer_launch(n, we, wd, nullptr, e0_f16, w3, b3, pred, o1, stream);
}

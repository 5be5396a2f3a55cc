// Weight gradient of a max-pooled ReLU layer (features.0: 3 -> 8 at 64x64, features.3: 8 -> 8 at 32x32) from the SPARSE
// pre-pool gradient.  Behind MaxPool2d(2) + ReLU only ONE position of every 2x2 window carries a gradient (and none where the
// pooled value is <= 0), per output channel: the dense product  dW[tap][ci][co] = sum_p X[p + tap][ci] dY[p][co]  (what the MFMA
// form computes, 3 of 4 rows of dY being zero, half of every 16-wide tile being padding at co = 8) collapses to
//     dW[tap][ci][co] = sum_cells X[p(cell, co) + tap][ci] * dE[cell][co],       p(cell, co) = the window's argmax position
// -- a quarter of the multiply-adds, on the VALU: one thread per (pool cell, output channel) pair reads the 3x3 input patch
// at ITS argmax position from the LDS tile (float4 per pixel slot: 9 or 18 reads) and keeps the full [9][CI] block of its
// output channel in registers; the pairs of a lane are reduced by lane shuffles once per workgroup.  LDS-read-bound.
#pragma once
#include "wgrad_body.h"

// H = W = 64 (CA = 3, sources: uint8 frames, the virtual mixes, fp32 images) or 32 (CA = 8, fp32); CO = 8; 256 threads
template <int H_, int CA_, int SRC_>
struct SpCfg {
    static constexpr int H = H_, W = H_, CA = CA_, SRC = SRC_, CO = 8, THREADS = 256;
    static constexpr int S = (CA + 3) / 4;                     // float4 slots per pixel
    static constexpr int CPR = W / 2, CROWS = 128 / CPR;       // pool cells per row; cell rows per tile (128 cells = 1024 pairs)
    static constexpr int TH = 2 * CROWS, STRIPS = H / TH, TRA = TH + 2;
    static constexpr int PW = (S == 1) ? W + 4 : W + 2;        // pixel slots per row
    // row stride in dwords.  CA = 3: 274 = 2 (mod 8): the lanes of a wave read the slots of 8 neighbouring cells (8 dwords apart),
    // either pixel column of the window (+4) and either row (+274 = 18 mod 64): all 32 addresses fall into different banks for
    // each of the 3 dwords read; rows are then 8-byte aligned (reads: b64 + b32).  CA = 8: 276 = 20 (mod 64), 16-byte aligned: the b128
    // reads of a 16-lane group (2 cells x 2 columns x 2 rows) cover banks 0-3, 8-11, 16-19, 24-27 | 20-23, 28-31, 36-39, 44-47 (272 put
    // the second row on the first row's banks: 2-way conflicts).
    static constexpr int RS = (S == 1) ? 274 : 276;
    static constexpr int NACC = 9 * CA + 1;                    // + the bias row
    static_assert((H == 64 && CA == 3) || (H == 32 && CA == 8), "features.0 / features.3");
    static_assert(PW * S * 4 <= RS, "row stride");
};

// The final reduction stages every lane's [NACC] block through LDS.  features.3 (73 rows x 256 lanes = 75 KB) does it in chunks of RCH rows
// that fit the X tile's 19.9 KB (round 5): as a rider of another launch its dynamic LDS size is the WHOLE launch's, and 75 KB would cut
// every role of that launch to two workgroups per CU (measured: wgrad_enc0u8_head 24.5 -> 56.8 us with the un-chunked rider).  The sums
// and their order are unchanged.  features.0 (28 rows = 28.7 KB against an 11 KB tile) keeps the one-pass form it was tuned with.
#ifndef CGS_SPRED_CHUNKED
#define CGS_SPRED_CHUNKED 1
#endif
#ifndef CGS_SPARSE8_LEAN
#define CGS_SPARSE8_LEAN 1
#endif
#ifndef CGS_SP8_GROUP
#define CGS_SP8_GROUP 1
#endif
template <class C>
struct SpRed {
    static constexpr size_t TILE = (size_t)C::TRA * C::RS * 4 + 16, FULL = (size_t)C::NACC * 256 * 4;
    static constexpr bool CHUNKED = (CGS_SPRED_CHUNKED != 0) && (C::CA == 8) && FULL > TILE;
    static constexpr int RCH = CHUNKED ? (int)(TILE / 1024) : C::NACC;       // rows per chunk (1 KB per row)
    static constexpr size_t BYTES = CHUNKED ? TILE : (TILE > FULL ? TILE : FULL);
};
template <class C>
static constexpr size_t wgrad_sparse_lds_bytes() { return SpRed<C>::BYTES; }

// Processes the tiles seq(0), seq(1), ... seq(cnt - 1) (tile = image * STRIPS + strip) and writes ONE slab [9 CA + 1][8].
// tid_bias: 0, or an OPAQUE zero when this body is one phase of a longer kernel (tail.hip) -- every address below derives from the thread
// id, and without it the compiler computes them at the top of the kernel and keeps them in registers through the phases before.
template <class C, class SEQ>
__device__ __forceinline__ void wgrad_sparse_body_seq(const WgradParams& P, const SEQ seq, const int cnt, float* slab, float4* smem,
                                                      const int tid_bias = 0) {
    constexpr int W = C::W, H = C::H, S = C::S, PW = C::PW, TH = C::TH, CA = C::CA;
    constexpr int HP = H / 2, WP = W / 2;
    float* xt = (float*)smem;                                  // [TRA] rows of RS dwords, [PW][S] float4 slots in a row
    constexpr int RS = C::RS;
    auto slot_store = [&](int r, int slot, const float4& v) {  // (rows may be only 8-byte aligned)
        float2* q = (float2*)(xt + r * RS + 4 * slot);
        q[0] = make_float2(v.x, v.y);
        q[1] = make_float2(v.z, v.w);
    };
    const int tid = threadIdx.x + tid_bias, lane = tid & 63, wave = tid >> 6;
    const int co = tid & 7, cell0 = tid >> 3;                  // pairs of this thread: cells cell0 + 32 k, k = 0..3
    const int N = P.n;

    // the zero halo columns are written once: no tile ever stores there
    for (int e = tid; e < C::TRA * (PW - W) * S; e += 256) {
        const int s = e % S, c = (e / S) % (PW - W), r = e / (S * (PW - W));
        slot_store(r, (c == 0 ? 0 : W + c) * S + s, f4zero());
    }

    // ---- X tile: fetch (global -> registers) / commit (registers -> LDS); interior columns 1..W of rows row0-1 .. row0+TH ----
    constexpr int NG = C::TRA * W / 4;                         // CA = 3: groups of 4 pixels (12 bytes / 3 dwords)
    constexpr int ITG = (NG + 255) / 256;
    constexpr int NF = C::TRA * W * S, ITF = (NF + 255) / 256; // CA = 8 (or fp32 images): float4 elements
    // what is fetched for a tile: its X rows (raw) and its pairs (gradient at the pooled output, argmax nibble).  TWO tiles are in
    // flight: a tile's FMA phase is much shorter than a memory round trip, one tile ahead leaves the loop latency-bound.
    struct Stage {
        uint32_t ga[ITG][3], gb[ITG][3];
        float4 gz[ITG];
        float4 gf[(C::SRC == WSRC_F32) ? ITF : 1];
        float pval[4];
        uint32_t pnib[4];
    };
    Stage stg[2];
    auto fetch = [&](int tile, Stage& R) {
        auto& ga = R.ga; auto& gb = R.gb; auto& gz = R.gz; auto& gf = R.gf; auto& pval = R.pval; auto& pnib = R.pnib;
        const int n = tile / C::STRIPS, row0 = (tile % C::STRIPS) * TH;
        {
            const int crow0 = (tile % C::STRIPS) * C::CROWS;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cell = cell0 + 32 * k, cy = crow0 + cell / C::CPR, cx = cell % C::CPR;
                const int pi = (n * HP + cy) * WP + cx;
                pval[k] = P.dy[(size_t)pi * 8 + co];
                pnib[k] = (P.amask[pi] >> (4 * co)) & 15u;
            }
        }
        if constexpr (C::SRC == WSRC_F32 && CA == 8) {
#pragma unroll
            for (int it = 0; it < ITF; ++it) {
                int e = tid + 256 * it; e = e < NF ? e : NF - 1;
                const int s = e % S, x = (e / S) % W, r = e / (S * W), y = row0 + r - 1;
                const bool in = y >= 0 && y < H;
                gf[it] = ((const float4*)P.src_a)[in ? ((n * H + y) * W + x) * S + s : 0];
            }
        } else {
#pragma unroll
            for (int it = 0; it < ITG; ++it) {
                int e = tid + 256 * it; e = e < NG ? e : NG - 1;
                const int g = e % (W / 4), r = e / (W / 4), y = row0 + r - 1;
                const bool in = y >= 0 && y < H;
                int nn = n;
                if constexpr (C::SRC == WSRC_MIX) nn = n >= P.mix_n_a ? n - P.mix_n_a : n;       // mixes of A-image nn
                const int pix = in ? (nn * H + y) * W + 4 * g : 0;
                if constexpr (C::SRC == WSRC_MIX) {
                    const uint32_t* a32 = (const uint32_t*)P.mix_a + (pix * 3) / 4;
                    const uint32_t* b32 = (const uint32_t*)P.mix_b + (pix * 3) / 4;
#pragma unroll
                    for (int d = 0; d < 3; ++d) { ga[it][d] = a32[d]; gb[it][d] = b32[d]; }
                    gz[it] = *(const float4*)(P.mix_z + pix);
                } else if constexpr (C::SRC == WSRC_U8) {
                    const uint32_t* a32 = (const uint32_t*)P.src_a + (pix * 3) / 4;
#pragma unroll
                    for (int d = 0; d < 3; ++d) ga[it][d] = a32[d];
                } else {      // fp32 image, 3 channels: 12 floats = 3 float4
                    const float4* f = (const float4*)((const float*)P.src_a + (size_t)pix * 3);
                    const float4 v0 = f[0], v1 = f[1], v2 = f[2];
                    ga[it][0] = __float_as_uint(v0.x); ga[it][1] = __float_as_uint(v0.y); ga[it][2] = __float_as_uint(v0.z);
                    gb[it][0] = __float_as_uint(v0.w); gb[it][1] = __float_as_uint(v1.x); gb[it][2] = __float_as_uint(v1.y);
                    gz[it] = make_float4(v1.z, v1.w, v2.x, v2.y);
                    gf[0] = make_float4(v2.z, v2.w, 0.f, 0.f);
                }
            }
        }
    };
    auto commit = [&](int tile, const Stage& R) {
        auto& ga = R.ga; auto& gb = R.gb; auto& gz = R.gz; auto& gf = R.gf;
        const int n = tile / C::STRIPS, row0 = (tile % C::STRIPS) * TH;
        if constexpr (C::SRC == WSRC_F32 && CA == 8) {
#pragma unroll
            for (int it = 0; it < ITF; ++it) {
                const int e = tid + 256 * it;
                if (e < NF) {
                    const int s = e % S, x = (e / S) % W, r = e / (S * W), y = row0 + r - 1;
                    slot_store(r, (x + 1) * S + s, (y >= 0 && y < H) ? gf[it] : f4zero());
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < ITG; ++it) {
                const int e = tid + 256 * it;
                if (e < NG) {
                    const int g = e % (W / 4), r = e / (W / 4), y = row0 + r - 1;
                    const bool in = y >= 0 && y < H;
                    float4 px[4];
                    if constexpr (C::SRC == WSRC_F32) {
                        static_assert(ITG == 1, "fp32 images: one group of 4 pixels per thread");
                        px[0] = make_float4(__uint_as_float(ga[it][0]), __uint_as_float(ga[it][1]), __uint_as_float(ga[it][2]), 0.f);
                        px[1] = make_float4(__uint_as_float(gb[it][0]), __uint_as_float(gb[it][1]), __uint_as_float(gb[it][2]), 0.f);
                        px[2] = make_float4(gz[it].x, gz[it].y, gz[it].z, 0.f);
                        px[3] = make_float4(gz[it].w, gf[0].x, gf[0].y, 0.f);
                    } else {
                        const float sc = 1.f / 255.f;
                        const bool inj = C::SRC == WSRC_MIX && n >= P.mix_n_a;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float m[3];
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                const int byte = 3 * k + c;
                                float av = ((ga[it][byte >> 2] >> (8 * (byte & 3))) & 255u) * sc;
                                if constexpr (C::SRC == WSRC_MIX) {
                                    float bv = ((gb[it][byte >> 2] >> (8 * (byte & 3))) & 255u) * sc;
                                    if (inj) { const float t = av; av = bv; bv = t; }
                                    const float zi = f4get(gz[it], k);
                                    m[c] = av * (1.f - zi) + zi * bv;
                                } else {
                                    m[c] = av;
                                }
                            }
                            px[k] = make_float4(m[0], m[1], m[2], 0.f);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) slot_store(r, 4 * g + k + 1, in ? px[k] : f4zero());
                }
            }
        }
    };

    float acc[9][CA];
    float bsum = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < CA; ++c) acc[t][c] = 0.f;

    auto process = [&](int kt, Stage& R) {
        const int tile = seq(kt);
        float val[4];
        uint32_t nib[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { val[k] = R.pval[k]; nib[k] = R.pnib[k]; }
        commit(tile, R);
        __syncthreads();
        if (kt + 2 < cnt) fetch(seq(kt + 2), R);      // the stage is free again: two tiles ahead
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cell = cell0 + 32 * k, cyl = cell / C::CPR, cx = cell % C::CPR;
            const bool dead = nib[k] > 3u;
            const float v = dead ? 0.f : val[k];
            const int pos = dead ? 0 : (int)nib[k];
            const float* p = xt + (2 * cyl + (pos >> 1)) * RS + (2 * cx + (pos & 1)) * 4 * S;      // patch origin (tile has the halo)
            bsum += v;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float* q = p + (t / 3) * RS + (t % 3) * 4 * S + 4 * s;
                    if constexpr (CA % 4 == 0) {
                        const float4 x = *(const float4*)q;
                        acc[t][4 * s] = fmaf(x.x, v, acc[t][4 * s]);
                        acc[t][4 * s + 1] = fmaf(x.y, v, acc[t][4 * s + 1]);
                        acc[t][4 * s + 2] = fmaf(x.z, v, acc[t][4 * s + 2]);
                        acc[t][4 * s + 3] = fmaf(x.w, v, acc[t][4 * s + 3]);
                    } else {
                        const float2 x01 = *(const float2*)q;
                        const float x2 = q[2];
                        acc[t][0] = fmaf(x01.x, v, acc[t][0]);
                        acc[t][1] = fmaf(x01.y, v, acc[t][1]);
                        acc[t][2] = fmaf(x2, v, acc[t][2]);
                    }
                }
            }
        }
        __syncthreads();
    };
    if (cnt > 0) fetch(seq(0), stg[0]);
    if (cnt > 1) fetch(seq(1), stg[1]);
    for (int k = 0; k < cnt; k += 2) {
        process(k, stg[0]);
        if (k + 1 >= cnt) break;
        process(k + 1, stg[1]);
    }

    // ---- every lane's block through LDS, then thread (row, co) sums the 32 lanes of its output channel in a fixed order ----
    float* red = (float*)smem;                                 // [NACC][256], or RCH rows of it at a time (SpRed)
    constexpr int RCH = SpRed<C>::RCH;                         // (process() ended with a barrier: the X tile is free)
#pragma unroll
    for (int r0 = 0; r0 < C::NACC; r0 += RCH) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < CA; ++c) {
                const int r = t * CA + c;                      // (compile-time after unrolling: rows outside the chunk fold away)
                if (r >= r0 && r < r0 + RCH) red[(r - r0) * 256 + tid] = acc[t][c];
            }
        if (9 * CA >= r0 && 9 * CA < r0 + RCH) red[(9 * CA - r0) * 256 + tid] = bsum;
        __syncthreads();
        const int rows = (C::NACC - r0) < RCH ? (C::NACC - r0) : RCH;
        for (int i = tid; i < rows * 8; i += 256) {
            const int r = i >> 3, c = i & 7;
            const float* src = red + r * 256 + c;              // lanes c, c + 8, ...: stride 8 floats
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 32; ++j) v += src[8 * j];
            slab[r0 * 8 + i] = v;
        }
        __syncthreads();   // the LDS region is reused by the next chunk / a following stage
    }
}

// ------------------------------------------------------------------------------------------------
// features.3 (CA = 8, 32x32), register-lean form (round 5).  The general body above keeps a thread's whole [9][8] block (72 accumulators) and
// two prefetch stages in registers: compiled as a ROLE of another kernel it takes 256 VGPRs (wgrad_enc0u8_head_kernel went from 60 to 256
// registers and from five workgroups per CU to two when it hosted the role: that, not the role's own work, made it a 24 us rider).  Here a
// (pool cell, output channel) pair is shared by TWO threads, one per half of the input channels: 36 accumulators, 9 b128 LDS reads per
// half-pair (the same LDS traffic and multiply-adds in total), ONE tile ahead in flight (X rows raw in 5 float4, the 8 gradient values, the 8
// argmax nibbles packed into one dword).  Thread tid: co = tid & 7, half = (tid >> 3) & 1, cell lane = tid >> 4; cells lane + 16 k, k = 0..7.
// Reduction: rows (tap, ci) of half h come from the 128 threads of that half -> [row][16 lanes][8 co] in LDS, two chunks of <= 38 rows.
// ------------------------------------------------------------------------------------------------
struct SpNoBetween { __device__ __forceinline__ void operator()() const {} };

// NSTG = prefetch stages (tiles in flight): 1 as a role of another launch (130 VGPRs), 2 inside the tail backward kernel, where a workgroup
// has only 2 - 4 tiles and the FIRST tiles' load latency is what it waits for.  `between` runs after the first NSTG tiles' global loads have
// been issued and before anything of this body touches LDS (the caller's own LDS epilogue hides the loads' round trip); it must end with
// the LDS region free and the workgroup synchronised.
template <int NSTG = 1, class SEQ, class BETWEEN = SpNoBetween>
__device__ __forceinline__ void wgrad_sparse8_body_seq(const WgradParams& P, const SEQ seq, const int cnt, float* slab, float4* smem,
                                                       const int tid_bias = 0, const BETWEEN between = BETWEEN{}) {
    using C = SpCfg<32, 8, WSRC_F32>;
    constexpr int W = C::W, H = C::H, S = C::S, PW = C::PW, TH = C::TH, RS = C::RS, HP = H / 2, WP = W / 2;
    float* xt = (float*)smem;
    const int tid = threadIdx.x + tid_bias;
    const int co = tid & 7, half = (tid >> 3) & 1, cl = tid >> 4;
    auto slot_store = [&](int r, int slot, const float4& v) { *(float4*)(xt + r * RS + 4 * slot) = v; };      // RS = 276: rows 16-byte aligned
    constexpr int NF = C::TRA * W * S, ITF = (NF + 255) / 256;      // 1152 float4 per tile: 4.5 per thread
    struct Stage { float4 gf[ITF]; float pval[8]; uint32_t pnibs; };
    Stage stg[NSTG];
    auto fetch = [&](int tile, Stage& R) {
        const int n = tile / C::STRIPS, row0 = (tile % C::STRIPS) * TH, crow0 = (tile % C::STRIPS) * C::CROWS;
        R.pnibs = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cell = cl + 16 * k, cy = crow0 + cell / C::CPR, cx = cell % C::CPR;
            const int pi = (n * HP + cy) * WP + cx;
            R.pval[k] = P.dy[(size_t)pi * 8 + co];
            R.pnibs |= ((P.amask[pi] >> (4 * co)) & 15u) << (4 * k);
        }
#pragma unroll
        for (int it = 0; it < ITF; ++it) {
            int e = tid + 256 * it; e = e < NF ? e : NF - 1;
            const int sidx = e % S, x = (e / S) % W, r = e / (S * W), y = row0 + r - 1;
            const bool in = y >= 0 && y < H;
            R.gf[it] = ((const float4*)P.src_a)[in ? ((n * H + y) * W + x) * S + sidx : 0];
        }
    };
    auto commit = [&](int tile, const Stage& R) {
        const int row0 = (tile % C::STRIPS) * TH;
#pragma unroll
        for (int it = 0; it < ITF; ++it) {
            const int e = tid + 256 * it;
            if (e < NF) {
                const int sidx = e % S, x = (e / S) % W, r = e / (S * W), y = row0 + r - 1;
                slot_store(r, (x + 1) * S + sidx, (y >= 0 && y < H) ? R.gf[it] : f4zero());
            }
        }
    };
    float acc[9][4];
    float bsum = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = 0.f;
#pragma unroll
    for (int sg = 0; sg < NSTG; ++sg)
        if (sg < cnt) fetch(seq(sg), stg[sg]);
    between();
    for (int e = tid; e < C::TRA * (PW - W) * S; e += 256) {        // zero halo columns, written once
        const int sidx = e % S, c = (e / S) % (PW - W), r = e / (S * (PW - W));
        slot_store(r, (c == 0 ? 0 : W + c) * S + sidx, f4zero());
    }
    auto process = [&](int kt, Stage& R) {
        float val[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) val[k] = R.pval[k];
        const uint32_t nibs = R.pnibs;
        commit(seq(kt), R);
        __syncthreads();
        if (kt + NSTG < cnt) fetch(seq(kt + NSTG), R);      // the stage is free again: NSTG tiles ahead, in flight during the multiply-adds
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cell = cl + 16 * k, cyl = cell / C::CPR, cx = cell % C::CPR;
            const uint32_t nib = (nibs >> (4 * k)) & 15u;
            const bool dead = nib > 3u;
            const float v = dead ? 0.f : val[k];
            const int pos = dead ? 0 : (int)nib;
            const float* p = xt + (2 * cyl + (pos >> 1)) * RS + (2 * cx + (pos & 1)) * 4 * S + 4 * half;
            bsum += v;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float4 x = *(const float4*)(p + (t / 3) * RS + (t % 3) * 4 * S);
                acc[t][0] = fmaf(x.x, v, acc[t][0]);
                acc[t][1] = fmaf(x.y, v, acc[t][1]);
                acc[t][2] = fmaf(x.z, v, acc[t][2]);
                acc[t][3] = fmaf(x.w, v, acc[t][3]);
            }
            if ((k % CGS_SP8_GROUP) == CGS_SP8_GROUP - 1)
                __builtin_amdgcn_sched_barrier(0);         // CGS_SP8_GROUP half-pairs at a time: each one interleaved costs 36 more registers
        }
        __syncthreads();
    };
    for (int kt = 0; kt < cnt; kt += NSTG) {
#pragma unroll
        for (int sg = 0; sg < NSTG; ++sg)
            if (kt + sg < cnt) process(kt + sg, stg[sg]);
    }
    // ---- reduction: row r = tap * 8 + 4 * half + c from the 128 threads of that half; [row][cell lane 0..15][co] -> sum over the lanes ----
    float* red = (float*)smem;
    constexpr int NROW = 73, RCH = 38;                      // 38 rows x 128 floats = 19.5 KB <= the X tile's 19.9 KB
    static_assert((size_t)RCH * 128 * 4 <= (size_t)C::TRA * C::RS * 4, "reduction chunk fits the tile");
#pragma unroll
    for (int r0 = 0; r0 < NROW; r0 += RCH) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // rows of this thread: t * 8 + 4 * half + c, half = 0 / 1 -> the chunk test is per half (uniform per 8-lane group)
                const int r = t * 8 + 4 * half + c;
                if (r >= r0 && r < r0 + RCH) red[(r - r0) * 128 + cl * 8 + co] = acc[t][c];
            }
        if (half == 0 && 72 >= r0 && 72 < r0 + RCH) red[(72 - r0) * 128 + cl * 8 + co] = bsum;
        __syncthreads();
        const int rows = (NROW - r0) < RCH ? (NROW - r0) : RCH;
        for (int i = tid; i < rows * 8; i += 256) {
            const float* src = red + (i >> 3) * 128 + (i & 7);
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) v += src[8 * j];
            slab[r0 * 8 + i] = v;
        }
        __syncthreads();
    }
}

// tiles tile0, tile0 + tstride, ... < tend
template <class C>
__device__ __forceinline__ void wgrad_sparse_body(const WgradParams& P, const int tile0, const int tstride, const int tend,
                                                  float* slab, float4* smem) {
    const int cnt = tile0 < tend ? (tend - tile0 + tstride - 1) / tstride : 0;
    if constexpr (C::CA == 8 && C::SRC == WSRC_F32 && CGS_SPARSE8_LEAN) wgrad_sparse8_body_seq(P, [=](int k) { return tile0 + k * tstride; }, cnt, slab, smem);
    else wgrad_sparse_body_seq<C>(P, [=](int k) { return tile0 + k * tstride; }, cnt, slab, smem);
}

// ---- which MFMA weight-gradient configurations have a sparse form (same tiles: image x strips of 8 / 16 rows) ----
template <class CWG> struct sparse_cfg { static constexpr bool ok = false; using type = void; };
template <> struct sparse_cfg<WEnc0U8> { static constexpr bool ok = true; using type = SpCfg<64, 3, WSRC_U8>; };
template <> struct sparse_cfg<WEnc0F32> { static constexpr bool ok = true; using type = SpCfg<64, 3, WSRC_F32>; };
template <> struct sparse_cfg<WEnc0Mix> { static constexpr bool ok = true; using type = SpCfg<64, 3, WSRC_MIX>; };
template <> struct sparse_cfg<WEnc1> { static constexpr bool ok = true; using type = SpCfg<32, 8, WSRC_F32>; };

// SPARSE is a compile-time choice per kernel instance (a kernel that carries both bodies pays the registers and the LDS of the
// bigger one); the host picks the instance (CGS_WGRAD_SPARSE, default on where a sparse form exists).
template <class CWG, bool SPARSE>
__device__ __forceinline__ void wgrad_dispatch(const WgradParams& P, const int tile0, const int tstride, const int tend,
                                               float* slab, float4* smem) {
    if constexpr (SPARSE) {
        static_assert(sparse_cfg<CWG>::ok && CWG::G::THREADS == 256 && CWG::G::STRIPS == sparse_cfg<CWG>::type::STRIPS, "same tiling");
        wgrad_sparse_body<typename sparse_cfg<CWG>::type>(P, tile0, tstride, tend, slab, smem);
    } else {
        wgrad_body<CWG>(P, tile0, tstride, tend, slab, smem);
    }
}

template <class CWG, bool SPARSE>
static constexpr size_t wgrad_any_lds_bytes() {
    if constexpr (SPARSE) return wgrad_sparse_lds_bytes<typename sparse_cfg<CWG>::type>();
    else return wgrad_lds_bytes<CWG>();
}

static inline int wgrad_sparse_enabled() { return 1; }     // (round 2 kept the dense MFMA form behind an environment switch)

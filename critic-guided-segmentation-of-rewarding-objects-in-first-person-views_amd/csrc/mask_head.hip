// Mask head backward (nets.py:488-491 backward) in ONE kernel:
//   * dH = d(masker.0 output) is rebuilt from the 1-channel dzpre and the LeakyReLU mask of the saved activation h
//     (never read from memory, and written out only if the caller asks for it),
//   * masker.2 weight gradient: dW2[ky][kx][ch] = sum_p h[p][ch] * dzpre[p + (1-ky, 1-kx)] uses exactly the (h pixel,
//     dzpre neighbour) pairs the dH rebuild has in registers,
//   * masker.0 data gradient w.r.t. its low-resolution input o0, with the nearest-upsample FOLDED into the weights:
//       d_o0[q][c] = sum over the 2x2 cell of q of the 3x3 data gradient
//                  = sum_{u,v in 0..3} sum_oc dH[2q + (u-1, v-1)][oc] * W4[u][v][oc][c],
//       W4[u][v][oc][c] = sum_{a in {0,1}, ky = a+2-u in 0..2} sum_{b in {0,1}, kx = b+2-v in 0..2} W[ky][kx][3+c][oc]
//     i.e. one stride-2 4x4 convolution (16 taps per output) instead of four 3x3 ones (36 taps): 2.25x fewer MACs;
//     GEMM: M = 16 PAIRS of horizontally adjacent low-res pixels, N = 2 x 8 channels (one column block per pixel of the
//     pair, no padding columns), K = 4 x 6 window positions (the union of the two 4x4 windows) x 16 channels,
//   * masker.0 weight gradient (optional): implicit GEMM over the pixels with dH read straight from the LDS tile it was
//     rebuilt into.  Rows: 27 image (tap, channel) rows + the bias row, and for the 8 upsampled channels the FOLDED rows:
//     per pixel parity (py, px) the 9 taps over the upsampled o0 hit only a 2x2 neighbourhood (a, b) of o0 itself, so
//     4 x 8 rows per parity class are accumulated (over that class's pixels only) and unfolded at the end:
//       dW[ky][kx][3+cb] = sum_{py,px} dW4[py][px][a(py,ky)][b(px,kx)][cb],  a(0,.) = {0,1,1}, a(1,.) = {0,0,1}
//     -> 4 instead of 7 MFMAs per 4 pixels.
// Workgroup = 8 waves with two roles, one tile (TH rows of one image) apart:
//   waves 0-3 ("builders", VALU + memory): rebuild dH of tile i into xt[i&1] (+ masker.2 weight-gradient partials),
//   waves 4-7 ("matrix" waves): the two GEMMs of tile i-1 out of xt[(i-1)&1] on the matrix cores.
// Each SIMD hosts one wave of each role, so the VALU and MFMA pipes and the memory system work at the same time.
// The role split is a SCALAR branch at the top level with one loop per role (the same number of workgroup barriers on
// both sides): the register allocator then sees each role's state only inside its own loop.  With both roles in one
// loop body every array of either role is live everywhere, the kernel spills, and each scratch reload carries an
// s_waitcnt vmcnt(0) that also drains the prefetched global loads.
// The dH tile is NHWC in LDS with a pixel stride of 17 floats so the stride-2 pixel reads of a wave hit 32 banks.
#include "wgrad_body.h"   // frag4, for_elems

struct MHeadParams {
    const float* dzpre; const float* h; const float* w2; const float* w0;
    const void* img;       // W0: masker.0 direct input (NHWC u8 or f32, 3 channels)
    const float* o0;       // W0: masker.0 low-resolution input [n,32,32,8]
    float* dh;             // optional
    float* d_o0;
    float* slab2;          // WG: masker.2 weight-gradient partials, one [145] slab per workgroup
    float* slab0;          // W0: masker.0 weight-gradient partials, one [1600] slab per workgroup
    int n, ntiles;
    unsigned long long* dbg;   // debug build only (tools/mh_stamps.py): per workgroup, cycles summed over its tiles [builder: p1, wait1, p2, wait2 | matrix: p1, wait1, dgrad, wgrad, wait2]
};

#ifdef CGS_DEBUG_STAMPS
#define MH_T() __builtin_amdgcn_s_memtime()
static unsigned long long* g_mh_stamps = nullptr;
extern "C" int dbg_mask_head_stamps(unsigned long long* stamps) { g_mh_stamps = stamps; return CGS_OK; }
#else
#define MH_T() 0ull
#endif

// (round 5's MH_IMG4 / MH_CVT32 staging variants of the matrix waves -- each measured +1 us -- went out with round 6's move of the staging to the builder waves)
#ifndef MH_WACC_PK
#define MH_WACC_PK 1
#endif
#ifndef MH_APK
#define MH_APK 1       // (round 6) builder: dH's four channel sums as v_pk_fma_f32 pairs
#endif
#ifndef MH_INTERLEAVE
#define MH_INTERLEAVE 1
#endif
#ifndef MH_DGRAD4
#define MH_DGRAD4 1    // (round 6) masker.0's data gradient on v_mfma_f32_4x4x1 with lane = low-resolution pixel (no padded window positions): see mask_head_matrix
#endif
#ifndef MH_DZWIN
#define MH_DZWIN 1     // (round 6) builder: rolling 3x3 dzpre window in registers (36 instead of 100 LDS reads per tile and thread)
#endif
template <int TH, bool W0>
struct MHeadGeo {
    static constexpr int H = 64, W = 64, TRA = TH + 2, PW = W + 2, PS = 17, DZW = W + 4, DZR = TH + 4, STRIPS = H / TH;
    static constexpr int XT = TRA * PW * PS, DZ = DZR * DZW;
    static constexpr int W4P = MH_DGRAD4 ? 16 * 16 * 8 : 24 * 16 * 16;  // folded masker.0 weights: [4x4 pos][oc][8] (MH_DGRAD4) / pair-folded [4x6 pos][oc][2x8]
    static constexpr int LR = TH / 2 + 2, LC = W / 2 + 2;              // o0 tile at its own resolution
    static constexpr int XIMG = W0 ? TRA * PW * 4 : 0, XO = W0 ? LR * LC * 8 : 0;   // masker.0 inputs (image [r,g,b,0])
    // (round 6) ONE workgroup barrier per tile: the dzpre tile and masker.0's input tiles are double buffered like the dH tile, so the builder
    // waves write tile i + 1's dzpre and tile i's inputs at the END of their phase (into the buffers nobody reads in this iteration) and the
    // barrier that used to separate "dzpre tile -> LDS" from the rebuild is gone -- by the stamps the matrix waves stood 0.9 k cycles of a
    // 13.7 k tile at it while the builders staged.  134.7 -> 155.0 KB of LDS (one workgroup per CU either way).
    static constexpr int FLOATS = 2 * XT + 2 * XIMG + 2 * XO + W4P + 2 * DZ;      // every tile buffer is double buffered
    static constexpr size_t LDS = (size_t)((FLOATS + 3) / 4) * 16;
    // reduction buffers (floats into the dH tiles, used after the last tile)
    static constexpr int RED2 = 0, REDA = 1024, REDB = REDA + 4 * 32 * 16, RED_END = REDB + 4 * 4 * 32 * 16;
    static_assert(RED_END <= 2 * XT, "reduction buffers fit in the tile storage");
};

struct MHeadLds { float *xt0, *ximg, *xo, *w4p, *dz; };

// ---------------------------------------------------------------------------------------------------------------
// Builder waves (threads 0..255).  Per tile i: rebuild dH(i) from dzpre buffer i & 1, then dzpre(i + 1) and masker.0's inputs of tile i -> LDS | barrier
// ---------------------------------------------------------------------------------------------------------------
template <int TH, bool WG, bool W0, int SRC>
__device__ __forceinline__ void mask_head_builder(const MHeadParams& P, const MHeadLds& L, const int btid, const int T,
                                                  const int bid, const int grid) {
    using G = MHeadGeo<TH, W0>;
    constexpr int H = G::H, W = G::W, TRA = G::TRA, PW = G::PW, PS = G::PS, DZW = G::DZW, LR = G::LR, LC = G::LC;
    constexpr int IT = TRA * W * 4 / 256, DIT = (G::DZ + 255) / 256;
    constexpr int NPIX = TRA * PW, NLO = LR * LC * 2, ITA = (NPIX + 255) / 256, ITB = (NLO + 255) / 256;
    static_assert((TRA * W * 4) % 256 == 0, "whole iterations: no element is visited twice");
    const int lane = btid & 63, wave = btid >> 6;
    const int pl = btid & 3;           // the 4-channel plane of dH this thread builds
    static_assert(4 * W == 256, "one tile row per 256-thread item");
    const int bx = btid >> 2, bx4 = btid;      // this thread's pixel column (every item) and its float4 index within a row (= 4 bx + pl)
    auto tile_of = [&](int i) { return bid + (i < T ? i : T - 1) * grid; };   // clamped: prefetches past the end re-read

    float w2r[9][4];
#if MH_APK
    typedef float mh_f2 __attribute__((ext_vector_type(2)));
    mh_f2 w2p[9][2];
#endif
    // masker.2 weight-gradient partials of this thread (plane pl).  MH_WACC_PK: held as register PAIRS and pinned as pairs, so that their FMAs are
    // v_pk_fma_f32 (the per-scalar pins below left them as 288 v_fmac_f32 per tile; round 5)
#if MH_WACC_PK
#if !MH_APK
    typedef float mh_f2 __attribute__((ext_vector_type(2)));
#endif
    mh_f2 wacc[9][2];
#else
    float wacc[9][4];
#endif
    float bacc = 0.f;
    float4 hvs[IT];
    float dzr[DIT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            w2r[t][c] = P.w2[t * 16 + 4 * pl + c];
#if MH_APK
            w2p[t][c >> 1][c & 1] = w2r[t][c];
#endif
#if MH_WACC_PK
            wacc[t][c >> 1][c & 1] = 0.f;
#else
            wacc[t][c] = 0.f;
#endif
        }

    // loads = address arithmetic + the load only; masking happens where the value is consumed (a select here would
    // wait for the load)
    auto load_h_into = [&](float4 (&dst)[IT], int tile) {
        const int n0 = tile / G::STRIPS, row0 = (tile % G::STRIPS) * TH;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            // item `it` is tile row `it` (256 threads = 64 pixels x 4 planes = one row): the row test is uniform, the thread's part of the
            // address (bx4) the same for every item -- written so, the selects and the per-item index arithmetic are scalar (round 5)
            const int y = row0 + it - 1;
            const bool in = y >= 0 && y < H;
            dst[it] = ((const float4*)P.h)[(in ? (n0 * H + y) * W * 4 : 0) + bx4];
        }
    };
    auto load_h = [&](int tile) { load_h_into(hvs, tile); };
    auto load_dz = [&](int tile) {
        const int n0 = tile / G::STRIPS, row0 = (tile % G::STRIPS) * TH;
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            int e = btid + it * 256;
            e = e < G::DZ ? e : G::DZ - 1;
            int y = row0 + e / DZW - 2, x = e % DZW - 2;
            bool in = y >= 0 && y < H && x >= 0 && x < W;
            dzr[it] = P.dzpre[in ? (n0 * H + y) * W + x : 0];
        }
    };
    auto store_dz = [&](int tile, int buf) {
        const int row0 = (tile % G::STRIPS) * TH;
        float* const dzb = L.dz + buf * G::DZ;
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            int e = btid + it * 256;
            e = e < G::DZ ? e : G::DZ - 1;
            int y = row0 + e / DZW - 2, x = e % DZW - 2;
            bool in = y >= 0 && y < H && x >= 0 && x < W;
            dzb[e] = in ? dzr[it] : 0.f;
        }
    };

    // (round 6) masker.0's INPUT tiles (image [r,g,b,0] and o0 at its own resolution) for the matrix waves' weight-gradient GEMM are staged by
    // the builder waves: by the stamps (tools/mh_stamps.py) the matrix waves are the tile's critical path (11.8 k cycles of a 12.7 k tile, 7.2 k
    // of them matrix instructions) and the fetch + commit of these tiles were 2 k of it, while the builder waves wait a third of their phase
    // for memory.  fetch = address arithmetic + loads only (raw dwords); decoding and zero padding happen in commit, one phase later.
    [[maybe_unused]] float4 ra[W0 ? ITA : 1], rb[W0 ? ITB : 1];
    auto fetch_x = [&](int tile) {
        const int n0 = tile / G::STRIPS, row0 = (tile % G::STRIPS) * TH;
#pragma unroll
        for (int it = 0; it < ITA; ++it) {
            int e = btid + it * 256; e = e < NPIX ? e : NPIX - 1;
            int y = row0 + e / PW - 1, x = e % PW - 1;
            bool in = y >= 0 && y < H && x >= 0 && x < W;
            int pix = in ? (n0 * H + y) * W + x : 0;
            if constexpr (SRC == WSRC_U8) {
                const uint32_t* s32 = (const uint32_t*)P.img;
                int off = pix * 3, last = P.n * H * W * 3 / 4 - 1, d = off >> 2;
                ra[it].x = __uint_as_float(s32[d]);
                ra[it].y = __uint_as_float(s32[d + 1 <= last ? d + 1 : last]);
            } else {
                const float* sf = (const float*)P.img;
                ra[it] = make_float4(sf[pix * 3], sf[pix * 3 + 1], sf[pix * 3 + 2], 0.f);
            }
        }
#pragma unroll
        for (int it = 0; it < ITB; ++it) {
            int e = btid + it * 256; e = e < NLO ? e : NLO - 1;
            int half = e & 1, pc = (e >> 1) % LC, pr = (e >> 1) / LC;
            int ly = row0 / 2 + pr - 1, lx = pc - 1;
            bool in = ly >= 0 && ly < H / 2 && lx >= 0 && lx < W / 2;
            rb[it] = ((const float4*)P.o0)[in ? ((n0 * (H / 2) + ly) * (W / 2) + lx) * 2 + half : 0];
        }
    };
    auto commit_x = [&](int tile, int buf) {
        const int n0 = tile / G::STRIPS, row0 = (tile % G::STRIPS) * TH;
        float4* const ximg4 = (float4*)(L.ximg + buf * G::XIMG);
        float4* const xo4 = (float4*)(L.xo + buf * G::XO);
#pragma unroll
        for (int it = 0; it < ITA; ++it) {
            int e = btid + it * 256; e = e < NPIX ? e : NPIX - 1;
            int y = row0 + e / PW - 1, x = e % PW - 1;
            bool in = y >= 0 && y < H && x >= 0 && x < W;
            float4 v = ra[it];
            if constexpr (SRC == WSRC_U8) {
                int pix = in ? (n0 * H + y) * W + x : 0;
                const uint32_t lo = __float_as_uint(v.x), hi = __float_as_uint(v.y);
                const uint32_t b3 = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)((pix * 3) & 3));      // the pixel's 3 bytes start at byte (3 pix) & 3 of the dword pair
                const float sc = 1.f / 255.f;
                v = make_float4((float)(b3 & 255u) * sc, (float)((b3 >> 8) & 255u) * sc, (float)((b3 >> 16) & 255u) * sc, 0.f);
            }
            v.w = 0.f;
            ximg4[e] = in ? v : f4zero();
        }
#pragma unroll
        for (int it = 0; it < ITB; ++it) {
            int e = btid + it * 256; e = e < NLO ? e : NLO - 1;
            int pc = (e >> 1) % LC, pr = (e >> 1) / LC;
            int ly = row0 / 2 + pr - 1, lx = pc - 1;
            bool in = ly >= 0 && ly < H / 2 && lx >= 0 && lx < W / 2;
            xo4[e] = in ? rb[it] : f4zero();       // [pr][pc][8]: e = (pr*LC + pc)*2 + half
        }
    };
    load_dz(tile_of(0));
    load_h(tile_of(0));
    store_dz(tile_of(0), 0);
    __syncthreads();                                       // once per workgroup (the matrix waves run the same one): tile 0's dzpre is in LDS

    [[maybe_unused]] unsigned long long tp = MH_T(), s_p1 = 0, s_w1 = 0, s_p2 = 0, s_w2 = 0;
    for (int i = 0; i <= T; ++i) {
        [[maybe_unused]] const unsigned long long tb = MH_T();
        [[maybe_unused]] unsigned long long tc = tb;
        if (i < T) {
            const int tile = tile_of(i);
            const int n0 = tile / G::STRIPS, row0 = (tile % G::STRIPS) * TH;
            float* xt = L.xt0 + (i & 1) * G::XT;
            const float* dz = L.dz + (i & 1) * G::DZ;
            load_dz(tile_of(i + 1));      // in flight during the rebuild
            if constexpr (W0) fetch_x(tile);               // committed in the next phase 1
            __builtin_amdgcn_sched_barrier(0);
#ifdef MH_WHATIF_NOBUILD
            if (P.n < 0)
#endif
            {
#if MH_DZWIN
            // (round 6) the 3x3 dzpre window of item `it` is rows it .. it+2, columns x+1 .. x+3 of the dz tile: consecutive items share two of the
            // three rows.  All 12 x 3 values of the tile are read up front (36 LDS reads instead of 100, ONE latency instead of one per item; the
            // compiler cannot do this itself: the dH stores between the items may alias dz).
            float dzw[IT + 2][3];
#pragma unroll
            for (int rr = 0; rr < IT + 2; ++rr)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) dzw[rr][cc] = dz[rr * DZW + bx + 1 + cc];
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int r = it;
                const int x = bx, y = row0 + it - 1;
                const bool in = (it >= 1 && it <= TH) || (y >= 0 && y < H);      // interior rows of a strip are always inside the image
                const int gi = (in ? (n0 * H + y) * W * 4 : 0) + bx4;
                const float4 hv = hvs[it];
                const bool own = it >= 1 && it <= TH;       // rows owned by this strip (halo rows: the neighbours')
                const float4 hw_ = own ? hv : f4zero();
#if MH_APK
                // (round 6) the four channel sums as two register pairs: v_pk_fma_f32 (same fused arithmetic per component: same bits)
                mh_f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#else
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#endif
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
#if MH_DZWIN
                        float d = dzw[r + 2 - ky][2 - kx];
#else
                        float d = dz[(r + 2 - ky) * DZW + x + 3 - kx];
#endif
#if MH_APK
                        {
                            const mh_f2 dd = {d, d};
                            a01 = __builtin_elementwise_fma(dd, w2p[ky * 3 + kx][0], a01);
                            a23 = __builtin_elementwise_fma(dd, w2p[ky * 3 + kx][1], a23);
                        }
#else
                        a0 = fmaf(d, w2r[ky * 3 + kx][0], a0); a1 = fmaf(d, w2r[ky * 3 + kx][1], a1);
                        a2 = fmaf(d, w2r[ky * 3 + kx][2], a2); a3 = fmaf(d, w2r[ky * 3 + kx][3], a3);
#endif
                        if (WG && it != 0 && it != IT - 1) {     // items 0 and IT-1 are the halo rows: never owned
#if MH_WACC_PK
                            const mh_f2 dd = {d, d};
                            wacc[ky * 3 + kx][0] = __builtin_elementwise_fma(dd, mh_f2{hw_.x, hw_.y}, wacc[ky * 3 + kx][0]);
                            wacc[ky * 3 + kx][1] = __builtin_elementwise_fma(dd, mh_f2{hw_.z, hw_.w}, wacc[ky * 3 + kx][1]);
#else
                            wacc[ky * 3 + kx][0] = fmaf(d, hw_.x, wacc[ky * 3 + kx][0]);
                            wacc[ky * 3 + kx][1] = fmaf(d, hw_.y, wacc[ky * 3 + kx][1]);
                            wacc[ky * 3 + kx][2] = fmaf(d, hw_.z, wacc[ky * 3 + kx][2]);
                            wacc[ky * 3 + kx][3] = fmaf(d, hw_.w, wacc[ky * 3 + kx][3]);
#endif
                        }
                    }
#if MH_APK
                const float a0 = a01[0], a1 = a01[1], a2 = a23[0], a3 = a23[1];
#endif
                static_assert(IT == TH + 2, "one item per tile row: item index == row index");
                if (WG && it != 0 && it != IT - 1) {
#if MH_DZWIN
                    bacc += (own && pl == 0) ? dzw[r + 1][1] : 0.f;
#else
                    bacc += (own && pl == 0) ? dz[(r + 1) * DZW + x + 2] : 0.f;
#endif
                    // pin the accumulators here: otherwise their FMAs are sunk past the whole item loop and every dz / h
                    // value of all items stays live (hundreds of registers)
#pragma unroll
                    for (int t = 0; t < 9; ++t)
#pragma unroll
#if MH_WACC_PK
                        for (int c = 0; c < 2; ++c) asm volatile("" : "+v"(wacc[t][c]));
#else
                        for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(wacc[t][c]));
#endif
                }
                float4 v = make_float4(a0 * (hv.x > 0.f ? 1.f : 0.01f), a1 * (hv.y > 0.f ? 1.f : 0.01f),
                                       a2 * (hv.z > 0.f ? 1.f : 0.01f), a3 * (hv.w > 0.f ? 1.f : 0.01f));
                v = in ? v : f4zero();
                float* d = xt + (r * PW + x + 1) * PS + 4 * pl;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                if (own && P.dh) ((float4*)P.dh)[gi] = v;
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            load_h(tile_of(i + 1));       // in flight until the next rebuild (requested at the START of this rebuild into a second register set, it
                                          // measured no faster: the rebuild does not wait for memory)
            tc = MH_T();
            // the NEXT tile's dzpre (requested at the start of this rebuild) and THIS tile's masker.0 inputs (for the matrix waves' weight-gradient
            // GEMM of tile i in the next iteration) go into the buffers of parity i + 1 / i: the matrix waves read the inputs of parity i - 1 now,
            // this thread's neighbours the dzpre of parity i -- the barrier below is the only one of the tile
            store_dz(tile_of(i + 1), (i + 1) & 1);
            if constexpr (W0) commit_x(tile, i & 1);
        }
        [[maybe_unused]] const unsigned long long te = MH_T();
        __syncthreads();                                   // the barrier of tile i
        if (CGS_STAMP_PTR(P.dbg)) { const unsigned long long td = MH_T(); s_p1 += te - tc; s_p2 += tc - tb; s_w2 += td - te; tp = td; }
    }
    if (CGS_STAMP_PTR(P.dbg) && btid == 0) {
        unsigned long long* o = CGS_STAMP_PTR(P.dbg) + (size_t)bid * 16;
        o[0] = s_p1; o[1] = s_w1; o[2] = s_p2; o[3] = s_w2; o[4] = (unsigned long long)T;
    }

    if constexpr (WG) {
        // lanes with equal (lane & 3) hold the same channel plane: butterfly over the other 16 lanes; the 4 waves' partials
        // go to LDS [wave][plane][37] (all tiles are done: the tile storage is free)
        float* red2 = L.xt0 + G::RED2;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#if MH_WACC_PK
                float v = wacc[t][c >> 1][c & 1];
#else
                float v = wacc[t][c];
#endif
                v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                if (lane < 4) red2[(wave * 4 + lane) * 37 + t * 4 + c] = v;
            }
        float v = bacc;
        v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (lane < 4) red2[(wave * 4 + lane) * 37 + 36] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Matrix waves (threads 256..511; mtid = 0..255).  Per tile i: data-gradient and weight-gradient GEMMs of tile i-1 (round 6: the builder waves
// stage masker.0's inputs) | barrier
// ---------------------------------------------------------------------------------------------------------------
template <int TH, bool W0, int SRC>
__device__ __forceinline__ void mask_head_matrix(const MHeadParams& P, const MHeadLds& L, const int mtid, const int T,
                                                 const int bid, const int grid) {
    using G = MHeadGeo<TH, W0>;
    constexpr int PW = G::PW, PS = G::PS, LC = G::LC;
    static_assert(TH == 8, "4 matrix waves: one low-res row (data gradient) and one even + one odd row (weight gradient) each");
    const int lane = mtid & 63, mwave = mtid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    auto tile_of = [&](int i) { return bid + (i < T ? i : T - 1) * grid; };
    // weight gradient accumulators: image rows + bias (2 row blocks), folded o0 rows per parity (py, px) (2 row blocks each)
    frag4 accA[2], accB[2][2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        accA[q] = frag4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) accB[c >> 1][c & 1][q] = frag4{0.f, 0.f, 0.f, 0.f};
    }
    // image rows r = 16q + l15: (tap, channel) = (r / 3, r % 3) for r < 27, r = 27 the bias row (operand 1), else padding
    int rimg[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int r = 16 * q + l15, tap = r / 3, c = r % 3;
        rimg[q] = r < 27 ? ((tap / 3) * PW + tap % 3) * 4 + c : 0;
    }
    const float m1 = (16 + l15 < 27) ? 1.f : 0.f, m0 = (16 + l15 == 27) ? 1.f : 0.f;   // second row block: valid / bias / pad

    // MH_DGRAD4 (round 6): the data gradient  d_o0[q][c] = sum_{u,v < 4} sum_oc dH[2 q + (u - 1, v - 1)][oc] W4[u][v][oc][c]  with lane = low-resolution
    // pixel q on v_mfma_f32_4x4x1 (A = four output-channel weights broadcast from block abid = oc of weight register (u, v), B = the lane's dH value):
    // 256 (position, oc) steps x 2 channel groups = 512 instructions of 8 cycles per 64 pixels = 2048 matrix cycles per low-resolution row, where the
    // pair form (16 pairs x 2 x 8 columns, K = the 4 x 6 union window: a third of it zero weights) ran 96 x 32 = 3072.  Waves 0 / 1 take two
    // low-resolution rows each (64 lanes = 64 pixels) and a smaller share of the weight gradient's chunks, waves 2 / 3 the larger share.
    [[maybe_unused]] float w4r[2][16];
    if constexpr (MH_DGRAD4) {
        if (mwave < 2) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int k = 0; k < 16; ++k) w4r[g][k] = L.w4p[(k * 16 + (lane >> 2)) * 8 + 4 * g + (lane & 3)];
        }
    }
    [[maybe_unused]] unsigned long long tp = MH_T(), s_p1 = 0, s_w1 = 0, s_dg = 0, s_wg = 0, s_w2 = 0, tm = 0;
    __syncthreads();                                       // (the builders' once-per-workgroup barrier behind tile 0's dzpre)
    for (int i = 0; i <= T; ++i) {
        // (the matrix waves stage nothing since round 6 -- the builder waves stage masker.0's input tiles -- and meet the builders at ONE barrier per tile)
        [[maybe_unused]] const unsigned long long ta = MH_T();
        [[maybe_unused]] const unsigned long long tb = ta;
        if (i > 0) {
            const int tile = tile_of(i - 1);
            const int n0 = tile / G::STRIPS, row0 = (tile % G::STRIPS) * TH;
            const float* xt = L.xt0 + ((i - 1) & 1) * G::XT;
            [[maybe_unused]] const float* const ximg = L.ximg + ((i - 1) & 1) * G::XIMG;
            [[maybe_unused]] const float* const xo = L.xo + ((i - 1) & 1) * G::XO;
            // Operands come from LDS one group AHEAD of the MFMAs that use them (register double buffer): a wave issues
            // in order, so reads placed after a group's MFMAs would only start once the last of them has issued and the
            // matrix pipe would idle for a full LDS round trip per group.  A group must stay below 16 LDS instructions:
            // s_waitcnt lgkmcnt counts to 15, so "wait for the previous group only" is not expressible beyond that.
#ifdef MH_WHATIF_NODGRAD
            if (P.n < 0)
#endif
            if constexpr (MH_DGRAD4) {
                if (mwave < 2) {
                    const int qyl = 2 * mwave + (lane >> 5), qx = lane & 31;
                    const float* bp = xt + ((2 * qyl) * PW + 2 * qx) * PS;      // window position (0, 0) of this lane's pixel, channel 0
                    frag4 d0 = frag4{0.f, 0.f, 0.f, 0.f}, d1 = frag4{0.f, 0.f, 0.f, 0.f};
                    float bv[2][16];
                    auto ldp = [&](int k, int buf) {                           // the 16 channels of window position k = (u, v)
#pragma unroll
                        for (int oc = 0; oc < 16; ++oc) bv[buf][oc] = bp[((k >> 2) * PW + (k & 3)) * PS + oc];
                    };
                    ldp(0, 0);
                    static_for<16>([&](auto K) {
                        constexpr int k = decltype(K)::value;
                        if constexpr (k + 1 < 16) ldp(k + 1, (k + 1) & 1);
                        __builtin_amdgcn_sched_barrier(0);
                        static_for<16>([&](auto OC) {
                            constexpr int oc = decltype(OC)::value;
                            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w4r[0][k], bv[k & 1][oc], d0, 4, oc, 0);
                            d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w4r[1][k], bv[k & 1][oc], d1, 4, oc, 0);
                        });
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    float4* o = (float4*)(P.d_o0 + ((size_t)(n0 * 32 + row0 / 2 + qyl) * 32 + qx) * 8);
                    o[0] = make_float4(d0[0], d0[1], d0[2], d0[3]);
                    o[1] = make_float4(d1[0], d1[1], d1[2], d1[3]);
                }
            } else {
                // data gradient of low-res row qyl = mwave: lane pair i = l15 covers low-res pixels 2i (columns 0-7 of the
                // tile) and 2i+1 (columns 8-15); window = rows 2qyl-1..2qyl+2, columns 4i-1..4i+4 of dH
                const int qyl = mwave;
                const int abase = ((2 * qyl) * PW + 4 * l15) * PS + kq;
                const float* wb = L.w4p + kq * 16 + l15;          // B: w4p[(pos*16 + 4*plane + kq)*16 + l15]
                // TWO accumulation chains (even / odd k-steps, added once at the end): v_mfma_f32_16x16x4_f32 issues every 32 cycles but a
                // dependent one only after 40 (cdna_hip_programming.md, "FP32-input MFMA"), and this role is the SIMD's only MFMA wave
                // (round 6: 96 x 8 cycles per tile)
                frag4 d = frag4{0.f, 0.f, 0.f, 0.f}, d1 = frag4{0.f, 0.f, 0.f, 0.f};
                constexpr int GS = 12, NG = 96 / GS;              // 24 dwords = 12 ds_read2 per group
                float av[2][GS], bw[2][GS];
                auto ld = [&](int g, int buf) {
#pragma unroll
                    for (int j = 0; j < GS; ++j) {
                        const int s = g * GS + j, pos = s >> 2, u = pos / 6, v6 = pos % 6;
                        av[buf][j] = xt[abase + (u * PW + v6) * PS + 4 * (s & 3)];
                        bw[buf][j] = wb[s * 64];
                    }
                };
                ld(0, 0);
                __builtin_amdgcn_sched_barrier(0);        // (group 0's reads all issue before its first MFMA: they must not be dealt out between them)
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g + 1 < NG) ld(g + 1, (g + 1) & 1);
#if !MH_INTERLEAVE
                    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                    for (int j = 0; j < GS; j += 2) {
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][j], bw[g & 1][j], d, 0, 0, 0);
                        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][j + 1], bw[g & 1][j + 1], d1, 0, 0, 0);
                    }
#if MH_INTERLEAVE
                    // (round 6) the next group's operand reads are issued BETWEEN this group's MFMAs (one LDS instruction per matrix instruction:
                    // a matrix instruction keeps the pipe busy for 32 cycles, the wave is free to issue meanwhile) instead of in front of them,
                    // where their issue time was a gap in the matrix pipe once per group
                    if (g + 1 < NG) {
#pragma unroll
                        for (int j = 0; j < GS; ++j) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, GS, 0);
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
                d += d1;
                // D[row = pair 4kq + j][col = l15 = 8*(pixel of the pair) + channel]: 16 contiguous floats per pair
                float* o = P.d_o0 + ((size_t)(n0 * 32 + row0 / 2 + qyl) * 32 + 2 * (4 * kq)) * 8 + l15;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j * 16] = d[j];
            }
            if (CGS_STAMP_PTR(P.dbg)) tm = MH_T();
#ifdef MH_WHATIF_NOWGRAD
            if (P.n < 0)
#endif
            if constexpr (W0) {
                // weight gradient: this wave's rows 2*mwave (py = 0) and 2*mwave + 1 (py = 1); a k-step = 4 same-parity
                // pixels x = 2*(4s + kq) + px of one row.  Chunk = (py, px, two k-steps): 5 ds_read2 + 8 MFMAs.
                constexpr int UU = 2, CPP = 8 / UU, NCH = 4 * CPP;
                const int bb = (PW + 1 + 2 * kq) * PS + l15;              // dH: ((yl+1)*PW + 1 + 2*(4s+kq) + px)*PS + l15
                const int ib = (2 * kq) * 4;                               // image: ((yl+ky)*PW + 2*(4s+kq) + px + kx)*4 + c
                const int ob = kq * 8 + l15;                               // o0: ((rp+a+py)*LC + 4s + kq + b + px)*8 + cb
                float ai[2][UU][2], ao[2][UU][2], b[2][UU];
                // A HALF = the 8 chunks (px = 0 / 1, s0 = 0, 2, 4, 6) of one row (row pair rp, row parity py): 64 matrix instructions.  The tile's 8
                // halves hh = 2 rp + py are dealt to the waves: two each, or -- MH_DGRAD4: waves 0 / 1 also carry the data gradient (4096 matrix
                // cycles) -- 1 / 1 / 3 / 3, so every wave issues 6144 matrix cycles per tile.  py is a template value of the half (the accumulator
                // arrays must be indexed statically), chosen by a wave-uniform branch.
                auto do_half = [&](int rp, auto PY) {
                    constexpr int py = decltype(PY)::value;
                    const int yl = 2 * rp + py;
                    auto ld = [&](int ch, int buf) {              // ch = 0 .. 7: px = ch / CPP, two k-steps from s0 = UU (ch % CPP)
                        const int px = ch / CPP, s0 = UU * (ch % CPP);
#pragma unroll
                        for (int u = 0; u < UU; ++u) b[buf][u] = xt[bb + (yl * PW + px + 8 * (s0 + u)) * PS];
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int u = 0; u < UU; ++u) ai[buf][u][q] = ximg[ib + rimg[q] + (yl * PW + px + 8 * (s0 + u)) * 4];
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int u = 0; u < UU; ++u) ao[buf][u][q] = xo[ob + ((rp + q + py) * LC + px + 4 * (s0 + u)) * 8];
                    };
                    ld(0, 0);
#pragma unroll
                    for (int ch = 0; ch < 2 * CPP; ++ch) {
                        if (ch + 1 < 2 * CPP) ld(ch + 1, (ch + 1) & 1);
                        __builtin_amdgcn_sched_barrier(0);
                        const int px = ch / CPP;
#pragma unroll
                        for (int u = 0; u < UU; ++u) {
                            const float bv = b[ch & 1][u];
                            // rows past the 27 image rows (bias row: constant 1, padding: 0) as arithmetic at the point of use:
                            // the load stays unconditional and pairs into ds_read2
                            accA[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[ch & 1][u][0], bv, accA[0], 0, 0, 0);
                            accA[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fmaf(ai[ch & 1][u][1], m1, m0), bv, accA[1], 0, 0, 0);
                            accB[py][px][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ao[ch & 1][u][0], bv, accB[py][px][0], 0, 0, 0);
                            accB[py][px][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ao[ch & 1][u][1], bv, accB[py][px][1], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                const int h_lo = MH_DGRAD4 ? (mwave < 2 ? mwave : 2 + 3 * (mwave - 2)) : 2 * mwave;
                const int h_n = MH_DGRAD4 ? (mwave < 2 ? 1 : 3) : 2;
#pragma unroll 1
                for (int hh = h_lo; hh < h_lo + h_n; ++hh) {
                    if (hh & 1) do_half(hh >> 1, std::integral_constant<int, 1>{});
                    else do_half(hh >> 1, std::integral_constant<int, 0>{});
                }
            }
        }
        [[maybe_unused]] const unsigned long long tc = MH_T();
        __syncthreads();                                   // the barrier of tile i
        if (CGS_STAMP_PTR(P.dbg)) {
            const unsigned long long td = MH_T();
            s_w2 += td - tc; tp = td;
            if (i > 0) { s_dg += tm - tb; s_wg += tc - tm; } else s_dg += tc - tb;
        }
    }
    if (CGS_STAMP_PTR(P.dbg) && mtid == 0) {
        unsigned long long* o = CGS_STAMP_PTR(P.dbg) + (size_t)bid * 16 + 8;
        o[0] = s_p1; o[1] = s_w1; o[2] = s_dg; o[3] = s_wg; o[4] = s_w2;
    }

    if constexpr (W0) {
        // D layout: col = lane & 15 (= oc), row = 16q + (lane >> 4) * 4 + reg; one partial set per matrix wave
        float* redA = L.xt0 + G::REDA + mwave * (32 * 16);
        float* redB = L.xt0 + G::REDB + mwave * (4 * 32 * 16);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 16 * q + 4 * kq + j;
                redA[r * 16 + l15] = accA[q][j];
#pragma unroll
                for (int c = 0; c < 4; ++c) redB[(c * 32 + r) * 16 + l15] = accB[c >> 1][c & 1][q][j];
            }
    }
}

// WG: produce slab2; W0: produce slab0 (needs img/o0); SRC: WSRC_U8 / WSRC_F32 image
template <int TH, bool WG, bool W0, int SRC>
__global__ void __launch_bounds__(512) mask_head_kernel(MHeadParams P) {
    if (CGS_KARG_PREFETCH) cgs_kernarg_prefetch<sizeof(MHeadParams)>();
    using G = MHeadGeo<TH, W0>;
    constexpr int TRA = G::TRA, PW = G::PW, PS = G::PS;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    MHeadLds L;
    L.xt0 = (float*)smem;              // dH tiles [2][TRA][PW][PS]
    L.ximg = L.xt0 + 2 * G::XT;        // masker.0 image tiles [2][TRA][PW][r,g,b,0] (W0)
    L.xo = L.ximg + 2 * G::XIMG;       // masker.0 low-resolution input tiles [2][LR][LC][8] (W0)
    L.w4p = L.xo + 2 * G::XO;          // pair-folded weights [u 0..3][v6 0..5][oc][8*g + c]
    L.dz = L.w4p + G::W4P;             // dzpre tiles [2] with a 2-pixel halo
    const int tid = threadIdx.x;
    [[maybe_unused]] const unsigned long long t_start = MH_T();

    // (every load of the table unconditional -- clamped index, selected afterwards -- and the loop unrolled: with `if (valid) s += P.w0[..]` each of the
    //  4 x 12 loads per thread was a branch, a load and a wait of its own: 48 dependent round trips at the start of every workgroup, the whole launch in
    //  lockstep behind them; the sum keeps its order, an absent term adds 0.f)
    static_assert(G::W4P % 512 == 0 && G::W4P / 512 <= 12, "table rounds");
    {
        float wv[G::W4P / 512][4];
        // entry e -> (window position (u, v), oc, output column): MH_DGRAD4: [pos 4x4][oc][c];  else pair-folded [pos 4x6][oc][8 g + c], pixel g of the
        // pair sees window column v6 as v = v6 - 2 g
        auto decode = [&](int e, int& u, int& v, int& oc, int& c) {
            if constexpr (MH_DGRAD4) { c = e & 7; oc = (e >> 3) & 15; const int pos = e >> 7; u = pos >> 2; v = pos & 3; }
            else { const int col = e & 15, pos = e >> 8; oc = (e >> 4) & 15; u = pos / 6; c = col & 7; v = pos % 6 - 2 * (col >> 3); }
        };
#pragma unroll
        for (int k = 0; k < G::W4P / 512; ++k) {
            int u, v, oc, c;
            decode(tid + 512 * k, u, v, oc, c);
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {
                const int ky = (ab >> 1) + 2 - u, kx = (ab & 1) + 2 - v;
                const bool ok = v >= 0 && v <= 3 && ky >= 0 && ky <= 2 && kx >= 0 && kx <= 2;
                wv[k][ab] = P.w0[ok ? ((ky * 3 + kx) * 11 + 3 + c) * 16 + oc : 0];
            }
        }
#pragma unroll
        for (int k = 0; k < G::W4P / 512; ++k) {
            int u, v, oc, c;
            decode(tid + 512 * k, u, v, oc, c);
            float sm = 0.f;
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {
                const int ky = (ab >> 1) + 2 - u, kx = (ab & 1) + 2 - v;
                const bool ok = v >= 0 && v <= 3 && ky >= 0 && ky <= 2 && kx >= 0 && kx <= 2;
                sm += ok ? wv[k][ab] : 0.f;
            }
            L.w4p[tid + 512 * k] = sm;
        }
    }
    for (int e = tid; e < 2 * TRA * 2 * 16; e += 512) {      // zero halo columns of both dH tiles (never written again)
        int ch = e & 15, side = (e >> 4) & 1, r = (e >> 5) % TRA, bufi = e / (32 * TRA);
        L.xt0[bufi * G::XT + (r * PW + (side ? PW - 1 : 0)) * PS + ch] = 0.f;
    }
    __syncthreads();

    const int grid = gridDim.x, bid = blockIdx.x;
    const int T = (P.ntiles - bid + grid - 1) / grid;      // tiles of this workgroup: bid, bid+grid, ...
    if (CGS_STAMP_PTR(P.dbg) && tid == 0) { CGS_STAMP_PTR(P.dbg)[(size_t)bid * 16 + 5] = t_start; CGS_STAMP_PTR(P.dbg)[(size_t)bid * 16 + 6] = MH_T(); }
    // SCALAR role branch: whole waves take one side, both sides execute 1 + (T+1) workgroup barriers
    if (__builtin_amdgcn_readfirstlane(tid) < 256) mask_head_builder<TH, WG, W0, SRC>(P, L, tid, T, bid, grid);
    else mask_head_matrix<TH, W0, SRC>(P, L, tid - 256, T, bid, grid);
    __syncthreads();
    [[maybe_unused]] const unsigned long long t_after_roles = MH_T();

    // ---------------- weight-gradient partials: one slab per workgroup ----------------
    if constexpr (W0) {
        // unfold: image rows and bias straight from the A partials; the 8 upsampled channels of tap (ky, kx) are the sum
        // over the 4 parity classes of their folded row a(py,ky), b(px,kx)   (fixed order: deterministic)
        const float* redA = L.xt0 + G::REDA;
        const float* redB = L.xt0 + G::REDB;
        auto sum4 = [&](const float* p, int stride) { return (p[0] + p[stride]) + (p[2 * stride] + p[3 * stride]); };
        for (int e = tid; e < 1600; e += 512) {
            float v;
            if (e >= 1584) {
                v = sum4(redA + 27 * 16 + (e - 1584), 32 * 16);
            } else {
                const int oc = e & 15, ci = (e >> 4) % 11, tap = (e >> 4) / 11, ky = tap / 3, kx = tap % 3;
                if (ci < 3) {
                    v = sum4(redA + (tap * 3 + ci) * 16 + oc, 32 * 16);
                } else {
                    v = 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int py = c >> 1, px = c & 1;
                        const int a = py == 0 ? (ky >= 1) : (ky == 2), b = px == 0 ? (kx >= 1) : (kx == 2);
                        v += sum4(redB + (c * 32 + (a * 2 + b) * 8 + (ci - 3)) * 16 + oc, 4 * 32 * 16);
                    }
                }
            }
            P.slab0[(size_t)blockIdx.x * 1600 + e] = v;
        }
    }
    if constexpr (WG) {
        const float* red2 = L.xt0 + G::RED2;
        if (tid < 145) {
            // slab layout = HWIO [9][16][1] weights then the bias
            int t = tid / 16, ch = tid % 16, p = ch >> 2, c = ch & 3;
            int idx = (tid < 144) ? t * 4 + c : 36;
            if (tid == 144) p = 0;
            float v = (red2[(0 * 4 + p) * 37 + idx] + red2[(1 * 4 + p) * 37 + idx]) + (red2[(2 * 4 + p) * 37 + idx] + red2[(3 * 4 + p) * 37 + idx]);
            P.slab2[(size_t)blockIdx.x * 145 + tid] = v;
        }
    }
    if (CGS_STAMP_PTR(P.dbg) && tid == 0) { CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16 + 7] = MH_T(); CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16 + 13] = t_after_roles; }
}

// one workgroup per CU (LDS), persistent over its tiles
static int mask_head_blocks(int n) { int t = n * (64 / 8); return t < 256 ? t : 256; }
int mask_head_slabs(int n) { return n <= 0 ? 0 : mask_head_blocks(n); }

template <bool WG, bool W0, int SRC>
static int launch_mask_head(MHeadParams P, hipStream_t st) {
    constexpr int TH = 8;
    using G = MHeadGeo<TH, W0>;
    P.ntiles = P.n * G::STRIPS;
    static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&mask_head_kernel<TH, WG, W0, SRC>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);   // > 64 KB
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL((mask_head_kernel<TH, WG, W0, SRC>), dim3(mask_head_blocks(P.n)), dim3(512), G::LDS, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// img_kind: CGS_SRC_U8 / CGS_SRC_F32 (only read when slab0 is requested)
int mask_head_launch(int n, int img_kind, const void* img, const float* o0, const float* dzpre, const float* h,
                     const float* w2, const float* w0, float* dh, float* d_o0, float* slab2, float* slab0, hipStream_t st) {
    if (n <= 0) return CGS_OK;
    MHeadParams P{dzpre, h, w2, w0, img, o0, dh, d_o0, slab2, slab0, n, 0, nullptr};
#ifdef CGS_DEBUG_STAMPS
    P.dbg = g_mh_stamps;
#endif
    if (slab0) {
        if (!slab2 || !img || !o0) return CGS_ERR_BADARG;
        return img_kind == CGS_SRC_U8 ? launch_mask_head<true, true, WSRC_U8>(P, st) : launch_mask_head<true, true, WSRC_F32>(P, st);
    }
    return slab2 ? launch_mask_head<true, false, WSRC_F32>(P, st) : launch_mask_head<false, false, WSRC_F32>(P, st);
}

// LDS-tiled, im2col-free direct 3x3 convolution machinery shared by the forward, data-gradient and
// weight-gradient kernels.
//
// Mapping (all sizes compile-time):
//   * one thread = one 2x2 quad of conv outputs, all output channels of the current chunk in registers
//     (the quad is the pooling window of the encoder and the nearest-upsample cell of the decoder, so
//     max-pool, argmax and upsample-backward sums never leave the thread);
//   * one workgroup = THREADS quads = a full-width strip of TH rows of one image, or IMGS whole images
//     when the image is small.  Full-width strips make the horizontal halo pure zero padding and the
//     global range a strip reads one contiguous NHWC block (coalesced 16-byte loads);
//   * source A (direct / skip input) is staged in LDS as float4 "planes" (4 channels per plane):
//       ldsA[plane][img][row 0..TH+1][col 0..W+1 (+1 pad slot per 16 columns)]
//     so a wave reads 64 x 16 B from distinct 16-byte slots;
//   * source B (the nearest-upsampled low-resolution input of a decoder conv) is staged at ITS OWN
//     resolution: a quad's 4x4 receptive field covers just 3x3 source pixels, the concat and the
//     upsampled tensor are never materialised;
//   * weights come from the constant address space (s_load -> SGPR operands of v_fmac_f32).
#pragma once
#include "cgs_common.h"
#include <utility>
#include <type_traits>

enum { SRC_F32 = 0, SRC_U8C3 = 1, SRC_F32C3 = 2, SRC_POOLEXP = 3, SRC_SCALAR = 4, SRC_DH = 5, SRC_MIXC3 = 6, SRC_POOLEXP_DIFF = 7 };

// THREADS = quads per workgroup; CW = waves-groups that split the output-channel chunks of those quads
// between them (small images: more waves per image); LT = threads that take part in the tile loads.
template <int H_, int W_, int THREADS_, int CW_ = 1>
struct Geo {
    static constexpr int H = H_, W = W_, THREADS = THREADS_, CW = CW_, LT = THREADS_ * CW_;
    static constexpr int QW = W / 2, QH = H / 2, Q = QW * QH;
    static constexpr int IMGS = (Q >= THREADS) ? 1 : THREADS / Q;
    static constexpr int RQ = (Q >= THREADS) ? THREADS / QW : QH;  // quad rows per workgroup
    static constexpr int TH = 2 * RQ;                               // conv rows per workgroup
    static constexpr int STRIPS = QH / RQ;
    static constexpr int TRA = TH + 2;                              // rows of the source-A tile
    static constexpr int PWA = (W + 1) + ((W + 1) >> 4) + 1;        // padded row length (slots)
    static constexpr int TRB = RQ + 2, PWB = QW + 2;                // source-B tile (half resolution)
    static_assert(THREADS % 64 == 0 && (Q >= THREADS ? (THREADS % QW == 0) : (THREADS % Q == 0)), "geometry");
    __device__ static __forceinline__ int pc(int c) { return c + (c >> 4); }
};

struct QuadPos {
    int img_l, qy_l, qx;  // image within workgroup, quad row within strip, quad column
    int n, row0;          // global image index, first conv row of the strip
};

template <class G>
__device__ __forceinline__ QuadPos quad_pos(int tid, int bid) {
    QuadPos q;
    constexpr int PER_IMG = G::RQ * G::QW;
    q.img_l = tid / PER_IMG;
    int r = tid % PER_IMG;
    q.qy_l = r / G::QW;
    q.qx = r % G::QW;
    if constexpr (G::IMGS == 1) {
        q.n = bid / G::STRIPS;
        q.row0 = (bid % G::STRIPS) * G::TH;
    } else {
        q.n = bid * G::IMGS + q.img_l;
        q.row0 = 0;
    }
    return q;
}

// ------------------------------------------------------------------------------------------------
// Loaders.  n0 = first image of the workgroup, row0 = first conv row.  All write float4 planes.
// Every loader is a fully unrolled, BRANCH-FREE loop (indices clamped, values zeroed by select, the
// tail iteration redoes the last element): the global loads of all iterations are independent, so the
// scheduler issues them back to back instead of paying one memory latency per iteration.
// ------------------------------------------------------------------------------------------------
template <int E, int THREADS, class F>
__device__ __forceinline__ void for_elems(int tid, F f) {
    constexpr int IT = (E + THREADS - 1) / THREADS;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        int e = tid + it * THREADS;
        f(e < E ? e : E - 1);
    }
}

template <class G, int PA>
__device__ __forceinline__ int ldsA_idx(int p, int img, int r, int c) {
    return ((p * G::IMGS + img) * G::TRA + r) * G::PWA + G::pc(c);
}

template <class G, int PA>
__device__ __forceinline__ void zero_halo_cols(float4* ldsA, int tid) {
    constexpr int E = PA * G::IMGS * G::TRA * 2;
    for_elems<E, G::LT>(tid, [&](int e) {
        int side = e & 1, rest = e >> 1;
        int r = rest % G::TRA, pi = rest / G::TRA;  // pi = p*IMGS + img
        ldsA[(pi * G::TRA + r) * G::PWA + G::pc(side ? G::W + 1 : 0)] = f4zero();
    });
}

// NHWC fp32 source with CA = 4*PA channels (optionally with dropout fused).
template <class G, int PA>
__device__ __forceinline__ void load_a_f32(float4* ldsA, const float4* __restrict__ src, int n0, int row0,
                                           int N, int tid, const DropCtx& dc) {
    constexpr int E = G::IMGS * G::TRA * G::W * PA;
    for_elems<E, G::LT>(tid, [&](int e) {
        int p = e % PA, x = (e / PA) % G::W, r = (e / (PA * G::W)) % G::TRA, img = e / (PA * G::W * G::TRA);
        int n = n0 + img, y = row0 + r - 1;
        bool in = n < N && y >= 0 && y < G::H;
        int gi = in ? ((n * G::H + y) * G::W + x) * PA + p : 0;
        float4 v = src[gi];
        if (dc.on) v = v * drop_mult4(dc, (uint32_t)gi);
        ldsA[ldsA_idx<G, PA>(p, img, r, x + 1)] = in ? v : f4zero();
    });
    zero_halo_cols<G, PA>(ldsA, tid);
}

// NHWC uint8 RGB source, /255 fused.  One thread converts 4 pixels (12 bytes = 3 dwords).
template <class G>
__device__ __forceinline__ void load_a_u8c3(float4* ldsA, const uint32_t* __restrict__ src, int n0, int row0,
                                            int N, int tid) {
    constexpr int GW = G::W / 4;
    constexpr int E = G::IMGS * G::TRA * GW;
    const float s = 1.f / 255.f;
    for_elems<E, G::LT>(tid, [&](int e) {
        int g = e % GW, r = (e / GW) % G::TRA, img = e / (GW * G::TRA);
        int n = n0 + img, y = row0 + r - 1;
        bool in = n < N && y >= 0 && y < G::H;
        int gi = in ? ((n * G::H + y) * G::W + g * 4) * 3 / 4 : 0;  // dword index (12 B per 4 pixels)
        uint32_t d0 = src[gi], d1 = src[gi + 1], d2 = src[gi + 2];
        d0 = in ? d0 : 0u; d1 = in ? d1 : 0u; d2 = in ? d2 : 0u;
        float4 p0 = make_float4((d0 & 255) * s, ((d0 >> 8) & 255) * s, ((d0 >> 16) & 255) * s, 0.f);
        float4 p1 = make_float4((d0 >> 24) * s, (d1 & 255) * s, ((d1 >> 8) & 255) * s, 0.f);
        float4 p2 = make_float4(((d1 >> 16) & 255) * s, (d1 >> 24) * s, (d2 & 255) * s, 0.f);
        float4 p3 = make_float4(((d2 >> 8) & 255) * s, ((d2 >> 16) & 255) * s, (d2 >> 24) * s, 0.f);
        int base = (img * G::TRA + r) * G::PWA;
        ldsA[base + G::pc(g * 4 + 1)] = p0;
        ldsA[base + G::pc(g * 4 + 2)] = p1;
        ldsA[base + G::pc(g * 4 + 3)] = p2;
        ldsA[base + G::pc(g * 4 + 4)] = p3;
    });
    zero_halo_cols<G, 1>(ldsA, tid);
}

// NHWC fp32 3-channel source (the replaced / injected mixes).  4 pixels = 3 float4.
template <class G>
__device__ __forceinline__ void load_a_f32c3(float4* ldsA, const float4* __restrict__ src, int n0, int row0,
                                             int N, int tid) {
    constexpr int GW = G::W / 4;
    constexpr int E = G::IMGS * G::TRA * GW;
    for_elems<E, G::LT>(tid, [&](int e) {
        int g = e % GW, r = (e / GW) % G::TRA, img = e / (GW * G::TRA);
        int n = n0 + img, y = row0 + r - 1;
        bool in = n < N && y >= 0 && y < G::H;
        int gi = in ? ((n * G::H + y) * G::W + g * 4) * 3 / 4 : 0;
        float4 a = src[gi], b = src[gi + 1], c = src[gi + 2];
        if (!in) { a = f4zero(); b = f4zero(); c = f4zero(); }
        int base = (img * G::TRA + r) * G::PWA;
        ldsA[base + G::pc(g * 4 + 1)] = make_float4(a.x, a.y, a.z, 0.f);
        ldsA[base + G::pc(g * 4 + 2)] = make_float4(a.w, b.x, b.y, 0.f);
        ldsA[base + G::pc(g * 4 + 3)] = make_float4(b.z, b.w, c.x, 0.f);
        ldsA[base + G::pc(g * 4 + 4)] = make_float4(c.y, c.z, c.w, 0.f);
    });
    zero_halo_cols<G, 1>(ldsA, tid);
}

// The replaced / injected mixes of main.py:395,406 computed on the fly from the uint8 frames and the mask (never stored):
// image n < n_a: A(1-Z) + Z B of A-image n;  image n >= n_a: B(1-Z) + Z A of A-image n - n_a.  One thread = 4 pixels.
template <class G>
__device__ __forceinline__ void load_a_mix(float4* ldsA, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                           const float4* __restrict__ z, int n_a, int n0, int row0, int N, int tid) {
    constexpr int GW = G::W / 4;
    constexpr int E = G::IMGS * G::TRA * GW;
    const float s = 1.f / 255.f;
    for_elems<E, G::LT>(tid, [&](int e) {
        int g = e % GW, r = (e / GW) % G::TRA, img = e / (GW * G::TRA);
        int n = n0 + img, y = row0 + r - 1;
        bool in = n < N && y >= 0 && y < G::H;
        const bool inj = n >= n_a;
        int src_n = in ? (inj ? n - n_a : n) : 0;
        int pg = in ? (src_n * G::H + y) * G::W / 4 + g : 0;          // 4-pixel group index
        uint32_t a0 = a[3 * pg], a1 = a[3 * pg + 1], a2 = a[3 * pg + 2];
        uint32_t b0 = b[3 * pg], b1 = b[3 * pg + 1], b2 = b[3 * pg + 2];
        float4 zz = z[pg];
        if (inj) { uint32_t t; t = a0; a0 = b0; b0 = t; t = a1; a1 = b1; b1 = t; t = a2; a2 = b2; b2 = t; }
        // byte k of the 12-byte group: channel k % 3 of pixel k / 3
        auto mixv = [&](uint32_t da, uint32_t db, int sh, float zi) {
            float av = ((da >> sh) & 255) * s, bv = ((db >> sh) & 255) * s;
            return av * (1.f - zi) + zi * bv;
        };
        float4 p0 = make_float4(mixv(a0, b0, 0, zz.x), mixv(a0, b0, 8, zz.x), mixv(a0, b0, 16, zz.x), 0.f);
        float4 p1 = make_float4(mixv(a0, b0, 24, zz.y), mixv(a1, b1, 0, zz.y), mixv(a1, b1, 8, zz.y), 0.f);
        float4 p2 = make_float4(mixv(a1, b1, 16, zz.z), mixv(a1, b1, 24, zz.z), mixv(a2, b2, 0, zz.z), 0.f);
        float4 p3 = make_float4(mixv(a2, b2, 8, zz.w), mixv(a2, b2, 16, zz.w), mixv(a2, b2, 24, zz.w), 0.f);
        int base = (img * G::TRA + r) * G::PWA;
        ldsA[base + G::pc(g * 4 + 1)] = in ? p0 : f4zero();
        ldsA[base + G::pc(g * 4 + 2)] = in ? p1 : f4zero();
        ldsA[base + G::pc(g * 4 + 3)] = in ? p2 : f4zero();
        ldsA[base + G::pc(g * 4 + 4)] = in ? p3 : f4zero();
    });
    zero_halo_cols<G, 1>(ldsA, tid);
}

__device__ __forceinline__ float4 nib_select(const float4& v, uint32_t nib16, uint32_t pos) {
    float4 o;
    o.x = ((nib16 & 15u) == pos) ? v.x : 0.f;
    o.y = (((nib16 >> 4) & 15u) == pos) ? v.y : 0.f;
    o.z = (((nib16 >> 8) & 15u) == pos) ? v.z : 0.f;
    o.w = (((nib16 >> 12) & 15u) == pos) ? v.w : 0.f;
    return o;
}

// Gradient of a conv+ReLU+maxpool stage, re-expanded to pre-pool resolution on the fly:
// tile(y,x,c) = dpooled(y/2,x/2,c) if amask nibble == 2*(y&1)+(x&1) else 0   (0xF nibble = ReLU dead).
// HALO = 1: rows row0-1 .. row0+TH (data-gradient tile); HALO = 0: rows row0 .. row0+TH-1.  idx(p,img,r,x)
// gives the destination slot; out-of-tile rows are redirected to the caller's dump slot idx(...) = `dump`.
template <class G, int PA, int HALO, class IdxF>
__device__ __forceinline__ void load_poolexp(float4* lds, const float4* __restrict__ dp,
                                             const uint32_t* __restrict__ am, int n0, int row0, int N,
                                             int tid, IdxF idx, int dump) {
    constexpr int HP = G::H / 2, WP = G::W / 2;
    constexpr int JR = G::RQ + 2 * HALO;  // pooled rows touched
    constexpr int E = G::IMGS * JR * WP * PA;
    constexpr int AMW = (PA + 1) / 2;     // amask words per pooled pixel
    const int pr0 = row0 / 2 - HALO;
    for_elems<E, G::LT>(tid, [&](int e) {
        int p = e % PA, px = (e / PA) % WP, j = (e / (PA * WP)) % JR, img = e / (PA * WP * JR);
        int n = n0 + img, pr = pr0 + j;
        bool in = n < N && pr >= 0 && pr < HP;
        int pi = in ? (n * HP + pr) * WP + px : 0;
        float4 v = dp[pi * PA + p];
        uint32_t nib = (am[pi * AMW + (p >> 1)] >> ((p & 1) * 16)) & 0xFFFFu;
        nib = in ? nib : 0xFFFFu;
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            int r = 2 * j + (pos >> 1) - HALO;  // row inside the tile
            bool ok = r >= 0 && r < G::TH + 2 * HALO;
            lds[ok ? idx(p, img, r, 2 * px + (pos & 1)) : dump] = nib_select(v, nib, pos);
        }
    });
}

// The same re-expansion for the DIFFERENCE of two images' gradients, image n0 minus image n0 + n_off (when sub): the data gradient
// of a convolution is linear in its input, so conv_bwd(dY_replaced) - conv_bwd(dY_injected) -- all the mix backward of
// main.py:395,406 needs -- is ONE pass over the difference tile (each image contributes at most one non-zero per pooling window).
template <class G, int PA, int HALO, class IdxF>
__device__ __forceinline__ void load_poolexp_diff(float4* lds, const float4* __restrict__ dp, const uint32_t* __restrict__ am, int n0,
                                                  int n_off, bool sub, int row0, int N, int tid, IdxF idx, int dump) {
    static_assert(G::IMGS == 1, "one image per workgroup");
    constexpr int HP = G::H / 2, WP = G::W / 2;
    constexpr int JR = G::RQ + 2 * HALO;
    constexpr int E = JR * WP * PA;
    constexpr int AMW = (PA + 1) / 2;
    const int pr0 = row0 / 2 - HALO;
    for_elems<E, G::LT>(tid, [&](int e) {
        int p = e % PA, px = (e / PA) % WP, j = e / (PA * WP);
        int pr = pr0 + j;
        bool in = n0 < N && pr >= 0 && pr < HP;
        int pi = in ? (n0 * HP + pr) * WP + px : 0;
        int pj = (in && sub) ? ((n0 + n_off) * HP + pr) * WP + px : 0;
        float4 v = dp[pi * PA + p], u = dp[pj * PA + p];
        uint32_t nib = (am[pi * AMW + (p >> 1)] >> ((p & 1) * 16)) & 0xFFFFu;
        uint32_t nbj = (am[pj * AMW + (p >> 1)] >> ((p & 1) * 16)) & 0xFFFFu;
        nib = in ? nib : 0xFFFFu;
        nbj = (in && sub) ? nbj : 0xFFFFu;
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            int r = 2 * j + (pos >> 1) - HALO;
            bool ok = r >= 0 && r < G::TH + 2 * HALO;
            const float4 a = nib_select(v, nib, pos), b = nib_select(u, nbj, pos);
            lds[ok ? idx(p, 0, r, 2 * px + (pos & 1)) : dump] = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
        }
    });
}

// Source B at its own (half) resolution with a 1-pixel zero halo: rows sy0-1 .. sy0+RQ, cols -1 .. QW.
template <class G, int PB>
__device__ __forceinline__ void load_b_half(float4* ldsB, const float4* __restrict__ src, int n0, int row0,
                                            int N, int tid) {
    constexpr int E = G::IMGS * G::TRB * G::PWB * PB;
    const int sy0 = row0 / 2;
    for_elems<E, G::LT>(tid, [&](int e) {
        int p = e % PB, c = (e / PB) % G::PWB, r = (e / (PB * G::PWB)) % G::TRB, img = e / (PB * G::PWB * G::TRB);
        int n = n0 + img, sy = sy0 + r - 1, sx = c - 1;
        bool in = n < N && sy >= 0 && sy < G::QH && sx >= 0 && sx < G::QW;
        float4 v = src[in ? ((n * G::QH + sy) * G::QW + sx) * PB + p : 0];
        ldsB[((p * G::IMGS + img) * G::TRB + r) * G::PWB + c] = in ? v : f4zero();
    });
}

// Source B that is a single pixel per image (the bottleneck, upsampled x4 to the 4x4 map).
template <class G, int PB>
__device__ __forceinline__ void load_b_pix(float4* ldsB, const float4* __restrict__ src, int n0, int N, int tid) {
    constexpr int E = G::IMGS * PB;
    for_elems<E, G::LT>(tid, [&](int e) {
        int p = e % PB, img = e / PB;
        int n = n0 + img;
        float4 v = src[n < N ? n * PB + p : 0];
        ldsB[p * G::IMGS + img] = (n < N) ? v : f4zero();
    });
}

template <int N, class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// ------------------------------------------------------------------------------------------------
// Split loaders (software pipeline over the tiles of a persistent workgroup, conv3x3_body_pipe): fetch_* issues the global loads of a
// tile into a REGISTER block, commit_* writes that block into the LDS tile -- the loads of tile t+1 are in flight while the matrix
// instructions of tile t run, and only the LDS stores sit between two tiles' compute phases.  Same element maps, clamping and
// select-to-zero as the one-piece loaders above (bit-identical tiles).  The halo columns are zeroed once per workgroup by the caller.
// ------------------------------------------------------------------------------------------------
#ifndef CGS_ELEMS_GUARD
#define CGS_ELEMS_GUARD 1
#endif
template <int E, int THREADS, class F>
__device__ __forceinline__ void for_elems_it(int tid, F f) {
    constexpr int IT = (E + THREADS - 1) / THREADS;
    static_for<IT>([&](auto I) {
        constexpr int it = decltype(I)::value;
        int e = tid + it * THREADS;
#if CGS_ELEMS_GUARD
        // the partial last round (features.0's strips: 288 groups = one full round + 32 threads) is skipped by the waves that have no element in it --
        // run with clamped indices by every wave it was half of the staging instructions of three waves in four (round 5)
        if constexpr ((it + 1) * THREADS <= E) f(std::integral_constant<int, it>{}, e);
        else if (e < E) f(std::integral_constant<int, it>{}, e);
#else
        f(std::integral_constant<int, it>{}, e < E ? e : E - 1);
#endif
    });
}

template <class G, int PA> struct FetchF32 { static constexpr int E = G::IMGS * G::TRA * G::W * PA, IT = (E + G::LT - 1) / G::LT; float4 v[IT]; };
template <class G, int PA>
__device__ __forceinline__ void fetch_a_f32(FetchF32<G, PA>& R, const float4* __restrict__ src, int n0, int row0, int N, int tid) {
    for_elems_it<FetchF32<G, PA>::E, G::LT>(tid, [&](auto I, int e) {
        int p = e % PA, x = (e / PA) % G::W, r = (e / (PA * G::W)) % G::TRA, img = e / (PA * G::W * G::TRA);
        int n = n0 + img, y = row0 + r - 1;
        bool in = n < N && y >= 0 && y < G::H;
        R.v[decltype(I)::value] = src[in ? ((n * G::H + y) * G::W + x) * PA + p : 0];
    });
}
template <class G, int PA>
__device__ __forceinline__ void commit_a_f32(const FetchF32<G, PA>& R, float4* ldsA, int n0, int row0, int N, int tid) {
    for_elems_it<FetchF32<G, PA>::E, G::LT>(tid, [&](auto I, int e) {
        int p = e % PA, x = (e / PA) % G::W, r = (e / (PA * G::W)) % G::TRA, img = e / (PA * G::W * G::TRA);
        int n = n0 + img, y = row0 + r - 1;
        bool in = n < N && y >= 0 && y < G::H;
        ldsA[ldsA_idx<G, PA>(p, img, r, x + 1)] = in ? R.v[decltype(I)::value] : f4zero();
    });
}

template <class G> struct FetchU8 { static constexpr int E = G::IMGS * G::TRA * (G::W / 4), IT = (E + G::LT - 1) / G::LT; uint32_t d[IT][3]; };
template <class G>
__device__ __forceinline__ void fetch_a_u8c3(FetchU8<G>& R, const uint32_t* __restrict__ src, int n0, int row0, int N, int tid) {
    constexpr int GW = G::W / 4;
    for_elems_it<FetchU8<G>::E, G::LT>(tid, [&](auto I, int e) {
        constexpr int it = decltype(I)::value;
        int g = e % GW, r = (e / GW) % G::TRA, img = e / (GW * G::TRA);
        int n = n0 + img, y = row0 + r - 1;
        bool in = n < N && y >= 0 && y < G::H;
        int gi = in ? ((n * G::H + y) * G::W + g * 4) * 3 / 4 : 0;
        R.d[it][0] = src[gi]; R.d[it][1] = src[gi + 1]; R.d[it][2] = src[gi + 2];
    });
}
template <class G>
__device__ __forceinline__ void commit_a_u8c3(const FetchU8<G>& R, float4* ldsA, int n0, int row0, int N, int tid) {
    constexpr int GW = G::W / 4;
    const float s = 1.f / 255.f;
    for_elems_it<FetchU8<G>::E, G::LT>(tid, [&](auto I, int e) {
        constexpr int it = decltype(I)::value;
        int g = e % GW, r = (e / GW) % G::TRA, img = e / (GW * G::TRA);
        int n = n0 + img, y = row0 + r - 1;
        bool in = n < N && y >= 0 && y < G::H;
        uint32_t d0 = in ? R.d[it][0] : 0u, d1 = in ? R.d[it][1] : 0u, d2 = in ? R.d[it][2] : 0u;
        float4 p0 = make_float4((d0 & 255) * s, ((d0 >> 8) & 255) * s, ((d0 >> 16) & 255) * s, 0.f);
        float4 p1 = make_float4((d0 >> 24) * s, (d1 & 255) * s, ((d1 >> 8) & 255) * s, 0.f);
        float4 p2 = make_float4(((d1 >> 16) & 255) * s, (d1 >> 24) * s, (d2 & 255) * s, 0.f);
        float4 p3 = make_float4(((d2 >> 8) & 255) * s, ((d2 >> 16) & 255) * s, (d2 >> 24) * s, 0.f);
        int base = (img * G::TRA + r) * G::PWA;
        ldsA[base + G::pc(g * 4 + 1)] = p0;
        ldsA[base + G::pc(g * 4 + 2)] = p1;
        ldsA[base + G::pc(g * 4 + 3)] = p2;
        ldsA[base + G::pc(g * 4 + 4)] = p3;
    });
}

template <class G> struct FetchMix { static constexpr int E = G::IMGS * G::TRA * (G::W / 4), IT = (E + G::LT - 1) / G::LT; uint32_t a[IT][3], b[IT][3]; float4 z[IT]; };
template <class G>
__device__ __forceinline__ void fetch_a_mix(FetchMix<G>& R, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                            const float4* __restrict__ z, int n_a, int n0, int row0, int N, int tid) {
    constexpr int GW = G::W / 4;
    for_elems_it<FetchMix<G>::E, G::LT>(tid, [&](auto I, int e) {
        constexpr int it = decltype(I)::value;
        int g = e % GW, r = (e / GW) % G::TRA, img = e / (GW * G::TRA);
        int n = n0 + img, y = row0 + r - 1;
        bool in = n < N && y >= 0 && y < G::H;
        int src_n = in ? (n >= n_a ? n - n_a : n) : 0;
        int pg = in ? (src_n * G::H + y) * G::W / 4 + g : 0;
        R.a[it][0] = a[3 * pg]; R.a[it][1] = a[3 * pg + 1]; R.a[it][2] = a[3 * pg + 2];
        R.b[it][0] = b[3 * pg]; R.b[it][1] = b[3 * pg + 1]; R.b[it][2] = b[3 * pg + 2];
        R.z[it] = z[pg];
    });
}
template <class G>
__device__ __forceinline__ void commit_a_mix(const FetchMix<G>& R, float4* ldsA, int n_a, int n0, int row0, int N, int tid) {
    constexpr int GW = G::W / 4;
    const float s = 1.f / 255.f;
    for_elems_it<FetchMix<G>::E, G::LT>(tid, [&](auto I, int e) {
        constexpr int it = decltype(I)::value;
        int g = e % GW, r = (e / GW) % G::TRA, img = e / (GW * G::TRA);
        int n = n0 + img, y = row0 + r - 1;
        bool in = n < N && y >= 0 && y < G::H;
        const bool inj = n >= n_a;
        uint32_t a0 = R.a[it][0], a1 = R.a[it][1], a2 = R.a[it][2], b0 = R.b[it][0], b1 = R.b[it][1], b2 = R.b[it][2];
        const float4 zz = R.z[it];
        if (inj) { uint32_t t; t = a0; a0 = b0; b0 = t; t = a1; a1 = b1; b1 = t; t = a2; a2 = b2; b2 = t; }
        auto mixv = [&](uint32_t da, uint32_t db, int sh, float zi) {
            float av = ((da >> sh) & 255) * s, bv = ((db >> sh) & 255) * s;
            return av * (1.f - zi) + zi * bv;
        };
        float4 p0 = make_float4(mixv(a0, b0, 0, zz.x), mixv(a0, b0, 8, zz.x), mixv(a0, b0, 16, zz.x), 0.f);
        float4 p1 = make_float4(mixv(a0, b0, 24, zz.y), mixv(a1, b1, 0, zz.y), mixv(a1, b1, 8, zz.y), 0.f);
        float4 p2 = make_float4(mixv(a1, b1, 16, zz.z), mixv(a1, b1, 24, zz.z), mixv(a2, b2, 0, zz.z), 0.f);
        float4 p3 = make_float4(mixv(a2, b2, 8, zz.w), mixv(a2, b2, 16, zz.w), mixv(a2, b2, 24, zz.w), 0.f);
        int base = (img * G::TRA + r) * G::PWA;
        ldsA[base + G::pc(g * 4 + 1)] = in ? p0 : f4zero();
        ldsA[base + G::pc(g * 4 + 2)] = in ? p1 : f4zero();
        ldsA[base + G::pc(g * 4 + 3)] = in ? p2 : f4zero();
        ldsA[base + G::pc(g * 4 + 4)] = in ? p3 : f4zero();
    });
}

// pooled gradient + argmax nibbles (HALO = 1 data-gradient tile); DIFF: also image n0 + n_off's (subtracted when sub)
template <class G, int PA, bool DIFF> struct FetchPool {
    static constexpr int JR = G::RQ + 2, E = G::IMGS * JR * (G::W / 2) * PA, IT = (E + G::LT - 1) / G::LT;
    float4 v[IT]; uint32_t nib[IT]; float4 u[DIFF ? IT : 1]; uint32_t nbj[DIFF ? IT : 1];
};
template <class G, int PA, bool DIFF>
__device__ __forceinline__ void fetch_poolexp(FetchPool<G, PA, DIFF>& R, const float4* __restrict__ dp, const uint32_t* __restrict__ am,
                                              int n0, int n_off, bool sub, int row0, int N, int tid) {
    constexpr int HP = G::H / 2, WP = G::W / 2, JR = G::RQ + 2, AMW = (PA + 1) / 2;
    const int pr0 = row0 / 2 - 1;
    for_elems_it<FetchPool<G, PA, DIFF>::E, G::LT>(tid, [&](auto I, int e) {
        constexpr int it = decltype(I)::value;
        int p = e % PA, px = (e / PA) % WP, j = (e / (PA * WP)) % JR, img = e / (PA * WP * JR);
        int n = n0 + img, pr = pr0 + j;
        bool in = n < N && pr >= 0 && pr < HP;
        int pi = in ? (n * HP + pr) * WP + px : 0;
        R.v[it] = dp[pi * PA + p];
        R.nib[it] = am[pi * AMW + (p >> 1)];
        if constexpr (DIFF) {
            int pj = (in && sub) ? ((n + n_off) * HP + pr) * WP + px : 0;
            R.u[it] = dp[pj * PA + p];
            R.nbj[it] = am[pj * AMW + (p >> 1)];
        }
    });
}
template <class G, int PA, bool DIFF, class IdxF>
__device__ __forceinline__ void commit_poolexp(const FetchPool<G, PA, DIFF>& R, float4* lds, int n0, bool sub, int row0, int N, int tid,
                                               IdxF idx, int dump) {
    constexpr int HP = G::H / 2, WP = G::W / 2, JR = G::RQ + 2;
    const int pr0 = row0 / 2 - 1;
    for_elems_it<FetchPool<G, PA, DIFF>::E, G::LT>(tid, [&](auto I, int e) {
        constexpr int it = decltype(I)::value;
        int p = e % PA, px = (e / PA) % WP, j = (e / (PA * WP)) % JR, img = e / (PA * WP * JR);
        int n = n0 + img, pr = pr0 + j;
        bool in = n < N && pr >= 0 && pr < HP;
        uint32_t nib = (R.nib[it] >> ((p & 1) * 16)) & 0xFFFFu;
        nib = in ? nib : 0xFFFFu;
        uint32_t nbj = 0xFFFFu;
        if constexpr (DIFF) {
            nbj = (R.nbj[it] >> ((p & 1) * 16)) & 0xFFFFu;
            nbj = (in && sub) ? nbj : 0xFFFFu;
        }
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            int r = 2 * j + (pos >> 1) - 1;
            bool ok = r >= 0 && r < G::TH + 2;
            float4 a = nib_select(R.v[it], nib, pos);
            if constexpr (DIFF) {
                const float4 b = nib_select(R.u[it], nbj, pos);
                a = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
            }
            lds[ok ? idx(p, img, r, 2 * px + (pos & 1)) : dump] = a;
        }
    });
}

template <class G, int PB> struct FetchBHalf { static constexpr int E = G::IMGS * G::TRB * G::PWB * PB, IT = (E + G::LT - 1) / G::LT; float4 v[IT]; };
template <class G, int PB>
__device__ __forceinline__ void fetch_b_half(FetchBHalf<G, PB>& R, const float4* __restrict__ src, int n0, int row0, int N, int tid) {
    const int sy0 = row0 / 2;
    for_elems_it<FetchBHalf<G, PB>::E, G::LT>(tid, [&](auto I, int e) {
        int p = e % PB, c = (e / PB) % G::PWB, r = (e / (PB * G::PWB)) % G::TRB, img = e / (PB * G::PWB * G::TRB);
        int n = n0 + img, sy = sy0 + r - 1, sx = c - 1;
        bool in = n < N && sy >= 0 && sy < G::QH && sx >= 0 && sx < G::QW;
        R.v[decltype(I)::value] = src[in ? ((n * G::QH + sy) * G::QW + sx) * PB + p : 0];
    });
}
template <class G, int PB>
__device__ __forceinline__ void commit_b_half(const FetchBHalf<G, PB>& R, float4* ldsB, int n0, int row0, int N, int tid) {
    const int sy0 = row0 / 2;
    for_elems_it<FetchBHalf<G, PB>::E, G::LT>(tid, [&](auto I, int e) {
        int p = e % PB, c = (e / PB) % G::PWB, r = (e / (PB * G::PWB)) % G::TRB, img = e / (PB * G::PWB * G::TRB);
        int n = n0 + img, sy = sy0 + r - 1, sx = c - 1;
        bool in = n < N && sy >= 0 && sy < G::QH && sx >= 0 && sx < G::QW;
        ldsB[((p * G::IMGS + img) * G::TRB + r) * G::PWB + c] = in ? R.v[decltype(I)::value] : f4zero();
    });
}

// ------------------------------------------------------------------------------------------------
// Compute core: accumulate one float4 plane (4 input channels) of a 4x4 receptive field into the
// 2x2 x OCB accumulator block.  WF(tap, ci, oc) returns the (wave-uniform) weight.
// ------------------------------------------------------------------------------------------------
template <int OCB, int NCH, class WF>
__device__ __forceinline__ void fma_plane(float (&acc)[4][OCB], const float4 (&pt)[4][4], WF wf, int ci0, int oc0) {
    // 9*NCH steps of (one tap, one input channel): OCB weights -> 4*OCB FMAs.  The weights of step s+1 are
    // read (LDS broadcast) before the FMAs of step s are issued; the compiler barrier after each step keeps
    // the reads of step s+2 from being hoisted further (bounded registers, one step of latency cover).
    constexpr int NS = 9 * NCH;
    float w[2][OCB];
#pragma unroll
    for (int o = 0; o < OCB; ++o) w[0][o] = wf(0, ci0, oc0 + o);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int tap = s / NCH, c = s % NCH, ky = tap / 3, kx = tap % 3;
        if (s + 1 < NS) {
#pragma unroll
            for (int o = 0; o < OCB; ++o) w[(s + 1) & 1][o] = wf((s + 1) / NCH, ci0 + (s + 1) % NCH, oc0 + o);
        }
#pragma unroll
        for (int oy = 0; oy < 2; ++oy)
#pragma unroll
            for (int ox = 0; ox < 2; ++ox) {
                float x = f4get(pt[oy + ky][ox + kx], c);
#pragma unroll
                for (int o = 0; o < OCB; ++o) acc[oy * 2 + ox][o] = fmaf(x, w[s & 1][o], acc[oy * 2 + ox][o]);
            }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------------------
// The same step on the matrix cores: v_mfma_f32_4x4x1_16B_f32 with the A operand broadcast from one of its 16 blocks
// (cbsz = 4, abid = block) computes   D[lane][r] += A[4*abid + r] * B[lane]   (r = 0..3): with B = the lane's input value of
// one (tap, input channel) and A = the four weights of one output-channel group, ONE instruction is the quad position's 4
// FMAs of that step for all 64 lanes, at the full fp32 rate (tools/mfma4_peak.hip: 150 TFLOP/s sustained vs 115 for
// v_fma_f32) with no padding rows or columns at 8 (or 3) channels.  A weight REGISTER holds 16 blocks = 16 (tap, channel)
// steps of a channel group: lane 4*b + i of register k carries w[step 16*k + b][group's channel i]; the whole 8 -> 8 layer
// is 9 registers.  An exact fp32 FMA chain in the same (tap, channel) order as fma_plane.
// ------------------------------------------------------------------------------------------------
typedef float frag4 __attribute__((ext_vector_type(4)));

// steps (tap, ci) are numbered tap * CIN + ci over ALL input channels of the layer (both sources)
template <int OCG, int NCH, int CIN, int CI0, int NREG>
__device__ __forceinline__ void mfma_plane(frag4 (&acc)[4][OCG], const float4 (&pt)[4][4], const float (&wreg)[OCG][NREG]) {
    static_for<9 * NCH>([&](auto S) {
        constexpr int s = decltype(S)::value, tap = s / NCH, c = s % NCH, ky = tap / 3, kx = tap % 3;
        constexpr int step = tap * CIN + CI0 + c, reg = step / 16, abid = step % 16;
#pragma unroll
        for (int oy = 0; oy < 2; ++oy)
#pragma unroll
            for (int ox = 0; ox < 2; ++ox) {
                const float x = f4get(pt[oy + ky][ox + kx], c);
#pragma unroll
                for (int g = 0; g < OCG; ++g)
                    acc[oy * 2 + ox][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[g][reg], x, acc[oy * 2 + ox][g], 4, abid, 0);
            }
    });
}

template <int OCB, class WF>
__device__ __forceinline__ void fma_scalar(float (&acc)[4][OCB], const float (&pt)[4][4], WF wf, int oc0) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            float wv[OCB];
#pragma unroll
            for (int o = 0; o < OCB; ++o) wv[o] = wf(ky * 3 + kx, 0, oc0 + o);
#pragma unroll
            for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                for (int ox = 0; ox < 2; ++ox) {
                    float x = pt[oy + ky][ox + kx];
#pragma unroll
                    for (int o = 0; o < OCB; ++o) acc[oy * 2 + ox][o] = fmaf(x, wv[o], acc[oy * 2 + ox][o]);
                }
        }
}

// Read the 4x4 receptive field of a quad from a source-A plane.
template <class G>
__device__ __forceinline__ void read_patch_a(float4 (&pt)[4][4], const float4* ldsA, int plane, const QuadPos& q,
                                             const int (&pcx)[4]) {
    int base = ((plane * G::IMGS + q.img_l) * G::TRA + 2 * q.qy_l) * G::PWA;
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) pt[dy][dx] = ldsA[base + dy * G::PWA + pcx[dx]];
}

// Receptive field of a quad in a nearest-upsampled source: 3x3 source pixels, patch rows/cols {0,1,1,2}.
template <class G>
__device__ __forceinline__ void read_patch_b2(float4 (&pt)[4][4], const float4* ldsB, int plane, const QuadPos& q) {
    float4 s[3][3];
    int base = ((plane * G::IMGS + q.img_l) * G::TRB + q.qy_l) * G::PWB + q.qx;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) s[j][i] = ldsB[base + j * G::PWB + i];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) pt[dy][dx] = s[(dy + 1) >> 1][(dx + 1) >> 1];
}

// x4 upsampling of a 1x1 source to the 4x4 map: every in-image position sees the same pixel.
template <class G>
__device__ __forceinline__ void read_patch_b4(float4 (&pt)[4][4], const float4* ldsB, int plane, const QuadPos& q) {
    float4 s = ldsB[plane * G::IMGS + q.img_l];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            int y = 2 * q.qy_l + dy - 1, x = 2 * q.qx + dx - 1;
            bool in = (y >= 0) && (y < G::H) && (x >= 0) && (x < G::W);
            pt[dy][dx] = in ? s : f4zero();
        }
}

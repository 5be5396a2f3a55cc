// The mask head forward in ONE kernel (round 3):  Upsample + cat(X, .) + masker.0 (11 -> 16, 3x3) + LeakyReLU + masker.2
// (16 -> 1, 3x3) + Sigmoid   (nets.py:519-521, 488-491), one workgroup per image, its four 16-row strips one after the other.
//
// Everything multiplies on v_mfma_f32_4x4x1_16B_f32 with the A operand broadcast from one block (conv_tile.h, mfma_plane):
// lane = one 2x2 quad of output pixels (= one pixel of the half-resolution decoder output o0), registers = output channels.
//   * image channels: 9 taps x 3 channels x 4 channel groups per quad position (432 instructions per quad);
//   * upsampled channels with the nearest-upsample FOLDED into the weights: quad position (oy, ox) sees only 2 x 2 pixels of
//     o0, rows {qy - 1 + oy, qy + oy}, with the taps that fall on the same pixel pre-summed (rows: oy = 0: {k0}, {k1 + k2};
//     oy = 1: {k0 + k1}, {k2}): 4 x 8 x 4 = 128 instead of 9 x 8 x 4 = 288 instructions per position (512 per quad); the pre-summed
//     weights of a position are 2 registers per channel group (32 in all); exact up to the order of the sums, zero padding
//     included (a tap that leaves the image is alone in its group);
//   * masker.2 without ever re-reading h: the lane holds h of its four pixels in registers (the B operand), one instruction
//     forms 4 of the 9 products  c[p][tap] = sum_ch h[p][ch] w2[tap][ch]  = what pixel p adds to output pixel p - tap (192 per
//     quad).  A thread pre-sums its pixels' contributions per output pixel of the 4x4 window around its quad, the windows
//     meet through a 16-float-per-quad LDS array, and a thread finalises rows {2qy - 1, 2qy}: these need the quad row above
//     (the previous strip's last quad row is carried in LDS) but never the one below.  Fixed summation order, no atomics.
// h (134 MB at N = 512) is written once for the backward pass and never read here; Z and the per-image (sum |z|, sum z^2)
// come out of the same launch.  Replaces mask0_fwd_kernel + conv3x3_kernel<FMask2> on the training path.
#include "conv_tile.h"

struct MaskFwdParams {
    const void* img; const float* o0;
    const float* w0; const float* b0; const float* w2; const float* b2;
    float* h; float* z; float* zpart;
    int n;
    unsigned long long* dbg;
    const float* w0_pack;         // optional: the weight registers below, built once by the launch before this one (tail.hip: mask0_pack_weights)      // debug: per-workgroup phase time stamps (tools/maskfwd_stamps.py), NULL in the product path
};

namespace {
using MG = Geo<64, 64, 256, 1>;                 // 8 quad rows x 32 quads per strip, 4 strips per image
constexpr int kPSlots = 9;                       // quad-row slots of the window array: 0 = carried from the previous strip, 1..8
constexpr int kWinPitch = 33;
unsigned long long* g_maskfwd_stamps = nullptr;
}

#define MF_STAMP(k) do { if (CGS_STAMP_PTR(P.dbg) && tid == 0) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 64 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#ifdef CGS_DEBUG_STAMPS
extern "C" int dbg_maskfwd_stamps(unsigned long long* stamps) { g_maskfwd_stamps = stamps; return CGS_OK; }
#endif

template <int SRC>     // SRC_U8C3 / SRC_F32C3
__global__ void __launch_bounds__(256, SRC == SRC_U8C3 ? 2 : 1) mask_fwd_kernel(MaskFwdParams P) {     // (fp32 frames: 9 more staging registers per pixel group than uint8 -- at two workgroups per SIMD they spilled 16 VGPRs)
    if (CGS_KARG_PREFETCH) cgs_kernarg_prefetch<sizeof(MaskFwdParams)>();
    using G = MG;
    // tiles: image strip (float4 planes, conv_tile.h layout) | o0 strip at its own resolution; after a strip's MFMAs the same
    // 32 KB are the waves' private areas for the h transposition (8 KB each)
    constexpr int A4 = (G::TRA * G::PWA + 1 + 3) / 4 * 4, B4 = 2 * G::TRB * G::PWB;
    static_assert(A4 + B4 <= 2048, "tiles fit the 32 KB staging area");
    __shared__ __attribute__((aligned(16))) float4 stage4[2048];
    float4* const ldsA = stage4;
    float4* const ldsB = stage4 + A4;
    __shared__ float win[kPSlots * 16 * kWinPitch];      // [slot][window element][quad column]: lane-contiguous, conflict-free
    __shared__ float zred[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x, N = P.n;
    const int lb = lane >> 2, li = lane & 3;
    MF_STAMP(0);

    // ---- tile loads split into fetch (global -> registers, raw) and commit (registers -> LDS, decoded): conv_tile.h's
    //      load_a_u8c3 / load_a_f32c3 / load_b_half, one strip ahead ----
    constexpr int GW = G::W / 4, EA = G::TRA * GW, ITA = (EA + 255) / 256;          // groups of 4 pixels
    constexpr int EB = G::TRB * G::PWB * 2, ITB = (EB + 255) / 256;                 // float4 of the o0 strip (2 planes)
    uint32_t ra[ITA][(SRC == SRC_U8C3) ? 3 : 12];
    float4 rb[ITB];
    auto fetch = [&](int row0) {
#pragma unroll
        for (int it = 0; it < ITA; ++it) {
            int e = tid + 256 * it; e = e < EA ? e : EA - 1;
            const int g = e % GW, r = e / GW, y = row0 + r - 1;
            const bool in = y >= 0 && y < G::H;
            const int gi = in ? ((n * G::H + y) * G::W + g * 4) * 3 / 4 : 0;
            if constexpr (SRC == SRC_U8C3) {
                const uint32_t* src = (const uint32_t*)P.img;
                ra[it][0] = src[gi]; ra[it][1] = src[gi + 1]; ra[it][2] = src[gi + 2];
            } else {
                const float4* src = (const float4*)P.img;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float4 v = src[gi + k];
                    ra[it][4 * k] = __float_as_uint(v.x); ra[it][4 * k + 1] = __float_as_uint(v.y);
                    ra[it][4 * k + 2] = __float_as_uint(v.z); ra[it][4 * k + 3] = __float_as_uint(v.w);
                }
            }
        }
        const int sy0 = row0 / 2;
#pragma unroll
        for (int it = 0; it < ITB; ++it) {
            int e = tid + 256 * it; e = e < EB ? e : EB - 1;
            const int p = e % 2, c = (e / 2) % G::PWB, r = e / (2 * G::PWB);
            const int sy = sy0 + r - 1, sx = c - 1;
            const bool in = sy >= 0 && sy < G::QH && sx >= 0 && sx < G::QW;
            rb[it] = ((const float4*)P.o0)[in ? ((n * G::QH + sy) * G::QW + sx) * 2 + p : 0];
        }
    };
    auto commit = [&](int row0) {
        const float sc = 1.f / 255.f;
#pragma unroll
        for (int it = 0; it < ITA; ++it) {
            int e = tid + 256 * it; e = e < EA ? e : EA - 1;
            const int g = e % GW, r = e / GW, y = row0 + r - 1;
            const bool in = y >= 0 && y < G::H;
            float4 p0, p1, p2, p3;
            if constexpr (SRC == SRC_U8C3) {
                const uint32_t d0 = ra[it][0], d1 = ra[it][1], d2 = ra[it][2];
                p0 = make_float4((d0 & 255) * sc, ((d0 >> 8) & 255) * sc, ((d0 >> 16) & 255) * sc, 0.f);
                p1 = make_float4((d0 >> 24) * sc, (d1 & 255) * sc, ((d1 >> 8) & 255) * sc, 0.f);
                p2 = make_float4(((d1 >> 16) & 255) * sc, (d1 >> 24) * sc, (d2 & 255) * sc, 0.f);
                p3 = make_float4(((d2 >> 8) & 255) * sc, ((d2 >> 16) & 255) * sc, (d2 >> 24) * sc, 0.f);
            } else {
                auto f = [&](int k) { return __uint_as_float(ra[it][k]); };
                p0 = make_float4(f(0), f(1), f(2), 0.f); p1 = make_float4(f(3), f(4), f(5), 0.f);
                p2 = make_float4(f(6), f(7), f(8), 0.f); p3 = make_float4(f(9), f(10), f(11), 0.f);
            }
            const int base = r * G::PWA;
            ldsA[base + G::pc(g * 4 + 1)] = in ? p0 : f4zero();
            ldsA[base + G::pc(g * 4 + 2)] = in ? p1 : f4zero();
            ldsA[base + G::pc(g * 4 + 3)] = in ? p2 : f4zero();
            ldsA[base + G::pc(g * 4 + 4)] = in ? p3 : f4zero();
        }
        zero_halo_cols<G, 1>(ldsA, tid);          // (the staging area was the h transposition buffer meanwhile)
        const int sy0 = row0 / 2;
#pragma unroll
        for (int it = 0; it < ITB; ++it) {
            int e = tid + 256 * it; e = e < EB ? e : EB - 1;
            const int p = e % 2, c = (e / 2) % G::PWB, r = e / (2 * G::PWB);
            const int sy = sy0 + r - 1, sx = c - 1;
            const bool in = sy >= 0 && sy < G::QH && sx >= 0 && sx < G::QW;
            ldsB[(p * G::TRB + r) * G::PWB + c] = in ? rb[it] : f4zero();
        }
    };
    fetch(0);                  // the first strip's loads fly while the weight registers are set up
    float wimg[4][2];        // [group][reg]: block b of reg k = step 16 k + b = tap * 3 + ci
    float wups[4][4][2];     // [position][group][reg]: block b of reg k: source pixel (ry = k, rx = b >> 3), channel b & 7
    if (P.w0_pack) {
        // built once per step by the previous launch: 40 coalesced loads instead of an LDS copy, a barrier and ~140 gathers per image
        const float* wp = P.w0_pack + lane;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int k = 0; k < 2; ++k) wimg[g][k] = wp[(g * 2 + k) * 64];
#pragma unroll
            for (int pos = 0; pos < 4; ++pos)
#pragma unroll
                for (int ry = 0; ry < 2; ++ry) wups[pos][g][ry] = wp[(8 + (pos * 4 + g) * 2 + ry) * 64];
        }
    } else {
    // ---- weight registers (once per image): masker.0's 1584 floats through LDS, each lane gathers / pre-sums its own ----
    float* wst = (float*)ldsA;
    for (int e = tid; e < 9 * 11 * 16; e += 256) wst[e] = P.w0[e];
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int step = 16 * k + lb;
            wimg[g][k] = step < 27 ? wst[((step / 3) * 11 + step % 3) * 16 + 4 * g + li] : 0.f;
        }
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            const int oy = pos >> 1, ox = pos & 1;
#pragma unroll
            for (int ry = 0; ry < 2; ++ry) {
                const int rx = lb >> 3, ci = lb & 7;
                // taps of row group (oy, ry): oy = 0: {0}, {1, 2};  oy = 1: {0, 1}, {2}; columns likewise
                const int ky0 = (oy == 0) ? (ry == 0 ? 0 : 1) : (ry == 0 ? 0 : 2), ky1 = (oy == 0) ? (ry == 0 ? 0 : 2) : (ry == 0 ? 1 : 2);
                const int kx0 = (ox == 0) ? (rx == 0 ? 0 : 1) : (rx == 0 ? 0 : 2), kx1 = (ox == 0) ? (rx == 0 ? 0 : 2) : (rx == 0 ? 1 : 2);
                // branch-free (a loop with lane-dependent bounds becomes a serial waterfall of LDS round trips): the group is
                // {k0} or {k0, k0 + 1} per dimension
                auto wv = [&](int ky, int kx) { return wst[((ky * 3 + kx) * 11 + 3 + ci) * 16 + 4 * g + li]; };
                const float w00 = wv(ky0, kx0), w01 = wv(ky0, kx1), w10 = wv(ky1, kx0), w11 = wv(ky1, kx1);
                const bool my = ky1 > ky0, mx = kx1 > kx0;
                float s = w00;
                s += mx ? w01 : 0.f;
                s += my ? w10 : 0.f;
                s += (mx && my) ? w11 : 0.f;
                wups[pos][g][ry] = s;
            }
            __builtin_amdgcn_sched_barrier(0);     // (bounded live range of the gathered values: the first strip's loads are in flight too)
        }
    }
    }
    float w2r[3];            // [tap group]: block b = channel b: w2[tap 4 g + i][channel]
#pragma unroll
    for (int g = 0; g < 3; ++g) w2r[g] = (4 * g + li < 9) ? P.w2[(4 * g + li) * 16 + lb] : 0.f;
    const cgs_cptr b0c = cgs_to_const(P.b0);       // (uniform: scalar loads at the point of use, no vector registers held)
    const float b2v = cgs_to_const(P.b2)[0];
    __syncthreads();

    MF_STAMP(1);
    const QuadPos q0 = quad_pos<G>(tid, 0);
    const int qy_l = q0.qy_l, qx = q0.qx;
    int pcx[4];
#pragma unroll
    for (int dx = 0; dx < 4; ++dx) pcx[dx] = G::pc(2 * qx + dx);
    float zs1 = 0.f, zs2 = 0.f;

#pragma unroll 1
    for (int strip = 0; strip < G::STRIPS; ++strip) {
        const int row0 = strip * G::TH;
        QuadPos q = q0;
        q.n = n; q.row0 = row0;
        MF_STAMP(2 + 10 * strip);
        commit(row0);
        __syncthreads();
        MF_STAMP(3 + 10 * strip);
        if (strip + 1 < G::STRIPS) fetch(row0 + G::TH);      // the next strip's loads fly during this strip's MFMAs (and are older
                                                             // than this strip's h stores: vmcnt retires them first)

        frag4 acc[4][4];           // start from the bias: one add per output saved
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[i][g] = frag4{b0c[4 * g], b0c[4 * g + 1], b0c[4 * g + 2], b0c[4 * g + 3]};
        {   // image channels
            float4 pt[4][4];
            read_patch_a<G>(pt, ldsA, 0, q, pcx);
            mfma_plane<4, 3, 3, 0, 2>(acc, pt, wimg);
        }
        MF_STAMP(4 + 10 * strip);
        // upsampled channels, folded: position (oy, ox) x source pixel (ry, rx) x 8 channels
        static_for<2>([&](auto PL) {
            constexpr int p = decltype(PL)::value;
            float4 s[3][3];
            const int base = (p * G::TRB + qy_l) * G::PWB + qx;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) s[j][i] = ldsB[base + j * G::PWB + i];
            static_for<16>([&](auto T) {
                constexpr int t = decltype(T)::value, ry = t >> 3, rx = (t >> 2) & 1, c = t & 3;
                constexpr int abid = rx * 8 + 4 * p + c;
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                    const float x = f4get(s[(pos >> 1) + ry][(pos & 1) + rx], c);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[pos][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wups[pos][g][ry], x, acc[pos][g], 4, abid, 0);
                }
            });
        });

        // ---- bias + LeakyReLU; h -> memory (backward needs it) ----
        float hv[4][16];
        const int y0 = row0 + 2 * qy_l, x0 = 2 * qx;
#pragma unroll
        for (int pos = 0; pos < 4; ++pos)
#pragma unroll
            for (int c = 0; c < 16; ++c) { const float v = acc[pos][c >> 2][c & 3]; hv[pos][c] = fmaxf(v, 0.01f * v); }    // LeakyReLU(0.01)
        MF_STAMP(5 + 10 * strip);
        __syncthreads();                               // every wave has read its patches: the staging area is free
        MF_STAMP(6 + 10 * strip);
        if (P.h) {
            // A lane holds 128 contiguous bytes of an image row (2 pixels x 16 channels); stored directly, every store
            // instruction would touch 64 different 128-byte lines (measured: +27 us).  Through the wave's private 8 KB of LDS (8
            // float4 per lane, XOR-swizzled so writes and reads are conflict-free) each instruction writes 1 KB contiguous.
            float4* sw4 = stage4 + wave * 512;
            const int pp = lane;                       // pixel pair = (quad row of the wave, quad column)
#pragma unroll
            for (int oy = 0; oy < 2; ++oy) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int pos = oy * 2 + (j >> 2), c0 = 4 * (j & 3);
                    sw4[pp * 8 + (j ^ (pp & 7))] = make_float4(hv[pos][c0], hv[pos][c0 + 1], hv[pos][c0 + 2], hv[pos][c0 + 3]);
                }
                // (LDS executes a wave's instructions in order: the reads below see the other lanes' writes)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = lane + 64 * k, rp = i >> 3, j = i & 7;            // rp: pixel pair, k >> 2 = its quad row
                    const float4 v = sw4[rp * 8 + (j ^ (rp & 7))];
                    const int y = row0 + 2 * (2 * wave + (k >> 2)) + oy;
                    ((float4*)P.h)[((size_t)(n * 64 + y) * 64) * 4 + (rp & 31) * 8 + j] = v;
                }
            }
        }
        MF_STAMP(7 + 10 * strip);
        // ---- masker.2: c[pos][tap] = sum_ch h[pos][ch] w2[tap][ch] ----
        frag4 c2[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 3; ++g) c2[i][g] = frag4{0.f, 0.f, 0.f, 0.f};
        static_for<16>([&](auto CH) {
            constexpr int ch = decltype(CH)::value;
#pragma unroll
            for (int pos = 0; pos < 4; ++pos)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    c2[pos][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(w2r[g], hv[pos][ch], c2[pos][g], 4, ch, 0);
        });
        MF_STAMP(8 + 10 * strip);
        // window sums: output pixel (wy, wx) in -1..2 relative to the quad's top-left pixel receives from quad pixel (oy, ox) the
        // product of tap (ky, kx) = (oy - wy + 1, ox - wx + 1)
        float sw[4][4];
#pragma unroll
        for (int wy = -1; wy <= 2; ++wy)
#pragma unroll
            for (int wx = -1; wx <= 2; ++wx) {
                float s = 0.f;
#pragma unroll
                for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                    for (int ox = 0; ox < 2; ++ox) {
                        const int ky = oy - wy + 1, kx = ox - wx + 1;
                        if (ky >= 0 && ky <= 2 && kx >= 0 && kx <= 2) {
                            const int tap = ky * 3 + kx;
                            s += c2[oy * 2 + ox][tap >> 2][tap & 3];
                        }
                    }
                sw[wy + 1][wx + 1] = s;
            }
        float* wown = win + (qy_l + 1) * 16 * kWinPitch + qx;       // element e of quad column qx + d: wown[e * kWinPitch + d]
#pragma unroll
        for (int e = 0; e < 16; ++e) wown[e * kWinPitch] = sw[e >> 2][e & 3];
        __syncthreads();
        MF_STAMP(9 + 10 * strip);
        // ---- finalise rows 2qy - 1 (wy = -1) and 2qy (wy = 0), columns 2qx (wx = 0) and 2qx + 1 (wx = 1) ----
        const bool has_top = strip > 0 || qy_l > 0;
        const float* wtop = win + qy_l * 16 * kWinPitch + qx;       // the quad above (slot qy_l; slot 0 = carried)
        // (branch-free: every read is issued -- redirected to the thread's own column where the neighbour does not exist -- and
        //  masked afterwards; conditional reads become exec-masked blocks with one LDS round trip each)
        float zz[3][2];
#pragma unroll
        for (int wy = -1; wy <= 1; ++wy) {
#pragma unroll
            for (int wx = 0; wx <= 1; ++wx) {
                const bool has_side = wx == 0 ? qx > 0 : qx < 31;    // the horizontal neighbour that reaches this column
                const int side = has_side ? (wx == 0 ? -1 : 1) : 0;
                const int wxs = wx == 0 ? 2 : -1;                    // this column in the neighbour's window
                float s = sw[wy + 1][wx + 1];
                const float a = wown[((wy + 1) * 4 + wxs + 1) * kWinPitch + side];
                s += has_side ? a : 0.f;
                if (wy <= 0) {
                    const float b = wtop[((wy + 3) * 4 + wx + 1) * kWinPitch];
                    const float c = wtop[((wy + 3) * 4 + wxs + 1) * kWinPitch + side];
                    s += has_top ? b : 0.f;
                    s += (has_top && has_side) ? c : 0.f;
                }
                zz[wy + 1][wx] = act_fwd<CGS_ACT_SIGMOID>(s + b2v);
            }
        }
        const bool last_row = strip == G::STRIPS - 1 && qy_l == 7;  // the image's last row has nothing below it: finalised here too
#pragma unroll
        for (int wy = -1; wy <= 1; ++wy) {
            const int y = y0 + wy;
            if (wy == 1 ? last_row : y >= 0) {
                *(float2*)(P.z + (size_t)(n * 64 + y) * 64 + x0) = make_float2(zz[wy + 1][0], zz[wy + 1][1]);
                zs1 += fabsf(zz[wy + 1][0]) + fabsf(zz[wy + 1][1]);
                zs2 += zz[wy + 1][0] * zz[wy + 1][0] + zz[wy + 1][1] * zz[wy + 1][1];
            }
        }
        MF_STAMP(10 + 10 * strip);
        __syncthreads();
        MF_STAMP(11 + 10 * strip);
        if (qy_l == 7) {                                             // carry this strip's last quad row
#pragma unroll
            for (int e = 0; e < 16; ++e) win[e * kWinPitch + qx] = sw[e >> 2][e & 3];
        }
        // (the next strip's tile loads are followed by a barrier before anybody reads slot 0)
    }
    // ---- per-image (sum |z|, sum z^2) for the L1 / L2 mask losses (main.py:421-429), fixed order ----
    if (P.zpart) {
        zs1 = wave_sum(zs1); zs2 = wave_sum(zs2);
        if (lane == 0) { zred[2 * wave] = zs1; zred[2 * wave + 1] = zs2; }
        __syncthreads();
        if (tid == 0) {
            P.zpart[2 * n] = (zred[0] + zred[2]) + (zred[4] + zred[6]);
            P.zpart[2 * n + 1] = (zred[1] + zred[3]) + (zred[5] + zred[7]);
        }
    }
}

// training form of the mask head forward: h [n,64,64,16], z [n,64,64], zpart [mask_train_partials(n)][2]
int mask_train_partials(int n) { return n; }
int mask_train_launch(int n, int img_kind, const void* img, const float* o0, const float* w0, const float* b0, const float* w2,
                      const float* b2, float* h, float* z, float* zpart, const float* w0_pack, hipStream_t st) {
    if (n <= 0) return CGS_OK;
    MaskFwdParams P{img, o0, w0, b0, w2, b2, h, z, zpart, n, g_maskfwd_stamps, w0_pack};
    if (img_kind == CGS_SRC_U8) hipLaunchKernelGGL(mask_fwd_kernel<SRC_U8C3>, dim3(n), dim3(256), 0, st, P);
    else hipLaunchKernelGGL(mask_fwd_kernel<SRC_F32C3>, dim3(n), dim3(256), 0, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

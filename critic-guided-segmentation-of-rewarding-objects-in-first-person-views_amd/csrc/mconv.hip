// 3x3 convolution (forward / data gradient) as an implicit GEMM on the matrix cores, for the layers where the
// GEMM shape fills an MFMA tile: 16 output channels with many input channels at tiny spatial size (dec_model.3
// forward: 48 -> 16 channels at 4x4, generic kernel below) and masker.0 forward (its own kernel at the end of the file).  (A data-gradient variant for
// dec_model.3 was measured slower than the shared-launch VALU kernel and is not kept.)
//
//   out[pixel][oc] = sum_k Xcol[pixel][k] * Wm[k][oc],  k = tap*PCI + channel   (v_mfma_f32_16x16x4_f32, exact fp32)
//
// A operand: 16 pixels x 4 k per instruction, read from an NHWC LDS tile with halo (concat of the skip input and
// the nearest-upsampled low-res input materialised, 3-channel images padded to 4): for a fixed k-step the address is
// (per-pixel base) + compile-time constant.  B operand: the weights, one register per (k-step, 16-channel block),
// loaded once per wave and kept for all tiles the (persistent) workgroup processes.
// The direct VALU kernels (conv_body.h) stay the choice for 8-channel layers, where half of an MFMA tile is padding.
#include "wgrad_body.h"   // WGeo, frag4, WSRC_*

struct MConvParams {
    const void* src_a;
    const float* src_b;
    const float* w;
    const float* bias;
    float* out;     // forward: output; dgrad: d_a
    float* out2;    // dgrad: d_b
    int n, ntiles;
};

enum { MEPI_PLAIN = 0 };

// C: G (WGeo), SRC, CA, CB, UPS, WT, WCI, WCO, NOUT, ACT, EPI, OUT_A
template <class C>
__global__ void __launch_bounds__(C::G::THREADS) mconv_kernel(MConvParams P) {
    using G = typename C::G;
    constexpr int SA = (C::CA + 3) / 4, SB = C::CB / 4, S = SA + SB, PCI = 4 * S;
    constexpr int NK = 9 * PCI / 4, NCB = (C::NOUT + 15) / 16;
    constexpr int NPIX = G::IMGS * G::TRA * G::PWA;
    constexpr int NT = G::IMGS * G::TH * G::W / 16;            // 16-pixel MFMA tiles per workgroup tile
    static_assert((G::TH * G::W) % 16 == 0 && (G::W >= 16 || (G::W == 4 && G::TH == 4)), "pixel tiling");
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float* xt = (float*)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int N = P.n;

    // ---- weights -> registers (B operand): lane (k = 4s+kq, oc = 16cb + l15) ----
    float wr[NK][NCB];
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        const int tap = (4 * s) / PCI, lch = (4 * s) % PCI + kq;     // PCI % 4 == 0: one tap per k-step
        int ci = lch < C::CA ? lch : ((lch >= 4 * SA) ? C::CA + (lch - 4 * SA) : -1);
        if (lch >= C::CA && lch < 4 * SA) ci = -1;                     // padding channel of a 3-channel image
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            int oc = cb * 16 + l15;
            float v = 0.f;
            if (ci >= 0 && oc < C::NOUT) {
                if constexpr (C::WT == 0) v = P.w[(tap * C::WCI + ci) * C::WCO + oc];
                else v = P.w[((8 - tap) * C::WCI + oc) * C::WCO + ci];
            }
            wr[s][cb] = v;
        }
    }

    for (int tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
        const int n0 = (G::IMGS == 1) ? tile / G::STRIPS : tile * G::IMGS;
        const int row0 = (G::IMGS == 1) ? (tile % G::STRIPS) * G::TH : 0;
        // ---- X tile (same layout as the weight-gradient kernels) ----
        if constexpr (C::CA % 4 == 0) {
            for_elems<NPIX * SA, G::THREADS>(tid, [&](int e) {
                int s = e % SA, c = (e / SA) % G::PWA, r = (e / (SA * G::PWA)) % G::TRA, img = e / (SA * G::PWA * G::TRA);
                int n = n0 + img, y = row0 + r - 1, x = c - 1;
                bool in = n < N && y >= 0 && y < G::H && x >= 0 && x < G::W;
                float4 v = ((const float4*)P.src_a)[in ? ((n * G::H + y) * G::W + x) * SA + s : 0];
                ((float4*)xt)[(e / SA) * S + s] = in ? v : f4zero();
            });
        } else {
            for_elems<NPIX, G::THREADS>(tid, [&](int e) {
                int c = e % G::PWA, r = (e / G::PWA) % G::TRA, img = e / (G::PWA * G::TRA);
                int n = n0 + img, y = row0 + r - 1, x = c - 1;
                bool in = n < N && y >= 0 && y < G::H && x >= 0 && x < G::W;
                int pix = in ? (n * G::H + y) * G::W + x : 0;
                float4 v;
                if constexpr (C::SRC == WSRC_U8) {
                    const uint32_t* s32 = (const uint32_t*)P.src_a;
                    int off = pix * 3, last = N * G::H * G::W * 3 / 4 - 1, d = off >> 2;
                    uint32_t lo = s32[d], hi = s32[d + 1 <= last ? d + 1 : last];
                    uint64_t both = (((uint64_t)hi << 32) | lo) >> ((off & 3) * 8);
                    const float sc = 1.f / 255.f;
                    const uint32_t b3 = (uint32_t)both;      // (32-bit conversions)
                    v = make_float4((b3 & 255u) * sc, ((b3 >> 8) & 255u) * sc, ((b3 >> 16) & 255u) * sc, 0.f);
                } else {
                    const float* sf = (const float*)P.src_a;
                    v = make_float4(sf[pix * 3], sf[pix * 3 + 1], sf[pix * 3 + 2], 0.f);
                }
                ((float4*)xt)[e * S] = in ? v : f4zero();
            });
        }
        if constexpr (SB > 0) {
            for_elems<NPIX * SB, G::THREADS>(tid, [&](int e) {
                int s = e % SB, c = (e / SB) % G::PWA, r = (e / (SB * G::PWA)) % G::TRA, img = e / (SB * G::PWA * G::TRA);
                int n = n0 + img, y = row0 + r - 1, x = c - 1;
                bool in = n < N && y >= 0 && y < G::H && x >= 0 && x < G::W;
                int gi;
                if constexpr (C::UPS == 2) gi = in ? ((n * G::QH + (y >> 1)) * G::QW + (x >> 1)) * SB + s : 0;
                else gi = in ? n * SB + s : 0;
                float4 v = ((const float4*)P.src_b)[gi];
                ((float4*)xt)[(e / SB) * S + SA + s] = in ? v : f4zero();
            });
        }
        __syncthreads();

        for (int t = wave; t < NT; t += G::NW) {
            // pixel of this lane's A row, and the 4 pixels of its D rows
            const int pa = 16 * t + l15;
            const int xa = pa % G::W, ya = (pa / G::W) % G::TH, ia = pa / (G::W * G::TH);
            const int abase = ((ia * G::TRA + ya) * G::PWA + xa) * PCI + kq;
            frag4 acc[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = frag4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NK; ++s) {
                constexpr int dummy = 0;
                const int tap = (4 * s) / PCI, lch0 = (4 * s) % PCI;
                float a = xt[abase + ((tap / 3) * G::PWA + (tap % 3)) * PCI + lch0];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wr[s][cb], acc[cb], 0, 0, 0);
                (void)dummy;
            }
            // ---- epilogue: D[row = 4kq + j][col = l15] ----
            {
                const float b = (l15 < C::NOUT) ? P.bias[l15] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int pd = 16 * t + 4 * kq + j;
                    int x = pd % G::W, yl = (pd / G::W) % G::TH, img = pd / (G::W * G::TH);
                    int n = n0 + img;
                    if (n < N && l15 < C::NOUT)
                        P.out[((size_t)(n * G::H + row0 + yl) * G::W + x) * C::NOUT + l15] = act_fwd<C::ACT>(acc[0][j] + b);
                }
            }
        }
        __syncthreads();
    }
}

struct MDec3F { using G = WGeo<4, 4, 4, 2, 128>; static constexpr int SRC = WSRC_F32, CA = 16, CB = 32, UPS = 4, WT = 0, WCI = 48, WCO = 16, NOUT = 16, ACT = CGS_ACT_NONE, EPI = MEPI_PLAIN, OUT_A = 0; };

// ------------------------------------------------------------------------------------------------
// masker.0 forward (nets.py:488-489: Upsample(o0) ++ image -> conv 11->16 -> LeakyReLU) with the nearest-upsample
// FOLDED into the weights.  For an output pixel of parity (py, px) the 3x3 taps over the upsampled o0 collapse onto a
// 2x2 neighbourhood of o0 itself:
//   out[y][x][oc] = sum_{tap, c<3} img[(y,x)+tap-1][c] * W[tap][c][oc]
//                 + sum_{a,b in {0,1}} sum_cb o0[(y>>1)+a-(1-py)][(x>>1)+b-(1-px)][cb] * W2[py][px][a][b][cb][oc],
//   W2[py][px][a][b] = sum_{ky in K(py,a)} sum_{kx in K(px,b)} W[ky][kx][3+cb],  K(0,0)={0} K(0,1)={1,2} K(1,0)={0,1} K(1,1)={2}
// (zero padding stays exact: a tap that leaves the image is the only member of its K set at that border).
// K per pixel: 9 taps x 4 (3 image channels + a zero) + 4 x 8 = 68 instead of 9 x 12 = 108: 17 MFMAs per 16 pixels, not 27.
// An MFMA tile = 16 same-parity pixels of one row (x = 2i + px), so its A addresses are (per-lane base) + constant.
// LDS: image tile [10][66][5] floats and o0 tile [6][34][10] floats at its own resolution: the odd pixel strides make
// the stride-2 pixel reads of a wave conflict-free.  Wave w owns the half hf = w&1 of rows 4*(w>>1) .. +3, both column
// parities (so it can store whole 2 KB row segments); the weights (9 image taps + 4 parities x 8) stay in registers
// for all tiles of the persistent workgroup.
// ------------------------------------------------------------------------------------------------
struct Mask0FwdParams {
    const void* img; const float* o0; const float* w; const float* bias; float* out;
    int n, ntiles;
};

template <int SRC>
__global__ void __launch_bounds__(256) mask0_fwd_kernel(Mask0FwdParams P) {
    constexpr int H = 64, W = 64, TH = 8, STRIPS = H / TH;
    constexpr int IR = TH + 2, IC = W + 2, IPS = 5;          // image tile rows, cols, pixel stride (floats)
    constexpr int LR = TH / 2 + 2, LC = W / 2 + 2, LPS = 10;  // o0 tile
    constexpr int NIMG = IR * IC, NLO = LR * LC * 2;
    __shared__ float ximg[IR * IC * IPS];
    __shared__ __attribute__((aligned(16))) float xo[LR * LC * LPS];
    __shared__ __attribute__((aligned(16))) float stage[4][32 * 16];   // per wave: 32 output pixels x 16 channels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int hf = wave & 1, rg = wave >> 1;   // this wave: columns 32*hf .. +31 of rows 4*rg .. 4*rg+3, both parities

    // ---- weights -> registers (B operand: k = kq, n = l15 = oc) ----
    float wimg[9], wo[2][2][2][2][2];     // [py][px][a][b][s]
#pragma unroll
    for (int t = 0; t < 9; ++t) wimg[t] = (kq < 3) ? P.w[(t * 11 + kq) * 16 + l15] : 0.f;
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int s4 = 0; s4 < 2; ++s4) {
                        float v = 0.f;
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {
                                const bool iny = py == 0 ? (a == 0 ? ky == 0 : ky >= 1) : (a == 0 ? ky <= 1 : ky == 2);
                                const bool inx = px == 0 ? (b == 0 ? kx == 0 : kx >= 1) : (b == 0 ? kx <= 1 : kx == 2);
                                if (iny && inx) v += P.w[((ky * 3 + kx) * 11 + 3 + 4 * s4 + kq) * 16 + l15];
                            }
                        wo[py][px][a][b][s4] = v;
                    }
    const float bias = P.bias[l15];

    for (int tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
        const int n0 = tile / STRIPS, row0 = (tile % STRIPS) * TH;
        // ---- tiles -> LDS ----
        for_elems<NIMG, 256>(tid, [&](int e) {
            int r = e / IC, c = e % IC;
            int y = row0 + r - 1, x = c - 1;
            bool in = y >= 0 && y < H && x >= 0 && x < W;
            int pix = in ? (n0 * H + y) * W + x : 0;
            float v0, v1, v2;
            if constexpr (SRC == WSRC_U8) {
                const uint32_t* s32 = (const uint32_t*)P.img;
                int off = pix * 3, last = P.n * H * W * 3 / 4 - 1, d = off >> 2;
                uint32_t lo = s32[d], hi = s32[d + 1 <= last ? d + 1 : last];
                uint64_t both = (((uint64_t)hi << 32) | lo) >> ((off & 3) * 8);
                const float sc = 1.f / 255.f;
                const uint32_t b3 = (uint32_t)both;      // (32-bit conversions)
                    v0 = (b3 & 255u) * sc; v1 = ((b3 >> 8) & 255u) * sc; v2 = ((b3 >> 16) & 255u) * sc;
            } else {
                const float* sf = (const float*)P.img;
                v0 = sf[pix * 3]; v1 = sf[pix * 3 + 1]; v2 = sf[pix * 3 + 2];
            }
            float* d = ximg + e * IPS;
            d[0] = in ? v0 : 0.f; d[1] = in ? v1 : 0.f; d[2] = in ? v2 : 0.f; d[3] = 0.f;
        });
        for_elems<NLO, 256>(tid, [&](int e) {
            int half = e & 1, pc = (e >> 1) % LC, pr = (e >> 1) / LC;
            int ly = row0 / 2 + pr - 1, lx = pc - 1;
            bool in = ly >= 0 && ly < H / 2 && lx >= 0 && lx < W / 2;
            float4 v = ((const float4*)P.o0)[in ? ((n0 * (H / 2) + ly) * (W / 2) + lx) * 2 + half : 0];
            v = in ? v : f4zero();
            float2* d = (float2*)(xo + (pr * LC + pc) * LPS + 4 * half);
            d[0] = make_float2(v.x, v.y); d[1] = make_float2(v.z, v.w);
        });
        __syncthreads();

        // ---- 4 rows x 2 column parities of this wave; operands of the next MFMA tile are read before the MFMAs of the
        //      current one (in-order issue: see mask_head.hip) ----
        const int ibase = ((4 * rg) * IC + 2 * (16 * hf + l15)) * IPS + kq;   // + ((r+ky)*IC + px + kx)*IPS
        const int obase = ((2 * rg) * LC + 16 * hf + l15) * LPS + kq;         // + (((r>>1)+a+py)*LC + px + b)*LPS + 4*s
        float ai[2][9], ao[2][8];
        auto ld = [&](int t, int buf) {         // MFMA tile t: row r = t>>1 of this wave's 4, parity px = t&1
            const int r = t >> 1, px = t & 1, py = r & 1;
#pragma unroll
            for (int k = 0; k < 9; ++k) ai[buf][k] = ximg[ibase + ((r + k / 3) * IC + px + k % 3) * IPS];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int a = q >> 2, b = (q >> 1) & 1, s4 = q & 1;
                ao[buf][q] = xo[obase + (((r >> 1) + a + py) * LC + px + b) * LPS + 4 * s4];
            }
        };
        ld(0, 0);
        frag4 res[2];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t + 1 < 8) ld(t + 1, (t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const int r = t >> 1, px = t & 1, py = r & 1;
            frag4 acc = frag4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[t & 1][k], wimg[k], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ao[t & 1][q], wo[py][px][q >> 2][(q >> 1) & 1][q & 1], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            res[px] = acc;
            if (px == 1) {
                // D[row = 4kq + j (pixel i)][col = l15 (oc)]: both parities of 16 i's = 32 consecutive pixels.  Through the
                // wave's LDS stage so that every lane stores 16 contiguous bytes and the wave 2 KB contiguous.
                float* st = stage[wave];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    st[(2 * (4 * kq + j)) * 16 + l15] = act_fwd<CGS_ACT_LRELU>(res[0][j] + bias);
                    st[(2 * (4 * kq + j) + 1) * 16 + l15] = act_fwd<CGS_ACT_LRELU>(res[1][j] + bias);
                }
                float4 v0 = ((const float4*)st)[lane], v1 = ((const float4*)st)[64 + lane];
                float4* o = (float4*)(P.out + ((size_t)(n0 * H + row0 + 4 * rg + r) * W + 32 * hf) * 16);
                o[lane] = v0; o[64 + lane] = v1;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Inference form of the whole mask head (nets.py:488-491, eval): masker.0 as above, but its output h (16 channels at
// 64x64, the largest activation of the network) goes to an LDS tile instead of memory, and masker.2 (conv 16 -> 1 + sigmoid)
// runs on that tile in the same workgroup.  h is needed only by the backward pass, so `-process` / `-eval` / infer() never
// write or re-read its 134 MB per 512 images.  A strip of 8 mask rows needs 10 rows of h (one halo row each side,
// recomputed by the neighbours): 25 % more MFMAs against two full passes over h saved.
// ------------------------------------------------------------------------------------------------
struct MaskInferParams {
    const void* img; const float* o0; const float* w0; const float* b0; const float* w2; const float* b2; float* z;
    int n, ntiles;
    // training form (TRAIN = true): h (the masker.0 output the backward pass needs) is stored once from the LDS tile, and every
    // tile leaves (sum |z|, sum z^2) for the L1 / L2 mask losses (main.py:421-429) at zpart[2 * tile]
    float* h_out; float* zpart;
    // (mask_infer_f16_kernel: whether o0 is fp16 NHWC -- the fused fp16 inference path, hconv.hip -- is the kernel's template parameter O16)
};

struct MaskInferGeo {
    static constexpr int H = 64, W = 64, TH = 8, STRIPS = H / TH, HR = TH + 2;       // h rows per tile
    static constexpr int IR = TH + 4, IC = W + 2, IPS = 5;                            // image tile
    static constexpr int LR = TH / 2 + 4, LC = W / 2 + 2, LPS = 10;                   // o0 tile
    static constexpr int HPS = 20;                                                    // h pixel stride: conflict-free float4 reads
    static constexpr int XIMG = IR * IC * IPS, XO = LR * LC * LPS, HS = HR * IC * HPS, W2S = 144;
    static constexpr size_t LDS = (size_t)((XIMG + 3) / 4 * 4 + XO + HS + W2S) * 4;
};

#ifndef MIF16_PREFETCH
#define MIF16_PREFETCH 0    // mask_infer_f16_kernel's own switch (A/B round 5)
#endif
#ifndef MIF_PREFETCH
#define MIF_PREFETCH 0      // 1 = the next tile's fetch issued before this tile's arithmetic (measured equal at 4 / 2 workgroups per CU: the other workgroups already cover the round trip)
#endif
template <int SRC, bool TRAIN>
__global__ void __launch_bounds__(256, 2) mask_infer_kernel(MaskInferParams P) {
    using G = MaskInferGeo;
    __shared__ float zred[8];
    constexpr int H = G::H, W = G::W, TH = G::TH, HR = G::HR, IR = G::IR, IC = G::IC, IPS = G::IPS, LR = G::LR, LC = G::LC,
                  LPS = G::LPS, HPS = G::HPS;
    constexpr int NIMG = IR * IC, NLO = LR * LC * 2;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float* ximg = (float*)smem;
    float* xo = ximg + (G::XIMG + 3) / 4 * 4;
    float* hs = xo + G::XO;            // [HR][IC][HPS]: h rows row0-1 .. row0+8, columns -1 .. 64
    float* w2s = hs + G::HS;           // masker.2 weights [9][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int hf = wave & 1, par = wave >> 1;   // this wave: columns 32*hf .. +31 of the h rows j = par, par+2, .. (5 rows)
    const int py = (par + 1) & 1;               // row parity of those rows (row0 is even, h row j is image row row0-1+j)

    // ---- masker.0 weights -> registers (B operand: k = kq, n = l15 = oc), for this wave's row parity ----
    float wimg[9], wo[2][2][2][2];     // [px][a][b][s]
#pragma unroll
    for (int t = 0; t < 9; ++t) wimg[t] = (kq < 3) ? P.w0[(t * 11 + kq) * 16 + l15] : 0.f;
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int s4 = 0; s4 < 2; ++s4) {
                    float v = 0.f;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const bool iny = py == 0 ? (a == 0 ? ky == 0 : ky >= 1) : (a == 0 ? ky <= 1 : ky == 2);
                            const bool inx = px == 0 ? (b == 0 ? kx == 0 : kx >= 1) : (b == 0 ? kx <= 1 : kx == 2);
                            const float w = P.w0[((ky * 3 + kx) * 11 + 3 + 4 * s4 + kq) * 16 + l15];
                            v += (iny && inx) ? w : 0.f;
                        }
                    wo[px][a][b][s4] = v;
                }
    const float bias0 = P.b0[l15], bias2 = P.b2[0];
    if (tid < 144) w2s[tid] = P.w2[tid];
    for (int e = tid; e < HR * 2 * 16; e += 256) {      // zero halo columns of the h tile (never written again)
        int ch = e & 15, side = (e >> 4) & 1, r = e >> 5;
        hs[(r * IC + (side ? IC - 1 : 0)) * HPS + ch] = 0.f;
    }

    __builtin_amdgcn_s_waitcnt(0);     // the preloads above land HERE: a first use inside the tile loop would wait with vmcnt(0) and drain the prefetch
    // Staging: ALL of a tile's global loads are issued back to back into registers (fetch), then written to LDS (commit): one
    // memory round trip per tile instead of seven dependent load -> wait -> store rounds; with MIF_PREFETCH the next tile's fetch
    // is issued before this tile's arithmetic.
    constexpr int RI = (NIMG + 255) / 256, RO = (NLO + 255) / 256;
    [[maybe_unused]] uint32_t ilo[RI], ihi[RI];
    [[maybe_unused]] float if0[RI], if1[RI], if2[RI];
    float4 of[RO];
    auto fetch = [&](int tile) {
        const int n0 = tile / G::STRIPS, row0 = (tile % G::STRIPS) * TH;
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            const int e = tid + 256 * r;       // (no branch around the loads: elements past the tile read pixel 0 and are dropped in commit)
            const int rr = e / IC, c = e % IC, y = row0 + rr - 2, x = c - 1;
            const bool in = e < NIMG && y >= 0 && y < H && x >= 0 && x < W;
            const int pix = in ? (n0 * H + y) * W + x : 0;
            if constexpr (SRC == WSRC_U8) {
                const uint32_t* s32 = (const uint32_t*)P.img;
                const int off = pix * 3, last = P.n * H * W * 3 / 4 - 1, d = off >> 2;
                ilo[r] = s32[d]; ihi[r] = s32[d + 1 <= last ? d + 1 : last];
            } else {
                const float* sf = (const float*)P.img;
                if0[r] = sf[pix * 3]; if1[r] = sf[pix * 3 + 1]; if2[r] = sf[pix * 3 + 2];
            }
        }
#pragma unroll
        for (int r = 0; r < RO; ++r) {
            const int e = tid + 256 * r;
            const int half = e & 1, pc = (e >> 1) % LC, pr = (e >> 1) / LC;
            const int ly = row0 / 2 + pr - 2, lx = pc - 1;
            const bool in = e < NLO && ly >= 0 && ly < H / 2 && lx >= 0 && lx < W / 2;
            of[r] = ((const float4*)P.o0)[in ? ((n0 * (H / 2) + ly) * (W / 2) + lx) * 2 + half : 0];
        }
    };
    auto commit = [&](int tile) {
        const int n0 = tile / G::STRIPS, row0 = (tile % G::STRIPS) * TH;
        (void)n0;
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            const int e = tid + 256 * r;
            if (e < NIMG) {
                const int rr = e / IC, c = e % IC, y = row0 + rr - 2, x = c - 1;
                const bool in = y >= 0 && y < H && x >= 0 && x < W;
                float v0, v1, v2;
                if constexpr (SRC == WSRC_U8) {
                    const int pix = in ? (n0 * H + y) * W + x : 0;
                    const uint64_t both = (((uint64_t)ihi[r] << 32) | ilo[r]) >> (((pix * 3) & 3) * 8);
                    const float sc = 1.f / 255.f;
                    const uint32_t b3 = (uint32_t)both;      // (32-bit conversions)
                    v0 = (b3 & 255u) * sc; v1 = ((b3 >> 8) & 255u) * sc; v2 = ((b3 >> 16) & 255u) * sc;
                } else {
                    v0 = if0[r]; v1 = if1[r]; v2 = if2[r];
                }
                float* d = ximg + e * IPS;
                d[0] = in ? v0 : 0.f; d[1] = in ? v1 : 0.f; d[2] = in ? v2 : 0.f; d[3] = 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < RO; ++r) {
            const int e = tid + 256 * r;
            if (e < NLO) {
                const int half = e & 1, pc = (e >> 1) % LC, pr = (e >> 1) / LC;
                const int ly = row0 / 2 + pr - 2, lx = pc - 1;
                const bool in = ly >= 0 && ly < H / 2 && lx >= 0 && lx < W / 2;
                const float4 v = in ? of[r] : f4zero();
                float2* d = (float2*)(xo + (pr * LC + pc) * LPS + 4 * half);
                d[0] = make_float2(v.x, v.y); d[1] = make_float2(v.z, v.w);
            }
        }
    };

    for (int tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
        const int n0 = tile / G::STRIPS, row0 = (tile % G::STRIPS) * TH;
        // ---- input tiles -> LDS: image rows row0-2 .. row0+9, o0 rows row0/2-2 .. row0/2+5 ----
        if (!MIF_PREFETCH || tile == (int)blockIdx.x) fetch(tile);
        commit(tile);
        __syncthreads();
        if (MIF_PREFETCH && tile + (int)gridDim.x < P.ntiles) fetch(tile + gridDim.x);      // in flight under this tile's arithmetic

        // ---- masker.0 on the matrix cores into the h tile: 5 rows x 2 column parities per wave ----
        // h row j = par + 2*jj: image tile row of tap ky = j + ky;  o0 tile row of fold a = jj + a + 1 (see derivation in
        // the header: (y>>1) + a - (1-py) relative to the tile's first low-res row, identical for both row parities)
        const int ibase = (par * IC + 2 * (16 * hf + l15)) * IPS + kq;     // + ((2*jj + ky)*IC + px + kx)*IPS
        const int obase = (16 * hf + l15) * LPS + kq;                      // + ((jj + a + 1)*LC + px + b)*LPS + 4*s
        float ai[2][9], ao[2][8];
        auto ld = [&](int t, int buf) {         // MFMA tile t: row jj = t>>1, column parity px = t&1
            const int jj = t >> 1, px = t & 1;
#pragma unroll
            for (int k = 0; k < 9; ++k) ai[buf][k] = ximg[ibase + ((2 * jj + k / 3) * IC + px + k % 3) * IPS];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int a = q >> 2, b = (q >> 1) & 1, s4 = q & 1;
                ao[buf][q] = xo[obase + ((jj + a + 1) * LC + px + b) * LPS + 4 * s4];
            }
        };
        ld(0, 0);
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            if (t + 1 < 10) ld(t + 1, (t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const int jj = t >> 1, px = t & 1;
            frag4 acc = frag4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[t & 1][k], wimg[k], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ao[t & 1][q], wo[px][q >> 2][(q >> 1) & 1][q & 1], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // D[row = 4kq + i (pixel)][col = l15 (oc)] -> h tile; rows outside the image are masker.2's ZERO padding
            const int j = par + 2 * jj, y = row0 - 1 + j;
            const float keep = (y >= 0 && y < H) ? 1.f : 0.f;
            float* hrow = hs + (j * IC + 1 + 2 * (16 * hf + 4 * kq) + px) * HPS + l15;
#pragma unroll
            for (int i = 0; i < 4; ++i) hrow[2 * i * HPS] = keep * act_fwd<CGS_ACT_LRELU>(acc[i] + bias0);
        }
        __syncthreads();

        if constexpr (TRAIN) {
            // h rows row0 .. row0+7 (tile rows 1 .. 8) -> memory, once: a row is 4 KB contiguous, 16 B per lane
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int e = tid + it * 256, c4 = e & 3, x = (e >> 2) & 63, yl = e >> 8;
                const float4 v = *(const float4*)(hs + ((yl + 1) * IC + 1 + x) * HPS + 4 * c4);
                ((float4*)P.h_out)[((size_t)(n0 * H + row0 + yl) * W + x) * 4 + c4] = v;
            }
        }
        [[maybe_unused]] float zs1 = 0.f, zs2 = 0.f;
        // ---- masker.2 + sigmoid on the tile: thread = pixels (yl, x) and (yl+4, x) ----
        {
            const int x = tid & 63, yl0 = tid >> 6;
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                const int yl = yl0 + 4 * rep;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float4* hp = (const float4*)(hs + ((yl + t / 3) * IC + x + t % 3) * HPS);
                    const auto* wp = cgs_to_const(P.w2) + t * 16;     // wave-uniform: scalar loads, SGPR operands
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        const float4 hv = hp[c4];
                        a0 = fmaf(hv.x, wp[4 * c4], a0); a1 = fmaf(hv.y, wp[4 * c4 + 1], a1);
                        a2 = fmaf(hv.z, wp[4 * c4 + 2], a2); a3 = fmaf(hv.w, wp[4 * c4 + 3], a3);
                    }
                    if (t % 3 == 2) __builtin_amdgcn_sched_barrier(0);      // at most one row of taps of operands in flight
                }
                const float zpre = ((a0 + a1) + (a2 + a3)) + bias2;
                const float zv = 1.f / (1.f + expf(-zpre));
                P.z[(size_t)(n0 * H + row0 + yl) * W + x] = zv;
                if constexpr (TRAIN) { zs1 += fabsf(zv); zs2 += zv * zv; }
            }
        }
        if constexpr (TRAIN) {
            zs1 = wave_sum(zs1); zs2 = wave_sum(zs2);
            if (lane == 0) { zred[2 * wave] = zs1; zred[2 * wave + 1] = zs2; }
        }
        __syncthreads();
        if constexpr (TRAIN) {
            if (tid == 0) {      // fixed order: reproducible
                P.zpart[2 * tile] = (zred[0] + zred[2]) + (zred[4] + zred[6]);
                P.zpart[2 * tile + 1] = (zred[1] + zred[3]) + (zred[5] + zred[7]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fp16-operand variant of the inference mask head (BASELINE config 4: "-process inference-only, fp16 conv kernels"):
// the masker.0 GEMM runs on v_mfma_f32_16x16x16_f16 (fp16 image / o0 / weight operands, fp32 accumulate): K = 16 per
// instruction, 5 instead of 17 MFMAs per 16 pixels, each 4x shorter than an fp32 MFMA.
// (round 4) masker.2 (16 -> 1 channels) no longer walks a 16-channel h tile in LDS (144 fp32 FMAs and 36 16-byte LDS reads per
// pixel: half of the kernel's 210 us at batch 2048).  The instruction is issued as D[oc][pixel] (weights = A operand), so a lane's
// four accumulators are four consecutive CHANNELS of one pixel -- exactly the B-operand layout of another 16x16x16 MFMA: after bias +
// LeakyReLU + the fp16 conversion they feed P[tap][pixel] = sum_c w2[tap][c] h[pixel][c] (A = masker.2's weights, 9 of 16 rows used)
// straight from registers.  h never reaches LDS; the nine per-tap planes P_t do (fp32, 9 floats per pixel), and
// Z = sigmoid(b2 + sum_t P_t[y + ky - 1][x + kx - 1]) is nine conflict-free dword reads and adds per pixel.
// OPT-IN only (engine.infer(..., fp16_mask_head=True) / fp16=True); the result differs from the fp32 path by ~1e-3
// absolute in Z (tests/test_gpu_kernels.py), so it is never used for training or for the parity-gated paths.
// K layout of one instruction = 4 lanes-groups (kq) x 4 halves: image: kq = tap (4m + kq), halves = (r, g, b, 0);
// o0: kq = (fold column b = kq>>1, channel group kq&1), instruction m = fold row a.
// ------------------------------------------------------------------------------------------------
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8m_t __attribute__((ext_vector_type(8)));
// (round 6) masker.0 on v_mfma_f32_16x16x32_f16 (K = 32 per instruction: lane (l15, kq) holds k = 8 kq + j): the image's nine taps x (r, g, b, 0)
// = 36 k-values are two instructions (taps (2 kq, 2 kq + 1) -- two 8-byte LDS reads; tap 8 + zero weights -- one), the folded o0 operand
// (fold row a, fold column b, 8 channels = 32) ONE instruction fed by one 16-byte LDS read per lane (a = kq >> 1, b = kq & 1): 3 matrix
// instructions + 4 LDS reads per 16 pixels where the K = 16 form ran 5 + 5.  Same products, same fp32 accumulator; only the order of the sum differs.
#ifndef MIF16_K32
#define MIF16_K32 0      // (A/B r06_k32, three interleaved runs at batch 2048: 0.1869 ms with the K = 32 form, 0.1818 with the K = 16 form -- see DESIGN.md)
#endif

// (round 5) The kernel was VALU-issue bound (SQ counters, profiles/r05_d_c4_sq.txt: 535 VALU instructions per wave and tile, 0.8 of
// the SIMDs' issue slots), a third of them the per-PIXEL staging of the uint8 frame.  Fast forms, taken when the operands allow:
//  * uint8 frame: a thread takes FOUR pixels = 12 bytes = three dwords (a 64-pixel row is 48 dwords, so groups never straddle rows;
//    the tile's two halo columns are the conv's zero padding for every tile and are zeroed once).  byte -> fp16 without a conversion:
//    v_perm_b32 places the byte under the exponent byte 0x3C (= 1 + b/1024, exact), one packed subtract of 1 leaves b/1024; the
//    1024/255 goes into masker.0's image weights.  18 VALU per four pixels instead of 4 x 40.
//  * fp16 o0: one 16-byte load / LDS store per low-res pixel (32 interior columns x 8 rows = one per thread).
//  * bias as the accumulator's initial value; LeakyReLU on packed fp16; the nine tap planes stored without branches (taps 9..15 of the
//    P instruction go to a tenth, never-read plane); rows of h outside the image zeroed after the loop by the wave that wrote them.
template <int SRC, bool O16>
__global__ void __launch_bounds__(256, 4) mask_infer_f16_kernel(MaskInferParams P) {
    using G = MaskInferGeo;
    constexpr int H = G::H, W = G::W, TH = G::TH, HR = G::HR, IR = G::IR, IC = G::IC, LR = G::LR, LC = G::LC;
    constexpr int NIMG = IR * IC, NLO = LR * LC * 2;
    constexpr bool IMG4 = SRC == WSRC_U8;                  // four-pixel staging of the uint8 frame
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    _Float16* xo = (_Float16*)smem;                        // [LR][LC][8]  (first: 16-byte aligned pixels)
    _Float16* ximg = xo + LR * LC * 8;                     // [IR][IC][4]  (r, g, b, 0)
    float* pl = (float*)(ximg + IR * IC * 4);              // [9 taps + 1][HR][IC] fp32: P_t of rows row0-1 .. row0+8, columns -1 .. 64
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int hf = wave & 1, par = wave >> 1;
    const int py = (par + 1) & 1;

    // ---- masker.0 weights -> fp16 registers (the A operand: m = l15 = oc, k = 4*kq + c) ----
    const float wscale = IMG4 ? 1024.f / 255.f : 1.f;
    half4_t wimg[3], wo[2][2];         // image: instruction m covers taps 4m..4m+3;  o0: [px][a = m]
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int t = 4 * m + kq;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            wimg[m][c] = (_Float16)((t < 9 && c < 3) ? wscale * P.w0[((t < 9 ? t : 0) * 11 + (c < 3 ? c : 0)) * 16 + l15] : 0.f);
    }
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int b = kq >> 1, ch = 4 * (kq & 1) + c;
                float v = 0.f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const bool iny = py == 0 ? (a == 0 ? ky == 0 : ky >= 1) : (a == 0 ? ky <= 1 : ky == 2);
                        const bool inx = px == 0 ? (b == 0 ? kx == 0 : kx >= 1) : (b == 0 ? kx <= 1 : kx == 2);
                        const float w = P.w0[((ky * 3 + kx) * 11 + 3 + ch) * 16 + l15];
                        v += (iny && inx) ? w : 0.f;
                    }
                wo[px][a][c] = (_Float16)v;
            }
    [[maybe_unused]] half8m_t wimg32[2], wo32[2];      // K = 32 form: image taps (2 kq, 2 kq + 1) | tap 8;  o0: [px], k = (a = kq >> 1, b = kq & 1, 8 channels)
    if constexpr (MIF16_K32) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = 2 * kq + (j >> 2), c = j & 3;
            wimg32[0][j] = (_Float16)(c < 3 ? wscale * P.w0[(t * 11 + (c < 3 ? c : 0)) * 16 + l15] : 0.f);
            wimg32[1][j] = (_Float16)((kq == 0 && j < 3) ? wscale * P.w0[(8 * 11 + (j < 3 ? j : 0)) * 16 + l15] : 0.f);
        }
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                const int a = kq >> 1, b = kq & 1;
                float v = 0.f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const bool iny = py == 0 ? (a == 0 ? ky == 0 : ky >= 1) : (a == 0 ? ky <= 1 : ky == 2);
                        const bool inx = px == 0 ? (b == 0 ? kx == 0 : kx >= 1) : (b == 0 ? kx <= 1 : kx == 2);
                        const float w = P.w0[((ky * 3 + kx) * 11 + 3 + ch) * 16 + l15];
                        v += (iny && inx) ? w : 0.f;
                    }
                wo32[px][ch] = (_Float16)v;
            }
    }
    const float bias2 = P.b2[0];
    float b0r[4];
    half4_t w2a;                       // masker.2 as the A operand of the P instruction: m = l15 = tap (9 of 16 rows), k = 4*kq + c = channel
    int poff[4];                       // plane of this lane's P rows 4 kq + r (taps 9 .. 15: the spare plane)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        b0r[c] = P.b0[4 * kq + c];
        w2a[c] = (_Float16)(l15 < 9 ? P.w2[(l15 < 9 ? l15 : 0) * 16 + 4 * kq + c] : 0.f);
        poff[c] = (4 * kq + c < 9 ? 4 * kq + c : 9) * HR * IC;
    }
    for (int e = tid; e < 9 * HR * 2; e += 256) {        // the two halo columns of every plane row: zero for every tile
        const int side = e & 1, r = e >> 1;
        pl[r * IC + (side ? IC - 1 : 0)] = 0.f;
    }
    if constexpr (IMG4) {                                // ... and of the image tile (columns -1 and 64 are the conv's padding)
        if (tid < IR * 2) *(uint2*)(ximg + ((tid >> 1) * IC + ((tid & 1) ? IC - 1 : 0)) * 4) = make_uint2(0u, 0u);
    }
    if constexpr (O16) {
        if (tid >= 64 && tid < 64 + LR * 2) {
            const int e = tid - 64;
            *(uint4*)(xo + ((e >> 1) * LC + ((e & 1) ? LC - 1 : 0)) * 8) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    // per-lane tap offsets of the three image instructions (halves): tap t = 4m + kq (t > 8: any valid address, zero weight)
    int toff[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) { int t = 4 * m + kq; t = t < 9 ? t : 8; toff[m] = ((t / 3) * IC + t % 3) * 4; }
    [[maybe_unused]] const int toffa = (((2 * kq) / 3) * IC + (2 * kq) % 3) * 4, toffb = (((2 * kq + 1) / 3) * IC + (2 * kq + 1) % 3) * 4, toff8 = (2 * IC + 2) * 4;

    __builtin_amdgcn_s_waitcnt(0);     // the preloads above land HERE: a first use inside the tile loop would wait with vmcnt(0) and drain the prefetch
    // Staging: ALL of a tile's global loads are issued back to back into registers (fetch), then converted and written to LDS
    // (commit) -- one memory round trip per tile.  With MIF_PREFETCH the next tile's fetch is issued before this tile's arithmetic.
    constexpr int RI = (NIMG + 255) / 256, RO = (NLO + 255) / 256;
    [[maybe_unused]] uint32_t d0 = 0, d1 = 0, d2 = 0;      // IMG4: this thread's four pixels
    [[maybe_unused]] uint4 ov = make_uint4(0u, 0u, 0u, 0u);  // O16 : this thread's low-res pixel
    [[maybe_unused]] float if0[RI], if1[RI], if2[RI];
    [[maybe_unused]] float4 of[RO];
    const int rr4 = tid >> 4, g4 = tid & 15;               // IMG4: tile row 0 .. 11 (tid < 192), pixel group
    const int pr8 = tid >> 5, lx8 = tid & 31;              // O16 : tile row 0 .. 7, low-res column
    auto fetch = [&](int tile) {
        const int n0 = tile / G::STRIPS, row0 = (tile % G::STRIPS) * TH;
        if constexpr (IMG4) {
            const int y = row0 + rr4 - 2;
            const bool in = tid < IR * 16 && y >= 0 && y < H;
            const uint32_t* p = (const uint32_t*)P.img + ((size_t)(n0 * H + (in ? y : 0)) * (W * 3 / 4) + 3 * g4);
            d0 = p[0]; d1 = p[1]; d2 = p[2];
        } else {
#pragma unroll
            for (int r = 0; r < RI; ++r) {
                const int e = tid + 256 * r;       // (no branch around the loads: elements past the tile read pixel 0 and are dropped in commit)
                const int rr = e / IC, c = e % IC, y = row0 + rr - 2, x = c - 1;
                const bool in = e < NIMG && y >= 0 && y < H && x >= 0 && x < W;
                const int pix = in ? (n0 * H + y) * W + x : 0;
                const float* sf = (const float*)P.img;
                if0[r] = sf[pix * 3]; if1[r] = sf[pix * 3 + 1]; if2[r] = sf[pix * 3 + 2];
            }
        }
        if constexpr (O16) {
            const int ly = row0 / 2 + pr8 - 2;
            const bool in = ly >= 0 && ly < H / 2;
            ov = ((const uint4*)P.o0)[(size_t)(n0 * (H / 2) + (in ? ly : 0)) * (W / 2) + lx8];
        } else {
#pragma unroll
            for (int r = 0; r < RO; ++r) {
                const int e = tid + 256 * r;
                const int half = e & 1, pc = (e >> 1) % LC, pr = (e >> 1) / LC;
                const int ly = row0 / 2 + pr - 2, lx = pc - 1;
                const bool in = e < NLO && ly >= 0 && ly < H / 2 && lx >= 0 && lx < W / 2;
                of[r] = ((const float4*)P.o0)[in ? ((n0 * (H / 2) + ly) * (W / 2) + lx) * 2 + half : 0];
            }
        }
    };
    auto commit = [&](int tile) {
        const int row0 = (tile % G::STRIPS) * TH;
        if constexpr (IMG4) {
            if (tid < IR * 16) {
                const int y = row0 + rr4 - 2;
                const bool in = y >= 0 && y < H;
                const uint32_t e0 = in ? d0 : 0u, e1 = in ? d1 : 0u, e2 = in ? d2 : 0u;      // rows outside the image: zero bytes -> 0.0
                const uint32_t q1 = __builtin_amdgcn_alignbyte(e1, e0, 3), q2 = __builtin_amdgcn_alignbyte(e2, e1, 2);
                constexpr uint32_t K = 0x3C3C3C3Cu;       // selectors 0 .. 3: bytes of the pixel dword; 4: 0x3C; 0x0c: 0x00
                auto cv = [&](uint32_t q, uint32_t sel_rg, uint32_t sel_b) {
                    const half2_t rg = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(K, q, sel_rg)) - half2_t{(_Float16)1.f, (_Float16)1.f};
                    const half2_t b0 = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(K, q, sel_b)) - half2_t{(_Float16)1.f, (_Float16)0.f};
                    return make_uint2(__builtin_bit_cast(uint32_t, rg), __builtin_bit_cast(uint32_t, b0));
                };
                uint2* d = (uint2*)(ximg + (rr4 * IC + 1 + 4 * g4) * 4);
                d[0] = cv(e0, 0x04010400u, 0x0c0c0402u);
                d[1] = cv(q1, 0x04010400u, 0x0c0c0402u);
                d[2] = cv(q2, 0x04010400u, 0x0c0c0402u);
                d[3] = cv(e2, 0x04020401u, 0x0c0c0403u);
            }
        } else {
#pragma unroll
            for (int r = 0; r < RI; ++r) {
                const int e = tid + 256 * r;
                if (e < NIMG) {
                    const int rr = e / IC, c = e % IC, y = row0 + rr - 2, x = c - 1;
                    const bool in = y >= 0 && y < H && x >= 0 && x < W;
                    half4_t hv;
                    hv[0] = (_Float16)(in ? if0[r] : 0.f); hv[1] = (_Float16)(in ? if1[r] : 0.f); hv[2] = (_Float16)(in ? if2[r] : 0.f); hv[3] = (_Float16)0.f;
                    *(half4_t*)(ximg + e * 4) = hv;
                }
            }
        }
        if constexpr (O16) {
            const int ly = row0 / 2 + pr8 - 2;
            const bool in = ly >= 0 && ly < H / 2;
            *(uint4*)(xo + (pr8 * LC + 1 + lx8) * 8) = in ? ov : make_uint4(0u, 0u, 0u, 0u);
        } else {
#pragma unroll
            for (int r = 0; r < RO; ++r) {
                const int e = tid + 256 * r;
                if (e < NLO) {
                    const int half = e & 1, pc = (e >> 1) % LC, pr = (e >> 1) / LC;
                    const int ly = row0 / 2 + pr - 2, lx = pc - 1;
                    const bool in = ly >= 0 && ly < H / 2 && lx >= 0 && lx < W / 2;
                    half4_t hv;
                    hv[0] = (_Float16)of[r].x; hv[1] = (_Float16)of[r].y; hv[2] = (_Float16)of[r].z; hv[3] = (_Float16)of[r].w;
                    if (!in) hv = half4_t{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                    *(half4_t*)(xo + ((pr * LC + pc) * 8 + 4 * half)) = hv;
                }
            }
        }
    };

    for (int tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
        const int n0 = tile / G::STRIPS, row0 = (tile % G::STRIPS) * TH;
        if (!MIF16_PREFETCH || tile == (int)blockIdx.x) fetch(tile);
        commit(tile);
        __syncthreads();
        if (MIF16_PREFETCH && tile + (int)gridDim.x < P.ntiles) fetch(tile + gridDim.x);      // in flight under this tile's arithmetic

        const int ibase = (par * IC + 2 * (16 * hf + l15)) * 4;            // halves; + (2*jj*IC + px)*4 + toff[m]
        const int obase = (16 * hf + l15) * 8 + 4 * kq;                    // halves; + ((jj + a + 1)*LC + px)*8
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            const int jj = t >> 1, px = t & 1;
            frag4 acc = frag4{b0r[0], b0r[1], b0r[2], b0r[3]};             // D[oc = 4 kq + r][pixel = l15], from the bias
            if constexpr (MIF16_K32) {
                const _Float16* ip = ximg + ibase + (2 * jj * IC + px) * 4;
                const half4_t t0 = *(const half4_t*)(ip + toffa), t1 = *(const half4_t*)(ip + toffb), t8 = *(const half4_t*)(ip + toff8);
                const half8m_t o8 = *(const half8m_t*)(xo + (16 * hf + l15) * 8 + ((jj + (kq >> 1) + 1) * LC + px + (kq & 1)) * 8);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wimg32[0], __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wimg32[1], __builtin_shufflevector(t8, t8, 0, 1, 2, 3, 4, 5, 6, 7), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wo32[px], o8, acc, 0, 0, 0);
            } else {
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const half4_t a = *(const half4_t*)(ximg + ibase + (2 * jj * IC + px) * 4 + toff[m]);
                    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(wimg[m], a, acc, 0, 0, 0);
                }
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2) {
                    const half4_t a = *(const half4_t*)(xo + obase + ((jj + a2 + 1) * LC + px) * 8);
                    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(wo[px][a2], a, acc, 0, 0, 0);
                }
            }
            half4_t hv = __builtin_convertvector(acc, half4_t);
            hv = __builtin_elementwise_max(hv, hv * (_Float16)0.01f);      // LeakyReLU on the packed halves
            const frag4 pt = __builtin_amdgcn_mfma_f32_16x16x16f16(w2a, hv, frag4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);      // P[tap = 4 kq + r][pixel]
            float* prow = pl + (par + 2 * jj) * IC + 1 + 2 * (16 * hf + l15) + px;
#pragma unroll
            for (int r = 0; r < 4; ++r) prow[poff[r]] = pt[r];
        }
        // rows of h outside the image are masker.2's ZERO padding: this wave wrote them (j = 0: par 0; j = 9: par 1), it clears them
        if ((row0 == 0 && par == 0) || (row0 == H - TH && par == 1)) {
            const int j = par ? HR - 1 : 0;
            for (int e = lane; e < 9 * 32; e += 64) pl[((e >> 5) * HR + j) * IC + 1 + 32 * hf + (e & 31)] = 0.f;
        }
        __syncthreads();
        {
            const int x = tid & 63, yl0 = tid >> 6;
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                const int yl = yl0 + 4 * rep;
                float zpre = bias2;
#pragma unroll
                for (int t = 0; t < 9; ++t) zpre += pl[(t * HR + yl + t / 3) * IC + x + t % 3];
                P.z[(size_t)(n0 * H + row0 + yl) * W + x] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * zpre));
            }
        }
        __syncthreads();
    }
}

static constexpr size_t kMaskInferF16Lds =
    (size_t)MaskInferGeo::IR * MaskInferGeo::IC * 4 * 2 + (size_t)MaskInferGeo::LR * MaskInferGeo::LC * 8 * 2 +
    (size_t)10 * MaskInferGeo::HR * MaskInferGeo::IC * 4;

int mask_infer_f16_launch(int n, int img_kind, const void* img, const float* o0, const float* w0, const float* b0,
                          const float* w2, const float* b2, float* z, hipStream_t st, int o0_f16) {
    if (n <= 0) return CGS_OK;
    MaskInferParams P{img, o0, w0, b0, w2, b2, z, n, n * MaskInferGeo::STRIPS, nullptr, nullptr};
    int blocks = P.ntiles < 1024 ? P.ntiles : 1024;
    const bool u8 = img_kind == CGS_SRC_U8;
    if (u8 && o0_f16) hipLaunchKernelGGL((mask_infer_f16_kernel<WSRC_U8, true>), dim3(blocks), dim3(256), kMaskInferF16Lds, st, P);
    else if (u8) hipLaunchKernelGGL((mask_infer_f16_kernel<WSRC_U8, false>), dim3(blocks), dim3(256), kMaskInferF16Lds, st, P);
    else if (o0_f16) hipLaunchKernelGGL((mask_infer_f16_kernel<WSRC_F32, true>), dim3(blocks), dim3(256), kMaskInferF16Lds, st, P);
    else hipLaunchKernelGGL((mask_infer_f16_kernel<WSRC_F32, false>), dim3(blocks), dim3(256), kMaskInferF16Lds, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

template <int SRC, bool TRAIN>
static int launch_mask_infer(const MaskInferParams& P, hipStream_t st) {
    static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&mask_infer_kernel<SRC, TRAIN>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)MaskInferGeo::LDS);
    if (attr != hipSuccess) return (int)attr;
    int blocks = P.ntiles < 1024 ? P.ntiles : 1024;
    hipLaunchKernelGGL((mask_infer_kernel<SRC, TRAIN>), dim3(blocks), dim3(256), MaskInferGeo::LDS, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

int mask_infer_launch(int n, int img_kind, const void* img, const float* o0, const float* w0, const float* b0, const float* w2,
                      const float* b2, float* z, hipStream_t st) {
    if (n <= 0) return CGS_OK;
    MaskInferParams P{img, o0, w0, b0, w2, b2, z, n, n * MaskInferGeo::STRIPS, nullptr, nullptr};
    return img_kind == CGS_SRC_U8 ? launch_mask_infer<WSRC_U8, false>(P, st) : launch_mask_infer<WSRC_F32, false>(P, st);
}

// (the training form -- h stored, Z and the mask-loss partial sums from the same launch -- is csrc/mask_fwd.hip)

template <int SRC>
static int launch_mask0_fwd(Mask0FwdParams P, hipStream_t st) {
    if (P.n <= 0) return CGS_OK;
    P.ntiles = P.n * 8;
    int blocks = P.ntiles < 2048 ? P.ntiles : 2048;
    hipLaunchKernelGGL(mask0_fwd_kernel<SRC>, dim3(blocks), dim3(256), 0, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

template <class C>
static int launch_mconv(MConvParams P, hipStream_t st) {
    using G = typename C::G;
    constexpr int S = (C::CA + 3) / 4 + C::CB / 4;
    if (P.n <= 0) return CGS_OK;
    P.ntiles = (G::IMGS == 1) ? P.n * G::STRIPS : (P.n + G::IMGS - 1) / G::IMGS;
    int blocks = P.ntiles < 2048 ? P.ntiles : 2048;
    size_t lds = (size_t)G::IMGS * G::TRA * G::PWA * S * 4 * sizeof(float);
    hipLaunchKernelGGL(mconv_kernel<C>, dim3(blocks), dim3(G::THREADS), lds, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// Called from cgs_conv3x3_fwd / cgs_conv3x3_bwd_data (conv_fwd.hip) for the layers routed to the matrix cores.
int mconv_fwd_dispatch(int which, int n, const void* src_a, const float* src_b, const float* w, const float* bias, float* out,
                       hipStream_t st) {
    MConvParams P{};
    P.src_a = src_a; P.src_b = src_b; P.w = w; P.bias = bias; P.out = out; P.n = n;
    switch (which) {
        case 0: return launch_mask0_fwd<WSRC_U8>(Mask0FwdParams{src_a, src_b, w, bias, out, n, 0}, st);
        case 1: return launch_mask0_fwd<WSRC_F32>(Mask0FwdParams{src_a, src_b, w, bias, out, n, 0}, st);
        case 2: return launch_mconv<MDec3F>(P, st);
    }
    return CGS_ERR_UNSUPPORTED;
}

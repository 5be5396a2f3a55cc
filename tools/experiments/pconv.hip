// Forward 3x3 convolutions of the 8-output-channel encoder layers at 64x64 / 32x32 (features.0, features.3: nets.py:170-175)
// as implicit GEMMs on v_mfma_f32_16x16x4_f32 with TWO horizontally adjacent output pixels per MFMA row:
//
//   D[pair p][(d, co)] = sum_{ky, kx' in 0..3, ci}  X[y + ky - 1][2p + kx' - 1][ci] * Wp[(ky, kx', ci)][(d, co)],
//   Wp[(ky, kx', ci)][(d, co)] = W[ky][kx' - d][ci][co] if 0 <= kx' - d <= 2 else 0          (d = 0, 1: left / right pixel)
//
// An 8-channel layer fills only half of a 16-column MFMA tile in the one-pixel-per-row form; here all 16 columns are outputs
// and K grows by 4/3 (a 3 x 4 window instead of 3 x 3): 2/3 of the matrix instructions per pixel, 3/4 of them useful.  The
// direct VALU kernels these replace (conv_body.h) issue one v_fma_f32 per 64 multiply-adds plus a weight broadcast per 4 of
// them and run at ~50 % of the vector issue rate; a matrix instruction carries 1024 multiply-adds.
// The zero entries of Wp add exactly 0 to an fp32 FMA chain, so pixels with equal receptive fields still get bit-equal sums
// in either column half: max-pool ties resolve as in the reference (first maximum wins).
//
// Workgroup = 4 waves = a 16-row (64x64) or 8-row (32x32) strip of one image; a wave owns row pairs, so the 2x2 pool window is
// (two accumulators) x (lane, lane ^ 8).  Weights live in registers for all tiles of the persistent workgroup.
#include "tail_common.h"
#include "conv_tile.h"      // SRC_* tags

struct PConvParams {
    const void* src;            // u8 frames [n,64,64,3] / fp32 NHWC [n,32,32,8]
    const uint8_t* mix_a; const uint8_t* mix_b; const float* mix_z; int mix_n_a;     // SRC_MIXC3
    const float* w; const float* bias;
    float* out; uint32_t* amask;
    int n, ntiles, nblocks;
    unsigned long long* dbg;
};
static unsigned long long* g_pconv_stamps = nullptr;
extern "C" int dbg_pconv_stamps(unsigned long long* p) { g_pconv_stamps = p; return 0; }
#define PSTAMP(k) do { if (P.dbg && tid == 0 && tile == (int)blockIdx.x) P.dbg[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)

// C: HW, CI, TH (strip rows), PS (LDS floats per pixel), SRC
template <class C>
__global__ void __launch_bounds__(256) pconv_fwd_kernel(PConvParams P) {
    constexpr int HW = C::HW, CI = C::CI, TH = C::TH, PS = C::PS, PW = HW + 2, STRIPS = HW / TH;
    constexpr int K = 12 * CI, NS = (K + 3) / 4;                  // k = (ky*4 + kx')*CI + ci
    constexpr int RPW = TH / 8;                                    // row pairs per wave
    constexpr int NXH = HW / 32;                                   // 32-pixel (16-pair) column blocks per row
    __shared__ __attribute__((aligned(16))) float xt[(TH + 2) * PW * PS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int dsel = l15 >> 3, co = l15 & 7;

    // ---- weights (B operand) and the per-lane A offsets of every k-step ----
    float bw[NS];
    int aoff[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = 4 * s + kq;
        const int ci = k % CI, t = k / CI, ky = t >> 2, kxp = t & 3, kx = kxp - dsel;
        const bool ok = k < K && kx >= 0 && kx <= 2;
        bw[s] = ok ? P.w[((ky * 3 + kx) * CI + ci) * 8 + co] : 0.f;
        aoff[s] = k < K ? (ky * PW + kxp) * PS + ci : 0;
    }
    const float bias = P.bias[co];
    // zero halo columns once (rows are rewritten per tile)
    for (int e = tid; e < (TH + 2) * 2 * PS; e += 256) {
        const int c = e % PS, side = (e / PS) & 1, r = e / (2 * PS);
        xt[(r * PW + (side ? PW - 1 : 0)) * PS + c] = 0.f;
    }

    if (P.dbg && tid == 0) P.dbg[(size_t)blockIdx.x * 8] = __builtin_amdgcn_s_memtime();
    for (int tile = blockIdx.x; tile < P.ntiles; tile += P.nblocks) {
        const int img = tile / STRIPS, row0 = (tile % STRIPS) * TH;
        PSTAMP(1);
        // ---- stage rows row0-1 .. row0+TH ----
        if constexpr (C::SRC == SRC_F32) {
            constexpr int NG = (TH + 2) * HW * (CI / 4);
            for (int e = tid; e < NG; e += 256) {
                const int g = e % (CI / 4), x = (e / (CI / 4)) % HW, r = e / ((CI / 4) * HW);
                const int y = row0 + r - 1;
                float4 v = f4zero();
                if (y >= 0 && y < HW) v = ((const float4*)P.src)[(((size_t)img * HW + y) * HW + x) * (CI / 4) + g];
                float* d = xt + (r * PW + x + 1) * PS + 4 * g;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            // 3-channel frames: one thread = 4 pixels = 12 bytes
            constexpr int NG = (TH + 2) * (HW / 4);
            const float sc = 1.f / 255.f;
            for (int e = tid; e < NG; e += 256) {
                const int g = e % (HW / 4), r = e / (HW / 4);
                const int y = row0 + r - 1;
                float v[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) v[i] = 0.f;
                if (y >= 0 && y < HW) {
                    if constexpr (C::SRC == SRC_U8C3) {
                        const uint32_t* s = (const uint32_t*)P.src + (((size_t)img * HW + y) * HW + 4 * g) * 3 / 4;
                        const uint32_t d0 = s[0], d1 = s[1], d2 = s[2];
                        const uint32_t dd[3] = {d0, d1, d2};
#pragma unroll
                        for (int i = 0; i < 12; ++i) v[i] = (float)((dd[i >> 2] >> (8 * (i & 3))) & 255u) * sc;
                    } else {      // SRC_MIXC3: image < n_a: A(1-Z) + Z B of A-image; image >= n_a: B(1-Z) + Z A of A-image - n_a
                        const bool inj = img >= P.mix_n_a;
                        const int src_n = inj ? img - P.mix_n_a : img;
                        const size_t pg = ((size_t)src_n * HW + y) * HW / 4 + g;
                        const uint32_t* pa = (const uint32_t*)P.mix_a + 3 * pg;
                        const uint32_t* pb = (const uint32_t*)P.mix_b + 3 * pg;
                        uint32_t da[3] = {pa[0], pa[1], pa[2]}, db[3] = {pb[0], pb[1], pb[2]};
                        const float4 zz = ((const float4*)P.mix_z)[pg];
                        const float zv[4] = {zz.x, zz.y, zz.z, zz.w};
#pragma unroll
                        for (int i = 0; i < 12; ++i) {
                            float av = (float)((da[i >> 2] >> (8 * (i & 3))) & 255u) * sc;
                            float bv = (float)((db[i >> 2] >> (8 * (i & 3))) & 255u) * sc;
                            if (inj) { const float t = av; av = bv; bv = t; }
                            const float zi = zv[i / 3];
                            v[i] = av * (1.f - zi) + zi * bv;
                        }
                    }
                }
                float* d = xt + (r * PW + 4 * g + 1) * PS;
#pragma unroll
                for (int i = 0; i < 12; ++i) d[(i / 3) * PS + (i % 3)] = v[i];
            }
        }
        __syncthreads();
        PSTAMP(2);
        // ---- this wave's row pairs ----
#pragma unroll 1
        for (int rp = 0; rp < RPW; ++rp) {
            const int ry = 2 * (wave * RPW + rp);                  // strip-local first row of the pair
#pragma unroll 1
            for (int xh = 0; xh < NXH; ++xh) {
                const int abase = (ry * PW + 2 * (16 * xh + l15)) * PS;      // tap (0,0) = pixel (y-1, x-1) = halo coords (y, x)
                frag4 acc0 = frag4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
                float a0[NS], a1[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) { a0[s] = xt[abase + aoff[s]]; a1[s] = xt[abase + PW * PS + aoff[s]]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], bw[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], bw[s], acc1, 0, 0, 0);
                }
                // ---- bias + ReLU + 2x2 max-pool (window = rows ry, ry+1 x column halves d = 0, 1), first maximum wins ----
                const int py = (row0 + ry) >> 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v00 = fmaxf(acc0[j] + bias, 0.f), v10 = fmaxf(acc1[j] + bias, 0.f);
                    const float o0 = dpp_ror8(v00), o1 = dpp_ror8(v10);                          // the other column half (lane ^ 8)
                    const float p0 = dsel ? o0 : v00, p1 = dsel ? v00 : o0, p2 = dsel ? o1 : v10, p3 = dsel ? v10 : o1;
                    float m = p0; uint32_t idx = 0;
                    if (p1 > m) { m = p1; idx = 1; }
                    if (p2 > m) { m = p2; idx = 2; }
                    if (p3 > m) { m = p3; idx = 3; }
                    if (!(m > 0.f)) idx = 15u;
                    const uint32_t word = pack_nibbles(idx, co);
                    const int px = 16 * xh + 4 * kq + j;
                    const size_t pi = ((size_t)img * (HW / 2) + py) * (HW / 2) + px;
                    if (dsel == 0) {
                        P.out[pi * 8 + co] = m;
                        if (co == 0 && P.amask) P.amask[pi] = word;
                    }
                }
            }
        }
        PSTAMP(3);
        __syncthreads();
        PSTAMP(4);
    }
}

struct PEnc0U8 { static constexpr int HW = 64, CI = 3, TH = 16, PS = 3, SRC = SRC_U8C3; };
struct PEnc0Mix { static constexpr int HW = 64, CI = 3, TH = 16, PS = 3, SRC = SRC_MIXC3; };
struct PEnc1 { static constexpr int HW = 32, CI = 8, TH = 8, PS = 10, SRC = SRC_F32; };

template <class C>
static int launch_pconv(PConvParams P, hipStream_t st) {
    if (P.n <= 0) return CGS_OK;
    P.ntiles = P.n * (C::HW / C::TH);
    P.nblocks = P.ntiles < 2048 ? P.ntiles : 2048;
    hipLaunchKernelGGL(pconv_fwd_kernel<C>, dim3(P.nblocks), dim3(256), 0, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// which: 0 = features.0 on uint8 frames, 1 = features.0 on the virtual mixes, 2 = features.3
int pconv_fwd_dispatch(int which, int n, const void* src, const uint8_t* mix_a, const uint8_t* mix_b, const float* mix_z, int mix_n_a,
                       const float* w, const float* bias, float* out, uint32_t* amask, hipStream_t st) {
    PConvParams P{src, mix_a, mix_b, mix_z, mix_n_a, w, bias, out, amask, n, 0, 0, g_pconv_stamps};
    switch (which) {
        case 0: return launch_pconv<PEnc0U8>(P, st);
        case 1: return launch_pconv<PEnc0Mix>(P, st);
        case 2: return launch_pconv<PEnc1>(P, st);
    }
    return CGS_ERR_UNSUPPORTED;
}

// Weight + bias gradient of the LARGE-MAP 3x3 layers of BASELINE config 5 (the build-defined 128x128 Hourglass, hourglass128.py, chfak 1):
// features.0 (3 -> 8 at 128x128), masker.0 (3 + 8 -> 16 at 128x128), masker.2 (16 -> 1 at 128x128), features.3 (8 -> 8 at 64x64),
// dec_model.0 (8 + 8 -> 8 at 64x64) and their 32x32 counterparts features.6 / dec_model.1 -- 86 % of the weight-gradient time of the step on the shape-generic bf16_wgrad_kernel
// (gen_bf16_train.hip: run-time shapes, 2 workgroups per CU, taps dealt to the waves so every wave re-reads dY).  Same arithmetic
// (bf16 operands, fp32 accumulation on v_mfma_f32_16x16x32_bf16 with K = 32 pixels through ds_read_b64_tr_b16), built like hconv.hip:
//   * compile-time shapes, workgroups persistent over row strips of one image, 3-4 workgroups per CU (one stages while another multiplies);
//   * the X tile is NHWC with a pixel of 8 / 16 / 32 bytes (4 / 8 / 16 channels): a transposing read's four column groups are four
//     (tap, channel quad) pairs, so a 16-row block holds 4 / 2 / 1 taps: 3 / 5 / 9 matrix instructions per 32 pixels instead of 9 / 9 / 9;
//   * the dY strip is contiguous in memory: staged by a flat 16-byte copy; a single-channel dY (masker.2: d Z fp32) is read from its
//     fp32 source and needs no transposing read (8 consecutive pixels = one 16-byte LDS read);
//   * the waves split the strip's 32-pixel blocks (every wave holds all row blocks), one cross-wave sum per workgroup at the end,
//     one slab row per workgroup -> cgs_reduce_slabs (fixed order: bitwise reproducible).
// No reference counterpart (the reference cannot run 128x128 frames): parity unpinned, see hourglass128.py.
#include "tail_common.h"

#ifndef HWG_XCD
#define HWG_XCD 1                // XCD-contiguous strip order (0: round-robin; r4 A/B)
#endif

namespace {

typedef short hs4_t __attribute__((ext_vector_type(4)));
typedef short hs8_t __attribute__((ext_vector_type(8)));
typedef __bf16 hbf8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) hs4_t lds_hs4_t;

__device__ __forceinline__ short hbf(float f) { return (short)__builtin_bit_cast(unsigned short, (__bf16)f); }      // round to nearest even
__device__ __forceinline__ hs4_t tr4(const uint16_t* addr) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_hs4_t*)addr); }
__device__ __forceinline__ hbf8_t pk8(hs4_t a, hs4_t b) {
    const hs8_t s = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return __builtin_bit_cast(hbf8_t, s);
}

struct HWgParams {
    const void* a;          // CA == 4: uint8 or fp32 frames [n,HW,HW,3]; else bf16 NHWC [n,HW,HW,CA]
    const uint16_t* b;      // CB == 8: bf16 [n,HW/2,HW/2,8], nearest-upsampled; else NULL
    const void* dy;         // CO > 1: bf16 [n,HW,HW,CO]; CO == 1: fp32 [n,HW,HW]; DYPOOL: the pooled gradient bf16 [n,HW/2,HW/2,8]
    const uint16_t* dy_add; // DYPOOL: optional addend of the pooled gradient (same shape) or NULL
    const uint8_t* codes;   // DYPOOL: argmax bytes of the forward pass [n,HW/2,HW/2,8]
    float* slab;            // [blocks][9 (CA_real + CB) CO + CO]
    int n, nstrips, a_f32;  // a_f32 == 2: source A = the virtual mixes of the frame pairs mix_a / mix_b [n / 2,HW,HW,3] with the mask mix_z [n / 2,HW,HW]
    const uint8_t* mix_a; const uint8_t* mix_b; const float* mix_z;
};

// HW: map size; CA: channels of source A in the LDS pixel (4 = rgb0 from uint8 / fp32 frames; 8 / 16 = bf16); CB: 0 / 8 (upsampled bf16
// source); CO: output channels (1, 8, 16); TH: rows per strip
// DYPOOL: dY = a pooled gradient re-expanded while it is staged (the gradient goes where the forward maximum was): replaces cgs_bf16_pool_expand
template <int HW, int CA, int CB, int CO, int TH, bool DYPOOL = false>
__global__ void __launch_bounds__(256) hwgrad_kernel(HWgParams P) {
    if (CGS_KARG_PREFETCH) cgs_kernarg_prefetch<sizeof(HWgParams)>();
    constexpr int CIN = CA + CB <= 4 ? 4 : (CA + CB <= 8 ? 8 : 16);             // channels of the LDS pixel
    constexpr int TPB = 16 / CIN, NB = (9 + TPB - 1) / TPB;                     // taps per 16-row block, row blocks
    constexpr int CA_REAL = CA == 4 ? 3 : CA, CI = CA_REAL + CB;
    constexpr int PW = HW + 2, PH = TH + 2, STRIPS = HW / TH, CPR = HW / 32, NCH = TH * CPR;
    constexpr int XT = (PH * PW * CIN + 7) & ~7;                                // elements of the X tile (16-byte multiple)
    static_assert(CO == 1 || CO == 8 || CO == 16, "dY pixel = 2 / 16 / 32 bytes");
    static_assert(CA == 4 || (CA & 7) == 0, "bf16 sources in 16-byte chunks");
    extern __shared__ __attribute__((aligned(16))) float4 hsm[];
    uint16_t* const xt = (uint16_t*)hsm;                                        // [PH][PW][CIN]
    uint16_t* const dt = xt + XT;                                               // [TH * HW][CO]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4, q = l15 >> 2, p = l15 & 3;

    // this lane's part of a transposing read of row block b: tap 4 b + p (CIN 4), 2 b + (p >> 1) + channels 4 (p & 1) (CIN 8), b + channels 4 p
    int aoff[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        int t = CIN == 4 ? 4 * b + p : (CIN == 8 ? 2 * b + (p >> 1) : b);
        t = t < 9 ? t : 8;                                                      // rows of taps >= 9: duplicates, dropped at the end
        aoff[b] = ((t / 3) * PW + t % 3) * CIN + (CIN == 4 ? 0 : (CIN == 8 ? 4 * (p & 1) : 4 * p));
    }
    frag4 acc[NB], accb = frag4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = frag4{0.f, 0.f, 0.f, 0.f};
    const short one = (short)0x3F80;                                            // bf16 1.0: bias gradient = column sums of dY
    const hbf8_t ones = __builtin_bit_cast(hbf8_t, hs8_t{one, one, one, one, one, one, one, one});

    for (int e = tid; e < XT / 8; e += 256) ((float4*)xt)[e] = f4zero();        // halo columns + padding channels: zero for every strip
    __syncthreads();                                                            // (before other threads stage the same addresses)

    for (int strip = HWG_XCD ? cgs_xcd_contiguous(blockIdx.x, gridDim.x) : (int)blockIdx.x; strip < P.nstrips; strip += gridDim.x) {   // (neighbouring strips on one XCD's L2)
        const int img = strip / STRIPS, row0 = (strip % STRIPS) * TH;
        // ---- staging: EVERY global load of the strip is issued back to back into registers (compile-time trip counts), then converted and stored
        //      to LDS: one memory round trip per strip instead of one per 256 items (these kernels multiply for well under a microsecond per
        //      strip: a strip's life was its four to six dependent load rounds) ----
        constexpr int GW = HW / 4, NGA = CA == 4 ? 1 : CA / 8, HP = HW / 2;
        constexpr int NXI = CA == 4 ? (PH * GW + 255) / 256 : (PH * HW * NGA + 255) / 256, NXV = CA == 4 ? 3 : 1;     // source A items, float4 each
        constexpr int NBI = CB == 8 ? (PH * HW + 255) / 256 : 1;                                                   // source B items
        constexpr int NDI = DYPOOL ? ((TH / 2) * HP + 255) / 256 : (CO == 1 ? (TH * HW / 4 + 255) / 256 : (TH * HW * CO / 8 + 255) / 256);
        constexpr int NDV = DYPOOL ? 3 : 1;
        float4 rxa[NXI][NXV], rxb[NBI], rdy[NDI][NDV];
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            const int e = tid + 256 * k;
            if constexpr (CA == 4) {        // frames: 4 pixels = 12 bytes (uint8) / 12 floats per item
                const int g = e % GW, r = e / GW, y = row0 + r - 1;
                const bool in = r < PH && y >= 0 && y < HW;
                const size_t gi = in ? (((size_t)img * HW + y) * HW + g * 4) * 3 / 4 : 0;
                if (P.a_f32 == 2) {
                    const int half = P.n >> 1, is = img < half ? img : img - half;
                    const size_t gs = in ? (((size_t)is * HW + y) * HW + g * 4) : 0;
                    const uint32_t* sa = (const uint32_t*)P.mix_a + gs * 3 / 4;
                    const uint32_t* sb = (const uint32_t*)P.mix_b + gs * 3 / 4;
                    rxa[k][0] = make_float4(__uint_as_float(sa[0]), __uint_as_float(sa[1]), __uint_as_float(sa[2]), 0.f);
                    rxa[k][1] = make_float4(__uint_as_float(sb[0]), __uint_as_float(sb[1]), __uint_as_float(sb[2]), 0.f);
                    rxa[k][2] = ((const float4*)P.mix_z)[gs / 4];
                } else if (P.a_f32) {
                    const float4* sf = (const float4*)P.a;
                    rxa[k][0] = sf[gi]; rxa[k][1] = sf[gi + 1]; rxa[k][2] = sf[gi + 2];
                } else {
                    const uint32_t* su = (const uint32_t*)P.a;
                    rxa[k][0] = make_float4(__uint_as_float(su[gi]), __uint_as_float(su[gi + 1]), __uint_as_float(su[gi + 2]), 0.f);
                }
            } else {
                const int g = e % NGA, x = (e / NGA) % HW, r = e / (NGA * HW), y = row0 + r - 1;
                const bool in = r < PH && y >= 0 && y < HW;
                rxa[k][0] = ((const float4*)P.a)[in ? (((size_t)img * HW + y) * HW + x) * NGA + g : 0];
            }
        }
        if constexpr (CB == 8) {
#pragma unroll
            for (int k = 0; k < NBI; ++k) {
                const int e = tid + 256 * k, x = e % HW, r = e / HW, y = row0 + r - 1;
                const bool in = r < PH && y >= 0 && y < HW;
                rxb[k] = ((const float4*)P.b)[in ? ((size_t)img * (HW / 2) + (y >> 1)) * (HW / 2) + (x >> 1) : 0];
            }
        }
#pragma unroll
        for (int k = 0; k < NDI; ++k) {
            const int e = tid + 256 * k;
            if constexpr (DYPOOL) {
                static_assert(!DYPOOL || CO == 8, "pooled gradient: 8 channels");
                const int ee = e < (TH / 2) * HP ? e : 0, xp = ee % HP, rp = ee / HP;
                const size_t gi = ((size_t)img * HP + row0 / 2 + rp) * HP + xp;
                rdy[k][0] = ((const float4*)P.dy)[gi];
                rdy[k][1] = P.dy_add ? ((const float4*)P.dy_add)[gi] : f4zero();
                const float2 c2 = ((const float2*)P.codes)[gi];
                rdy[k][2] = make_float4(c2.x, c2.y, 0.f, 0.f);
            } else if constexpr (CO == 1) {
                const float4* sd = (const float4*)((const float*)P.dy + ((size_t)img * HW + row0) * HW);
                rdy[k][0] = sd[e < TH * HW / 4 ? e : 0];
            } else {
                const float4* sd = (const float4*)((const uint16_t*)P.dy + ((size_t)img * HW + row0) * HW * CO);
                rdy[k][0] = sd[e < TH * HW * CO / 8 ? e : 0];
            }
        }
        // ---- commit ----
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            const int e = tid + 256 * k;
            if constexpr (CA == 4) {
                const int g = e % GW, r = e / GW, y = row0 + r - 1;
                if (r >= PH) continue;
                const bool in = y >= 0 && y < HW;
                float f[12];
                if (P.a_f32 == 2) {         // the formula of cgs_mix_fwd
                    const uint32_t da[3] = {__float_as_uint(rxa[k][0].x), __float_as_uint(rxa[k][0].y), __float_as_uint(rxa[k][0].z)};
                    const uint32_t db[3] = {__float_as_uint(rxa[k][1].x), __float_as_uint(rxa[k][1].y), __float_as_uint(rxa[k][1].z)};
                    const float zv[4] = {rxa[k][2].x, rxa[k][2].y, rxa[k][2].z, rxa[k][2].w};
                    const bool inj = img >= (P.n >> 1);
#pragma unroll
                    for (int j = 0; j < 12; ++j) {
                        const float av = (float)((da[j >> 2] >> (8 * (j & 3))) & 255u) * (1.f / 255.f);
                        const float bv = (float)((db[j >> 2] >> (8 * (j & 3))) & 255u) * (1.f / 255.f);
                        const float zi = zv[j / 3];
                        f[j] = inj ? bv * (1.f - zi) + zi * av : av * (1.f - zi) + zi * bv;
                    }
                } else if (P.a_f32) {
                    const float4 f0 = rxa[k][0], f1 = rxa[k][1], f2 = rxa[k][2];
                    f[0] = f0.x; f[1] = f0.y; f[2] = f0.z; f[3] = f0.w; f[4] = f1.x; f[5] = f1.y; f[6] = f1.z; f[7] = f1.w;
                    f[8] = f2.x; f[9] = f2.y; f[10] = f2.z; f[11] = f2.w;
                } else {
                    const uint32_t d[3] = {__float_as_uint(rxa[k][0].x), __float_as_uint(rxa[k][0].y), __float_as_uint(rxa[k][0].z)};
#pragma unroll
                    for (int j = 0; j < 12; ++j) f[j] = (float)((d[j >> 2] >> (8 * (j & 3))) & 255u) * (1.f / 255.f);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const hs4_t v = in ? hs4_t{hbf(f[3 * j]), hbf(f[3 * j + 1]), hbf(f[3 * j + 2]), 0} : hs4_t{0, 0, 0, 0};
                    *(hs4_t*)(xt + ((size_t)r * PW + 1 + 4 * g + j) * CIN) = v;
                }
            } else {
                const int g = e % NGA, x = (e / NGA) % HW, r = e / (NGA * HW), y = row0 + r - 1;
                if (r >= PH) continue;
                *(float4*)(xt + ((size_t)r * PW + 1 + x) * CIN + 8 * g) = (y >= 0 && y < HW) ? rxa[k][0] : f4zero();
            }
        }
        if constexpr (CB == 8) {            // the nearest-upsampled source: channels CA .. CA + 7 of the pixel
#pragma unroll
            for (int k = 0; k < NBI; ++k) {
                const int e = tid + 256 * k, x = e % HW, r = e / HW, y = row0 + r - 1;
                if (r >= PH) continue;
                const float4 v = (y >= 0 && y < HW) ? rxb[k] : f4zero();
                uint16_t* d = xt + ((size_t)r * PW + 1 + x) * CIN + CA;       // 8-byte aligned (CA = 4) or 16
                *(float2*)d = make_float2(v.x, v.y);
                *(float2*)(d + 4) = make_float2(v.z, v.w);
            }
        }
#pragma unroll
        for (int k = 0; k < NDI; ++k) {
            const int e = tid + 256 * k;
            if constexpr (DYPOOL) {         // one pooled pixel -> its 2 x 2 pixels: the gradient goes where the forward maximum was
                if (e >= (TH / 2) * HP) continue;
                const int xp = e % HP, rp = e / HP;
                const float4 d4 = rdy[k][0], a4 = rdy[k][1];
                const uint32_t dw[4] = {__float_as_uint(d4.x), __float_as_uint(d4.y), __float_as_uint(d4.z), __float_as_uint(d4.w)};
                const uint32_t aw[4] = {__float_as_uint(a4.x), __float_as_uint(a4.y), __float_as_uint(a4.z), __float_as_uint(a4.w)};
                const uint32_t cw[2] = {__float_as_uint(rdy[k][2].x), __float_as_uint(rdy[k][2].y)};
                short sv[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t sh = 16 * (c & 1);
                    sv[c] = hbf(__uint_as_float(((dw[c >> 1] >> sh) & 0xffffu) << 16) + __uint_as_float(((aw[c >> 1] >> sh) & 0xffffu) << 16));
                }
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                    hs8_t v;
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = ((cw[c >> 2] >> (8 * (c & 3))) & 255u) == (uint32_t)pos ? sv[c] : (short)0;
                    *(hs8_t*)(dt + ((size_t)(2 * rp + (pos >> 1)) * HW + 2 * xp + (pos & 1)) * 8) = v;
                }
            } else if constexpr (CO == 1) {
                if (e >= TH * HW / 4) continue;
                const float4 v = rdy[k][0];
                *(hs4_t*)(dt + 4 * e) = hs4_t{hbf(v.x), hbf(v.y), hbf(v.z), hbf(v.w)};
            } else {
                if (e >= TH * HW * CO / 8) continue;
                ((float4*)dt)[e] = rdy[k][0];
            }
        }
        __syncthreads();
        // ---- 32-pixel blocks of the strip (x0 .. x0 + 31 of row y): lane group kq supplies pixels 8 kq .. + 7 (two transposing reads) ----
        for (int c = wave; c < NCH; c += 4) {
            const int y = c / CPR, x0 = (c % CPR) * 32;
            hbf8_t B;
            if constexpr (CO == 1) {
                B = __builtin_bit_cast(hbf8_t, *(const hs8_t*)(dt + c * 32 + 8 * kq));          // every column the same: column 0 is used
            } else {
                const uint16_t* db = dt + (size_t)(c * 32 + 8 * kq + q) * CO + (CO == 8 ? 4 * (p & 1) : 4 * p);
                B = pk8(tr4(db), tr4(db + 4 * CO));
            }
            accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, B, accb, 0, 0, 0);
            const uint16_t* ab = xt + (size_t)(y * PW + x0 + 8 * kq + q) * CIN;
#pragma unroll
            for (int b = 0; b < NB; ++b)
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pk8(tr4(ab + aoff[b]), tr4(ab + aoff[b] + 4 * CIN)), B, acc[b], 0, 0, 0);
        }
        __syncthreads();            // every wave is done with the tiles before the next strip is staged
    }

    // ---- cross-wave sum, slab row [9][CI][CO] + [CO]: D[m = 4 kq + r][n = l15]: m = row of the block, n = output channel ----
    float* const red = (float*)hsm;                                             // [4 waves][NB + 1][64 lanes] float4
#pragma unroll
    for (int b = 0; b < NB; ++b) *(frag4*)(red + (((size_t)wave * (NB + 1) + b) * 64 + lane) * 4) = acc[b];
    *(frag4*)(red + (((size_t)wave * (NB + 1) + NB) * 64 + lane) * 4) = accb;
    __syncthreads();
    float* const sl = P.slab + (size_t)blockIdx.x * (9 * CI * CO + CO);
    for (int e = tid; e < (NB + 1) * 256; e += 256) {
        const int b = e >> 8, ln = (e >> 2) & 63, r = e & 3, n = ln & 15, g = ln >> 4;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[(((size_t)w * (NB + 1) + b) * 64 + ln) * 4 + r];
        if (n >= CO) continue;
        if (b == NB) {
            if (g == 0 && r == 0) sl[9 * CI * CO + n] = s;
            continue;
        }
        const int tap = CIN == 4 ? 4 * b + g : (CIN == 8 ? 2 * b + (g >> 1) : b);
        const int ch = CIN == 4 ? r : (CIN == 8 ? 4 * (g & 1) + r : 4 * g + r);  // channel of the LDS pixel
        const int ci = ch < CA ? (ch < CA_REAL ? ch : -1) : (ch < CA + CB ? CA_REAL + (ch - CA) : -1);
        if (tap < 9 && ci >= 0) sl[(tap * CI + ci) * CO + n] = s;
    }
}

template <int HW, int CA, int CB, int CO, int TH, bool DYPOOL = false>
struct HWg {
    static constexpr int CIN = CA + CB <= 4 ? 4 : (CA + CB <= 8 ? 8 : 16), TPB = 16 / CIN, NB = (9 + TPB - 1) / TPB;
    static constexpr size_t tiles = (size_t)(((TH + 2) * (HW + 2) * CIN + 7) & ~7) * 2 + (size_t)TH * HW * CO * 2;
    static constexpr size_t red = (size_t)4 * (NB + 1) * 64 * 16;
    static constexpr size_t lds = tiles > red ? tiles : red;
    static int blocks(int n) {
        const int per_cu = (int)((160 * 1024) / lds) < 4 ? (int)((160 * 1024) / lds) : 4;
        const int nstrips = n * (HW / TH), cap = 256 * per_cu;          // as many as stay resident, then as few as walk the same
        if (nstrips <= 0) return 0;                                     // number of strips each
        const int rounds = (nstrips + cap - 1) / cap;
        return (nstrips + rounds - 1) / rounds;
    }
    static int launch(HWgParams P, hipStream_t st) {
        P.nstrips = P.n * (HW / TH);
        auto k = hwgrad_kernel<HW, CA, CB, CO, TH, DYPOOL>;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(k, dim3(blocks(P.n)), dim3(256), lds, st, P);
        CGS_HIP_CHECK_LAUNCH();
        return CGS_OK;
    }
};

using HWgEnc0 = HWg<128, 4, 0, 8, 8>;        // features.0
#ifndef HWG_TH_M0
#define HWG_TH_M0 4
#endif
#ifndef HWG_TH_M2
#define HWG_TH_M2 4
#endif
using HWgMask0 = HWg<128, 4, 8, 16, HWG_TH_M0>;      // masker.0
using HWgMask2 = HWg<128, 16, 0, 1, HWG_TH_M2>;      // masker.2
using HWgEnc1 = HWg<64, 8, 0, 8, 16>;        // features.3
using HWgDec0 = HWg<64, 8, 8, 8, 8>;         // dec_model.0
using HWgEnc0P = HWg<128, 4, 0, 8, 8, true>; // features.0 / features.3 from the pooled gradient + argmax bytes
using HWgEnc1P = HWg<64, 8, 0, 8, 16, true>;
using HWgEnc2 = HWg<32, 8, 0, 8, 32>;        // features.6 / dec_model.1: one image per strip
using HWgEnc2P = HWg<32, 8, 0, 8, 32, true>;
using HWgDec1 = HWg<32, 8, 8, 8, 32>;

int hwg_which(int hw, int ca, int cb, int co) {
    if (hw == 128 && ca == 3 && cb == 0 && co == 8) return 1;
    if (hw == 128 && ca == 3 && cb == 8 && co == 16) return 2;
    if (hw == 128 && ca == 16 && cb == 0 && co == 1) return 3;
    if (hw == 64 && ca == 8 && cb == 0 && co == 8) return 4;
    if (hw == 64 && ca == 8 && cb == 8 && co == 8) return 5;
    if (hw == 32 && ca == 8 && cb == 0 && co == 8) return 6;
    if (hw == 32 && ca == 8 && cb == 8 && co == 8) return 7;
    return 0;
}

}  // namespace

// Slab rows cgs_bf16_hwgrad writes for this shape (0: the shape is not one of the dedicated ones -> cgs_bf16_conv3x3_bwd_weight).
extern "C" int cgs_bf16_hwgrad_slabs(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co) {
    if (n < 0) return CGS_ERR_BADARG;
    switch (hwg_which(hw, ca, cb, co)) {
        case 1: return HWgEnc0::blocks(n);
        case 2: return HWgMask0::blocks(n);
        case 3: return HWgMask2::blocks(n);
        case 4: return HWgEnc1::blocks(n);
        case 5: return HWgDec0::blocks(n);
        case 6: return HWgEnc2::blocks(n);
        case 7: return HWgDec1::blocks(n);
    }
    return 0;
}

// dW / db slabs of conv3x3(cat(A, nearest-up2(B))): a_kind 0 = bf16 [n,hw,hw,ca], 1 = uint8, 2 = fp32 frames [n,hw,hw,3];
// dy: bf16 [n,hw,hw,co] (co > 1) or fp32 [n,hw,hw] (co == 1); slab [cgs_bf16_hwgrad_slabs][9 (ca + cb) co + co].
extern "C" int cgs_bf16_hwgrad(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_kind, const void* src_a, const void* src_b,
                               const void* dy, float* slab, cgs_stream_t stream) {
    const int which = hwg_which(hw, ca, cb, co);
    if (n < 0 || !src_a || !dy || !slab || (cb > 0 && !src_b) || a_kind < 0 || a_kind > 2) return CGS_ERR_BADARG;
    if (!which) return CGS_ERR_UNSUPPORTED;
    if ((ca == 3) != (a_kind != 0)) return CGS_ERR_BADARG;                     // frames are uint8 / fp32, activations bf16
    if (n == 0) return CGS_OK;
    const HWgParams P{src_a, (const uint16_t*)src_b, dy, nullptr, nullptr, slab, n, 0, a_kind == 2 ? 1 : 0, nullptr, nullptr, nullptr};
    switch (which) {
        case 1: return HWgEnc0::launch(P, (hipStream_t)stream);
        case 2: return HWgMask0::launch(P, (hipStream_t)stream);
        case 3: return HWgMask2::launch(P, (hipStream_t)stream);
        case 4: return HWgEnc1::launch(P, (hipStream_t)stream);
        case 5: return HWgDec0::launch(P, (hipStream_t)stream);
        case 6: return HWgEnc2::launch(P, (hipStream_t)stream);
        default: return HWgDec1::launch(P, (hipStream_t)stream);
    }
}

// The same for features.0 (hw 128, ca 3) / features.3 (hw 64, ca 8) with dY given as the pooled gradient dp bf16 [n,hw/2,hw/2,8] (+ addend of
// the same shape or NULL) and the forward pass's argmax bytes: no re-expanded copy of dY in memory.  Slab rows: cgs_bf16_hwgrad_slabs(.., co = 8).
extern "C" int cgs_bf16_hwgrad_pooled(int32_t n, int32_t hw, int32_t ca, int32_t a_kind, const void* src_a, const void* dp, const void* addend,
                                      const uint8_t* codes, float* slab, cgs_stream_t stream) {
    const int which = hwg_which(hw, ca, 0, 8);
    if (n < 0 || !src_a || !dp || !codes || !slab || a_kind < 0 || a_kind > 2) return CGS_ERR_BADARG;
    if (which != 1 && which != 4 && which != 6) return CGS_ERR_UNSUPPORTED;
    if ((ca == 3) != (a_kind != 0)) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    const HWgParams P{src_a, nullptr, dp, (const uint16_t*)addend, codes, slab, n, 0, a_kind == 2 ? 1 : 0, nullptr, nullptr, nullptr};
    return which == 1 ? HWgEnc0P::launch(P, (hipStream_t)stream) : (which == 4 ? HWgEnc1P::launch(P, (hipStream_t)stream) : HWgEnc2P::launch(P, (hipStream_t)stream));
}

// features.0's weight gradient on the VIRTUAL mixes: the 2 n images [replaced | injected] formed from the frame pairs a_u8 / b_u8 [n,128,128,3] and
// the mask z [n,128,128] while the tile is staged (see cgs_bf16_enc0_fwd_mix); dp / codes: the pooled gradient bf16 [2n,64,64,8] / argmax bytes.
// Slab rows: cgs_bf16_hwgrad_slabs(2 n, 128, 3, 0, 8).
extern "C" int cgs_bf16_hwgrad_pooled_mix(int32_t n, const uint8_t* a_u8, const uint8_t* b_u8, const float* z, const void* dp, const uint8_t* codes,
                                          float* slab, cgs_stream_t stream) {
    if (n < 0 || !a_u8 || !b_u8 || !z || !dp || !codes || !slab) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    const HWgParams P{a_u8, nullptr, dp, nullptr, codes, slab, 2 * n, 0, 2, a_u8, b_u8, z};
    return HWgEnc0P::launch(P, (hipStream_t)stream);
}


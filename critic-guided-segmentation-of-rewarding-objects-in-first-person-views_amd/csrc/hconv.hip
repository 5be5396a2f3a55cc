// Whole-strip 16-bit convolutions (h5conv_kernel): the large-map 3x3 layers of
//   * BASELINE config 4, the FUSED fp16 inference path ("-process inference-only mask path, batch 2048, fp16 conv kernels"; main.py:1130-1151,
//     nets.py:170-176, 516-517): features.0 (uint8 frames -> e0), features.3 (e0 -> e1) and dec_model.0 (cat(e0, up(o1)) -> o0) in IEEE half;
//   * BASELINE config 5, the build-defined 128x128 variant (hourglass128.py, chfak 1): every forward layer and data gradient of its 128x128 /
//     64x64 / 32x32 levels in bfloat16 (no reference counterpart: parity unpinned),
// with 16-bit activations in HBM / LDS, 16-bit weights, fp32 accumulation on v_mfma_f32_16x16x32_{f16,bf16} -- gfx950's K = 32 form: one
// instruction covers 8 / 4 / 2 taps of a 4- / 8- / 16-channel input, so a 3x3 layer is 2 / 3 / 5 instructions per 16 pixels and the matrix
// time is negligible; the kernel is built for the memory side and the per-tile instruction count: compile-time shapes (H5Cfg), a workgroup
// persistent over row strips of one image stages a strip once as an NHWC tile whose pixel is ONE 8- / 16- / 32-byte LDS read per lane and tap
// group (every global access a full 16-byte lane access, all loads of a strip issued back to back), the weights live in registers (converted
// from the fp32 master copy; flipped + transposed for a data gradient).  Pooled epilogues issue the instruction as D[pixel][output channel]
// with a tile = four 2x2 pool windows (tile pixel 4 w + pos): a lane's four accumulators are ONE window, so ReLU + MaxPool2d(2) (+ its argmax
// byte) or the 2x2 cell sum is in-lane; plain epilogues issue D[output channel][pixel]: four consecutive channels of the lane's pixel = one
// vector store.  8-channel outputs share an accumulator between two pixel tiles (PAIR).
// The 16x16-and-smaller layers of both configurations stay on the fp32 tail kernels (tail.hip): features.3 of config 4 and features.6 of
// config 5 therefore write fp32.  OPT-IN precisions: never used by the fp32 training path or the parity-gated fp32 paths.
#include "tail_common.h"

#ifndef H5_PREFETCH
#define H5_PREFETCH 0             // h5conv: 1 = the next strip's loads issued before the current strip's matrix loop; 0 = right before their commit
                                  // (r4 A/B at config 5: 1.035 vs 1.048 ms per step -- the 40 more live registers cost more than the overlap returns;
                                  // what pays is issuing a strip's loads back to back: one memory round trip per strip either way)
#endif
#ifndef H5_PER_CU
#define H5_PER_CU 4               // h5conv: persistent workgroups per CU (at most; LDS may allow fewer)
#endif
#ifndef H5_PAIR16
#define H5_PAIR16 0              // h5conv: 1 = tile pairing also for 16-channel pixels with 8 outputs (20 more weight registers; r4 A/B: 0.8 % slower per config-5 step)
#endif
#ifndef H5_QUAD
#define H5_QUAD 0                // h5conv: 1 = three-channel outputs share an accumulator between four pixel tiles (r4 A/B: the four weight sets push the
                                  // kernel past 128 registers -- three instead of four workgroups per CU: 81 vs 64 us for the fused mix backward)
#endif
#ifndef H5_XCD
#define H5_XCD 1                 // h5conv: XCD-contiguous strip order (0: round-robin; r4 A/B: no difference -- the halo rows two strips share come out of
                                  // the memory-side cache either way)
#endif

namespace {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4h_t __attribute__((ext_vector_type(4)));
typedef short bshort8_t __attribute__((ext_vector_type(8)));
typedef short bshort4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bbf16x8_t __attribute__((ext_vector_type(8)));

// 16-bit element of the tile / operands: IEEE half (config 4) or bfloat16 (config 5)
struct ElF16 {
    static constexpr bool IS_F16 = true;
    using V8 = half8_t; using V4 = half4h_t; using S = _Float16;
    __device__ static __forceinline__ S cvt(float f) { return (_Float16)f; }
    __device__ static __forceinline__ frag4 mfma(V8 a, V8 b, frag4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
struct ElBF16 {
    static constexpr bool IS_F16 = false;
    using V8 = bshort8_t; using V4 = bshort4_t; using S = short;
    __device__ static __forceinline__ S cvt(float f) { return (short)__builtin_bit_cast(unsigned short, (__bf16)f); }
    __device__ static __forceinline__ frag4 mfma(V8 a, V8 b, frag4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bbf16x8_t, a), __builtin_bit_cast(bbf16x8_t, b), c, 0, 0, 0);
    }
};

// The epilogues the config-5 training step needs are fused: ReLU + MaxPool2d(2) + argmax bytes, bias + LeakyReLU / Sigmoid, the 2x2 cell sum of
// a nearest-upsampled source's gradient (replaces cgs_bf16_cat_split), the LeakyReLU' factor of the layer below (replaces cgs_bf16_lrelu_bwd),
// a pooled gradient re-expanded while it is staged (replaces cgs_bf16_pool_expand).
struct H5Params {
    const void* a;          // CA == 4: uint8 / fp32 frames [n,HW,HW,3]; CA == 1: fp32 [n,HW,HW]; else bf16 NHWC [n,HW,HW,CA]
    const uint16_t* b;      // CB == 8: bf16 [n,HW/2,HW/2,8], nearest-upsampled; else NULL
    const float* w;         // the layer's HWIO fp32 weights [9][CIL][COL]
    const float* bias;      // [CO] or NULL
    void* out;
    uint8_t* codes;         // EPI_POOLMAX: argmax bytes or NULL (written); APOOL: the argmax bytes of source A (read)
    const uint16_t* hm;     // EPI_LRELU_BWD: the layer's own output of the forward pass, bf16 [n,HW,HW,CO]
    int n, nstrips, a_f32;
    // AMIX / EPI_MIXBWD (features.0's data gradient + the mix backward, main.py:395,406 differentiated): source = the pooled gradient of the replaced
    // mix of image i MINUS that of its injected mix (images n + i of a / codes): d rep - d inj, which is all the mask gradient needs
    // (mix_rep = A + Z (B - A), mix_inj = B - Z (B - A)); epilogue: out = dzpre [n,HW,HW] = (sum_c d_c (B_c - A_c) + l1s sign(Z) + 2 l2s Z) Z (1 - Z)
    const uint8_t* mix_a; const uint8_t* mix_b; const float* mix_z;
    float mix_l1s, mix_l2s;
};

enum { EPI_PLAIN = 0, EPI_POOLMAX = 1, EPI_POOLSUM = 2, EPI_LRELU_BWD = 3, EPI_MIXBWD = 4 };

// CA: LDS channels of source A (4 = rgb0 frames, 1 = fp32 single channel, 8 / 16 = bf16); CB: 0 / 8; CO: output channels of the kernel;
// CIL, COL: the LAYER's input / output channels (weight strides); DGRAD: the kernel computes the data gradient of the layer with respect to
// its input channels O0 .. O0 + CO - 1 (source = dY with COL channels); EPI / ACT: epilogue; OUT_F32: fp32 output (else bf16)
template <int HW_, int TH_, int CA_, int CB_, int CO_, int CIL_, int COL_, int O0_, bool DGRAD_, int EPI_, int ACT_, bool OUT_F32_, bool APOOL_ = false,
          class EL_ = ElBF16, bool BF32_ = false, bool AMIX_ = false>
struct H5Cfg {
    static constexpr bool AMIX = AMIX_;     // APOOL with the source of image i = (pooled gradient of image i) - (that of image n + i): see H5Params.mix_*
    using EL = EL_;                         // 16-bit element: bfloat16 (config 5) or IEEE half (config 4: the fused fp16 inference path)
    static constexpr bool BF32 = BF32_;     // source B is fp32 [n,HW/2,HW/2,8] (config 4: o1 comes out of the fp32 tail kernel)
    static constexpr bool APOOL = APOOL_;   // source A = a pooled gradient re-expanded while it is staged (a = dP bf16 [n,HW/2,HW/2,CA], b = the optional
                                            // addend of the same shape, codes = the forward pass's argmax bytes): replaces cgs_bf16_pool_expand
    static constexpr int HW = HW_, TH = TH_, CA = CA_, CB = CB_, CO = CO_, CIL = CIL_, COL = COL_, O0 = O0_, EPI = EPI_, ACT = ACT_;
    static constexpr bool DGRAD = DGRAD_, OUT_F32 = OUT_F32_, POOL = EPI_ == EPI_POOLMAX || EPI_ == EPI_POOLSUM;
    static constexpr int CIN = CA + CB <= 1 ? 1 : (CA + CB <= 4 ? 4 : (CA + CB <= 8 ? 8 : 16));
    static constexpr int TPM = 32 / CIN, NM = (9 + TPM - 1) / TPM, CA_REAL = CA == 4 ? 3 : CA;
    static constexpr int PW = HW + 2, PH = TH + 2;
    static constexpr size_t lds = (size_t)((PH * PW * CIN + 7) & ~7) * 2;
};

// (round 6, config 4) uint8 frames into an fp16 tile WITHOUT conversions: v_perm_b32 places a byte under the exponent byte 0x3C (= 1 + b / 1024, exact),
// one packed subtract of 1 leaves b / 1024 (as mask_infer_f16_kernel stages its frame tile, round 5); the 1024 / 255 goes into the layer's image
// weights.  18 vector instructions per four pixels instead of ~60 (12 x byte extract + int -> float + scale + fp16 conversion, 4 packs).
#ifndef H5_U8_PERM
#define H5_U8_PERM 1
#endif
template <class C>
__global__ void __launch_bounds__(256) h5conv_kernel(H5Params P) {
    if (CGS_KARG_PREFETCH) cgs_kernarg_prefetch<sizeof(H5Params)>();
    using EL = typename C::EL;
    using v8 = typename EL::V8; using v4 = typename EL::V4; using S = typename EL::S;
    constexpr int HW = C::HW, TH = C::TH, CA = C::CA, CB = C::CB, CO = C::CO, CIN = C::CIN, TPM = C::TPM, NM = C::NM, PW = C::PW, PH = C::PH;
    constexpr int STRIPS = HW / TH, NT = TH * HW / 16, XT = (PH * PW * CIN + 7) & ~7;
    extern __shared__ __attribute__((aligned(16))) float4 hsm[];
    S* const tile = (S*)hsm;                                            // [PH][PW][CIN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;

    // ---- weights -> registers: lane (row l15 = output channel, kq): k = 8 kq + j, tap = g TPM + k / CIN, LDS channel k % CIN.
    //      PAIR (8 output channels, short K): the 16-wide output side of the instruction holds TWO pixel tiles -- tile h of a pair is
    //      multiplied by the weight set whose rows 8 h .. 8 h + 7 are the layer's (the other rows zero) into the SAME accumulator, so every
    //      lane of the epilogue carries a real output (the per-tile instruction count, not the matrix rate, bounds these kernels) ----
    constexpr bool PAIR = CO == 8 && (CIN <= 8 || H5_PAIR16);
    constexpr bool QUAD = CO == 3 && !C::POOL && H5_QUAD;           // three outputs: FOUR tiles per accumulator (tile h = rows 4 h .. 4 h + 2)
    constexpr int NP = PAIR ? 2 : (QUAD ? 4 : 1);
    const int ocl = PAIR ? (l15 & 7) : (QUAD ? (l15 & 3) : l15);
    v8 wa[NP][NM];
#pragma unroll
    for (int h = 0; h < NP; ++h)
#pragma unroll
    for (int g = 0; g < NM; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * kq + j, tap = g * TPM + k / CIN, c = k % CIN;
            int idx = -1;
            if (tap < 9 && (PAIR ? (l15 >> 3) == h : (QUAD ? (l15 >> 2) == h && ocl < CO : l15 < CO))) {
                if constexpr (C::DGRAD) {       // d x[ci = O0 + oc] = sum over (tap', co) of dY[p + off(tap')][co] W[8 - tap'][ci][co]
                    if (c < C::COL) idx = ((8 - tap) * C::CIL + C::O0 + ocl) * C::COL + c;
                } else {
                    const int cr = c < CA ? (c < C::CA_REAL ? c : -1) : (c < CA + CB ? C::CA_REAL + (c - CA) : -1);
                    if (cr >= 0) idx = (tap * C::CIL + cr) * C::COL + ocl;
                }
            }
            // (unconditional load + select: `idx >= 0 ? P.w[idx] : 0` is a branch around the load, and every one of the 8 NM NP loads is then
            //  waited for before the next is issued -- up to 72 dependent L1 / L2 round trips at the start of every workgroup)
            const float wv = P.w[idx >= 0 ? idx : 0];
            // (H5_U8_PERM: the uint8 frame's tile holds b / 1024, not b / 255)
            const float wsc = (H5_U8_PERM && EL::IS_F16 && CA == 4 && CIN == 4 && !C::DGRAD && c < CA) ? 1024.f / 255.f : 1.f;
            wa[h][g][j] = EL::cvt(idx >= 0 ? wv * wsc : 0.f);
        }
    float br[4] = {0.f, 0.f, 0.f, 0.f};
    float bl = 0.f;
    if (P.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = PAIR ? (4 * kq + r) & 7 : (QUAD ? r : 4 * kq + r);
            const float bv = P.bias[ch < CO ? ch : 0];
            br[r] = ch < CO ? bv : 0.f;
        }
        const float bv = P.bias[ocl < CO ? ocl : 0];
        bl = ocl < CO ? bv : 0.f;
    }
    int toff1[8];                               // CIN == 1: this lane's eight taps (8 kq + j, clamped: the weights of taps >= 9 are zero)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int t = 8 * kq + j;
        t = t < 9 ? t : 8;
        toff1[j] = (t / 3) * PW + t % 3;
    }

    for (int e = tid; e < XT / 8; e += 256) ((float4*)tile)[e] = f4zero();       // halo columns + padding channels: zero for every strip

    // ---- staging, split: fetch = every global load of a strip issued back to back into registers (ONE memory round trip per strip, in
    //      optionally -- H5_PREFETCH -- in flight while the previous strip multiplies), commit = conversion + LDS stores.  Items: source A = 4 frame pixels (3 dwords /
    //      3 float4), 4 fp32 pixels or 8 bf16 channels; source B = one low-resolution pixel (8 channels) -> the two tile pixels above it.
    constexpr int GW = HW / 4;
    constexpr int HP = HW / 2, PR = TH / 2 + 2;                                  // APOOL: pooled rows under a strip's tile
    constexpr int NA = C::APOOL ? (PR * HP + 255) / 256 : (CA == 4 || CA == 1 ? (PH * GW + 255) / 256 : (PH * HW * (CA / 8) + 255) / 256);
    constexpr int NAV = C::AMIX ? 4 : (CA == 4 || C::APOOL ? 3 : 1);               // float4 registers per item
    static_assert(!C::APOOL || (CA == 8 && CB == 0 && sizeof(S) == 2 && !C::BF32), "pooled source: 8 bf16 channels");
    constexpr int NBI = CB == 8 ? (PH * (HW / 2) + 255) / 256 : 1;
    float4 ra[NA][NAV];
    float4 rb[NBI][C::BF32 ? 2 : 1];
    auto fetch = [&](int strip) {
        const int img = strip / STRIPS, row0 = (strip % STRIPS) * TH;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + 256 * i;
            if constexpr (C::APOOL) {
                const int xp = e % HP, rp = e / HP, yp = row0 / 2 - 1 + rp;
                const bool in = rp < PR && yp >= 0 && yp < HP;
                const size_t gi = in ? ((size_t)img * HP + yp) * HP + xp : 0;
                ra[i][0] = ((const float4*)P.a)[gi];
                const float2 cd = ((const float2*)P.codes)[gi];
                if constexpr (C::AMIX) {        // the injected mix of the same A-image: n images further
                    const size_t gj = gi + (size_t)P.n * HP * HP;
                    ra[i][1] = ((const float4*)P.a)[gj];
                    const float2 ce = ((const float2*)P.codes)[gj];
                    ra[i][2] = make_float4(cd.x, cd.y, ce.x, ce.y);
                } else {
                    ra[i][1] = P.b ? ((const float4*)P.b)[gi] : f4zero();
                    ra[i][2] = make_float4(cd.x, cd.y, 0.f, 0.f);
                }
            } else if constexpr (CA == 4) {
                const int g = e % GW, r = e / GW, y = row0 + r - 1;
                const bool in = r < PH && y >= 0 && y < HW;
                const size_t gi = in ? (((size_t)img * HW + y) * HW + g * 4) * 3 / 4 : 0;
                if constexpr (H5_U8_PERM && EL::IS_F16 && CIN == 4) {      // (the fp16 instance: uint8 frames only, see commit below)
                    const uint32_t* su = (const uint32_t*)P.a;
                    ra[i][0] = make_float4(__uint_as_float(su[gi]), __uint_as_float(su[gi + 1]), __uint_as_float(su[gi + 2]), 0.f);
                } else if (P.a_f32 == 2) {             // virtual mixes (main.py:395,406): image img < n / 2 = A (1 - Z) + Z B of frame pair img, else B (1 - Z) + Z A
                    const int half = P.n >> 1, is = img < half ? img : img - half;
                    const size_t gs = in ? (((size_t)is * HW + y) * HW + g * 4) : 0;
                    const uint32_t* sa = (const uint32_t*)P.mix_a + gs * 3 / 4;
                    const uint32_t* sb = (const uint32_t*)P.mix_b + gs * 3 / 4;
                    ra[i][0] = make_float4(__uint_as_float(sa[0]), __uint_as_float(sa[1]), __uint_as_float(sa[2]), 0.f);
                    ra[i][1] = make_float4(__uint_as_float(sb[0]), __uint_as_float(sb[1]), __uint_as_float(sb[2]), 0.f);
                    ra[i][2] = ((const float4*)P.mix_z)[gs / 4];
                } else if (P.a_f32) {
                    const float4* sf = (const float4*)P.a;
                    ra[i][0] = sf[gi]; ra[i][1] = sf[gi + 1]; ra[i][2] = sf[gi + 2];
                } else {
                    const uint32_t* su = (const uint32_t*)P.a;
                    ra[i][0] = make_float4(__uint_as_float(su[gi]), __uint_as_float(su[gi + 1]), __uint_as_float(su[gi + 2]), 0.f);
                }
            } else if constexpr (CA == 1) {
                const int g = e % GW, r = e / GW, y = row0 + r - 1;
                const bool in = r < PH && y >= 0 && y < HW;
                ra[i][0] = ((const float4*)P.a)[in ? (((size_t)img * HW + y) * HW) / 4 + g : 0];
            } else {
                constexpr int NG = CA / 8;
                const int g = e % NG, x = (e / NG) % HW, r = e / (NG * HW), y = row0 + r - 1;
                const bool in = r < PH && y >= 0 && y < HW;
                ra[i][0] = ((const float4*)P.a)[in ? (((size_t)img * HW + y) * HW + x) * NG + g : 0];
            }
        }
        if constexpr (CB == 8) {
#pragma unroll
            for (int i = 0; i < NBI; ++i) {
                const int e = tid + 256 * i, xc = e % (HW / 2), r = e / (HW / 2), y = row0 + r - 1;
                const bool in = r < PH && y >= 0 && y < HW;
                const size_t gi = in ? ((size_t)img * (HW / 2) + (y >> 1)) * (HW / 2) + xc : 0;
                if constexpr (C::BF32) { rb[i][0] = ((const float4*)P.b)[2 * gi]; rb[i][1] = ((const float4*)P.b)[2 * gi + 1]; }
                else rb[i][0] = ((const float4*)P.b)[gi];
            }
        }
    };
    auto commit = [&](int strip) {
        const int row0 = (strip % STRIPS) * TH;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + 256 * i;
            if constexpr (C::APOOL) {           // one pooled pixel -> the 2 x 2 tile pixels above it: the gradient goes where the maximum was
                const int xp = e % HP, rp = e / HP, yp = row0 / 2 - 1 + rp;
                if (rp >= PR) continue;
                const bool in = yp >= 0 && yp < HP;
                const uint32_t dw[4] = {__float_as_uint(ra[i][0].x), __float_as_uint(ra[i][0].y), __float_as_uint(ra[i][0].z), __float_as_uint(ra[i][0].w)};
                const uint32_t aw[4] = {__float_as_uint(ra[i][1].x), __float_as_uint(ra[i][1].y), __float_as_uint(ra[i][1].z), __float_as_uint(ra[i][1].w)};
                const uint32_t cw[2] = {__float_as_uint(ra[i][2].x), __float_as_uint(ra[i][2].y)};
                if constexpr (C::AMIX) {        // (replaced - injected), each where its own maximum was: rounded to bf16 once
                    const uint32_t ce[2] = {__float_as_uint(ra[i][2].z), __float_as_uint(ra[i][2].w)};
                    float fr[8], fi[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const uint32_t sh = 16 * (c & 1);
                        fr[c] = __uint_as_float(((dw[c >> 1] >> sh) & 0xffffu) << 16);
                        fi[c] = __uint_as_float(((aw[c >> 1] >> sh) & 0xffffu) << 16);
                    }
#pragma unroll
                    for (int pos = 0; pos < 4; ++pos) {
                        const int r = 2 * rp - 1 + (pos >> 1);
                        if (r < 0 || r >= PH) continue;
                        v8 v;
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const float a = ((cw[c >> 2] >> (8 * (c & 3))) & 255u) == (uint32_t)pos ? fr[c] : 0.f;
                            const float b = ((ce[c >> 2] >> (8 * (c & 3))) & 255u) == (uint32_t)pos ? fi[c] : 0.f;
                            v[c] = in ? EL::cvt(a - b) : (short)0;
                        }
                        *(v8*)(tile + ((size_t)r * PW + 1 + 2 * xp + (pos & 1)) * CIN) = v;
                    }
                    continue;
                }
                short sv[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t sh = 16 * (c & 1);
                    sv[c] = EL::cvt(__uint_as_float(((dw[c >> 1] >> sh) & 0xffffu) << 16) + __uint_as_float(((aw[c >> 1] >> sh) & 0xffffu) << 16));
                }
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                    const int r = 2 * rp - 1 + (pos >> 1);
                    if (r < 0 || r >= PH) continue;
                    v8 v;
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = in && ((cw[c >> 2] >> (8 * (c & 3))) & 255u) == (uint32_t)pos ? sv[c] : (short)0;
                    *(v8*)(tile + ((size_t)r * PW + 1 + 2 * xp + (pos & 1)) * CIN) = v;
                }
            } else if constexpr (CA == 4) {
                const int g = e % GW, r = e / GW, y = row0 + r - 1;
                if (r >= PH) continue;
                const bool in = y >= 0 && y < HW;
                if constexpr (H5_U8_PERM && EL::IS_F16 && CIN == 4) {
                    {          // four uint8 pixels = three dwords -> four (r, g, b, 0) half quadruples (the fp16 instance only ever reads uint8 frames:
                               // cgs_f16_enc0_fwd; a run-time test of P.a_f32 here kept both forms' registers alive: 128 instead of 120, three waves per SIMD)
                        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                        const uint32_t e0 = in ? __float_as_uint(ra[i][0].x) : 0u, e1 = in ? __float_as_uint(ra[i][0].y) : 0u,
                                       e2 = in ? __float_as_uint(ra[i][0].z) : 0u;      // rows outside the image: zero bytes -> 0.0
                        const uint32_t q1 = __builtin_amdgcn_alignbyte(e1, e0, 3), q2 = __builtin_amdgcn_alignbyte(e2, e1, 2);
                        constexpr uint32_t K = 0x3C3C3C3Cu;       // selectors 0 .. 3: bytes of the pixel dword; 4: 0x3C; 0x0c: 0x00
                        auto cv = [&](uint32_t q, uint32_t sel_rg, uint32_t sel_b) {
                            const h2_t rg = __builtin_bit_cast(h2_t, __builtin_amdgcn_perm(K, q, sel_rg)) - h2_t{(_Float16)1.f, (_Float16)1.f};
                            const h2_t b0 = __builtin_bit_cast(h2_t, __builtin_amdgcn_perm(K, q, sel_b)) - h2_t{(_Float16)1.f, (_Float16)0.f};
                            return make_uint2(__builtin_bit_cast(uint32_t, rg), __builtin_bit_cast(uint32_t, b0));
                        };
                        uint2* d = (uint2*)(tile + ((size_t)r * PW + 1 + 4 * g) * CIN);
                        d[0] = cv(e0, 0x04010400u, 0x0c0c0402u);
                        d[1] = cv(q1, 0x04010400u, 0x0c0c0402u);
                        d[2] = cv(q2, 0x04010400u, 0x0c0c0402u);
                        d[3] = cv(e2, 0x04020401u, 0x0c0c0403u);
                        continue;
                    }
                }
                float f[12];
                if (P.a_f32 == 2) {             // (the formula of cgs_mix_fwd: the fp32 mix it used to write, up to FMA contraction)
                    const uint32_t da[3] = {__float_as_uint(ra[i][0].x), __float_as_uint(ra[i][0].y), __float_as_uint(ra[i][0].z)};
                    const uint32_t db[3] = {__float_as_uint(ra[i][1].x), __float_as_uint(ra[i][1].y), __float_as_uint(ra[i][1].z)};
                    const float zv[4] = {ra[i][2].x, ra[i][2].y, ra[i][2].z, ra[i][2].w};
                    const bool inj = (strip / STRIPS) >= (P.n >> 1);
#pragma unroll
                    for (int j = 0; j < 12; ++j) {
                        const float av = (float)((da[j >> 2] >> (8 * (j & 3))) & 255u) * (1.f / 255.f);
                        const float bv = (float)((db[j >> 2] >> (8 * (j & 3))) & 255u) * (1.f / 255.f);
                        const float zi = zv[j / 3];
                        f[j] = inj ? bv * (1.f - zi) + zi * av : av * (1.f - zi) + zi * bv;
                    }
                } else if (P.a_f32) {
                    const float4 f0 = ra[i][0], f1 = ra[i][1], f2 = ra[i][2];
                    f[0] = f0.x; f[1] = f0.y; f[2] = f0.z; f[3] = f0.w; f[4] = f1.x; f[5] = f1.y; f[6] = f1.z; f[7] = f1.w;
                    f[8] = f2.x; f[9] = f2.y; f[10] = f2.z; f[11] = f2.w;
                } else {
                    const uint32_t d[3] = {__float_as_uint(ra[i][0].x), __float_as_uint(ra[i][0].y), __float_as_uint(ra[i][0].z)};
#pragma unroll
                    for (int j = 0; j < 12; ++j) f[j] = (float)((d[j >> 2] >> (8 * (j & 3))) & 255u) * (1.f / 255.f);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v4 v = in ? v4{EL::cvt(f[3 * j]), EL::cvt(f[3 * j + 1]), EL::cvt(f[3 * j + 2]), 0} : v4{0, 0, 0, 0};
                    *(v4*)(tile + ((size_t)r * PW + 1 + 4 * g + j) * CIN) = v;
                }
            } else if constexpr (CA == 1) {     // (the tile row starts 2 bytes after an 8-byte boundary: four 2-byte stores)
                const int g = e % GW, r = e / GW, y = row0 + r - 1;
                if (r >= PH) continue;
                const bool in = y >= 0 && y < HW;
                const float4 v = in ? ra[i][0] : f4zero();
                S* d = tile + (size_t)r * PW + 1 + 4 * g;
                d[0] = EL::cvt(v.x); d[1] = EL::cvt(v.y); d[2] = EL::cvt(v.z); d[3] = EL::cvt(v.w);
            } else {
                constexpr int NG = CA / 8;
                const int g = e % NG, x = (e / NG) % HW, r = e / (NG * HW), y = row0 + r - 1;
                if (r >= PH) continue;
                const bool in = y >= 0 && y < HW;
                *(float4*)(tile + ((size_t)r * PW + 1 + x) * CIN + 8 * g) = in ? ra[i][0] : f4zero();
            }
        }
        if constexpr (CB == 8) {
#pragma unroll
            for (int i = 0; i < NBI; ++i) {
                const int e = tid + 256 * i, xc = e % (HW / 2), r = e / (HW / 2), y = row0 + r - 1;
                if (r >= PH) continue;
                const bool in = y >= 0 && y < HW;
                S* d = tile + ((size_t)r * PW + 1 + 2 * xc) * CIN + CA;    // 8-byte aligned (CA = 4) or 16
                float2 lo, hi;
                if constexpr (C::BF32) {
                    const float4 u0 = in ? rb[i][0] : f4zero(), u1 = in ? rb[i][1] : f4zero();
                    const v4 l4 = {EL::cvt(u0.x), EL::cvt(u0.y), EL::cvt(u0.z), EL::cvt(u0.w)}, h4 = {EL::cvt(u1.x), EL::cvt(u1.y), EL::cvt(u1.z), EL::cvt(u1.w)};
                    lo = __builtin_bit_cast(float2, l4); hi = __builtin_bit_cast(float2, h4);
                } else {
                    const float4 v = in ? rb[i][0] : f4zero();
                    lo = make_float2(v.x, v.y); hi = make_float2(v.z, v.w);
                }
                *(float2*)d = lo; *(float2*)(d + 4) = hi;
                *(float2*)(d + CIN) = lo; *(float2*)(d + CIN + 4) = hi;
            }
        }
    };

    const int vb = H5_XCD ? cgs_xcd_contiguous(blockIdx.x, gridDim.x) : (int)blockIdx.x;   // neighbouring strips (shared halo rows) on one XCD's L2
    if (H5_PREFETCH && vb < P.nstrips) fetch(vb);
    __syncthreads();                                                             // (the zeroes above, before other threads stage the same addresses)
    for (int strip = vb; strip < P.nstrips; strip += gridDim.x) {
    const int img = strip / STRIPS, row0 = (strip % STRIPS) * TH;
    if (!H5_PREFETCH) fetch(strip);
    commit(strip);
    __syncthreads();
    if (H5_PREFETCH && strip + (int)gridDim.x < P.nstrips) fetch(strip + gridDim.x);   // in flight while this strip multiplies

    // ---- tiles: 16 pixels = 4 pool windows adjacent in x (lane = 4 window + position), or 16 consecutive pixels of a row ----
    auto tile_pixel = [&](int t, int& y, int& x) {      // strip-local pixel of this lane in tile t
        if constexpr (C::POOL) {
            constexpr int TPR = HW / 8;
            const int wy = t / TPR, tx = t % TPR, win = l15 >> 2, pos = l15 & 3;
            y = 2 * wy + (pos >> 1); x = 8 * tx + 2 * win + (pos & 1);
        } else {
            constexpr int TPR = HW / 16;
            y = t / TPR; x = 16 * (t % TPR) + l15;
        }
    };
    for (int t = NP * wave; t < NT; t += 4 * NP) {
        frag4 acc = frag4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < NP; ++h) {
        int y, x;
        tile_pixel(t + h, y, x);
        const S* pix = tile + ((size_t)y * PW + x) * CIN;                   // tap (0,0) of the lane's 3x3 window
#pragma unroll
        for (int g = 0; g < NM; ++g) {
            v8 b;
            if constexpr (CIN == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] = pix[toff1[j]];
            } else if constexpr (CIN == 4) {
                int t0 = g * 8 + 2 * kq, t1 = t0 + 1;
                t0 = t0 < 9 ? t0 : 8; t1 = t1 < 9 ? t1 : 8;
                const v4 lo = *(const v4*)(pix + ((t0 / 3) * PW + t0 % 3) * 4), hi = *(const v4*)(pix + ((t1 / 3) * PW + t1 % 3) * 4);
                b = v8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            } else if constexpr (CIN == 8) {
                int tp = g * 4 + kq;
                tp = tp < 9 ? tp : 8;
                b = *(const v8*)(pix + ((tp / 3) * PW + tp % 3) * 8);
            } else {
                int tp = g * 2 + (kq >> 1);
                tp = tp < 9 ? tp : 8;
                b = *(const v8*)(pix + ((tp / 3) * PW + tp % 3) * 16 + 8 * (kq & 1));
            }
            if constexpr (C::POOL) acc = EL::mfma(b, wa[h][g], acc);            // D[pixel = 4 kq + r][oc = l15]: a lane's four values = one 2x2 window
            else acc = EL::mfma(wa[h][g], b, acc);                              // D[oc = 4 kq + r][pixel = l15]
        }
        }
        if constexpr (C::POOL) {
            constexpr int TPR = HW / 8;
            const int te = t + (PAIR ? l15 >> 3 : 0), wy = te / TPR, tx = te % TPR;     // the tile this lane's column belongs to
            if (PAIR || l15 < CO) {
                const size_t o = ((((size_t)img * (HW / 2) + row0 / 2 + wy) * (HW / 2) + 4 * tx + kq) * CO) + ocl;
                float m;
                if constexpr (C::EPI == EPI_POOLSUM) m = (acc[0] + acc[1]) + (acc[2] + acc[3]);
                else m = fmaxf(fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])) + bl, 0.f);
                if constexpr (C::OUT_F32) ((float*)P.out)[o] = m;
                else ((S*)P.out)[o] = EL::cvt(m);
                if constexpr (C::EPI == EPI_POOLMAX) {
                    if (P.codes) {              // first maximum wins (max_pool2d); 4 = the pooled value is not positive: no gradient
                        uint32_t code = 0;
                        float mm = acc[0] + bl;
#pragma unroll
                        for (int j = 1; j < 4; ++j) if (acc[j] + bl > mm) { mm = acc[j] + bl; code = j; }
                        P.codes[o] = (uint8_t)(m > 0.f ? code : 4u);
                    }
                }
            }
        } else if (NP > 1 || 4 * kq < CO) {
            int y, x;
            tile_pixel(t + (PAIR ? kq >> 1 : (QUAD ? kq : 0)), y, x);            // the tile this lane's rows belong to
            const size_t o = ((((size_t)img * HW + row0 + y) * HW + x) * CO) + (PAIR ? 4 * (kq & 1) : (QUAD ? 0 : 4 * kq));
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[r] + br[r];
                if constexpr (C::ACT == CGS_ACT_LRELU) v[r] = v[r] > 0.f ? v[r] : 0.01f * v[r];
                if constexpr (C::ACT == CGS_ACT_SIGMOID) v[r] = 1.f / (1.f + __expf(-v[r]));
                if constexpr (C::ACT == CGS_ACT_RELU) v[r] = fmaxf(v[r], 0.f);
            }
            if constexpr (C::EPI == EPI_MIXBWD) {                                // (CO = 3: the lanes kq == 0 hold the pixel's d rep - d inj)
                const size_t pxi = ((size_t)img * HW + row0 + y) * HW + x;
                const uint8_t* pa = P.mix_a + 3 * pxi;
                const uint8_t* pb = P.mix_b + 3 * pxi;
                const float zi = P.mix_z[pxi];
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) d = fmaf((float)pb[c] * (1.f / 255.f) - (float)pa[c] * (1.f / 255.f), v[c], d);
                const float sg = zi > 0.f ? 1.f : (zi < 0.f ? -1.f : 0.f);
                d += P.mix_l1s * sg + 2.f * P.mix_l2s * zi;
                ((float*)P.out)[pxi] = d * zi * (1.f - zi);
                continue;
            }
            if constexpr (C::EPI == EPI_LRELU_BWD) {                             // x LeakyReLU'(h): h = the layer's forward output (bf16: sign is exact)
                const v4 h = *(const v4*)((const S*)P.hm + o);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = h[r] > 0 ? v[r] : 0.01f * v[r];
            }
            if constexpr (CO % 4 == 0) {
                if constexpr (C::OUT_F32) *(float4*)((float*)P.out + o) = make_float4(v[0], v[1], v[2], v[3]);
                else *(v4*)((S*)P.out + o) = v4{EL::cvt(v[0]), EL::cvt(v[1]), EL::cvt(v[2]), EL::cvt(v[3])};
            } else {
#pragma unroll
                for (int r = 0; r < CO; ++r) {
                    if constexpr (C::OUT_F32) ((float*)P.out)[o + r] = v[r];
                    else ((S*)P.out)[o + r] = EL::cvt(v[r]);
                }
            }
        }
    }
    __syncthreads();            // every wave is done with the tile before the next strip is staged
    }
}

template <class C>
int h5_launch(H5Params P, hipStream_t st) {
    if (P.n <= 0) return CGS_OK;
    P.nstrips = P.n * (C::HW / C::TH);
    // persistent workgroups: as many as stay resident (LDS; at most 4 per CU), then as few as walk the same number of strips each
    const int per_cu = (int)((160 * 1024) / C::lds) < H5_PER_CU ? (int)((160 * 1024) / C::lds) : H5_PER_CU, cap = 256 * per_cu;
    const int rounds = (P.nstrips + cap - 1) / cap, blocks = (P.nstrips + rounds - 1) / rounds;
    hipLaunchKernelGGL((h5conv_kernel<C>), dim3(blocks), dim3(256), C::lds, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

//                      HW   TH  CA CB  CO CIL COL O0 DGRAD  EPI            ACT              OUT_F32
using H4Enc0F  = H5Cfg< 64, 32,  4, 0,  8,  3,  8, 0, false, EPI_POOLMAX,   CGS_ACT_RELU,    false, false, ElF16>;          // config 4 (fp16): features.0
using H4Enc1F  = H5Cfg< 32, 32,  8, 0,  8,  8,  8, 0, false, EPI_POOLMAX,   CGS_ACT_RELU,    true,  false, ElF16>;          // features.3 -> fp32 (tail kernels)
using H4Dec0F  = H5Cfg< 32, 32,  8, 8,  8, 16,  8, 0, false, EPI_PLAIN,     CGS_ACT_NONE,    false, false, ElF16, true>;    // dec_model.0 (o1 fp32)
using H5Enc0F  = H5Cfg<128, 16,  4, 0,  8,  3,  8, 0, false, EPI_POOLMAX,   CGS_ACT_RELU,    false>;   // features.0 forward (+ argmax bytes)
using H5Mask0F = H5Cfg<128,  8,  4, 8, 16, 11, 16, 0, false, EPI_PLAIN,     CGS_ACT_LRELU,   false>;   // masker.0 forward
using H5Mask2F = H5Cfg<128,  8, 16, 0,  1, 16,  1, 0, false, EPI_PLAIN,     CGS_ACT_SIGMOID, true>;    // masker.2 forward
using H5Enc0D  = H5Cfg<128, 16,  8, 0,  3,  3,  8, 0, true,  EPI_PLAIN,     CGS_ACT_NONE,    true>;    // features.0: image gradient
using H5Enc0DP = H5Cfg<128, 16,  8, 0,  3,  3,  8, 0, true,  EPI_PLAIN,     CGS_ACT_NONE,    true,  true>;   // ... from the pooled gradient
using H5Enc0DM = H5Cfg<128, 16,  8, 0,  3,  3,  8, 0, true,  EPI_MIXBWD,    CGS_ACT_NONE,    true,  true, ElBF16, false, true>;   // ... + the mix backward
using H5Mask2D = H5Cfg<128, 16,  1, 0, 16, 16,  1, 0, true,  EPI_LRELU_BWD, CGS_ACT_NONE,    false>;   // masker.2: d hm (x LeakyReLU')
using H5Mask0D = H5Cfg<128,  8, 16, 0,  8, 11, 16, 3, true,  EPI_POOLSUM,   CGS_ACT_NONE,    false>;   // masker.0: d o0 (2x2 cell sums)
using H5Enc1F  = H5Cfg< 64, 16,  8, 0,  8,  8,  8, 0, false, EPI_POOLMAX,   CGS_ACT_RELU,    false>;   // features.3 forward (+ argmax bytes)
using H5Enc1D  = H5Cfg< 64, 16,  8, 0,  8,  8,  8, 0, true,  EPI_PLAIN,     CGS_ACT_NONE,    false>;   // features.3: d e0
using H5Enc1DP = H5Cfg< 64, 16,  8, 0,  8,  8,  8, 0, true,  EPI_PLAIN,     CGS_ACT_NONE,    false, true>;   // ... from the pooled gradient
using H5Dec0F  = H5Cfg< 64,  8,  8, 8,  8, 16,  8, 0, false, EPI_PLAIN,     CGS_ACT_NONE,    false>;   // dec_model.0 forward
using H5Dec0DS = H5Cfg< 64, 16,  8, 0,  8, 16,  8, 0, true,  EPI_PLAIN,     CGS_ACT_NONE,    false>;   // dec_model.0: d skip (e0)
using H5Dec0DL = H5Cfg< 64, 16,  8, 0,  8, 16,  8, 8, true,  EPI_POOLSUM,   CGS_ACT_NONE,    false>;   // dec_model.0: d o1 (2x2 cell sums)
using H5Enc2F  = H5Cfg< 32, 32,  8, 0,  8,  8,  8, 0, false, EPI_POOLMAX,   CGS_ACT_RELU,    true>;    // features.6 forward -> fp32 (the tail kernels' input)
using H5Enc2DP = H5Cfg< 32, 32,  8, 0,  8,  8,  8, 0, true,  EPI_PLAIN,     CGS_ACT_NONE,    false, true>;   // features.6: d e1 from the pooled gradient
using H5Dec1F  = H5Cfg< 32, 16,  8, 8,  8, 16,  8, 0, false, EPI_PLAIN,     CGS_ACT_NONE,    false>;   // dec_model.1 forward
using H5Dec1FF = H5Cfg< 32, 16,  8, 8,  8, 16,  8, 0, false, EPI_PLAIN,     CGS_ACT_NONE,    false, false, ElBF16, true>;   // ... with o2 in fp32 (straight from the tail kernel)
using H5Dec1DS = H5Cfg< 32, 32,  8, 0,  8, 16,  8, 0, true,  EPI_PLAIN,     CGS_ACT_NONE,    false>;   // dec_model.1: d skip (e1)
using H5Dec1DL = H5Cfg< 32, 32,  8, 0,  8, 16,  8, 8, true,  EPI_POOLSUM,   CGS_ACT_NONE,    true>;    // dec_model.1: d o2 (2x2 cell sums, fp32: the tail kernels' input)

}  // namespace

extern "C" int cgs_f16_enc0_fwd(int32_t n, const uint8_t* x_u8, const float* w_hwio, const float* bias, void* e0_f16, cgs_stream_t stream) {
    if (n < 0 || !x_u8 || !w_hwio || !bias || !e0_f16) return CGS_ERR_BADARG;
    return h5_launch<H4Enc0F>(H5Params{x_u8, nullptr, w_hwio, bias, e0_f16, nullptr, nullptr, n, 0, 0 /* a_f32: this instance reads uint8 frames only (H5_U8_PERM) */}, (hipStream_t)stream);
}

extern "C" int cgs_f16_enc1_fwd(int32_t n, const void* e0_f16, const float* w_hwio, const float* bias, float* e1_f32, cgs_stream_t stream) {
    if (n < 0 || !e0_f16 || !w_hwio || !bias || !e1_f32) return CGS_ERR_BADARG;
    return h5_launch<H4Enc1F>(H5Params{e0_f16, nullptr, w_hwio, bias, e1_f32, nullptr, nullptr, n, 0, 0}, (hipStream_t)stream);
}

extern "C" int cgs_f16_dec0_fwd(int32_t n, const void* e0_f16, const float* o1_f32, const float* w_hwio, const float* bias, void* o0_f16,
                                cgs_stream_t stream) {
    if (n < 0 || !e0_f16 || !o1_f32 || !w_hwio || !bias || !o0_f16) return CGS_ERR_BADARG;
    return h5_launch<H4Dec0F>(H5Params{e0_f16, (const uint16_t*)o1_f32, w_hwio, bias, o0_f16, nullptr, nullptr, n, 0, 0}, (hipStream_t)stream);
}

// config 5 (the build-defined 128x128 variant, hourglass128.py): features.0 (3 -> 8 at 128x128 + ReLU + MaxPool2d(2)) with bf16 output
// [n,64,64,8] on v_mfma_f32_16x16x32_bf16; x: uint8 frames (x_is_f32 = 0) or fp32 frames [n,128,128,3] (the replaced / injected mixes);
// codes (optional, training): the argmax bytes cgs_bf16_pool_expand consumes.
extern "C" int cgs_bf16_enc0_fwd(int32_t n, const void* x, int32_t x_is_f32, const float* w_hwio, const float* bias, void* e0_bf16,
                                 uint8_t* codes, cgs_stream_t stream) {
    if (n < 0 || !x || !w_hwio || !bias || !e0_bf16) return CGS_ERR_BADARG;
    return h5_launch<H5Enc0F>(H5Params{x, nullptr, w_hwio, bias, e0_bf16, codes, nullptr, n, 0, x_is_f32 ? 1 : 0}, (hipStream_t)stream);
}

// ---- config 5, the 128x128 layers of the training step (h5conv_kernel above); weights / bias: the layer's fp32 HWIO master parameters ----
// masker.0 forward: LeakyReLU(conv3x3(cat(frames uint8 [n,128,128,3] / 255, nearest-up2(o0 bf16 [n,64,64,8])))) -> hm bf16 [n,128,128,16]
extern "C" int cgs_bf16_mask0_fwd(int32_t n, const uint8_t* x_u8, const void* o0_bf16, const float* w_hwio, const float* bias, void* hm_bf16,
                                  cgs_stream_t stream) {
    if (n < 0 || !x_u8 || !o0_bf16 || !w_hwio || !bias || !hm_bf16) return CGS_ERR_BADARG;
    return h5_launch<H5Mask0F>(H5Params{x_u8, (const uint16_t*)o0_bf16, w_hwio, bias, hm_bf16, nullptr, nullptr, n, 0, 0}, (hipStream_t)stream);
}
// masker.2 forward: Sigmoid(conv3x3(hm)) -> Z fp32 [n,128,128]
extern "C" int cgs_bf16_mask2_fwd(int32_t n, const void* hm_bf16, const float* w_hwio, const float* bias, float* z, cgs_stream_t stream) {
    if (n < 0 || !hm_bf16 || !w_hwio || !bias || !z) return CGS_ERR_BADARG;
    return h5_launch<H5Mask2F>(H5Params{hm_bf16, nullptr, w_hwio, bias, z, nullptr, nullptr, n, 0, 0}, (hipStream_t)stream);
}
// features.0 data gradient: dy bf16 [n,128,128,8] (the re-expanded pooled gradient) -> d frames fp32 [n,128,128,3]
extern "C" int cgs_bf16_enc0_bwd_data(int32_t n, const void* dy_bf16, const float* w_hwio, float* dx, cgs_stream_t stream) {
    if (n < 0 || !dy_bf16 || !w_hwio || !dx) return CGS_ERR_BADARG;
    return h5_launch<H5Enc0D>(H5Params{dy_bf16, nullptr, w_hwio, nullptr, dx, nullptr, nullptr, n, 0, 0}, (hipStream_t)stream);
}
// the same from the pooled gradient: dp bf16 [n,64,64,8] (+ addend or NULL), codes = the argmax bytes of the forward pass (no re-expanded copy)
extern "C" int cgs_bf16_enc0_bwd_data_pooled(int32_t n, const void* dp_bf16, const void* addend_bf16, const uint8_t* codes, const float* w_hwio,
                                             float* dx, cgs_stream_t stream) {
    if (n < 0 || !dp_bf16 || !codes || !w_hwio || !dx) return CGS_ERR_BADARG;
    return h5_launch<H5Enc0DP>(H5Params{dp_bf16, (const uint16_t*)addend_bf16, w_hwio, nullptr, dx, const_cast<uint8_t*>(codes), nullptr, n, 0, 0},
                               (hipStream_t)stream);
}
// features.0's data gradient on the two mixes AND the mix backward (cgs_mix_bwd) in one pass over n images: dp / codes hold the 2 n mix images
// [replaced | injected] (pooled gradient bf16 [2n,64,64,8], argmax bytes), a_u8 / b_u8 the frames [n,128,128,3], z the mask [n,128,128];
// dzpre [n,128,128] = (sum_c (conv^T(d rep - d inj))_c (B_c - A_c) / 255 + l1s sign(Z) + 2 l2s Z) Z (1 - Z) -- no d mix tensor in memory.
extern "C" int cgs_bf16_enc0_bwd_mix(int32_t n, const void* dp_bf16, const uint8_t* codes, const float* w_hwio, const uint8_t* a_u8, const uint8_t* b_u8,
                                     const float* z, float l1s, float l2s, float* dzpre, cgs_stream_t stream) {
    if (n < 0 || !dp_bf16 || !codes || !w_hwio || !a_u8 || !b_u8 || !z || !dzpre) return CGS_ERR_BADARG;
    H5Params P{dp_bf16, nullptr, w_hwio, nullptr, dzpre, const_cast<uint8_t*>(codes), nullptr, n, 0, 0, a_u8, b_u8, z, l1s, l2s};
    return h5_launch<H5Enc0DM>(P, (hipStream_t)stream);
}
// masker.2 data gradient through masker.0's LeakyReLU: dz fp32 [n,128,128] (pre-Sigmoid gradient), hm bf16 [n,128,128,16] -> d (masker.0
// pre-activation) bf16 [n,128,128,16]
extern "C" int cgs_bf16_mask2_bwd_data(int32_t n, const float* dz, const void* hm_bf16, const float* w_hwio, void* dhm_bf16, cgs_stream_t stream) {
    if (n < 0 || !dz || !hm_bf16 || !w_hwio || !dhm_bf16) return CGS_ERR_BADARG;
    return h5_launch<H5Mask2D>(H5Params{dz, nullptr, w_hwio, nullptr, dhm_bf16, nullptr, (const uint16_t*)hm_bf16, n, 0, 0}, (hipStream_t)stream);
}
// masker.0 data gradient with respect to the upsampled source: dhm bf16 [n,128,128,16] -> d o0 bf16 [n,64,64,8] (summed over each 2x2 cell)
extern "C" int cgs_bf16_mask0_bwd_data(int32_t n, const void* dhm_bf16, const float* w_hwio, void* do0_bf16, cgs_stream_t stream) {
    if (n < 0 || !dhm_bf16 || !w_hwio || !do0_bf16) return CGS_ERR_BADARG;
    return h5_launch<H5Mask0D>(H5Params{dhm_bf16, nullptr, w_hwio, nullptr, do0_bf16, nullptr, nullptr, n, 0, 0}, (hipStream_t)stream);
}

// features.0 on the VIRTUAL mixes (main.py:395,406): the 2 n images [replaced | injected] are formed from the frame pairs (a_u8, b_u8 [n,128,128,3])
// and the mask z [n,128,128] while the tile is staged -- the fp32 mixes are never written (cgs_mix_fwd with mixed = NULL leaves only its partial
// sums of |Z| and Z^2).  e0 bf16 [2n,64,64,8] + argmax bytes, as cgs_bf16_enc0_fwd on the materialised mixes (up to which product of A (1 - Z) + Z B the compiler fuses into the add).
extern "C" int cgs_bf16_enc0_fwd_mix(int32_t n, const uint8_t* a_u8, const uint8_t* b_u8, const float* z, const float* w_hwio, const float* bias,
                                     void* e0_bf16, uint8_t* codes, cgs_stream_t stream) {
    if (n < 0 || !a_u8 || !b_u8 || !z || !w_hwio || !bias || !e0_bf16) return CGS_ERR_BADARG;
    H5Params P{a_u8, nullptr, w_hwio, bias, e0_bf16, codes, nullptr, 2 * n, 0, 2, a_u8, b_u8, z, 0.f, 0.f};
    return h5_launch<H5Enc0F>(P, (hipStream_t)stream);
}

// The 64x64 and 32x32 layers of the same step (which: CGS_H5_*, cgs_hip.h; the 32x32 forms: same roles one level down, see the header): features.3 forward (src_a bf16 [n,64,64,8] -> e1 bf16 [n,32,32,8] + argmax
// bytes) and data gradient (dy bf16 [n,64,64,8] -> d e0), dec_model.0 forward (cat(e0, nearest-up2(o1 bf16 [n,32,32,8])) -> o0) and its two
// data gradients (d o0 bf16 [n,64,64,8] -> the skip gradient [n,64,64,8] / the cell-summed low-resolution gradient [n,32,32,8]).
extern "C" int cgs_bf16_h5conv(int32_t which, int32_t n, const void* src_a, const void* src_b, const float* w_hwio, const float* bias, void* out,
                               uint8_t* codes, cgs_stream_t stream) {
    if (n < 0 || !src_a || !w_hwio || !out) return CGS_ERR_BADARG;
    const H5Params P{src_a, (const uint16_t*)src_b, w_hwio, bias, out, codes, nullptr, n, 0, 0};
    switch (which) {
        case CGS_H5_ENC1_FWD: return bias ? h5_launch<H5Enc1F>(P, (hipStream_t)stream) : CGS_ERR_BADARG;
        case CGS_H5_ENC1_BWD_DATA: return h5_launch<H5Enc1D>(P, (hipStream_t)stream);
        case CGS_H5_ENC1_BWD_DATA_POOLED: return codes ? h5_launch<H5Enc1DP>(P, (hipStream_t)stream) : CGS_ERR_BADARG;
        case CGS_H5_DEC0_FWD: return bias && src_b ? h5_launch<H5Dec0F>(P, (hipStream_t)stream) : CGS_ERR_BADARG;
        case CGS_H5_DEC0_BWD_SKIP: return h5_launch<H5Dec0DS>(P, (hipStream_t)stream);
        case CGS_H5_DEC0_BWD_LOW: return h5_launch<H5Dec0DL>(P, (hipStream_t)stream);
        case CGS_H5_ENC2_FWD: return bias ? h5_launch<H5Enc2F>(P, (hipStream_t)stream) : CGS_ERR_BADARG;
        case CGS_H5_ENC2_BWD_DATA_POOLED: return codes ? h5_launch<H5Enc2DP>(P, (hipStream_t)stream) : CGS_ERR_BADARG;
        case CGS_H5_DEC1_FWD: return bias && src_b ? h5_launch<H5Dec1F>(P, (hipStream_t)stream) : CGS_ERR_BADARG;
        case CGS_H5_DEC1_FWD_F32B: return bias && src_b ? h5_launch<H5Dec1FF>(P, (hipStream_t)stream) : CGS_ERR_BADARG;
        case CGS_H5_DEC1_BWD_SKIP: return h5_launch<H5Dec1DS>(P, (hipStream_t)stream);
        case CGS_H5_DEC1_BWD_LOW: return h5_launch<H5Dec1DL>(P, (hipStream_t)stream);
    }
    return CGS_ERR_UNSUPPORTED;
}

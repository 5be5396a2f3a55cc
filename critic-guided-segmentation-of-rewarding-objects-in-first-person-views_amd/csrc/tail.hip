// "Tail" kernels: every 16x16-and-smaller layer of the Hourglass, image by image inside one workgroup.
//
//   tail_enc_fwd : features.6 (+ReLU+pool) -> Dropout -> features.10 (+ReLU+pool) -> Dropout -> features.14 (4x4 valid conv)
//                  -> ReLU -> crit.1 -> ReLU -> Dropout -> crit.4 -> Sigmoid           nets.py:176-194
//                  (+ the decoder's 1x1 bottleneck conv dec_model.4 of e4, nets.py:501, riding along)
//   tail_dec_fwd : dec_model.3 / .2 / .1 with their Upsample + cat inputs                  nets.py:503-513
//   tail_enc_bwd, tail_dec_bwd : what loss.backward() does for those layers (main.py:462)
//
// Before: one launch per layer over the whole batch (6 + 3 launches forward, 7 + 3 backward per critic pass), each bound by
// its own launch + load + store latency chain (7-17 us for < 1 us of arithmetic at N = 512).  Here a workgroup of 4 waves
// owns an image: the layer inputs / outputs live in LDS tiles, the convolutions run on the matrix cores (tail_common.h), the
// head runs on the vector ALU, only what the backward pass or another kernel needs is written to memory.  The grid is
// persistent (image = blockIdx.x, += number of workgroups; the count is a kernel ARGUMENT: reading gridDim.x loads the dispatch packet
// from host-visible memory, measured ~10 us at the top of these latency-bound kernels) so the weight-gradient accumulators of the backward kernels stay in
// registers across a workgroup's images and every workgroup writes ONE slab per layer.
//
// Dropout: the same Philox indexing as the per-layer kernels (site, element / 4 + base), so cgs_dropout_mask exports the
// masks these kernels draw.
#include "tail4.h"
#include "tail_h16.h"
#include "head_wgrad.h"
#include "wgrad_dec0.h"
#include "conv_body.h"
#include "wgrad_sparse.h"

namespace {

using T16x8 = Tile<16, 16, 8>;      // e1 / d(features.6 pre-pool)
using T8x8 = Tile<8, 8, 8>;         // dropout(e2) / do2
using T8x16 = Tile<8, 8, 16>;       // d(features.10 pre-pool)
using T16x16 = Tile<16, 16, 16>;    // cat(e1, up(o2))
using T8x24 = Tile<8, 8, 24>;       // cat(e2, up(o3))
using T4x48 = Tile<4, 4, 48>;       // cat(e3, up4(o4))
using T4x16 = Tile<4, 4, 20>;       // do3 (16 channels in 20-float pixel slots: the dword reads of dec_model.3's data gradient -- 16 pixels x 2 channels per
                                    // 32 lanes -- fall on 16 banks instead of 4; round 6)

// gradient of conv+ReLU+pool re-expanded to one position of the 2x2 window: nibble == pos ? v : 0 (0xF = ReLU dead)
__device__ __forceinline__ float4 nib_select4(const float4& v, uint32_t nib16, uint32_t pos) {
    float4 r;
    r.x = ((nib16 & 15u) == pos) ? v.x : 0.f;
    r.y = (((nib16 >> 4) & 15u) == pos) ? v.y : 0.f;
    r.z = (((nib16 >> 8) & 15u) == pos) ? v.z : 0.f;
    r.w = (((nib16 >> 12) & 15u) == pos) ? v.w : 0.f;
    return r;
}

__device__ __forceinline__ float drop1(const DropCtx& dc, uint32_t i) {
    return dc.on ? f4get(drop_mult4(dc, i >> 2), i & 3) : 1.f;
}

int tail_blocks(int n, int cap) { return n < cap ? n : cap; }
int tail_fwd_cap() { return 1024; }      // persistent workgroups: swept on the step (round 2)
// the stand-alone decoder tail forward (inference at large batches: 84 registers / 48.6 KB of LDS = three workgroups per CU)
#ifndef CGS_TAIL_DEC_FWD_CAP
#define CGS_TAIL_DEC_FWD_CAP 768
#endif
int tail_dec_fwd_cap() { return CGS_TAIL_DEC_FWD_CAP; }
int tail_bwd_cap() { return 512; }       // decoder tail backward
// encoder tail backward: 768 = three workgroups per CU (its 168 registers / 51 KB of LDS allow exactly that).  Round 5, with features.3's
// data and weight gradients inside the kernel: the mixes' pass (1024 images at N = 512) on 768 workgroups -- 256 of them take two images --
// measures 0.5577 ms per step against 0.5633 with 512 x 2 images, 0.5657 at 640, 0.5687 at 1024 (r05r / r05s, three interleaved runs each)
#ifndef CGS_TAIL_ENC_BWD_CAP
#define CGS_TAIL_ENC_BWD_CAP 768
#endif
int tail_enc_bwd_cap() { return CGS_TAIL_ENC_BWD_CAP; }

unsigned long long* g_tail_stamps = nullptr;     // debug: per-workgroup stage time stamps (tools/tail_stamps.py)

}  // namespace

// Debug hook (not part of the product path): when set, thread 0 of every tail workgroup records s_memtime at its stage
// boundaries of its FIRST image into stamps[(kernel * 2048 + block) * 16 + stage].
#ifdef CGS_DEBUG_STAMPS
extern "C" int dbg_tail_stamps(unsigned long long* stamps) { g_tail_stamps = stamps; return CGS_OK; }
#endif
#define TAIL_STAMP(k)                                                                                     \
    do {                                                                                                  \
        if (CGS_STAMP_PTR(P.dbg) && tid == 0 && img == (int)blockIdx.x) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// encoder tail + critic head, forward
// ------------------------------------------------------------------------------------------------
// s_sleep units (x 64 cycles) of the phase stagger in the fused critic forward, the fused decoder forward and the fused decoder backward (cgs_stagger;
// 0 = none).  Round 4 shipped 127 in all three (-5 us then); after round 5's instruction-level passes the same A/B reads the other way.
#ifndef CGS_TAIL_STAGGER
#define CGS_TAIL_STAGGER 0
#endif
#ifndef CGS_STAGGER_ENC_FWD
#define CGS_STAGGER_ENC_FWD CGS_TAIL_STAGGER
#endif
#ifndef CGS_STAGGER_DEC_FWD
#define CGS_STAGGER_DEC_FWD CGS_TAIL_STAGGER
#endif
#ifndef CGS_STAGGER_DEC_BWD
#define CGS_STAGGER_DEC_BWD CGS_TAIL_STAGGER
#endif
struct TailEncFwdParams {
    cgs_tail_enc_weights w;
    const float* e1;
    float* e2; uint32_t* am2; float* e3; uint32_t* am3; float* e4; float* h1; float* pred; float* o4;
    cgs_dropout drop_e2, drop_e3, drop_h1;
    int n;
    int nblocks;
    unsigned long long* dbg;
};

// (round 3) features.6 and features.10 on v_mfma_f32_4x4x1 with lane = pixel (tail4.h): 216 instead of 90 x 4 MFMA-cycles-worth
// of instructions per wave and image, 36 16-byte LDS reads instead of ~360 dword reads.
using X1P = TileP<16, 16, 8, 8, 148>;     // e1
using X2P = TileP<8, 8, 8, 8, 84>;        // dropout(e2)

// FUSED (round 4): the workgroup first runs features.3 of ITS image (conv3x3_body_pipe<FEnc1P>: both 16-row strips, software-pipelined;
// e1 + argmax nibbles go to memory for the backward pass and the decoder as before) with the pooled epilogue also writing the x1 tile,
// then the tail stages: no e1 round trip, no launch boundary, and the tail's prologue (kernel arguments, Dropout counters, LDS zeroing,
// cold instruction fetch: 38 % of the stand-alone kernel's life by the stamps) runs while the convolution's loads are in flight.
// ENC0 (with FUSED): 1 / 2 = features.0 of the image as well (uint8 frames / the virtual replaced | injected mixes: four 16-row strips,
// conv3x3_body_pipe<FEnc0U8P / FEnc0MixP>; e0 + am0 to memory as before, features.3 reads that e0 back): the whole critic forward of an
// image in one workgroup.
// H16 (round 6, config 4, stand-alone form only): features.6 / features.10 with fp16 operands on v_mfma_f32_16x16x16_f16 (tail_h16.h)
template <bool FUSED, int ENC0 = 0, bool H16 = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(H16 ? 3 : 4, H16 ? 3 : 4))) tail_enc_fwd_kernel(TailEncFwdParams P, ConvParams PC, ConvParams PC0) {
    static_assert(!H16 || (!FUSED && ENC0 == 0), "fp16 operands: the stand-alone inference form");
    extern __shared__ __attribute__((aligned(16))) float4 conv_smem[];       // FUSED: the convolution's tiles + weights
    __shared__ __attribute__((aligned(16))) float x1[X1P::FLOATS];
    __shared__ __attribute__((aligned(16))) float x2[X2P::FLOATS];
    __shared__ __attribute__((aligned(16))) float xs[256];               // dropout(e3), flat NHWC
    __shared__ __attribute__((aligned(16))) float red[8][32];          // (also: the 64 float4 Dropout multipliers of e3 between stages 3 and 4)
    __shared__ float es[32], hs[32];                                    // hs: the 32 Dropout multipliers of h1 (stage 3 -> stage 10)
    if (CGS_KARG_PREFETCH) cgs_kernarg_prefetch<sizeof(TailEncFwdParams) + 2 * sizeof(ConvParams)>();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (CGS_STAMP_PTR(P.dbg) && tid == 0) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16] = __builtin_amdgcn_s_memtime();
    const int o = tid & 31, kg = tid >> 5;
    // the first image's loads and the Dropout step counters are requested before anything else: they fly during the set-up below
    int img = blockIdx.x;
    float4 pe[2] = {f4zero(), f4zero()};
    if constexpr (!FUSED) {
        if (img < P.n) { pe[0] = ((const float4*)P.e1)[(size_t)img * 512 + tid]; pe[1] = ((const float4*)P.e1)[(size_t)img * 512 + tid + 256]; }
    }
#if CGS_DROPCTX3
    DropCtx d2, d3, dh;
    drop_ctx3(P.drop_e2, P.drop_e3, P.drop_h1, P.w.b6, d2, d3, dh);
#else
    const DropCtx d2 = drop_ctx(P.drop_e2, P.w.b6), d3 = drop_ctx(P.drop_e3, P.w.b6), dh = drop_ctx(P.drop_h1, P.w.b6);
#endif

    // ---- once per workgroup: halos -> 0, conv weights -> registers (features.6: both channel groups; features.10: this wave's
    //      four output channels), head weights -> registers ----
    tilep_zero<X1P>(x1, tid);
    tilep_zero<X2P>(x2, tid);
    if constexpr (FUSED) {
        // every other co-resident workgroup starts ~4 us late: its neighbours' latency-bound tail stages then run under its matrix
        // instructions instead of all four workgroups of a CU moving through the phases in lockstep (r4 A/B, five runs each:
        // 0.590-0.593 ms/step against 0.596-0.607 without; the un-staggered step is bimodal)
        cgs_stagger<8, CGS_STAGGER_ENC_FWD>();      // (sweep r4: 64 x 64 cycles is too short -- the step stays bimodal --, 190 / 254 and bit 9 measure the same)
        if constexpr (ENC0 == 1) { conv3x3_body_pipe<FEnc0U8P>(PC0, 4 * (int)blockIdx.x, conv_smem); __syncthreads(); }
        if constexpr (ENC0 == 2) { conv3x3_body_pipe<FEnc0MixP>(PC0, 4 * (int)blockIdx.x, conv_smem); __syncthreads(); }
        // features.3 of image blockIdx.x (strips 2 b, 2 b + 1); its first barrier separates the zeroing above from the epilogue's tile writes.
        // The tail's weight registers are loaded AFTER it: held across the convolution they would push the kernel past 128 registers.
        conv3x3_body_pipe<FEnc1P>(PC, 2 * (int)blockIdx.x, conv_smem, PoolLds{x1, X1P::PITCH, X1P::PS});
    }
    if (CGS_STAMP_PTR(P.dbg) && tid == 0) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16 + 12] = __builtin_amdgcn_s_memtime();
    [[maybe_unused]] float wr6[H16 ? 1 : 2][5], wr10[1][5];
    [[maybe_unused]] th4_t w6h[5], w10h[5];
    [[maybe_unused]] int off6[5], off10[5];
    if constexpr (H16) {
        const int l15w = lane & 15;
        h16_fill_w<8, 5, 1>(w6h, lane, 0, l15w < 8, [&](int tap, int c) { return P.w.w6[(tap * 8 + c) * 8 + (l15w & 7)]; });
        h16_fill_w<8, 5, 1>(w10h, lane, 0, true, [&](int tap, int c) { return P.w.w10[(tap * 8 + c) * 16 + l15w]; });
        h16_fill_off<X1P, 8, 5, 1>(off6, lane, 0);
        h16_fill_off<X2P, 8, 5, 1>(off10, lane, 0);
    } else {
        fill_wreg<2, 5, 72>(wr6, lane, [&](int step, int co) { return P.w.w6[step * 8 + co]; });
        fill_wreg<1, 5, 72>(wr10, lane, [&](int step, int co) { return P.w.w10[step * 16 + 4 * wave + co]; });
    }
    if (CGS_STAMP_PTR(P.dbg) && tid == 0) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16 + 13] = __builtin_amdgcn_s_memtime();
    float w1r[4], wpr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w1r[j] = P.w.wl1[(kg * 4 + j) * 32 + o];
        wpr[j] = P.o4 ? P.w.wpw[(kg * 4 + j) * 32 + o] : 0.f;
    }
    const cgs_cptr b6c = cgs_to_const(P.w.b6), b10c = cgs_to_const(P.w.b10);
    const float b14 = P.w.b14[o], bl1 = P.w.bl1[o], wl2 = P.w.wl2[o], bl2 = P.w.bl2[0], bpw = P.o4 ? P.w.bpw[o] : 0.f;
    const PxPos pa = px16(wave, lane), pb = px8(lane);
    if (CGS_STAMP_PTR(P.dbg) && tid == 0) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memtime();
    __syncthreads();

    for (; img < P.n; img += P.nblocks) {
        TAIL_STAMP(1);
        // ---- e1 -> tile interior (512 float4), then the next image's loads (FUSED: the convolution's epilogue filled the tile) ----
        if constexpr (!FUSED) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int e = tid + 256 * it, p = e & 1, x = (e >> 1) & 15, y = e >> 5;
                *(float4*)(x1 + X1P::at(y, x) + 4 * p) = pe[it];
            }
            __syncthreads();
            if (img + P.nblocks < P.n) {
                pe[0] = ((const float4*)P.e1)[(size_t)(img + P.nblocks) * 512 + tid];
                pe[1] = ((const float4*)P.e1)[(size_t)(img + P.nblocks) * 512 + tid + 256];
            }
        }
        TAIL_STAMP(2);
        // ---- features.6 + ReLU + pool: this wave's 64 pixels, all 8 channels ----
        if constexpr (H16) {
            // four 16-pixel tiles per wave; D[pixel 4 kq + r][channel l15]: a lane's four values are one 2x2 quad = one pooled pixel
            const int l15 = lane & 15, kq = lane >> 4;
            const float b6v = b6c[l15 & 7];
#pragma unroll 1
            for (int u = 0; u < 4; ++u) {
                const int t = wave + 4 * u;
                int y, x;
                h16_tile_px<16>(t, l15, y, x);
                const frag4 acc = h16_conv<5>(x1 + X1P::at(y - 1, x - 1), off6, w6h, frag4{0.f, 0.f, 0.f, 0.f});
                uint32_t idx;
                const float m = pool_quad(acc, b6v, idx);
                const int q = 4 * t + kq;
                const uint32_t word = pack_nibbles(idx, l15);       // (lanes 8 .. 15 of a row hold padding columns: their group's word is never stored)
                if (l15 < 8) {
                    P.e2[((size_t)img * 64 + q) * 8 + l15] = m;
                    x2[X2P::at(q >> 3, q & 7) + l15] = m;
                    if (l15 == 0) P.am2[(size_t)img * 64 + q] = word;
                }
            }
        } else {
            frag4 a6[2] = {frag4{0.f, 0.f, 0.f, 0.f}, frag4{0.f, 0.f, 0.f, 0.f}};
            conv_px<X1P, 0, 8, 8, 0, 2, 5>(a6, x1, pa.y, pa.x, wr6);
            float m[8];
            uint32_t word = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                uint32_t nib;
                m[c] = quad_pool(fmaxf(a6[c >> 2][c & 3] + b6c[c], 0.f), pa.pos, nib);
                word |= nib << (4 * c);
            }
            // the quad's four lanes hold the same pooled values: lanes 0 / 1 store the two halves to e2, lanes 2 / 3 to the tile
            const float4 sel = (pa.pos & 1) ? make_float4(m[4], m[5], m[6], m[7]) : make_float4(m[0], m[1], m[2], m[3]);
            if (pa.pos < 2) ((float4*)P.e2)[((size_t)img * 64 + pa.q) * 2 + pa.pos] = sel;
            else *(float4*)(x2 + X2P::at(pa.qy, pa.qx) + 4 * (pa.pos & 1)) = sel;
            if (pa.pos == 0) P.am2[(size_t)img * 64 + pa.q] = word;
        }
        // features.14's weights of this thread (L2-resident): requested here, used two stages later (held across the image loop
        // they cost the registers of the convolution stages)
        float w4r[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) w4r[j] = P.w.w14[(kg * 32 + j) * 32 + o];
        __syncthreads();
        TAIL_STAMP(3);
        // The image's three sets of Dropout multipliers, ONE Philox evaluation per wave and all at the same time (round 5; they were three in a row on
        // wave 0 -- e2's here, e3's in the next stage's epilogue on 16 lanes of every wave, h1's at the head): waves 0 / 1 = e2's 128 float4
        // (applied in place), wave 2 = e3's 64 float4 -> red, wave 3 = h1's 32 -> hs.  Same counters, same values.
        if (tid < 128) {                // Dropout on features.10's input (the stored e2 stays undropped: it is the skip)
            if (d2.on) {
                const int q = tid >> 1, p = tid & 1;
                float4* v = (float4*)(x2 + X2P::at(q >> 3, q & 7) + 4 * p);
                *v = *v * drop_mult4(d2, (uint32_t)(img * 128 + tid));
            }
        } else if (tid < 192) {
            if (d3.on) ((float4*)red)[tid - 128] = drop_mult4(d3, (uint32_t)(img * 64 + tid - 128));     // float4 index (img * 16 + q) * 4 + wave'
        } else if (tid < 224) {
            if (dh.on) hs[tid - 192] = drop1(dh, (uint32_t)(img * 32 + tid - 192));
        }
        __syncthreads();
        TAIL_STAMP(4);
        // ---- features.10 + ReLU + pool: all 64 pixels of the 8x8 map, this wave's 4 output channels ----
        if constexpr (H16) {
            // one 16-pixel tile per wave, all 16 output channels (column l15)
            const int l15 = lane & 15, kq = lane >> 4;
            int y, x;
            h16_tile_px<8>(wave, l15, y, x);
            const frag4 acc = h16_conv<5>(x2 + X2P::at(y - 1, x - 1), off10, w10h, frag4{0.f, 0.f, 0.f, 0.f});
            uint32_t idx;
            const float m = pool_quad(acc, b10c[l15], idx);
            const int q = 4 * wave + kq;                             // pooled pixel of the 4x4 map
            const uint32_t word = pack_nibbles(idx, l15);           // channels 8 g .. 8 g + 7: word g of the pixel
            P.e3[((size_t)img * 16 + q) * 16 + l15] = m;
            xs[q * 16 + l15] = m;                                   // (eval mode: no Dropout in front of features.14)
            if ((l15 & 7) == 0) P.am3[((size_t)img * 16 + q) * 2 + (l15 >> 3)] = word;
        } else {
            frag4 a10[1] = {frag4{0.f, 0.f, 0.f, 0.f}};
            conv_px<X2P, 0, 8, 8, 0, 1, 5>(a10, x2, pb.y, pb.x, wr10);
            float m[4];
            uint32_t half = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                uint32_t nib;
                m[r] = quad_pool(fmaxf(a10[0][r] + b10c[4 * wave + r], 0.f), pb.pos, nib);
                half |= nib << (4 * r);
            }
            if (pb.pos == 0) {
                const float4 mv = make_float4(m[0], m[1], m[2], m[3]);
                const uint32_t i4 = (uint32_t)((img * 16 + pb.q) * 4 + wave);      // float4 index of channels 4w .. 4w+3 of pooled pixel q
                ((float4*)P.e3)[i4] = mv;
                ((float4*)xs)[pb.q * 4 + wave] = d3.on ? mv * ((const float4*)red)[pb.q * 4 + wave] : mv;
                ((uint16_t*)P.am3)[(size_t)i4] = (uint16_t)half;                    // nibbles of channels 4w .. 4w+3: 16-bit quarter of the pixel's two words
            }
        }
        __syncthreads();
        TAIL_STAMP(5);
        // ---- features.14 (256 -> 32) + ReLU: thread (o, kg) sums k = 32*kg .. +31 ----
        {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int j = 0; j < 32; j += 4) {
                const float4 xv = *(const float4*)(xs + kg * 32 + j);
                a0 = fmaf(xv.x, w4r[j], a0); a1 = fmaf(xv.y, w4r[j + 1], a1);
                a2 = fmaf(xv.z, w4r[j + 2], a2); a3 = fmaf(xv.w, w4r[j + 3], a3);
            }
            red[kg][o] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        TAIL_STAMP(6);
        if (tid < 32) {
            float s = b14;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += red[g][o];
            s = fmaxf(s, 0.f);
            es[o] = s;
            P.e4[(size_t)img * 32 + o] = s;
        }
        __syncthreads();
        TAIL_STAMP(7);
        // ---- crit.1 (32 -> 32) and the decoder's 1x1 conv of e4: thread (o, kg) sums k = 4*kg .. +3 ----
        {
            float s1 = 0.f, sp = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ev = es[kg * 4 + j];
                s1 = fmaf(ev, w1r[j], s1);
                sp = fmaf(ev, wpr[j], sp);
            }
            red[kg][o] = s1;
            __syncthreads();
            TAIL_STAMP(8);
            float h = 0.f;
            if (tid < 32) {
                float s = bl1;
#pragma unroll
                for (int g = 0; g < 8; ++g) s += red[g][o];
                h = fmaxf(s, 0.f);
                P.h1[(size_t)img * 32 + o] = h;
            }
            __syncthreads();
            TAIL_STAMP(9);
            red[kg][o] = sp;
            __syncthreads();
            TAIL_STAMP(10);
            if (tid < 32) {
                if (P.o4) {
                    float s = bpw;
#pragma unroll
                    for (int g = 0; g < 8; ++g) s += red[g][o];
                    P.o4[(size_t)img * 32 + o] = s;
                }
                // ---- Dropout -> crit.4 (32 -> 1) -> Sigmoid ----
                float t = h * (dh.on ? hs[o] : 1.f) * wl2;
#pragma unroll
                for (int m = 16; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
                if (o == 0) P.pred[img] = 1.f / (1.f + expf(-(t + bl2)));
            }
        }
        __syncthreads();
        TAIL_STAMP(11);
    }
    if (CGS_STAMP_PTR(P.dbg) && tid == 0) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16 + 15] = __builtin_amdgcn_s_memtime();
}

extern "C" int cgs_tail_enc_fwd(int32_t n, const cgs_tail_enc_weights* w, const float* e1, float* e2, uint32_t* am2, float* e3,
                                uint32_t* am3, float* e4, float* h1, float* pred, float* o4, cgs_dropout drop_e2,
                                cgs_dropout drop_e3, cgs_dropout drop_h1, cgs_stream_t stream) {
    if (n < 0 || !w || !e1 || !e2 || !am2 || !e3 || !am3 || !e4 || !h1 || !pred) return CGS_ERR_BADARG;
    if (!w->w6 || !w->b6 || !w->w10 || !w->b10 || !w->w14 || !w->b14 || !w->wl1 || !w->bl1 || !w->wl2 || !w->bl2) return CGS_ERR_BADARG;
    if (o4 && (!w->wpw || !w->bpw)) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    TailEncFwdParams P{*w, e1, e2, am2, e3, am3, e4, h1, pred, o4, drop_e2, drop_e3, drop_h1, n, tail_blocks(n, tail_fwd_cap()), g_tail_stamps ? g_tail_stamps + 0 * 2048 * 16 : nullptr};
    const int blocks = P.nblocks;
    hipLaunchKernelGGL((tail_enc_fwd_kernel<false, 0>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, ConvParams{}, ConvParams{});
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// cgs_tail_enc_fwd in eval mode with fp16 OPERANDS in features.6 / features.10 (fp32 accumulation, fp32 tensors in and out; BASELINE config 4:
// the -process path with fp16 conv kernels, nets.py:176-194 in eval mode).  Dropout is not supported here: CGS_ERR_UNSUPPORTED when any p > 0.
extern "C" int cgs_tail_enc_fwd_h16(int32_t n, const cgs_tail_enc_weights* w, const float* e1, float* e2, uint32_t* am2, float* e3,
                                    uint32_t* am3, float* e4, float* h1, float* pred, float* o4, cgs_stream_t stream) {
    if (n < 0 || !w || !e1 || !e2 || !am2 || !e3 || !am3 || !e4 || !h1 || !pred) return CGS_ERR_BADARG;
    if (!w->w6 || !w->b6 || !w->w10 || !w->b10 || !w->w14 || !w->b14 || !w->wl1 || !w->bl1 || !w->wl2 || !w->bl2) return CGS_ERR_BADARG;
    if (o4 && (!w->wpw || !w->bpw)) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    const cgs_dropout nd{};
    TailEncFwdParams P{*w, e1, e2, am2, e3, am3, e4, h1, pred, o4, nd, nd, nd, n, tail_blocks(n, 768), nullptr};      // (three workgroups per CU)
    hipLaunchKernelGGL((tail_enc_fwd_kernel<false, 0, true>), dim3(P.nblocks), dim3(256), 0, (hipStream_t)stream, P, ConvParams{}, ConvParams{});
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// features.3 (conv 8 -> 8 at 32x32 + ReLU + MaxPool2d(2), nets.py:173-175) AND the encoder tail in one launch, one workgroup per image:
// e0 [n,32,32,8] -> e1 [n,16,16,8] + am1 (to memory, as cgs_conv3x3_fwd writes them) -> everything cgs_tail_enc_fwd produces.
extern "C" int cgs_enc1_tail_fwd(int32_t n, const cgs_tail_enc_weights* w, const float* e0, const float* w3, const float* b3, float* e1,
                                 uint32_t* am1, float* e2, uint32_t* am2, float* e3, uint32_t* am3, float* e4, float* h1, float* pred, float* o4,
                                 cgs_dropout drop_e2, cgs_dropout drop_e3, cgs_dropout drop_h1, cgs_stream_t stream) {
    if (n < 0 || !w || !e0 || !w3 || !b3 || !e1 || !am1 || !e2 || !am2 || !e3 || !am3 || !e4 || !h1 || !pred) return CGS_ERR_BADARG;
    if (!w->w6 || !w->b6 || !w->w10 || !w->b10 || !w->w14 || !w->b14 || !w->wl1 || !w->bl1 || !w->wl2 || !w->bl2) return CGS_ERR_BADARG;
    if (o4 && (!w->wpw || !w->bpw)) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    TailEncFwdParams P{*w, e1, e2, am2, e3, am3, e4, h1, pred, o4, drop_e2, drop_e3, drop_h1, n, n, g_tail_stamps ? g_tail_stamps + 0 * 2048 * 16 : nullptr};
    ConvParams PC{};
    PC.src_a = e0; PC.w = w3; PC.bias = b3; PC.out = e1; PC.amask_out = am1; PC.n = n;
    const size_t lds = conv_lds_bytes<FEnc1P>();
    hipLaunchKernelGGL((tail_enc_fwd_kernel<true, 0>), dim3(n), dim3(256), lds, (hipStream_t)stream, P, PC, ConvParams{});
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// The WHOLE critic forward of an image in one workgroup: features.0 (cgs_conv3x3_fwd of the 3 -> 8 layer at 64x64 on uint8 frames, or -- x_is_mix --
// on the virtual replaced | injected mixes of a cgs_mix_src) -> e0 + am0, then everything cgs_enc1_tail_fwd does.  w0 / b0: features.0's
// HWIO weights and bias.
extern "C" int cgs_critic_fwd_fused(int32_t n, const cgs_tail_enc_weights* w, const void* x, int32_t x_is_mix, const float* w0, const float* b0,
                                    float* e0, uint32_t* am0, const float* w3, const float* b3, float* e1, uint32_t* am1, float* e2, uint32_t* am2,
                                    float* e3, uint32_t* am3, float* e4, float* h1, float* pred, float* o4, cgs_dropout drop_e2,
                                    cgs_dropout drop_e3, cgs_dropout drop_h1, cgs_stream_t stream) {
    if (n < 0 || !w || !x || !w0 || !b0 || !e0 || !am0 || !w3 || !b3 || !e1 || !am1 || !e2 || !am2 || !e3 || !am3 || !e4 || !h1 || !pred) return CGS_ERR_BADARG;
    if (!w->w6 || !w->b6 || !w->w10 || !w->b10 || !w->w14 || !w->b14 || !w->wl1 || !w->bl1 || !w->wl2 || !w->bl2) return CGS_ERR_BADARG;
    if (o4 && (!w->wpw || !w->bpw)) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    TailEncFwdParams P{*w, e1, e2, am2, e3, am3, e4, h1, pred, o4, drop_e2, drop_e3, drop_h1, n, n, g_tail_stamps ? g_tail_stamps + 0 * 2048 * 16 : nullptr};
    ConvParams PC{}, PC0{};
    PC.src_a = e0; PC.w = w3; PC.bias = b3; PC.out = e1; PC.amask_out = am1; PC.n = n;
    PC0.w = w0; PC0.bias = b0; PC0.out = e0; PC0.amask_out = am0; PC0.n = n;
    size_t lds = conv_lds_bytes<FEnc1P>();
    if (x_is_mix) {
        const cgs_mix_src* m = (const cgs_mix_src*)x;        // HOST struct
        if (!m->a || !m->b || !m->z || m->n_a <= 0 || n > 2 * m->n_a) return CGS_ERR_BADARG;
        PC0.mix_a = m->a; PC0.mix_b = m->b; PC0.mix_z = m->z; PC0.mix_n_a = m->n_a;
        if (conv_lds_bytes<FEnc0MixP>() > lds) lds = conv_lds_bytes<FEnc0MixP>();
        hipLaunchKernelGGL((tail_enc_fwd_kernel<true, 2>), dim3(n), dim3(256), lds, (hipStream_t)stream, P, PC, PC0);
    } else {
        PC0.src_a = x;
        if (conv_lds_bytes<FEnc0U8P>() > lds) lds = conv_lds_bytes<FEnc0U8P>();
        hipLaunchKernelGGL((tail_enc_fwd_kernel<true, 1>), dim3(n), dim3(256), lds, (hipStream_t)stream, P, PC, PC0);
    }
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// ------------------------------------------------------------------------------------------------
// decoder tail, forward: dec_model.3 (cat(e3, up4(o4)) @4x4) -> dec_model.2 (cat(e2, up(o3)) @8x8) -> dec_model.1
// (cat(e1, up(o2)) @16x16); the decoder trunk is linear (no activation, nets.py:503-513).
// ------------------------------------------------------------------------------------------------
struct TailDecFwdParams {
    cgs_tail_dec_weights w;
    const float* e1; const float* e2; const float* e3; const float* o4;
    float* o3; float* o2; float* o1;
    int n;
    int nblocks;
    unsigned long long* dbg;
    const float* m0_w; float* m0_pack;      // rider (optional): masker.0's weight registers of the mask-head forward kernel, packed once
};

// The mask head forward (mask_fwd.hip) multiplies with per-lane weight REGISTERS: wimg[4][2] (image channels, block b of register k
// = step 16 k + b = tap * 3 + channel) and wups[4][4][2] (upsampled channels with the nearest-upsample folded: per output position
// the taps that fall on the same source pixel pre-summed).  Each of its 512 workgroups used to rebuild them from masker.0's HWIO
// weights (LDS copy + ~140 gathers + sums: 11 k of a workgroup's 122 k cycles); one spare workgroup of the launch BEFORE it builds
// them once: pack[k * 64 + lane], k = 0..7 wimg[g][r] (g * 2 + r), 8..39 wups[pos][g][ry] ((pos * 4 + g) * 2 + ry).
__device__ __forceinline__ void mask0_pack_weights(const float* __restrict__ w0, float* __restrict__ pack, int lane) {
    const int lb = lane >> 2, li = lane & 3;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int step = 16 * k + lb;
            pack[(g * 2 + k) * 64 + lane] = step < 27 ? w0[((step / 3) * 11 + step % 3) * 16 + 4 * g + li] : 0.f;
        }
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            const int oy = pos >> 1, ox = pos & 1;
#pragma unroll
            for (int ry = 0; ry < 2; ++ry) {
                const int rx = lb >> 3, ci = lb & 7;
                const int ky0 = (oy == 0) ? (ry == 0 ? 0 : 1) : (ry == 0 ? 0 : 2), ky1 = (oy == 0) ? (ry == 0 ? 0 : 2) : (ry == 0 ? 1 : 2);
                const int kx0 = (ox == 0) ? (rx == 0 ? 0 : 1) : (rx == 0 ? 0 : 2), kx1 = (ox == 0) ? (rx == 0 ? 0 : 2) : (rx == 0 ? 1 : 2);
                auto wv = [&](int ky, int kx) { return w0[((ky * 3 + kx) * 11 + 3 + ci) * 16 + 4 * g + li]; };
                const float w00 = wv(ky0, kx0), w01 = wv(ky0, kx1), w10 = wv(ky1, kx0), w11 = wv(ky1, kx1);
                const bool my = ky1 > ky0, mx = kx1 > kx0;
                float sum = w00;                     // (the order of mask_fwd_kernel's own set-up: bit-identical registers)
                sum += mx ? w01 : 0.f;
                sum += my ? w10 : 0.f;
                sum += (mx && my) ? w11 : 0.f;
                pack[(8 + (pos * 4 + g) * 2 + ry) * 64 + lane] = sum;
            }
        }
    }
}

using T1F = TileP<16, 16, 16, 16, 292>;     // cat(e1, up(o2)) of the forward decoder tail (2-way conflicts on the b128 reads accepted)

// FUSED (round 4): one workgroup per image; after the tail stages (o1 written) the same workgroup runs dec_model.0 of ITS image
// (conv3x3_body_pipe<FDec0P>: cat(e0, up(o1)) -> o0), every other co-resident workgroup ~4 us late (cgs_stagger).
// H16 (round 6, config 4, stand-alone form only): the three layers with fp16 operands on v_mfma_f32_16x16x16_f16 (tail_h16.h)
template <bool FUSED, bool H16 = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) tail_dec_fwd_kernel(TailDecFwdParams P, ConvParams PC) {
    static_assert(!H16 || !FUSED, "fp16 operands: the stand-alone inference form");
    if (CGS_KARG_PREFETCH) cgs_kernarg_prefetch<sizeof(TailDecFwdParams) + sizeof(ConvParams)>();
    extern __shared__ __attribute__((aligned(16))) float4 dec_conv_smem[];     // FUSED: dec_model.0's tiles + weights
    __shared__ __attribute__((aligned(16))) float t1[T1F::FLOATS];
    __shared__ __attribute__((aligned(16))) float t2[T8x24::FLOATS];
    __shared__ __attribute__((aligned(16))) float t3[T4x48::FLOATS];
    __shared__ __attribute__((aligned(16))) float w2s[14 * 64 * 2];       // dec_model.2's weights [216][8] (H16: the fp16 operand table [14][64] x 8 bytes)
    static_assert(14 * 64 * 2 >= 216 * 8, "either form fits");
    __shared__ float part[4][16][16];       // dec_model.3: the waves split K, partial [pixel][co]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x == P.nblocks) {          // the rider workgroup (launched only when m0_pack is given)
        if (wave == 0) mask0_pack_weights(P.m0_w, P.m0_pack, lane);
        return;
    }
    if constexpr (FUSED) cgs_stagger<8, CGS_STAGGER_DEC_FWD>();
    if (CGS_STAMP_PTR(P.dbg) && tid == 0) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16] = __builtin_amdgcn_s_memtime();
    const int l15 = lane & 15, kq = lane >> 4;
    // the first image's loads are requested before the set-up (and the next image's as soon as the tiles are filled)
    int img = blockIdx.x;
    float4 pe1a = f4zero(), pe1b = f4zero(), pe2 = f4zero(), pe3 = f4zero(), po4 = f4zero();
#define DEC_FWD_FETCH(im)                                                                                         \
    do {                                                                                                          \
        pe1a = ((const float4*)P.e1)[(size_t)(im) * 512 + tid]; pe1b = ((const float4*)P.e1)[(size_t)(im) * 512 + tid + 256]; \
        pe2 = ((const float4*)P.e2)[(size_t)(im) * 128 + (tid & 127)];                                            \
        pe3 = ((const float4*)P.e3)[(size_t)(im) * 64 + (tid & 63)];                                              \
        po4 = ((const float4*)P.o4)[(size_t)(im) * 8 + (tid & 7)];                                                \
    } while (0)
    if (img < P.n) DEC_FWD_FETCH(img);

    // (every load of the set-up requested before the first LDS store: as a copy loop dec_model.2's weights were two load -> wait -> store rounds
    //  in front of the register images' loads -- three dependent round trips per workgroup, the whole launch in lockstep)
    const float4 w2a = ((const float4*)P.w.w2)[tid], w2b = ((const float4*)P.w.w2)[tid + 256 < 216 * 8 / 4 ? tid + 256 : 0];      // (scalars: a small array here lands in scratch)
    // dec_model.1 on v_mfma_f32_4x4x1 with lane = pixel (tail4.h): both channel groups' weights in 18 registers
    [[maybe_unused]] float wr1[2][H16 ? 1 : 9];
    // H16: dec_model.3's 27 instructions dealt to the waves (i = wave, wave + 4, ...), dec_model.2's 14 and dec_model.1's 9 per tile in every wave
    // (dec_model.3's and dec_model.1's operands in registers -- 16- and 48-channel taps: their A offsets fold to constants + 4 kq --, dec_model.2's
    //  in an LDS table [14][64 lanes] in place of the fp32 weights: one tile per wave and image reads it once)
    [[maybe_unused]] th4_t w3h[7], w1h[9];
    [[maybe_unused]] int off2[14];
    [[maybe_unused]] th4_t w2t[H16 ? 4 : 1];          // this thread's entries of the table (14 x 64 = 3.5 x 256)
    if constexpr (H16) {
        h16_fill_w<48, 7, 4>(w3h, lane, wave, true, [&](int tap, int c) { return P.w.w3[(tap * 48 + c) * 16 + l15]; });
        h16_fill_w<16, 9, 1>(w1h, lane, 0, l15 < 8, [&](int tap, int c) { return P.w.w1[(tap * 16 + c) * 8 + (l15 & 7)]; });
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = tid + 256 * r, i = e >> 6, ln = e & 63;       // (i = 14, 15: zero operands, never stored)
            w2t[r] = h16_w1<24>(i, ln >> 4, (ln & 15) < 8, [&](int tap, int c) { return P.w.w2[(tap * 24 + c) * 8 + (ln & 7)]; });
        }
        h16_fill_off<T8x24, 24, 14, 1>(off2, lane, 0);
    } else {
        fill_wreg<2, 9, 144>(wr1, lane, [&](int step, int co) { return P.w.w1[step * 8 + co]; });
    }
    const float b2 = P.w.b2[l15 & 7];
    [[maybe_unused]] const float b1v = P.w.b1[l15 & 7];
    __builtin_amdgcn_sched_barrier(0);       // (the loads above stay above: the scheduler sinks them to their first use otherwise)
    tilep_zero<T1F>(t1, tid);
    tile_zero<T8x24>(t2, tid);
    tile_zero<T4x48>(t3, tid);
    if constexpr (H16) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (tid + 256 * r < 14 * 64) ((th4_t*)w2s)[tid + 256 * r] = w2t[r];
    } else {
        ((float4*)w2s)[tid] = w2a;
        if (tid + 256 < 216 * 8 / 4) ((float4*)w2s)[tid + 256] = w2b;
    }
    const PxPos pa = px16(wave, lane);
    const cgs_cptr b1c = cgs_to_const(P.w.b1);
    __syncthreads();

    for (; img < P.n; img += P.nblocks) {
        TAIL_STAMP(1);
        int lz = 0;                         // opaque zero: keeps the lane-only LDS addresses from being hoisted out of the image loop
        asm volatile("" : "+v"(lz));
        const int lane_i = lane + lz;
        // ---- skip inputs and the bottleneck -> tiles ----
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int e = tid + 256 * it, p = e & 1, x = (e >> 1) & 15, y = e >> 5;
            *(float4*)(t1 + T1F::at(y, x) + 4 * p) = it ? pe1b : pe1a;
        }
        if (tid < 128) {
            const int p = tid & 1, x = (tid >> 1) & 7, y = tid >> 4;
            *(float4*)(t2 + T8x24::at(y, x) + 4 * p) = pe2;
        } else if (tid < 192) {
            const int e = tid - 128, p = e & 3, x = (e >> 2) & 3, y = e >> 4;
            *(float4*)(t3 + T4x48::at(y, x) + 4 * p) = pe3;
        }
        if (tid < 128) {         // up4(o4): every pixel of the 4x4 map sees the bottleneck vector
            const int p = tid & 7, pix = tid >> 3;
            *(float4*)(t3 + T4x48::at(pix >> 2, pix & 3) + 16 + 4 * p) = po4;
        }
        if (img + P.nblocks < P.n) DEC_FWD_FETCH(img + P.nblocks);
        __syncthreads();
        TAIL_STAMP(2);
        // ---- dec_model.3: one 16-pixel tile, K = 9 x 48; wave w takes channels 12w .. 12w+11 of every tap ----
        if constexpr (H16) {
            int y, x;
            h16_tile_px<4>(0, l15, y, x);
            const frag4 acc = h16_conv_f<7>(t3 + T4x48::at(y - 1, x - 1) + 4 * kq, [&](int n) { return h16_off<T4x48, 48>(wave + 4 * n, 0); },
                                             [&](int n) { return w3h[n]; }, frag4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int j = 0; j < 4; ++j) part[wave][4 * kq + j][l15] = acc[j];
        } else {
            const int q = l15 >> 2, y = 2 * (q >> 1) + ((l15 >> 1) & 1), x = 2 * (q & 1) + (l15 & 1);
            const int abase = (y * T4x48::PW + x) * 48 + 12 * wave + kq;
            const float* wg = P.w.w3 + (12 * wave + kq) * 16 + l15;
            frag4 acc = frag4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = ((tap / 3) * T4x48::PW + tap % 3) * 48;
#pragma unroll
                for (int c0 = 0; c0 < 12; c0 += 4)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t3[abase + toff + c0], wg[(tap * 48 + c0) * 16], acc, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) part[wave][4 * kq + j][l15] = acc[j];
        }
        __syncthreads();
        TAIL_STAMP(3);
        {
            const int i = tid >> 4, co = tid & 15;                       // tile pixel i = 4*quad + 2*dy + dx
            const int q = i >> 2, y = 2 * (q >> 1) + ((i >> 1) & 1), x = 2 * (q & 1) + (i & 1);
            const float v = ((part[0][i][co] + part[1][i][co]) + (part[2][i][co] + part[3][i][co])) + P.w.b3[co];
            P.o3[((size_t)img * 16 + y * 4 + x) * 16 + co] = v;
#pragma unroll
            for (int d = 0; d < 4; ++d) t2[T8x24::at(2 * y + (d >> 1), 2 * x + (d & 1)) + 8 + co] = v;
        }
        __syncthreads();
        TAIL_STAMP(4);
        // ---- dec_model.2: 4 tiles ----
        auto epi2 = [&](int q, const frag4 (&acc)[1]) {
            if (l15 < 8) {
                const int qy = q >> 2, qx = q & 3;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = 2 * qy + (j >> 1), x = 2 * qx + (j & 1);
                    const float v = acc[0][j] + b2;
                    P.o2[((size_t)img * 64 + y * 8 + x) * 8 + l15] = v;
#pragma unroll
                    for (int d = 0; d < 4; ++d) t1[T1F::at(2 * y + (d >> 1), 2 * x + (d & 1)) + 8 + l15] = v;
                }
            }
        };
        if constexpr (H16) {
            int y, x;
            h16_tile_px<8>(wave, l15, y, x);
            const frag4 acc[1] = {h16_conv_f<14>(t2 + T8x24::at(y - 1, x - 1), [&](int n) { return off2[n]; },
                                               [&](int n) { return ((const th4_t*)w2s)[n * 64 + lane_i]; }, frag4{0.f, 0.f, 0.f, 0.f})};
            epi2(4 * wave + kq, acc);
        } else {
            conv_tiles<T8x24, 0, 24, 1>(t2, [&](int tap, int c, int) { return w2s[(tap * 24 + c) * 8 + (l15 & 7)]; }, epi2, wave, lane_i);
        }
        __syncthreads();
        TAIL_STAMP(5);
        // ---- dec_model.1: this wave's 64 pixels, all 8 channels ----
        if constexpr (H16) {
            // four 16-pixel tiles per wave; D[pixel 4 kq + j][channel l15]: lanes l15 < 8 store their channel of the quad's four pixels
#pragma unroll 1
            for (int u = 0; u < 4; ++u) {
                const int t = wave + 4 * u;
                int y, x;
                h16_tile_px<16>(t, l15, y, x);
                const frag4 acc = h16_conv_f<9>(t1 + T1F::at(y - 1, x - 1) + 4 * kq, [&](int n) { return h16_off<T1F, 16>(n, 0); },
                                                 [&](int n) { return w1h[n]; }, frag4{b1v, b1v, b1v, b1v});
                if (l15 < 8) {
                    const int q = 4 * t + kq, qy = q >> 3, qx = q & 7;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        P.o1[((size_t)img * 256 + (2 * qy + (j >> 1)) * 16 + 2 * qx + (j & 1)) * 8 + l15] = acc[j];
                }
            }
        } else {
            frag4 a[2] = {frag4{b1c[0], b1c[1], b1c[2], b1c[3]}, frag4{b1c[4], b1c[5], b1c[6], b1c[7]}};
            conv_px<T1F, 0, 16, 16, 0, 2, 9, false, true>(a, t1, pa.y, pa.x, wr1);
            float4* dst = (float4*)(P.o1 + ((size_t)img * 256 + pa.y * 16 + pa.x) * 8);
            dst[0] = make_float4(a[0][0], a[0][1], a[0][2], a[0][3]);
            dst[1] = make_float4(a[1][0], a[1][1], a[1][2], a[1][3]);
        }
        __syncthreads();
        TAIL_STAMP(6);
    }
    if constexpr (FUSED) {          // o1 of image blockIdx.x is in memory (the loop's last barrier): dec_model.0 of that image, both strips
        conv3x3_body_pipe<FDec0P>(PC, 2 * (int)blockIdx.x, dec_conv_smem);
    }
    if (CGS_STAMP_PTR(P.dbg) && tid == 0) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16 + 15] = __builtin_amdgcn_s_memtime();
}

extern "C" int cgs_tail_dec_fwd_pack(int32_t n, const cgs_tail_dec_weights* w, const float* e1, const float* e2, const float* e3,
                                     const float* o4, float* o3, float* o2, float* o1, const float* w_m0, float* m0_pack,
                                     cgs_stream_t stream) {
    if (n < 0 || !w || !e1 || !e2 || !e3 || !o4 || !o3 || !o2 || !o1 || (m0_pack && !w_m0)) return CGS_ERR_BADARG;
    if (!w->w3 || !w->b3 || !w->w2 || !w->b2 || !w->w1 || !w->b1) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    TailDecFwdParams P{*w, e1, e2, e3, o4, o3, o2, o1, n, tail_blocks(n, tail_dec_fwd_cap()), g_tail_stamps ? g_tail_stamps + 1 * 2048 * 16 : nullptr,
                       w_m0, m0_pack};
    const int blocks = tail_blocks(n, tail_dec_fwd_cap());
    hipLaunchKernelGGL(tail_dec_fwd_kernel<false>, dim3(blocks + (m0_pack ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, P, ConvParams{});
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// cgs_tail_dec_fwd_pack AND dec_model.0 (cgs_conv3x3_fwd of the 16 -> 8 layer at 32x32: cat(e0 [n,32,32,8], nearest-up(o1)) -> o0
// [n,32,32,8], linear; nets.py:516-517) in one launch, one workgroup per image (n <= 1024, else CGS_ERR_UNSUPPORTED).
extern "C" int cgs_tail_dec_fwd_dec0(int32_t n, const cgs_tail_dec_weights* w, const float* e0, const float* e1, const float* e2, const float* e3,
                                     const float* o4, float* o3, float* o2, float* o1, const float* w0, const float* b0, float* o0,
                                     const float* w_m0, float* m0_pack, cgs_stream_t stream) {
    if (n < 0 || !w || !e0 || !e1 || !e2 || !e3 || !o4 || !o3 || !o2 || !o1 || !w0 || !b0 || !o0 || (m0_pack && !w_m0)) return CGS_ERR_BADARG;
    if (!w->w3 || !w->b3 || !w->w2 || !w->b2 || !w->w1 || !w->b1) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    if (n > tail_fwd_cap()) return CGS_ERR_UNSUPPORTED;
    TailDecFwdParams P{*w, e1, e2, e3, o4, o3, o2, o1, n, n, g_tail_stamps ? g_tail_stamps + 1 * 2048 * 16 : nullptr, w_m0, m0_pack};
    ConvParams PC{};
    PC.src_a = e0; PC.src_b = o1; PC.w = w0; PC.bias = b0; PC.out = o0; PC.n = n;
    const size_t lds = conv_lds_bytes<FDec0P>();
    static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_dec_fwd_kernel<true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)(conv_lds_bytes<FDec0P>()));
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL(tail_dec_fwd_kernel<true>, dim3(n + (m0_pack ? 1 : 0)), dim3(256), lds, (hipStream_t)stream, P, PC);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// cgs_tail_dec_fwd with fp16 OPERANDS in dec_model.3 / .2 / .1 (fp32 accumulation, fp32 tensors in and out; BASELINE config 4: nets.py:501-513 in
// the -process path with fp16 conv kernels)
extern "C" int cgs_tail_dec_fwd_h16(int32_t n, const cgs_tail_dec_weights* w, const float* e1, const float* e2, const float* e3,
                                    const float* o4, float* o3, float* o2, float* o1, cgs_stream_t stream) {
    if (n < 0 || !w || !e1 || !e2 || !e3 || !o4 || !o3 || !o2 || !o1) return CGS_ERR_BADARG;
    if (!w->w3 || !w->b3 || !w->w2 || !w->b2 || !w->w1 || !w->b1) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    const int blocks = tail_blocks(n, tail_dec_fwd_cap());
    TailDecFwdParams P{*w, e1, e2, e3, o4, o3, o2, o1, n, blocks, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL((tail_dec_fwd_kernel<false, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, ConvParams{});
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_tail_dec_fwd(int32_t n, const cgs_tail_dec_weights* w, const float* e1, const float* e2, const float* e3,
                                const float* o4, float* o3, float* o2, float* o1, cgs_stream_t stream) {
    return cgs_tail_dec_fwd_pack(n, w, e1, e2, e3, o4, o3, o2, o1, nullptr, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------
// encoder tail + critic head, backward (one critic pass).  Per image:
//   head backward (crit.4, crit.1, features.14, and the decoder's 1x1 conv when d_o4 is given) -> d e3
//   -> re-expansion through the pool/ReLU nibbles -> features.10 weight + data gradient -> Dropout mask, + skip gradient
//   -> re-expansion -> features.6 weight + data gradient (+ skip gradient) -> d e1 (memory, for features.3's backward).
// The convolutions' weight-gradient partials stay in registers across the workgroup's images (one slab per layer per
// workgroup: slab10 [1152 | 16], slab6 [576 | 8]); the head's weight gradients -- sums of outer products over the batch -- are
// left to cgs_tail_head_wgrad below, which reads the per-image vectors this kernel writes to hvec.
// ------------------------------------------------------------------------------------------------
static constexpr int kTailSlab10 = 1168, kTailSlab6 = 584;
// (round 6) the two pre-pool gradient tiles are read by the DATA gradients as 16-byte channel planes with lane = pixel (conv_px, tail4.h, as the
// forward kernel reads x1 / x2): padded slots / pitches instead of the dword reads of conv_tiles, which were 4-way (8-channel tile: 32 lanes on 8
// banks) and 8-way (16-channel tile: 4 banks) bank-conflicted.  DY3P: 16-float slots, pitch / 4 = 41 = 9 (mod 16): the 16 lanes a ds_read_b128
// serves together ({0-3, 12-15, 20-27}: quads 0, 3, 5, 6 = rows 0-1 columns {0,1,6,7} + rows 2-3 columns {2..5} of the 8-wide map) fall on
// slot groups {0,4,8,12} + 9 * row: 16 different ones.
using DY2P = TileP<16, 16, 8, 8, 148>;    // d(features.6 pre-pool)
using DY3P = TileP<8, 8, 16, 16, 164>;    // d(features.10 pre-pool)
// floats of tail_enc_bwd_kernel's one static LDS block (its tiles + weights + per-image scratch, or the dec_model.0 rider's tiles)
static constexpr int kTailEncBwdTileFloats = T16x8::FLOATS + T8x8::FLOATS + DY2P::FLOATS + DY3P::FLOATS;
static constexpr int kTailEncBwdOwnFloats = ((kTailEncBwdTileFloats + 3) & ~3) + 72 * 8 + 72 * 16 + 256 + 512 + 512 + 2048 + 64 + 32;
static constexpr int kTailEncBwdLdsFloats = kTailEncBwdOwnFloats > kWD0LdsFloats ? kTailEncBwdOwnFloats : kWD0LdsFloats;

struct TailEncBwdParams {
    cgs_tail_enc_weights w;
    const float* e1; const float* e2; const uint32_t* am2; const float* e3; const uint32_t* am3;
    const float* e4; const float* h1; const float* pred; const float* dpred;
    const float* target; float loss_scale; int bce;
    const float* dE1; const float* dE2; const float* dE3; const float* d_o4; int n_add;
    float* de1;
    float* hvec; float* slab10; float* slab6;
    cgs_dropout drop_e2, drop_e3, drop_h1;
    int n;
    int nblocks;
    unsigned long long* dbg;
    WDec0Params rider;      // optional: dec_model.0's weight gradient as spare workgroups of this launch (rider.slab != NULL)
};

// ENC1 (round 5): the workgroup also runs features.3's DATA gradient of its image(s) after the tail stages (conv3x3_body_pipe<DEnc1P>, the
// body cgs_conv3x3_bwd_both launches for that layer: d e1 read back from memory inside the workgroup, argmax nibbles re-expanded in the
// loader, the decoder's skip gradient added in the epilogue; the static LDS block is its scratch) -- one launch boundary of the step's
// dependent chain less per critic pass.  features.3's sparse weight gradient (only the final reduction waits for it) moves into the
// features.0 backward launch that follows (cgs_enc0_bwd_mix_enc1 / cgs_enc0_wgrad_u8_with_head_enc1).
#ifndef CGS_ENC1_STAGGER
#define CGS_ENC1_STAGGER 64
#endif
#ifndef CGS_DROPCTX_SPLIT
#define CGS_DROPCTX_SPLIT 0      // (r05x A/B: 0.5570 ms either way; the split form spills 9 more SGPRs)
#endif
#ifndef CGS_ENC1_WGRAD_STAGES
#define CGS_ENC1_WGRAD_STAGES 1
#endif
// MODE 2 / 3 (round 5): ... and features.0's sparse WEIGHT gradient of the same images on the uint8 frames (2) / the virtual mixes (3), after
// their d e0 has been written: the critic's whole backward pass of an image except features.0's data gradient in one workgroup; the
// stand-alone launches then hold no weight-gradient role of this pass (cgs_enc0_bwd_mix(slab = NULL), cgs_tail_head_wgrad alone).
template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) tail_enc_bwd_kernel(TailEncBwdParams P, ConvParams PC, WgradParams PW1,
                                                                                                       WgradParams PW0) {
    constexpr bool ENC1 = MODE >= 1;
    if (CGS_KARG_PREFETCH) cgs_kernarg_prefetch<sizeof(TailEncBwdParams) + (ENC1 ? sizeof(ConvParams) + sizeof(WgradParams) : 0) + (MODE >= 2 ? sizeof(WgradParams) : 0)>();
    // The chain of one image is latency-bound, so: every global load of an image is issued at the top of its iteration (one
    // memory latency instead of one per stage), the head runs redundantly in all waves on shuffles (no single-wave sections),
    // four barriers per image.
    constexpr int OX1 = 0, OX2 = OX1 + T16x8::FLOATS, ODY2 = OX2 + T8x8::FLOATS, ODY3 = ODY2 + DY2P::FLOATS, OEND = ODY3 + DY3P::FLOATS;
    static_assert(OX2 % 4 == 0 && ODY2 % 4 == 0 && ODY3 % 4 == 0, "16-byte aligned tiles");
    // one LDS block, carved by hand: the rider workgroups below (dec_model.0's weight gradient) use the same bytes their own way
    constexpr int OW6 = (OEND + 3) & ~3, OW10 = OW6 + 72 * 8, OXS = OW10 + 72 * 16, OM2 = OXS + 256, OD2 = OM2 + 512, OO1 = OD2 + 512,
                  OAM2 = OO1 + 2048, ODZ4 = OAM2 + 64, OALL = ODZ4 + 32;
    constexpr int LDS_ALL = OALL > kWD0LdsFloats ? OALL : kWD0LdsFloats;
    static_assert(LDS_ALL == kTailEncBwdLdsFloats, "the launcher checks the fused convolution's scratch against this size");
    __shared__ __attribute__((aligned(16))) float lds_all[LDS_ALL];
    if ((int)blockIdx.x >= P.nblocks) {          // riders (launched only when rider.slab is given): see cgs_tail_enc_bwd_rider
        wgrad_dec0_body(P.rider, (int)blockIdx.x - P.nblocks, (int)gridDim.x - P.nblocks, lds_all);
        return;
    }
#if CGS_ENC1_STAGGER
    if constexpr (ENC1) cgs_stagger<8, CGS_ENC1_STAGGER>();   // (A/B) latency-bound tail stages, then the issue-bound convolution: offset the co-resident workgroups
#endif
    float* tiles = lds_all;                                              // after the loop: scratch of the weight-gradient reduction
    float* x1 = tiles + OX1;       // e1: X of features.6
    float* x2 = tiles + OX2;       // dropout(e2): X of features.10
    float* dy2 = tiles + ODY2;     // gradient at features.6's pre-pool output
    float* dy3 = tiles + ODY3;     // gradient at features.10's pre-pool output
    float* w6s = lds_all + OW6; float* w10s = lds_all + OW10;
    float* xs = lds_all + OXS;                                           // dropout(e3), flat
    float* m2s = lds_all + OM2; float* d2s = lds_all + OD2;              // Dropout multipliers of e2; skip gradient dE2
    float* o1s = lds_all + OO1;                                          // d e1 before the coalesced store
    uint32_t* am2s = (uint32_t*)(lds_all + OAM2);
    float* dz4s = lds_all + ODZ4;
    static_assert(OEND >= 4 * 5 * 256, "reduction scratch fits the tile area");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (CGS_STAMP_PTR(P.dbg) && tid == 0) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16] = __builtin_amdgcn_s_memtime();
    const int l15 = lane & 15;
    const int o = tid & 31, kg = tid >> 5, half = lane & 32;
    const int hk = tid >> 3, part = tid & 7;          // d e4 mapping: row k = hk, columns 4*part .. +3
    // the three Dropout step counters are REQUESTED here and consumed after the first image's loads have been issued (drop_ctx3_fill below)
    DropCtx d2_{}, d3_{}, dh_{};
    uint64_t sv2, sv3, svh;
    drop_ctx3_load(P.drop_e2, P.drop_e3, P.drop_h1, P.w.w6, sv2, sv3, svh);
#if !CGS_DROPCTX_SPLIT
    drop_ctx3_fill(P.drop_e2, P.drop_e3, P.drop_h1, sv2, sv3, svh, d2_, d3_, dh_);      // (A/B: the round-4 placement)
#endif
    const bool has_pw = P.d_o4 != nullptr;

    // (the once-per-workgroup LDS set-up -- zero halos, convolution weights -- runs inside the first iteration, BEHIND the first
    //  image's global loads: the set-up then costs no memory latency of its own)
    WgradAccK<T8x8, DY3P, 16> wg10;
    WgradAccK<T16x8, DY2P, 8> wg6;
    wg10.init(lane);
    wg6.init(lane);

    for (int img = blockIdx.x; img < P.n; img += P.nblocks) {
        TAIL_STAMP(1);
        const bool add = img < P.n_add;
        // Opaque zero: the LDS addresses of the MFMA stages depend only on the lane, so the compiler hoists ALL of them out of
        // the image loop and keeps them in registers (-> 256 VGPRs + scratch spills).  Deriving them from lane + lz keeps them
        // per-iteration temporaries.
        int lz = 0;
        asm volatile("" : "+v"(lz));
        const int lane_i = lane + lz;
        // ---- every global load of this image, back to back ----
        // (scalars, not arrays: an array the optimiser fails to fully unroll becomes a per-thread LDS array indexed by the flat
        //  work-item id, which needs the workgroup size from the dispatch packet -- a ~10 us load from host-visible memory)
        float4 le23 = f4zero(), ldE2 = f4zero(), lde1a = f4zero(), lde1b = f4zero();
        const float4 le1a = ((const float4*)P.e1)[(size_t)img * 512 + tid], le1b = ((const float4*)P.e1)[(size_t)img * 512 + tid + 256];
        if (P.dE1 && add) {       // parked in o1s below
            lde1a = ((const float4*)P.dE1)[(size_t)img * 512 + tid];
            lde1b = ((const float4*)P.dE1)[(size_t)img * 512 + tid + 256];
        }
        uint32_t lam2 = 0;
        int tl = tid;             // opaque per image: as loop invariants the per-thread 64-bit addresses of these guarded loads were hoisted out of the
        asm volatile("" : "+v"(tl));      // image loop and two of them spilled -- each reload an s_waitcnt vmcnt(0) in front of the load it feeds
        if (tid < 128) {
            le23 = ((const float4*)P.e2)[(size_t)img * 128 + tl];
            if (P.dE2 && add) ldE2 = ((const float4*)P.dE2)[(size_t)img * 128 + tl];
            if (tid < 64) lam2 = P.am2[(size_t)img * 64 + tl];
        } else if (tid < 192) {
            le23 = ((const float4*)P.e3)[(size_t)img * 64 + tl - 128];
        }
        const uint32_t lam3 = P.am3[(size_t)img * 32 + (tid >> 3)];           // word of pooled pixel tid >> 4, channel half (tid >> 3) & 1
        const float ldE3 = (P.dE3 && add) ? P.dE3[(size_t)img * 256 + tid] : 0.f;
        const float ev = P.e4[(size_t)img * 32 + o], hv = P.h1[(size_t)img * 32 + o];
        const float pr = P.pred[img];
        float dpr = 0.f;          // d loss / d pred: given, or derived from the target (MSE / BCE terms of main.py:380-411)
        if (P.dpred) dpr = P.dpred[img];
        else if (P.target) {
            const float tg = P.target[img];
            dpr = P.bce ? P.loss_scale * (pr - tg) / fmaxf((1.f - pr) * pr, 1e-12f) : P.loss_scale * 2.f * (pr - tg);
        }
        const float go4 = (has_pw && add) ? P.d_o4[(size_t)img * 32 + o] : 0.f;
        // head weights of this thread (L2-resident; reloaded per image rather than held in registers across the loop)
        const float4 w1v = *(const float4*)(P.w.wl1 + hk * 32 + 4 * part);
        const float4 wpv = has_pw ? *(const float4*)(P.w.wpw + hk * 32 + 4 * part) : f4zero();
        const float wl2 = P.w.wl2[o];
#if CGS_DROPCTX_SPLIT
        if (img == (int)blockIdx.x) drop_ctx3_fill(P.drop_e2, P.drop_e3, P.drop_h1, sv2, sv3, svh, d2_, d3_, dh_);
#endif
        // (the Philox round keys: 3 contexts x 20 loop-invariant scalars otherwise live across the loop -> SGPR spills)
        DropCtx d2 = d2_, d3 = d3_, dh = dh_;
        asm volatile("" : "+s"(d2.key.x), "+s"(d2.key.y), "+s"(d3.key.x), "+s"(d3.key.y), "+s"(dh.key.x), "+s"(dh.key.y));
        if (img == (int)blockIdx.x) {        // first image of this workgroup (uniform): the set-up, with the loads above in flight
            const int ts = tid + lz;         // (opaque: the set-up's addresses must not become loop invariants held in registers)
            tile_zero<T16x8>(x1, ts);
            tile_zero<T8x8>(x2, ts);
            tilep_zero<DY2P>(dy2, ts);
            tilep_zero<DY3P>(dy3, ts);
            TAIL_STAMP(12);
            for (int e = ts; e < 72 * 8 / 4; e += 256) ((float4*)w6s)[e] = ((const float4*)P.w.w6)[e];
            for (int e = ts; e < 72 * 16 / 4; e += 256) ((float4*)w10s)[e] = ((const float4*)P.w.w10)[e];
            __syncthreads();                 // the zeroes land before any thread's tile commit below
            TAIL_STAMP(13);
        }
        // Dropout multipliers while the loads are in flight
        __builtin_amdgcn_sched_barrier(0);        // (one Philox at a time: interleaved they spill)
        float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
        if (tid < 128) { if (d2.on) mk = drop_mult4(d2, (uint32_t)(img * 128 + tid)); }
        else if (tid < 192) { if (d3.on) mk = drop_mult4(d3, (uint32_t)(img * 64 + tid - 128)); }
        __builtin_amdgcn_sched_barrier(0);
        const float m2 = drop1(dh, (uint32_t)(img * 32 + o));
        __builtin_amdgcn_sched_barrier(0);
        const float m3 = drop1(d3, (uint32_t)(img * 256 + tid));
        __builtin_amdgcn_sched_barrier(0);
        // ---- commit to LDS ----
        {
            const int p = tid & 1, x = (tid >> 1) & 15, y = tid >> 5;          // float4 index e = tid and tid + 256 (8 rows further)
            *(float4*)(x1 + T16x8::at(y, x) + 4 * p) = le1a;
            *(float4*)(x1 + T16x8::at(y + 8, x) + 4 * p) = le1b;
            ((float4*)o1s)[tid] = lde1a;               // features.6's data gradient is added on top of the skip gradient
            ((float4*)o1s)[tid + 256] = lde1b;
        }
        if (tid < 128) {
            const int q = tid >> 1, p = tid & 1;
            *(float4*)(x2 + T8x8::at(q >> 3, q & 7) + 4 * p) = le23 * mk;
            ((float4*)m2s)[tid] = mk;
            ((float4*)d2s)[tid] = ldE2;
            if (tid < 64) am2s[tid] = lam2;
        } else if (tid < 192) {
            ((float4*)xs)[tid - 128] = le23 * mk;
        }
        // ---- head on shuffles, redundantly in every wave (a half-wave holds all 32 values of o) ----
        const float dz2 = dpr * pr * (1.f - pr);
        const float dh1 = hv > 0.f ? dz2 * wl2 * m2 : 0.f;
        // the head's weight gradients are sums of outer products over the batch: they are left to cgs_tail_head_wgrad (one small
        // MFMA GEMM over all images) -- here only the per-image vectors it needs are written (hvec [n][384]):
        //   [0,256) dropout(e3)   [256,288) dz4   [288,320) dh1   [320,352) dz2 * h1 * mask   [352] dz2
        if (P.hvec) {
            float* hvp = P.hvec + (size_t)img * 384;
            if (tid >= 128 && tid < 192) ((float4*)hvp)[tid - 128] = le23 * mk;
            if (kg == 0) {
                hvp[288 + o] = dh1;
                hvp[320 + o] = dz2 * hv * m2;
                if (o == 0) hvp[352] = dz2;
            }
        }
        {   // d e4[hk] = sum_o' wl1[hk][o'] dh1[o'] + sum_j wpw[hk][j] d o4[j]: 8 lanes x 4 columns, then ReLU of features.14
            const float w1a[4] = {w1v.x, w1v.y, w1v.z, w1v.w}, wpa[4] = {wpv.x, wpv.y, wpv.z, wpv.w};
            float part_sum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                part_sum = fmaf(w1a[j], __shfl(dh1, half + 4 * part + j, 64), part_sum);
                part_sum = fmaf(wpa[j], __shfl(go4, half + 4 * part + j, 64), part_sum);
            }
            part_sum += __shfl_xor(part_sum, 1, 64);
            part_sum += __shfl_xor(part_sum, 2, 64);
            part_sum += __shfl_xor(part_sum, 4, 64);
            const float ek = __shfl(ev, half + hk, 64);
            if (part == 0) {
                const float dz4 = ek > 0.f ? part_sum : 0.f;
                dz4s[hk] = dz4;
                if (P.hvec) P.hvec[(size_t)img * 384 + 256 + hk] = dz4;
            }
        }
        __syncthreads();
        TAIL_STAMP(2);
        // ---- d e3[k = tid] -> straight into features.10's pre-pool gradient tile ----
        {
            float4 w14r[8];                       // features.14 row k = tid (L2-resident)
#pragma unroll
            for (int q4 = 0; q4 < 8; ++q4) w14r[q4] = ((const float4*)(P.w.w14 + (size_t)tid * 32))[q4];
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int q4 = 0; q4 < 8; ++q4) {
                const float4 dv = *(const float4*)(dz4s + 4 * q4);
                s0 = fmaf(w14r[q4].x, dv.x, s0); s1 = fmaf(w14r[q4].y, dv.y, s1);
                s2 = fmaf(w14r[q4].z, dv.z, s2); s3 = fmaf(w14r[q4].w, dv.w, s3);
            }
            const float r = ((s0 + s1) + (s2 + s3)) * m3 + ldE3;
            const int q = tid >> 4, c = tid & 15, qy = q >> 2, qx = q & 3;
            const uint32_t nib = (lam3 >> (4 * (c & 7))) & 15u;
#pragma unroll
            for (int pos = 0; pos < 4; ++pos)
                dy3[DY3P::at(2 * qy + (pos >> 1), 2 * qx + (pos & 1)) + c] = (nib == (uint32_t)pos) ? r : 0.f;
        }
        __syncthreads();
        TAIL_STAMP(3);
        // ---- features.10: weight gradient (waves 2, 3) NEXT TO the data gradient (waves 0, 1: four of e2's eight channels each, lane = pixel
        //      of the 8x8 map, v_mfma_f32_4x4x1 on 16-byte planes of dy3) -> Dropout mask, + skip gradient, re-expansion through features.6's
        //      argmax nibbles into dy2 (four 16-byte stores per lane) ----
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        if (wv >= 2) {
            wg10.template accumulate<2>(x2, dy3, wv - 2, lane_i);
        } else {
            float wd10[1][9];           // W^T of this wave's four channels: step = (tap, co), re-read from LDS per image
            fill_wreg<1, 9, 144>(wd10, lane_i, [&](int step, int ci) { return w10s[((step >> 4) * 8 + 4 * wv + ci) * 16 + (step & 15)]; });
            frag4 a[1] = {frag4{0.f, 0.f, 0.f, 0.f}};
            const PxPos pb = px8(lane_i);
            conv_px<DY3P, 0, 16, 16, 0, 1, 9, true, true>(a, dy3, pb.y, pb.x, wd10);
            const int pp = pb.y * 8 + pb.x;                     // pixel of the 8x8 map = pooled pixel of features.6
            const float4 mk4 = *(const float4*)(m2s + pp * 8 + 4 * wv), sk4 = *(const float4*)(d2s + pp * 8 + 4 * wv);
            const float4 r = make_float4(a[0][0] * mk4.x + sk4.x, a[0][1] * mk4.y + sk4.y, a[0][2] * mk4.z + sk4.z, a[0][3] * mk4.w + sk4.w);
            const uint32_t nib16 = (am2s[pp] >> (16 * wv)) & 0xFFFFu;
#pragma unroll
            for (int pos = 0; pos < 4; ++pos)
                *(float4*)(dy2 + DY2P::at(2 * pb.y + (pos >> 1), 2 * pb.x + (pos & 1)) + 4 * wv) = nib_select4(r, nib16, (uint32_t)pos);
        }
        __syncthreads();
        TAIL_STAMP(4);
        // ---- features.6: data gradient (lane = pixel, this wave's 64 pixels x 8 channels) + skip gradient -> d e1 straight to memory;
        //      then the weight gradient ----
        {
            float wd6[2][5];
            fill_wreg<2, 5, 72>(wd6, lane_i, [&](int step, int ci) { return w6s[((step >> 3) * 8 + ci) * 8 + (step & 7)]; });
            frag4 a[2] = {frag4{0.f, 0.f, 0.f, 0.f}, frag4{0.f, 0.f, 0.f, 0.f}};
            const PxPos pa = px16(wave, lane_i);
            conv_px<DY2P, 0, 8, 8, 0, 2, 5, true>(a, dy2, pa.y, pa.x, wd6);
            const int pix = pa.y * 16 + pa.x;
            const float4 s0 = ((const float4*)o1s)[pix * 2], s1 = ((const float4*)o1s)[pix * 2 + 1];      // the skip gradient parked at commit
            float4* out = (float4*)P.de1 + (size_t)img * 512 + pix * 2;
            out[0] = make_float4(s0.x + a[0][0], s0.y + a[0][1], s0.z + a[0][2], s0.w + a[0][3]);
            out[1] = make_float4(s1.x + a[1][0], s1.y + a[1][1], s1.z + a[1][2], s1.w + a[1][3]);
        }
        wg6.accumulate(x1, dy2, wave, lane_i);
        __syncthreads();
        TAIL_STAMP(5);
    }

    // ---- one slab per layer for this workgroup ----
    __syncthreads();
    const size_t b = blockIdx.x;
    wg10.reduce_store(P.slab10 ? P.slab10 + b * kTailSlab10 : nullptr, tiles, wave, lane, tid);
    wg6.reduce_store(P.slab6 ? P.slab6 + b * kTailSlab6 : nullptr, tiles, wave, lane, tid);
    if (CGS_STAMP_PTR(P.dbg) && tid == 0) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16 + 6] = __builtin_amdgcn_s_memtime();
    if constexpr (ENC1) {
        // features.3's data gradient of this workgroup's images: their d e1 is in memory (stored by this workgroup, visible after the
        // barrier), every tile and the reduction scratch above are dead
        auto enc1_dgrad = [&]() {
            for (int img = blockIdx.x; img < P.n; img += P.nblocks) {
                __syncthreads();
                conv3x3_body_pipe<DEnc1P>(PC, 2 * img, (float4*)lds_all);
            }
            if (CGS_STAMP_PTR(P.dbg) && tid == 0) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16 + 7] = __builtin_amdgcn_s_memtime();
        };
        if (PW1.slab) {
            // features.3's WEIGHT gradient of this workgroup's images, sparse form (wgrad_sparse.h: one non-zero of the pre-pool gradient
            // per pooling window and channel), one slab row per workgroup.  As rider workgroups of the features.0 backward launches the
            // same work cost 24 us per step (r05k).  Here its first tile's global loads (e0 rows, d e1, argmax words) are issued, THEN the
            // data gradient runs (the `between` stage: its 108 registers + the 29 of a prefetch stage fit the kernel's 168), then the
            // tiles are multiplied: by the stamps the phase waited ~2 us per tile for its loads when nothing ran in between.
            // (opaque block index / zero: nothing of this phase can be computed -- and held in registers -- before this point)
            int b0 = (int)blockIdx.x, tz = 0;
            asm volatile("" : "+s"(b0));
            asm volatile("" : "+v"(tz));
            const int nb = P.nblocks;
            const int nimg = b0 < P.n ? (P.n - b0 + nb - 1) / nb : 0;
            wgrad_sparse8_body_seq<CGS_ENC1_WGRAD_STAGES>(
                PW1, [=](int k) { return 2 * (b0 + (k >> 1) * nb) + (k & 1); }, 2 * nimg, PW1.slab + (size_t)b0 * ((9 * 8 + 1) * 8),
                (float4*)lds_all, tz, [&]() { enc1_dgrad(); __syncthreads(); });
        } else {
            enc1_dgrad();
        }
        if constexpr (MODE >= 2) {
            if (PW0.slab) {
                // features.0's weight gradient of this workgroup's images (8 strips each) from the d e0 just written (same workgroup: visible
                // after the barrier) and the frames / virtual mixes; one slab row per workgroup
                __syncthreads();
                int b0 = (int)blockIdx.x, tz = 0;
                asm volatile("" : "+s"(b0));
                asm volatile("" : "+v"(tz));
                const int nb = P.nblocks;
                const int nimg = b0 < P.n ? (P.n - b0 + nb - 1) / nb : 0;
                using SP0 = SpCfg<64, 3, MODE == 2 ? WSRC_U8 : WSRC_MIX>;
                wgrad_sparse_body_seq<SP0>(PW0, [=](int k) { return 8 * (b0 + (k >> 3) * nb) + (k & 7); }, 8 * nimg,
                                           PW0.slab + (size_t)b0 * ((9 * 3 + 1) * 8), (float4*)lds_all, tz);
            }
        }
    }
    if (CGS_STAMP_PTR(P.dbg) && tid == 0) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16 + 15] = __builtin_amdgcn_s_memtime();
}

extern "C" int cgs_tail_enc_bwd_slabs(int32_t n) { return n < 0 ? CGS_ERR_BADARG : tail_blocks(n, tail_enc_bwd_cap()); }

// cgs_tail_enc_bwd with dec_model.0's weight gradient (cgs_conv3x3_bwd_weight of that layer: n_r images, skip input e0_r [n_r,32,32,8],
// low-resolution input o1_r [n_r,16,16,8], output gradient dy_r [n_r,32,32,8], slab_r [nslab_r][1160]) as nslab_r SPARE workgroups of
// the launch: a tail launch at N = 512 is one image per workgroup and two workgroups per CU -- a third fits (168 registers, 50 KB of
// LDS each) and only the step's final reduction waits for that gradient.  slab_r = NULL: plain cgs_tail_enc_bwd.
static int tail_enc_bwd_launch(int32_t n, const cgs_tail_enc_weights* w, const float* e1, const float* e2, const uint32_t* am2,
                               const float* e3, const uint32_t* am3, const float* e4, const float* h1, const float* pred,
                               const float* dpred, const float* target, float loss_scale, int32_t bce, const float* dE1,
                               const float* dE2, const float* dE3, const float* d_o4, int32_t n_add, float* de1, float* hvec,
                               float* slab10, float* slab6, cgs_dropout drop_e2, cgs_dropout drop_e3, cgs_dropout drop_h1,
                               int32_t n_r, const float* e0_r, const float* o1_r, const float* dy_r, float* slab_r, int32_t nslab_r,
                               const ConvParams* enc1, const WgradParams* enc1w, const WgradParams* enc0w, int enc0_mix, cgs_stream_t stream) {
    if (n < 0 || !w || !e1 || !e2 || !am2 || !e3 || !am3 || !e4 || !h1 || !pred || !de1) return CGS_ERR_BADARG;
    if (!w->w6 || !w->w10 || !w->w14 || !w->wl1 || !w->wl2) return CGS_ERR_BADARG;
    if (d_o4 && !w->wpw) return CGS_ERR_BADARG;
    if (slab_r && (n_r <= 0 || !e0_r || !o1_r || !dy_r || nslab_r <= 0 || nslab_r > n_r * kStrips)) return CGS_ERR_BADARG;
    if (n == 0) return slab_r ? CGS_ERR_BADARG : CGS_OK;
    TailEncBwdParams P{*w, e1, e2, am2, e3, am3, e4, h1, pred, dpred, target, loss_scale, bce, dE1, dE2, dE3, d_o4, n_add, de1,
                       hvec, slab10, slab6, drop_e2, drop_e3, drop_h1, n, tail_blocks(n, tail_enc_bwd_cap()), g_tail_stamps ? g_tail_stamps + 2 * 2048 * 16 : nullptr,
                       WDec0Params{e0_r, o1_r, dy_r, slab_r, n_r, n_r * kStrips}};
    const int riders = slab_r ? nslab_r : 0;
    const dim3 grid(tail_blocks(n, tail_enc_bwd_cap()) + riders);
    const WgradParams w1 = enc1w ? *enc1w : WgradParams{}, w0 = enc0w ? *enc0w : WgradParams{};
    if (enc1 && enc0w && enc0_mix) hipLaunchKernelGGL(tail_enc_bwd_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, P, *enc1, w1, w0);
    else if (enc1 && enc0w) hipLaunchKernelGGL(tail_enc_bwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, P, *enc1, w1, w0);
    else if (enc1) hipLaunchKernelGGL(tail_enc_bwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, P, *enc1, w1, w0);
    else hipLaunchKernelGGL(tail_enc_bwd_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, P, ConvParams{}, w1, w0);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_tail_enc_bwd_rider(int32_t n, const cgs_tail_enc_weights* w, const float* e1, const float* e2, const uint32_t* am2,
                                      const float* e3, const uint32_t* am3, const float* e4, const float* h1, const float* pred,
                                      const float* dpred, const float* target, float loss_scale, int32_t bce, const float* dE1,
                                      const float* dE2, const float* dE3, const float* d_o4, int32_t n_add, float* de1, float* hvec,
                                      float* slab10, float* slab6, cgs_dropout drop_e2, cgs_dropout drop_e3, cgs_dropout drop_h1,
                                      int32_t n_r, const float* e0_r, const float* o1_r, const float* dy_r, float* slab_r, int32_t nslab_r,
                                      cgs_stream_t stream) {
    return tail_enc_bwd_launch(n, w, e1, e2, am2, e3, am3, e4, h1, pred, dpred, target, loss_scale, bce, dE1, dE2, dE3, d_o4, n_add, de1, hvec,
                               slab10, slab6, drop_e2, drop_e3, drop_h1, n_r, e0_r, o1_r, dy_r, slab_r, nslab_r, nullptr, nullptr, nullptr, 0, stream);
}

// cgs_tail_enc_bwd_rider AND features.3's data gradient (the data-gradient half of cgs_conv3x3_bwd_both for the 8 -> 8 layer at 32x32 with
// ReLU + pool: d e1 [n,16,16,8] re-expanded by the argmax nibbles am1, weights w_enc1 (HWIO), + addend0 [n_addend,32,32,8] (the decoder's
// skip gradient at e0, may be NULL) -> de0 [n,32,32,8]) in one launch: every workgroup continues with the convolution of the image(s)
// whose tail it just ran (nets.py:173-194 backward; round 5).
extern "C" int cgs_tail_enc_bwd_enc1(int32_t n, const cgs_tail_enc_weights* w, const float* e1, const float* e2, const uint32_t* am2,
                                     const float* e3, const uint32_t* am3, const float* e4, const float* h1, const float* pred,
                                     const float* dpred, const float* target, float loss_scale, int32_t bce, const float* dE1,
                                     const float* dE2, const float* dE3, const float* d_o4, int32_t n_add, float* de1, float* hvec,
                                     float* slab10, float* slab6, cgs_dropout drop_e2, cgs_dropout drop_e3, cgs_dropout drop_h1,
                                     int32_t n_r, const float* e0_r, const float* o1_r, const float* dy_r, float* slab_r, int32_t nslab_r,
                                     const uint32_t* am1, const float* w_enc1, const float* addend0, int32_t n_addend, float* de0,
                                     const float* e0, float* slab1,
                                     int32_t x_kind, const void* x, const uint32_t* am0, float* slab0, cgs_stream_t stream) {
    if (!am1 || !w_enc1 || !de0 || n_addend < 0 || (n_addend > 0 && !addend0) || (slab1 && !e0)) return CGS_ERR_BADARG;
    if (slab0 && (!x || !am0 || (x_kind != CGS_SRC_U8 && x_kind != CGS_SRC_MIX))) return CGS_ERR_BADARG;
    if (slab0 && wgrad_sparse_lds_bytes<SpCfg<64, 3, WSRC_U8>>() > sizeof(float) * (size_t)kTailEncBwdLdsFloats) return CGS_ERR_UNSUPPORTED;
    if (conv_lds_bytes<DEnc1P>() > sizeof(float) * (size_t)kTailEncBwdLdsFloats) return CGS_ERR_UNSUPPORTED;
    if (wgrad_sparse_lds_bytes<SpCfg<32, 8, WSRC_F32>>() > sizeof(float) * (size_t)kTailEncBwdLdsFloats) return CGS_ERR_UNSUPPORTED;
    ConvParams pd{};
    pd.src_a = de1; pd.amask_in = am1; pd.w = w_enc1; pd.out = de0; pd.addend = addend0; pd.n_addend = n_addend; pd.n = n;
    WgradParams pw{};
    pw.src_a = e0; pw.dy = de1; pw.amask = am1; pw.slab = slab1; pw.n = n; pw.ntiles = 2 * n;
    WgradParams p0{};
    if (slab0) {
        p0.dy = de0; p0.amask = am0; p0.slab = slab0; p0.n = n; p0.ntiles = 8 * n;
        if (x_kind == CGS_SRC_MIX) {
            const cgs_mix_src* m = (const cgs_mix_src*)x;
            if (!m->a || !m->b || !m->z || m->n_a <= 0 || (n != m->n_a && n != 2 * m->n_a)) return CGS_ERR_BADARG;
            p0.mix_a = m->a; p0.mix_b = m->b; p0.mix_z = m->z; p0.mix_n_a = m->n_a;
        } else {
            p0.src_a = x;
        }
    }
    return tail_enc_bwd_launch(n, w, e1, e2, am2, e3, am3, e4, h1, pred, dpred, target, loss_scale, bce, dE1, dE2, dE3, d_o4, n_add, de1, hvec,
                               slab10, slab6, drop_e2, drop_e3, drop_h1, n_r, e0_r, o1_r, dy_r, slab_r, nslab_r, &pd, &pw, slab0 ? &p0 : nullptr,
                               x_kind == CGS_SRC_MIX, stream);
}

extern "C" int cgs_tail_enc_bwd(int32_t n, const cgs_tail_enc_weights* w, const float* e1, const float* e2, const uint32_t* am2,
                                const float* e3, const uint32_t* am3, const float* e4, const float* h1, const float* pred,
                                const float* dpred, const float* target, float loss_scale, int32_t bce, const float* dE1,
                                const float* dE2, const float* dE3, const float* d_o4, int32_t n_add, float* de1, float* hvec, float* slab10, float* slab6,
                                cgs_dropout drop_e2, cgs_dropout drop_e3, cgs_dropout drop_h1, cgs_stream_t stream) {
    return cgs_tail_enc_bwd_rider(n, w, e1, e2, am2, e3, am3, e4, h1, pred, dpred, target, loss_scale, bce, dE1, dE2, dE3, d_o4, n_add, de1, hvec,
                                  slab10, slab6, drop_e2, drop_e3, drop_h1, 0, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

// ------------------------------------------------------------------------------------------------
// Weight gradients of the critic head (features.14, crit.1, crit.4) and of the decoder's 1x1 conv (dec_model.4) for up to two
// image ranges (the two critic passes of a step) as ONE small GEMM over the images on the matrix cores:
//   dW14[k][o] = sum_img dropout(e3)[img][k] dz4[img][o]   dWl1[k][o] = sum e4[img][k] dh1[img][o]   dWpw[k][j] = sum e4[img][k] d_o4[img][j]
//   db14 = sum dz4, dbl1 = sum dh1, dwl2 = sum dz2 h1 mask, dbl2 = sum dz2, dbpw = sum d_o4
// from the vectors tail_enc_bwd left in hvec.  16 images per workgroup; slab_head [8192 | 32 | 1024 | 32 | 32 | 1], slab_pw [1024 | 32].
// (Inside tail_enc_bwd these sums cost 40 accumulator registers per thread and one 37 KB slab per workgroup and pass.)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) tail_head_wgrad_kernel(HeadWgradParams P) {
    __shared__ __attribute__((aligned(16))) float lds[kHwLdsFloats];
    tail_head_wgrad_body(P, blockIdx.x, lds);
}

extern "C" int cgs_tail_head_wgrad_slabs(int32_t n_total) { return n_total < 0 ? CGS_ERR_BADARG : (n_total + kHwIpb - 1) / kHwIpb; }

extern "C" int cgs_tail_head_wgrad(int32_t n0, const float* hvec0, const float* e4_0, const float* d_o4_0, int32_t n_o4_0, int32_t n1,
                                   const float* hvec1, const float* e4_1, const float* d_o4_1, int32_t n_o4_1, float* slab_head,
                                   float* slab_pw, cgs_stream_t stream) {
    if (n0 < 0 || n1 < 0 || !slab_head || (n0 > 0 && (!hvec0 || !e4_0)) || (n1 > 0 && (!hvec1 || !e4_1))) return CGS_ERR_BADARG;
    if ((d_o4_0 || d_o4_1) && !slab_pw) return CGS_ERR_BADARG;
    if (n0 + n1 == 0) return CGS_OK;
    HeadWgradParams P{{{hvec0, e4_0, d_o4_0, n0, n_o4_0}, {hvec1, e4_1, d_o4_1, n1, n_o4_1}}, slab_head, slab_pw};
    hipLaunchKernelGGL(tail_head_wgrad_kernel, dim3((n0 + n1 + kHwIpb - 1) / kHwIpb), dim3(256), 0, (hipStream_t)stream, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// ------------------------------------------------------------------------------------------------
// decoder tail, backward.  Per image, from d o1 (gradient w.r.t. dec_model.1's output, produced by dec_model.0's backward):
//   dec_model.1: weight gradient; data gradient -> dE1 (skip, memory) and, summed over the 2x2 upsample cell, d o2 (LDS)
//   dec_model.2: likewise -> dE2, d o3;   dec_model.3: likewise -> dE3, d o4 (sum over the 4x4 map, memory)
// Slabs: slab1 [9*16*8 | 8], slab2 [9*24*8 | 8], slab3 [9*48*16 | 16], one per workgroup.
// ------------------------------------------------------------------------------------------------
static constexpr int kTailSlabD1 = 9 * 16 * 8 + 8, kTailSlabD2 = 9 * 24 * 8 + 8, kTailSlabD3 = 9 * 48 * 16 + 16;

struct TailDecBwdParams {
    cgs_tail_dec_weights w;
    const float* e1; const float* e2; const float* e3; const float* o4; const float* o3; const float* o2;
    const float* do1;
    float* dE1; float* dE2; float* dE3; float* d_o4;
    float* slab3; float* slab2; float* slab1;
    int n;
    int nblocks;
    unsigned long long* dbg;
    float* do3;           // W3 = false: d o3 [n,4,4,16] for the dec_model.3 weight-gradient riders
};

using T1Q = TileP<16, 16, 16, 16, 304>;   // cat(e1, up(o2)): only the weight gradient's dword reads (2 taps x 4 channel groups on 32 banks)
using D1Q = TileP<16, 16, 8, 8, 148>;     // d o1: data gradient (b128 reads) and the weight gradient's B operand
struct TailDecBwdLds {
    static constexpr int T1 = 0, T2 = T1 + T1Q::FLOATS, T3 = T2 + T8x24::FLOATS, D1 = T3 + T4x48::FLOATS,
                         D2 = D1 + D1Q::FLOATS, D3 = D2 + T8x8::FLOATS, W1 = D3 + T4x16::FLOATS, W2 = W1 + 144 * 8,
                         RED = W2 + 216 * 8, FLOATS = RED + 4 * 32;
    static constexpr size_t BYTES = (size_t)FLOATS * 4;
};

// FUSED (round 4): one workgroup per image first runs dec_model.0's data gradient of ITS image (conv3x3_body_pipe<DDec0P>: the skip
// gradient d e0 and the cell-summed d o1 go to memory as cgs_conv3x3_bwd_data writes them; the tile region of this kernel is its
// scratch), then the tail stages read that d o1 back (same workgroup: visible after the barrier).  Every other co-resident workgroup
// starts ~4 us late, so the latency-bound chains of one half run under the matrix instructions of the other (cgs_stagger).
// W3 = false (round 5): dec_model.3's weight gradient is NOT formed here -- d o3 [n,4,4,16] goes to P.do3 and a few rider workgroups of a later launch
// form it as a GEMM over the images (wgrad_dec3.h): 64 slab rows of 27.7 KB instead of one per image.
template <bool FUSED, bool W3 = true>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) tail_dec_bwd_kernel(TailDecBwdParams P, ConvParams PC) {
    if (CGS_KARG_PREFETCH) cgs_kernarg_prefetch<sizeof(TailDecBwdParams) + sizeof(ConvParams)>();
    using L = TailDecBwdLds;
    extern __shared__ __attribute__((aligned(16))) float4 smem4[];
    if (CGS_STAMP_PTR(P.dbg) && threadIdx.x == 0) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memtime();      // kernel entry
    // B operand of dec_model.3's data gradient for this wave's 16 input channels: B[k = (tap, co)][col = ci] = w3[(8 - tap)][ci][co] (flipped taps);
    // requested FIRST: the 36 loads land while the convolution below runs
    const int wv3 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    float w3b[36];
    if (wv3 < 3) {
        const float* wp = P.w.w3 + (size_t)(16 * wv3 + ((int)threadIdx.x & 15)) * 16 + (((int)threadIdx.x & 63) >> 4);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) w3b[tap * 4 + sx] = wp[(size_t)(8 - tap) * 48 * 16 + 4 * sx];
    }
    if constexpr (FUSED) {
        cgs_stagger<8, CGS_STAGGER_DEC_BWD>();
        conv3x3_body_pipe<DDec0P>(PC, 2 * (int)blockIdx.x, smem4);
        __syncthreads();            // d o1 of this image is in memory; the convolution's tiles are dead
    }
    float* sm = (float*)smem4;
    float* t1 = sm + L::T1; float* t2 = sm + L::T2; float* t3 = sm + L::T3;
    float* dy1 = sm + L::D1; float* dy2 = sm + L::D2; float* dy3 = sm + L::D3;
    float* w1s = sm + L::W1; float* w2s = sm + L::W2; float* red = sm + L::RED;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (CGS_STAMP_PTR(P.dbg) && tid == 0) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16] = __builtin_amdgcn_s_memtime();
    const int l15 = lane & 15;

    // (the once-per-workgroup LDS set-up -- zero halos, dec_model.1 / .2 weights -- runs inside the first iteration, behind the
    //  first image's global loads)
    // dec_model.1 on v_mfma_f32_4x4x1 (tail4.h): data gradient with lane = pixel (weights W^T in 20 registers), weight gradient as
    // outer products with the blocks as (tap, channel group) combinations (6 accumulators, pixels split over the waves)
    WgradTapBlk<T1Q, D1Q, 4, 2> wg1;
    wg1.init(lane);
    float bs1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // bias gradient of dec_model.1: this lane's pixels
    const PxPos pa = px16(wave, lane);
    WgradAcc<T8x24, T8x8, 8, 4> wg2;
    WgradAcc<T4x48, T4x16, 16, 7> wg3;
    wg2.init(wave, lane); wg3.init(wave, lane);
    for (int img = blockIdx.x; img < P.n; img += P.nblocks) {
        TAIL_STAMP(1);
        int lz = 0;                         // opaque zero: keeps the lane-only LDS addresses from being hoisted out of the image loop
        asm volatile("" : "+v"(lz));
        const int lane_i = lane + lz;
        // ---- every global load of this image, back to back (scalars, not arrays: see tail_enc_bwd) ----
        const float4 le1a = ((const float4*)P.e1)[(size_t)img * 512 + tid], le1b = ((const float4*)P.e1)[(size_t)img * 512 + tid + 256];
        const float4 ld1a = ((const float4*)P.do1)[(size_t)img * 512 + tid], ld1b = ((const float4*)P.do1)[(size_t)img * 512 + tid + 256];
        float4 lsk = f4zero(), lup = f4zero(), lo4 = f4zero();
        if (tid < 128) {
            lsk = ((const float4*)P.e2)[(size_t)img * 128 + tid];
            lup = ((const float4*)P.o2)[(size_t)img * 128 + tid];
            lo4 = ((const float4*)P.o4)[(size_t)img * 8 + (tid & 7)];
        } else if (tid < 192) {
            lsk = ((const float4*)P.e3)[(size_t)img * 64 + tid - 128];
            lup = ((const float4*)P.o3)[(size_t)img * 64 + tid - 128];
        }
        if (img == (int)blockIdx.x) {        // first image of this workgroup (uniform): the set-up, with the loads above in flight
            const int ts = tid + lz;         // (opaque: the set-up's addresses must not become loop invariants held in registers)
            float4 wv[3];                    // dec_model.1: 288 float4, dec_model.2: 432 float4 -> 720 = 2 full rounds + 208
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int e = ts + 256 * k;
                wv[k] = e < 288 ? ((const float4*)P.w.w1)[e] : (e < 720 ? ((const float4*)P.w.w2)[e - 288] : f4zero());
            }
            tilep_zero<T1Q>(t1, ts); tile_zero<T8x24>(t2, ts); tile_zero<T4x48>(t3, ts);
            tilep_zero<D1Q>(dy1, ts); tile_zero<T8x8>(dy2, ts); tile_zero<T4x16>(dy3, ts);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int e = ts + 256 * k;
                if (e < 720) ((float4*)w1s)[e] = wv[k];          // w1s | w2s are contiguous (L::W2 = L::W1 + 144 * 8)
            }
            __syncthreads();                 // the zeroes land before any thread's tile commit below
        }
        // ---- layer inputs (skip ++ upsampled) and d o1 -> tiles ----
        {
            const int p = tid & 1, x = (tid >> 1) & 15, y = tid >> 5;          // float4 index e = tid and tid + 256 (8 rows further)
            *(float4*)(t1 + T1Q::at(y, x) + 4 * p) = le1a;
            *(float4*)(t1 + T1Q::at(y + 8, x) + 4 * p) = le1b;
            *(float4*)(dy1 + D1Q::at(y, x) + 4 * p) = ld1a;
            *(float4*)(dy1 + D1Q::at(y + 8, x) + 4 * p) = ld1b;
        }
        if (tid < 128) {
            const int p = tid & 1, x = (tid >> 1) & 7, y = tid >> 4;
            *(float4*)(t2 + T8x24::at(y, x) + 4 * p) = lsk;
#pragma unroll
            for (int d = 0; d < 4; ++d) *(float4*)(t1 + T1Q::at(2 * y + (d >> 1), 2 * x + (d & 1)) + 8 + 4 * p) = lup;     // up(o2) -> channels 8..15 of t1
            const int p4 = tid & 7, pix = tid >> 3;
            *(float4*)(t3 + T4x48::at(pix >> 2, pix & 3) + 16 + 4 * p4) = lo4;
        } else if (tid < 192) {
            const int e = tid - 128, p = e & 3, x = (e >> 2) & 3, y = e >> 4;
            *(float4*)(t3 + T4x48::at(y, x) + 4 * p) = lsk;
#pragma unroll
            for (int d = 0; d < 4; ++d) *(float4*)(t2 + T8x24::at(2 * y + (d >> 1), 2 * x + (d & 1)) + 8 + 4 * p) = lup;   // up(o3) -> channels 8..23 of t2
        }
        __syncthreads();
        TAIL_STAMP(2);
        // ---- dec_model.1 ----
        wg1.accumulate(t1 + T1Q::at(4 * wave - 1, -1), dy1 + D1Q::at(4 * wave, 0), lane_i);
        {
            frag4 a[4] = {frag4{0.f, 0.f, 0.f, 0.f}, frag4{0.f, 0.f, 0.f, 0.f}, frag4{0.f, 0.f, 0.f, 0.f}, frag4{0.f, 0.f, 0.f, 0.f}};
            float wd1[4][5];        // W^T in 20 registers, re-read from LDS per image (held across the loop they cost the occupancy)
            fill_wreg<4, 5, 72>(wd1, lane_i, [&](int step, int ci) { return w1s[((step >> 3) * 16 + ci) * 8 + (step & 7)]; });
            conv_px<D1Q, 0, 8, 8, 0, 4, 5, true>(a, dy1, pa.y, pa.x, wd1);
            float4* de = (float4*)(P.dE1 + ((size_t)img * 256 + pa.y * 16 + pa.x) * 8);       // skip gradient: channels 0..7
            de[0] = make_float4(a[0][0], a[0][1], a[0][2], a[0][3]);
            de[1] = make_float4(a[1][0], a[1][1], a[1][2], a[1][3]);
            float s[8];                                                                   // upsample backward: sum over the 2x2 cell
#pragma unroll
            for (int c = 0; c < 8; ++c) s[c] = quad_sum(a[2 + (c >> 2)][c & 3]);
            if (pa.pos < 2)
                *(float4*)(dy2 + T8x8::at(pa.qy, pa.qx) + 4 * pa.pos) = pa.pos ? make_float4(s[4], s[5], s[6], s[7]) : make_float4(s[0], s[1], s[2], s[3]);
            const float4 d0 = *(const float4*)(dy1 + D1Q::at(pa.y, pa.x)), d1v = *(const float4*)(dy1 + D1Q::at(pa.y, pa.x) + 4);
            bs1[0] += d0.x; bs1[1] += d0.y; bs1[2] += d0.z; bs1[3] += d0.w;
            bs1[4] += d1v.x; bs1[5] += d1v.y; bs1[6] += d1v.z; bs1[7] += d1v.w;
        }
        __syncthreads();
        TAIL_STAMP(3);
        // ---- dec_model.2 ----
        wg2.accumulate(t2, dy2, lane_i);
        conv_tiles<T8x8, 0, 8, 2>(
            dy2, [&](int tap, int c, int cb) { const int ci = 16 * cb + l15; return w2s[((8 - tap) * 24 + (ci < 24 ? ci : 23)) * 8 + c]; },
            [&](int q, const frag4 (&acc)[2]) {
                const int qy = q >> 2, qx = q & 3;
                if (l15 < 8) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        P.dE2[((size_t)img * 64 + (2 * qy + (j >> 1)) * 8 + 2 * qx + (j & 1)) * 8 + l15] = acc[0][j];
                    dy3[T4x16::at(qy, qx) + 8 + l15] = (acc[1][0] + acc[1][1]) + (acc[1][2] + acc[1][3]);
                } else {
                    dy3[T4x16::at(qy, qx) + l15 - 8] = (acc[0][0] + acc[0][1]) + (acc[0][2] + acc[0][3]);
                }
            },
            wave, lane_i);
        __syncthreads();
        TAIL_STAMP(4);
        if constexpr (!W3) {
            if (tid < 64) {
                const int p4 = tid & 3, x = (tid >> 2) & 3, y = tid >> 4;
                ((float4*)P.do3)[(size_t)img * 64 + tid] = *(const float4*)(dy3 + T4x16::at(y, x) + 4 * p4);
            }
        }
        // ---- dec_model.3: weight gradient on the matrix cores; data gradient on the vector ALU (48 input channels x 4 quads:
        //      thread = (ci, quad), weights read in their natural [tap][ci][co] order, 16 contiguous floats per (tap, ci)) ----
        if constexpr (W3) wg3.accumulate(t3, dy3, lane_i);
        // (round 6) dec_model.3's data gradient on the matrix cores (it was 576 multiply-adds per thread on the vector ALU behind nine dependent L2
        // round trips for the weights: 16.5 k of the kernel's 105 k cycles by the stamps).  GEMM: M = the 16 pixels of the 4x4 map (four 2x2 quads),
        // N = 48 input channels = one 16-column block per wave (waves 0, 1, 2), K = 9 taps x 16 channels of d o3; the B operand (W^T, flipped taps) sits
        // in 36 registers per lane, loaded once per workgroup in front of the image loop.
        if (wv3 < 3) {
            const int q = l15 >> 2, y = 2 * (q >> 1) + ((l15 >> 1) & 1), x = 2 * (q & 1) + (l15 & 1);
            const int kq = lane_i >> 4;
            const float* ap = dy3 + (y * T4x16::PW + x) * T4x16::PCI + kq;        // tap (0,0) = pixel (y-1, x-1) = halo coordinates (y, x)
            frag4 acc = frag4{0.f, 0.f, 0.f, 0.f}, acc1 = frag4{0.f, 0.f, 0.f, 0.f};      // two chains: a dependent 16x16x4 issues after 40 cycles, not 32
            float av[36];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) av[tap * 4 + sx] = ap[((tap / 3) * T4x16::PW + tap % 3) * T4x16::PCI + 4 * sx];
#pragma unroll
            for (int k = 0; k < 36; k += 2) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k], w3b[k], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k + 1], w3b[k + 1], acc1, 0, 0, 0);
            }
            acc += acc1;
            // D[row = pixel 4 kq + j = (quad kq, position j)][col = l15 = channel 16 wv3 + l15]
            if (wv3 == 0) {          // channels 0..15: the skip gradient d e3
                const int qy = kq >> 1, qx = kq & 1;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    P.dE3[((size_t)img * 16 + (2 * qy + (j >> 1)) * 4 + 2 * qx + (j & 1)) * 16 + l15] = acc[j];
            } else {                  // channels 16..47: the upsampled o4 -- summed over the 4x4 map (per quad here, over the quads below)
                red[kq * 32 + 16 * (wv3 - 1) + l15] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            }
        }
        __syncthreads();
        TAIL_STAMP(5);
        if (tid < 32) P.d_o4[(size_t)img * 32 + tid] = (red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid]);
        __syncthreads();
        TAIL_STAMP(6);
    }
    const size_t b = blockIdx.x;
    // dec_model.1's slab [9*16*8 | 8]; the bias row is the sum of the lanes' pixel sums
    __syncthreads();
    wg1.reduce_store(P.slab1 ? P.slab1 + b * kTailSlabD1 : nullptr, sm, wave, lane, tid);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float v = wave_sum(bs1[c]);
        if (lane == 0) red[c * 4 + wave] = v;
    }
    __syncthreads();
    if (tid < 8 && P.slab1) P.slab1[b * kTailSlabD1 + 1152 + tid] = (red[tid * 4] + red[tid * 4 + 1]) + (red[tid * 4 + 2] + red[tid * 4 + 3]);
    if (P.slab2) wg2.store(P.slab2 + b * kTailSlabD2, wave, lane);
    if constexpr (W3) { if (P.slab3) wg3.store(P.slab3 + b * kTailSlabD3, wave, lane); }
    if (CGS_STAMP_PTR(P.dbg) && tid == 0) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 16 + 15] = __builtin_amdgcn_s_memtime();
}

extern "C" int cgs_tail_dec_bwd_slabs(int32_t n) { return n < 0 ? CGS_ERR_BADARG : tail_blocks(n, tail_bwd_cap()); }

extern "C" int cgs_tail_dec_bwd(int32_t n, const cgs_tail_dec_weights* w, const float* e1, const float* e2, const float* e3,
                                const float* o4, const float* o3, const float* o2, const float* do1, float* dE1, float* dE2,
                                float* dE3, float* d_o4, float* slab3, float* slab2, float* slab1, cgs_stream_t stream) {
    if (n < 0 || !w || !e1 || !e2 || !e3 || !o4 || !o3 || !o2 || !do1 || !dE1 || !dE2 || !dE3 || !d_o4) return CGS_ERR_BADARG;
    if (!w->w3 || !w->w2 || !w->w1) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_dec_bwd_kernel<false>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)TailDecBwdLds::BYTES);
    if (attr != hipSuccess) return (int)attr;
    TailDecBwdParams P{*w, e1, e2, e3, o4, o3, o2, do1, dE1, dE2, dE3, d_o4, slab3, slab2, slab1, n, tail_blocks(n, tail_bwd_cap()), g_tail_stamps ? g_tail_stamps + 3 * 2048 * 16 : nullptr};
    hipLaunchKernelGGL(tail_dec_bwd_kernel<false>, dim3(tail_blocks(n, tail_bwd_cap())), dim3(256), TailDecBwdLds::BYTES, (hipStream_t)stream, P, ConvParams{});
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// dec_model.0's data gradient (cgs_conv3x3_bwd_data of the 16 -> 8 layer at 32x32: dy = d o0 [n,32,32,8] -> d e0 [n,32,32,8] and the
// cell-summed d o1 [n,16,16,8]) AND cgs_tail_dec_bwd in one launch, one workgroup per image (n <= cgs_tail_dec_bwd_slabs' cap: the
// tail's slabs are one row per image).  w0: dec_model.0's HWIO weights.  CGS_ERR_UNSUPPORTED for larger n (the caller launches the two).
// do3 != NULL (round 5; then slab3 must be NULL): dec_model.3's weight gradient is left to the riders of cgs_enc0_wgrad_u8_with_head_riders --
// this launch writes d o3 [n,4,4,16] for them instead of one 27.7 KB slab row per image.
extern "C" int cgs_dec0_tail_dec_bwd_do3(int32_t n, const cgs_tail_dec_weights* w, const float* dy_o0, const float* w0, float* dE0, const float* e1,
                                         const float* e2, const float* e3, const float* o4, const float* o3, const float* o2, float* do1,
                                         float* dE1, float* dE2, float* dE3, float* d_o4, float* slab3, float* slab2, float* slab1,
                                         float* do3, cgs_stream_t stream) {
    if (n < 0 || !w || !dy_o0 || !w0 || !dE0 || !e1 || !e2 || !e3 || !o4 || !o3 || !o2 || !do1 || !dE1 || !dE2 || !dE3 || !d_o4) return CGS_ERR_BADARG;
    if (do3 && slab3) return CGS_ERR_BADARG;
    if (!w->w3 || !w->w2 || !w->w1) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    if (n > tail_bwd_cap()) return CGS_ERR_UNSUPPORTED;
    static_assert(TailDecBwdLds::BYTES >= 26 * 1024, "the convolution's tiles fit the tail's LDS block");
    if (conv_lds_bytes<DDec0P>() > TailDecBwdLds::BYTES) return CGS_ERR_UNSUPPORTED;
    static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_dec_bwd_kernel<true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)TailDecBwdLds::BYTES);
    if (attr != hipSuccess) return (int)attr;
    static hipError_t attr3 = hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_dec_bwd_kernel<true, false>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)TailDecBwdLds::BYTES);
    if (attr3 != hipSuccess) return (int)attr3;
    TailDecBwdParams P{*w, e1, e2, e3, o4, o3, o2, do1, dE1, dE2, dE3, d_o4, slab3, slab2, slab1, n, n, g_tail_stamps ? g_tail_stamps + 3 * 2048 * 16 : nullptr, do3};
    ConvParams PC{};
    PC.src_a = dy_o0; PC.w = w0; PC.out = dE0; PC.out2 = do1; PC.n = n;
    if (do3) hipLaunchKernelGGL((tail_dec_bwd_kernel<true, false>), dim3(n), dim3(256), TailDecBwdLds::BYTES, (hipStream_t)stream, P, PC);
    else hipLaunchKernelGGL((tail_dec_bwd_kernel<true, true>), dim3(n), dim3(256), TailDecBwdLds::BYTES, (hipStream_t)stream, P, PC);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_dec0_tail_dec_bwd(int32_t n, const cgs_tail_dec_weights* w, const float* dy_o0, const float* w0, float* dE0, const float* e1,
                                     const float* e2, const float* e3, const float* o4, const float* o3, const float* o2, float* do1,
                                     float* dE1, float* dE2, float* dE3, float* d_o4, float* slab3, float* slab2, float* slab1,
                                     cgs_stream_t stream) {
    return cgs_dec0_tail_dec_bwd_do3(n, w, dy_o0, w0, dE0, e1, e2, e3, o4, o3, o2, do1, dE1, dE2, dE3, d_o4, slab3, slab2, slab1, nullptr, stream);
}

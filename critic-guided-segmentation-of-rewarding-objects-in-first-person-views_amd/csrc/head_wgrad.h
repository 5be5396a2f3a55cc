// Weight gradients of the critic head as one small GEMM over the images (body shared by tail.hip's stand-alone launch and by the
// launch it shares with features.0's uint8 weight gradient, conv_wgrad.hip).
#pragma once
#include "cgs_common.h"

typedef float frag4 __attribute__((ext_vector_type(4)));
static constexpr int kTailHeadSlab = 8192 + 32 + 1024 + 32 + 32 + 1, kTailPwSlab = 1024 + 32;

struct HeadWgradRange { const float* hvec; const float* e4; const float* d_o4; int n, n_o4; };
struct HeadWgradParams { HeadWgradRange r[2]; float* slab_head; float* slab_pw; };
static constexpr int kHwIpb = 8;      // images per workgroup (8: 192 workgroups at 1536 images; 16 left most CUs idle)

// LDS of the body, carved from a caller-supplied block (round 5): as STATIC arrays inside a launch that also carries a role with dynamic LDS the
// two add up for every workgroup -- wgrad_enc0u8_head_kernel ran at 13.6 + 28.7 KB = three workgroups per CU instead of five.
static constexpr int kHwLdsFloats = kHwIpb * (260 + 32 + 32 + 32 + 36 + 32) + kHwIpb + 8;

__device__ __forceinline__ void tail_head_wgrad_body(const HeadWgradParams& P, const int bid, float* lds) {
    float (*xs)[260] = (float (*)[260])lds;                              // +4: conflict-free column reads
    float (*dz4)[32] = (float (*)[32])(lds + kHwIpb * 260);
    float (*dh1)[32] = (float (*)[32])(lds + kHwIpb * (260 + 32));
    float (*qv)[32] = (float (*)[32])(lds + kHwIpb * (260 + 64));
    float (*e4s)[36] = (float (*)[36])(lds + kHwIpb * (260 + 96));
    float (*do4)[32] = (float (*)[32])(lds + kHwIpb * (260 + 96 + 36));
    float* dz2s = lds + kHwIpb * (260 + 96 + 36 + 32);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int total = P.r[0].n + P.r[1].n, img0 = bid * kHwIpb;
    // ---- this workgroup's images -> LDS (zeros beyond the end) ----
    for (int e = tid; e < kHwIpb * 96; e += 256) {       // 96 float4 of hvec per image
        const int il = e / 96, q = e % 96, g = img0 + il;
        float4 v = f4zero();
        if (g < total) {
            const HeadWgradRange& R = g < P.r[0].n ? P.r[0] : P.r[1];
            const int i = g < P.r[0].n ? g : g - P.r[0].n;
            if (q < 88 || q == 88) v = ((const float4*)(R.hvec + (size_t)i * 384))[q];
        }
        if (q < 64) *(float4*)&xs[il][4 * q] = v;
        else if (q < 72) *(float4*)&dz4[il][4 * (q - 64)] = v;
        else if (q < 80) *(float4*)&dh1[il][4 * (q - 72)] = v;
        else if (q < 88) *(float4*)&qv[il][4 * (q - 80)] = v;
        else if (q == 88) dz2s[il] = v.x;
    }
    for (int e = tid; e < kHwIpb * 16; e += 256) {        // e4 (8 float4) and d_o4 (8 float4) per image
        const int il = e / 16, q = e % 16, g = img0 + il;
        float4 v = f4zero();
        if (g < total) {
            const HeadWgradRange& R = g < P.r[0].n ? P.r[0] : P.r[1];
            const int i = g < P.r[0].n ? g : g - P.r[0].n;
            if (q < 8) v = ((const float4*)(R.e4 + (size_t)i * 32))[q];
            else if (R.d_o4 && i < R.n_o4) v = ((const float4*)(R.d_o4 + (size_t)i * 32))[q - 8];
        }
        if (q < 8) *(float4*)&e4s[il][4 * q] = v; else *(float4*)&do4[il][4 * (q - 8)] = v;
    }
    __syncthreads();
    float* sh = P.slab_head + (size_t)bid * kTailHeadSlab;
    float* sp = P.slab_pw ? P.slab_pw + (size_t)bid * kTailPwSlab : nullptr;
    // ---- 40 output tiles of 16 x 16 over 4 waves: 32 of dW14, 4 of dWl1, 4 of dWpw; K = 16 images = 4 k-steps ----
    for (int t = wave; t < 40; t += 4) {
        const bool w14 = t < 32, wl1 = t >= 32 && t < 36;
        const int tt = w14 ? t : (wl1 ? t - 32 : t - 36);
        const int mb = tt >> 1, nb = tt & 1;            // row block (k), column block (o)
        frag4 acc = frag4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < kHwIpb / 4; ++s4) {
            const int il = 4 * s4 + kq;
            const float a = w14 ? xs[il][16 * mb + l15] : e4s[il][16 * mb + l15];
            const float b = w14 ? dz4[il][16 * nb + l15] : (wl1 ? dh1[il][16 * nb + l15] : do4[il][16 * nb + l15]);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
        float* dst = w14 ? sh : (wl1 ? sh + 8192 + 32 : sp);
        if (dst) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[(16 * mb + 4 * kq + j) * 32 + 16 * nb + l15] = acc[j];
        }
    }
    // ---- bias-like sums ----
    if (tid < 32) {
        float s4 = 0.f, s1 = 0.f, sq = 0.f, so = 0.f;
#pragma unroll
        for (int il = 0; il < kHwIpb; ++il) { s4 += dz4[il][tid]; s1 += dh1[il][tid]; sq += qv[il][tid]; so += do4[il][tid]; }
        sh[8192 + tid] = s4;
        sh[8192 + 32 + 1024 + tid] = s1;
        sh[8192 + 32 + 1024 + 32 + tid] = sq;
        if (sp) sp[1024 + tid] = so;
        if (tid == 0) {
            float sz = 0.f;
#pragma unroll
            for (int il = 0; il < kHwIpb; ++il) sz += dz2s[il];
            sh[8192 + 32 + 1024 + 32 + 32] = sz;
        }
    }
}


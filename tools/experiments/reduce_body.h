// Slab reduction -> gradient -> Adam update of one 32-column block of one reduction job: the body shared by reduce_adam_kernel
// (elementwise.hip: the step's last launch) and by the "early" rider workgroups of the launch before it (conv_wgrad.hip:
// cgs_enc0_wgrad_u8_with_head_reduce), which reduce + update every parameter whose slabs are already complete while the
// latency-bound last weight-gradient launch runs.
#pragma once
#include "cgs_common.h"
#include "cgs_hip.h"

struct AdamArgs {
    float* param; const float* grad_base; float* m; float* v;
    float lr, b1, b2, eps;
    unsigned int* ticket;
};

// Adam's bias corrections for t = s_old + 1: 1 - b^t = -expm1(t ln b) (no cancellation at small t, no double-precision pow)
__device__ __forceinline__ void adam_bias_corrections(const AdamArgs& A, uint64_t s_old, float* bc) {
    const float t = (float)(s_old + 1);
    bc[0] = -expm1f(t * logf(A.b1));
    bc[1] = sqrtf(-expm1f(t * logf(A.b2)));
}

// SL slab lanes x 32 columns (threadIdx.x = sl * 32 + col); red = float[SL][33], bc = float[2] in LDS (bc written by thread 0 before
// the call's internal barrier).  i0 = first column of the block inside the job.
template <int SL>
__device__ __forceinline__ void reduce_adam_block(const cgs_reduce_job& j, const int i0, const AdamArgs& A, float (*red)[33], const float* bc) {
    const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = i0 + col;
    float s[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) s[u] = 0.f;
    if (i < j.count) {
        const float* p = j.slab + i;
        int b = sl;
        for (; b + 15 * SL < j.nslab; b += 16 * SL) {
#pragma unroll
            for (int u = 0; u < 16; ++u) s[u] += p[(size_t)(b + u * SL) * j.stride];
        }
        for (; b + 3 * SL < j.nslab; b += 4 * SL) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += p[(size_t)(b + u * SL) * j.stride];
        }
        for (; b < j.nslab; b += SL) s[0] += p[(size_t)b * j.stride];
    }
    red[sl][col] = (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]))) +
                   (((s[8] + s[9]) + (s[10] + s[11])) + ((s[12] + s[13]) + (s[14] + s[15])));
    __syncthreads();
    if (sl == 0 && i < j.count) {
        float g = 0.f;
#pragma unroll
        for (int k = 0; k < SL; ++k) g += red[k][col];
        j.dst[i] = g;                                          // the gradient stays observable (tests, DP)
        if (A.param) {      // (NULL: data parallel -- the all-reduce of the gradient comes first, Adam is a launch of its own)
            const size_t e = (size_t)(j.dst + i - A.grad_base);      // element of the flat buffers
            const float c1 = bc[0], c2s = bc[1];
            const float mi = A.b1 * A.m[e] + (1.f - A.b1) * g;
            const float vi = A.b2 * A.v[e] + (1.f - A.b2) * g * g;
            A.m[e] = mi;
            A.v[e] = vi;
            A.param[e] -= (A.lr / c1) * (mi / (sqrtf(vi) / c2s + A.eps));
        }
    }
}

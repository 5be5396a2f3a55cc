// Bodies of the critic-head and 1x1-bottleneck kernels.  Included by head.hip and fused_small.hip.
#pragma once
#include "cgs_common.h"

typedef float frag16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float drop_mult1(const DropCtx& dc, uint32_t i) {
    return dc.on ? f4get(drop_mult4(dc, i >> 2), i & 3) : 1.f;
}

// ------------------------------------------------------------------------------------------------
// head forward: 8 images x 32 outputs per workgroup
// ------------------------------------------------------------------------------------------------
static constexpr int HEAD_FWD_LDS = (8 * 256 + 256 * 32 + 32 * 32 + 8 * 32 + 8 * 33 + 32 * 32) * 4;   // bytes

// 8 images (img0 .. img0+7) x 32 outputs by 256 threads; lds = HEAD_FWD_LDS bytes, 16-byte aligned
__device__ __forceinline__ void head_fwd_body(int n, int img0, const float* __restrict__ e3, const float* __restrict__ w4,
                                              const float* __restrict__ b4, const float* __restrict__ w1,
                                              const float* __restrict__ b1, const float* __restrict__ w2,
                                              const float* __restrict__ b2, const cgs_dropout& drop_in,
                                              const cgs_dropout& drop_h, float* __restrict__ e4,
                                              float* __restrict__ h1, float* __restrict__ pred,
                                              const float* __restrict__ wpw, const float* __restrict__ bpw,
                                              float* __restrict__ o4, float* lds) {
    float (*xs)[256] = (float (*)[256])lds;
    float* w4s = lds + 8 * 256;                 // 32 KB: the 4x4-conv weights, read 8x per block
    float* w1s = w4s + 256 * 32;
    float (*es)[32] = (float (*)[32])(w1s + 32 * 32);
    float (*hs)[33] = (float (*)[33])(w1s + 32 * 32 + 8 * 32);
    float* wps = w1s + 32 * 32 + 8 * 32 + 8 * 33;   // optional: the decoder's 1x1 bottleneck conv (dec_model.4), k-major
    const int tid = threadIdx.x, il = tid >> 5, o = tid & 31;
    const int nn = img0 + il;
    const DropCtx di = drop_ctx(drop_in), dh = drop_ctx(drop_h);
#pragma unroll
    for (int i = 0; i < 8; ++i) ((float4*)w4s)[tid + i * 256] = ((const float4*)w4)[tid + i * 256];
    ((float4*)w1s)[tid] = ((const float4*)w1)[tid];
    if (o4) ((float4*)wps)[tid] = ((const float4*)wpw)[tid];
    for (int e = tid; e < 8 * 64; e += 256) {
        int img = e >> 6, q = e & 63, m = img0 + img;
        float4 v = f4zero();
        if (m < n) {
            int gi = m * 64 + q;
            v = ((const float4*)e3)[gi];
            if (di.on) v = v * drop_mult4(di, (uint32_t)gi);
        }
        ((float4*)&xs[img][0])[q] = v;
    }
    __syncthreads();
    float acc0 = b4[o], acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll 8
    for (int k = 0; k < 256; k += 4) {
        float4 xv = *(const float4*)&xs[il][k];
        acc0 = fmaf(xv.x, w4s[k * 32 + o], acc0);
        acc1 = fmaf(xv.y, w4s[(k + 1) * 32 + o], acc1);
        acc2 = fmaf(xv.z, w4s[(k + 2) * 32 + o], acc2);
        acc3 = fmaf(xv.w, w4s[(k + 3) * 32 + o], acc3);
    }
    float acc = (acc0 + acc1) + (acc2 + acc3);
    float e = acc > 0.f ? acc : 0.f;
    es[il][o] = e;
    if (nn < n) e4[nn * 32 + o] = e;
    __syncthreads();
    if (o4) {   // decoder bottleneck o4 = W_pw e4 + b_pw (nets.py:501) from the e4 row already in LDS
        float a = bpw[o];
#pragma unroll
        for (int k = 0; k < 32; ++k) a = fmaf(es[il][k], wps[k * 32 + o], a);
        if (nn < n) o4[nn * 32 + o] = a;
    }
    acc = b1[o];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc = fmaf(es[il][k], w1s[k * 32 + o], acc);
    float h = acc > 0.f ? acc : 0.f;
    if (nn < n) h1[nn * 32 + o] = h;
    float hd = h * drop_mult1(dh, (uint32_t)(nn * 32 + o));
    hs[il][o] = hd * w2[o];
    __syncthreads();
    if (o == 0 && nn < n) {
        float s = b2[0];
#pragma unroll
        for (int k = 0; k < 32; ++k) s += hs[il][k];
        pred[nn] = 1.f / (1.f + expf(-s));
    }
}

__global__ void __launch_bounds__(256) head_fwd_kernel(int n, const float* e3, const float* w4, const float* b4,
                                                       const float* w1, const float* b1, const float* w2, const float* b2,
                                                       cgs_dropout drop_in, cgs_dropout drop_h, float* e4, float* h1,
                                                       float* pred, const float* wpw, const float* bpw, float* o4) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    head_fwd_body(n, blockIdx.x * 8, e3, w4, b4, w1, b1, w2, b2, drop_in, drop_h, e4, h1, pred, wpw, bpw, o4, (float*)smem);
}

// ------------------------------------------------------------------------------------------------
// head backward: each workgroup walks HB_IPB images in chunks of 8 and keeps its weight-gradient
// partials in registers; one slab per workgroup: [w4 8192 | b4 32 | w1 1024 | b1 32 | w2 32 | b2 1].
// ------------------------------------------------------------------------------------------------
static constexpr int HB_IPB = 8;
static constexpr int HB_SLAB = 8192 + 32 + 1024 + 32 + 32 + 1;
static constexpr int HEAD_BWD_LDS = (8 * 256 + 3 * 8 * 32 + 5 * 8 * 32 + 256 * 33 + 32 * 33 + 32 * 33 + 8 * 32) * 4;   // bytes

// HB_IPB images starting at `base`; writes ONE slab at `sl`
__device__ __forceinline__ void head_bwd_body(int n, int base, const float* __restrict__ e3, const float* __restrict__ e4,
                                              const float* __restrict__ h1, const float* __restrict__ pred,
                                              const float* __restrict__ dpred, const float* __restrict__ d_e4_extra,
                                              const float* d_e3_extra, int n_extra, const float* __restrict__ w4,
                                              const float* __restrict__ w1, const float* __restrict__ w2,
                                              const cgs_dropout& drop_in, const cgs_dropout& drop_h,
                                              float* d_e3, float* __restrict__ sl, const float* __restrict__ d_o4,
                                              const float* __restrict__ wpw, float* __restrict__ slpw, float* lds) {
    float (*xs)[256] = (float (*)[256])lds;
    float (*es)[32] = (float (*)[32])(lds + 8 * 256);
    float (*dh1s)[32] = (float (*)[32])(lds + 8 * 256 + 8 * 32);
    float (*dz4s)[32] = (float (*)[32])(lds + 8 * 256 + 2 * 8 * 32);
    float (*red)[8][32] = (float (*)[8][32])(lds + 8 * 256 + 3 * 8 * 32);
    float* w4s = lds + 8 * 256 + 3 * 8 * 32 + 5 * 8 * 32;        // rows padded to 33 floats: row-per-lane reads
    float* w1s = w4s + 256 * 33;
    float* wps = w1s + 32 * 33;                                    // optional dec_model.4 weights, rows padded to 33
    float (*do4s)[32] = (float (*)[32])(wps + 32 * 33);
    const int tid = threadIdx.x, il = tid >> 5, o = tid & 31, kg = il;
    const DropCtx di = drop_ctx(drop_in), dh = drop_ctx(drop_h);
    // all weight loads of a thread are independent 16-byte loads issued back to back (one memory latency, not 36)
    {
        float4 v4[8], v1 = ((const float4*)w1)[tid];
#pragma unroll
        for (int i = 0; i < 8; ++i) v4[i] = ((const float4*)w4)[tid + i * 256];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + i * 256;
            float* d = w4s + (idx >> 3) * 33 + (idx & 7) * 4;
            d[0] = v4[i].x; d[1] = v4[i].y; d[2] = v4[i].z; d[3] = v4[i].w;
        }
        float* d = w1s + (tid >> 3) * 33 + (tid & 7) * 4;
        d[0] = v1.x; d[1] = v1.y; d[2] = v1.z; d[3] = v1.w;
        if (d_o4) {
            float4 vp = ((const float4*)wpw)[tid];
            float* dp = wps + (tid >> 3) * 33 + (tid & 7) * 4;
            dp[0] = vp.x; dp[1] = vp.y; dp[2] = vp.z; dp[3] = vp.w;
        }
    }
    float accpw[4] = {0.f, 0.f, 0.f, 0.f}, pbpw = 0.f;             // dec_model.4 weight / bias gradient partials
    float acc4[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc4[j] = 0.f;
    float accw1[4] = {0.f, 0.f, 0.f, 0.f};
    float pb4 = 0.f, pb1 = 0.f, pw2 = 0.f, pb2 = 0.f;
    for (int ch = 0; ch < HB_IPB / 8; ++ch) {
        const int img0 = base + ch * 8, nn = img0 + il;
        for (int e = tid; e < 8 * 64; e += 256) {
            int img = e >> 6, q = e & 63, m = img0 + img;
            float4 v = f4zero();
            if (m < n) {
                int gi = m * 64 + q;
                v = ((const float4*)e3)[gi];
                if (di.on) v = v * drop_mult4(di, (uint32_t)gi);
            }
            ((float4*)&xs[img][0])[q] = v;
        }
        float ev = 0.f, hv = 0.f, dz2 = 0.f;
        if (nn < n) {
            ev = e4[nn * 32 + o];
            hv = h1[nn * 32 + o];
            float p = pred[nn];
            dz2 = dpred[nn] * p * (1.f - p);
        }
        es[il][o] = ev;
        if (d_o4) {
            float g = (nn < n_extra) ? d_o4[nn * 32 + o] : 0.f;   // the decoder saw the first n_extra images only
            do4s[il][o] = g;
            pbpw += g;
        }
        float m2 = drop_mult1(dh, (uint32_t)(nn * 32 + o));
        pw2 = fmaf(dz2, hv * m2, pw2);
        if (o == 0) pb2 += dz2;
        float dh1 = (hv > 0.f) ? dz2 * w2[o] * m2 : 0.f;
        dh1s[il][o] = dh1;
        pb1 += dh1;
        __syncthreads();
        // dW1[k][o], k = kg*4 + j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s = fmaf(es[i][kg * 4 + j], dh1s[i][o], s);
            accw1[j] += s;
        }
        // d e4[il][k = o] = sum_o' w1[k][o'] dh1[il][o'] (+ decoder gradient), through ReLU
        float de = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) de = fmaf(w1s[o * 33 + q], dh1s[il][q], de);
        if (d_e4_extra && nn < n_extra) de += d_e4_extra[nn * 32 + o];
        if (d_o4) {   // decoder bottleneck backward: d e4[k = o] += sum_j W_pw[k][j] d o4[j];  dW_pw[k][j] += e4[k] d o4[j]
#pragma unroll
            for (int q = 0; q < 32; ++q) de = fmaf(wps[o * 33 + q], do4s[il][q], de);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s = fmaf(es[i][kg * 4 + j], do4s[i][o], s);
                accpw[j] += s;
            }
        }
        float dz4 = (ev > 0.f) ? de : 0.f;
        dz4s[il][o] = dz4;
        pb4 += dz4;
        __syncthreads();
        // dW4[k][o], k = kg*32 + j
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s = fmaf(xs[i][kg * 32 + j], dz4s[i][o], s);
            acc4[j] += s;
        }
        // d e3[il][k] = sum_o w4[k][o] dz4[il][o], k = o + 32 m  (dropout mask of the input applied)
        if (nn < n) {
#pragma unroll 2
            for (int m = 0; m < 8; ++m) {
                int k = o + 32 * m;
                const float* wr = w4s + k * 33;
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 32; ++q) s = fmaf(wr[q], dz4s[il][q], s);
                int gi = nn * 256 + k;
                float r = s * drop_mult1(di, (uint32_t)gi);
                if (d_e3_extra && nn < n_extra) r += d_e3_extra[gi];
                d_e3[gi] = r;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) sl[(kg * 32 + j) * 32 + o] = acc4[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) sl[8192 + 32 + (kg * 4 + j) * 32 + o] = accw1[j];
    if (d_o4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) slpw[(kg * 4 + j) * 32 + o] = accpw[j];
    }
    red[0][il][o] = pb4; red[1][il][o] = pb1; red[2][il][o] = pw2; red[3][il][o] = pb2; red[4][il][o] = pbpw;
    __syncthreads();
    if (il == 0) {
        float s4 = 0.f, s1 = 0.f, s2 = 0.f, sb = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { s4 += red[0][i][o]; s1 += red[1][i][o]; s2 += red[2][i][o]; sb += red[3][i][o]; }
        sl[8192 + o] = s4;
        sl[8192 + 32 + 1024 + o] = s1;
        sl[8192 + 32 + 1024 + 32 + o] = s2;
        if (o == 0) sl[8192 + 32 + 1024 + 32 + 32] = sb;
        if (d_o4) {   // bias gradient of dec_model.4
            float sp = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sp += red[4][i][o];
            slpw[1024 + o] = sp;
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) head_bwd_kernel(int n, const float* e3, const float* e4, const float* h1,
                                                       const float* pred, const float* dpred, const float* d_e4_extra,
                                                       const float* d_e3_extra, int n_extra, const float* w4,
                                                       const float* w1, const float* w2, cgs_dropout drop_in,
                                                       cgs_dropout drop_h, float* d_e3, float* slab, const float* d_o4,
                                                       const float* wpw, float* slab_pw) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    head_bwd_body(n, blockIdx.x * HB_IPB, e3, e4, h1, pred, dpred, d_e4_extra, d_e3_extra, n_extra, w4, w1, w2, drop_in,
                  drop_h, d_e3, slab + (size_t)blockIdx.x * HB_SLAB, d_o4, wpw,
                  slab_pw ? slab_pw + (size_t)blockIdx.x * (32 * 32 + 32) : nullptr, (float*)smem);
}

// ------------------------------------------------------------------------------------------------
// 32 -> 32 pointwise conv on the bottleneck: v_mfma_f32_32x32x2_f32, one wave = 32 images.
//   A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31];  D: col = lane&31,
//   row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int d32_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// stage a [32][32] fp32 tile (rows of 32 contiguous floats, `rows_valid` of them real) into LDS with coalesced 16-B loads
__device__ __forceinline__ void pw_stage(float (*dst)[33], const float* __restrict__ src, int rows_valid, int lane) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        int idx = lane + 64 * it, row = idx >> 3, c4 = idx & 7;
        float4 v = row < rows_valid ? ((const float4*)src)[row * 8 + c4] : f4zero();
        dst[row][c4 * 4] = v.x; dst[row][c4 * 4 + 1] = v.y; dst[row][c4 * 4 + 2] = v.z; dst[row][c4 * 4 + 3] = v.w;
    }
}

__global__ void __launch_bounds__(64) pointwise32_fwd_kernel(int n, const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y) {
    __shared__ float xs[32][33], ws[32][33];
    const int lane = threadIdx.x, i = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.x * 32;
    pw_stage(xs, x + (size_t)n0 * 32, n - n0 < 32 ? n - n0 : 32, lane);
    pw_stage(ws, w, 32, lane);
    __syncthreads();
    frag16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        int k = 2 * s + half;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[i][k], ws[k][i], acc, 0, 0, 0);
    }
    float bias = b[i];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int m = n0 + d32_row(r, half);
        if (m < n) y[m * 32 + i] = acc[r] + bias;
    }
}

static constexpr int PW_SLAB = 32 * 32 + 32;

__global__ void __launch_bounds__(64) pointwise32_bwd_kernel(int n, const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ w, float* __restrict__ dx,
                                                            float* __restrict__ slab) {
    __shared__ float xs[32][33], ds[32][33], ws[32][33];
    const int lane = threadIdx.x, i = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int rows = n - n0 < 32 ? n - n0 : 32;
    pw_stage(xs, x + (size_t)n0 * 32, rows, lane);
    pw_stage(ds, dy + (size_t)n0 * 32, rows, lane);
    pw_stage(ws, w, 32, lane);
    __syncthreads();
    // dx[img][k] = sum_o dy[img][o] w[k][o]
    frag16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        int oo = 2 * s + half;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[i][oo], ws[i][oo], acc, 0, 0, 0);
    }
    if (dx) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int m = n0 + d32_row(r, half);
            if (m < n) dx[m * 32 + i] = acc[r];
        }
    }
    // dW[k][o] = sum_img x[img][k] dy[img][o]  (rows = k, reduction over the 32 images of this wave)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        int m = 2 * s + half;
        float bb = ds[m][i];
        bsum += bb;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[m][i], bb, acc, 0, 0, 0);
    }
    float* sl = slab + (size_t)blockIdx.x * PW_SLAB;
#pragma unroll
    for (int r = 0; r < 16; ++r) sl[d32_row(r, half) * 32 + i] = acc[r];
    bsum += __shfl_xor(bsum, 32, 64);
    if (half == 0) sl[1024 + i] = bsum;
}


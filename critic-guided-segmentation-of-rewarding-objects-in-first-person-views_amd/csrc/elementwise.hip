// Memory-bound glue of the training step: mask replace/inject mix (main.py:395,406) and its backward
// with the mask regularisers fused, the four losses, slab reduction (+ per-iteration tick), flat Adam,
// NCHW<->NHWC boundary conversion and the dropout-mask test hook.
#include "cgs_common.h"

__device__ __forceinline__ void unpack12(uint32_t d0, uint32_t d1, uint32_t d2, float (&v)[12]) {
    const float s = 1.f / 255.f;
    v[0] = (d0 & 255) * s; v[1] = ((d0 >> 8) & 255) * s; v[2] = ((d0 >> 16) & 255) * s; v[3] = (d0 >> 24) * s;
    v[4] = (d1 & 255) * s; v[5] = ((d1 >> 8) & 255) * s; v[6] = ((d1 >> 16) & 255) * s; v[7] = (d1 >> 24) * s;
    v[8] = (d2 & 255) * s; v[9] = ((d2 >> 8) & 255) * s; v[10] = ((d2 >> 16) & 255) * s; v[11] = (d2 >> 24) * s;
}

// one thread = 4 pixels (12 bytes of A and of B, one float4 of Z, 3 float4 of each mix); grid-stride.
// Each workgroup writes its partial (sum |Z|, sum Z^2) to zpart[2*block .. 2*block+1]: no float atomics.
__global__ void __launch_bounds__(256) mix_fwd_kernel(int groups, int rep_groups, const uint32_t* __restrict__ a,
                                                      const uint32_t* __restrict__ b, const float4* __restrict__ z,
                                                      int inject, float4* __restrict__ mixed, float* __restrict__ zpart) {
    __shared__ float red[2][4];
    float s1 = 0.f, s2 = 0.f;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < groups; g += gridDim.x * 256) {
        float4 zz = z[g];
        float zv[4] = {zz.x, zz.y, zz.z, zz.w};
        if (!mixed) {       // partial sums only: the consumers form the mixes themselves (cgs_bf16_enc0_fwd_mix / cgs_bf16_hwgrad_pooled_mix)
#pragma unroll
            for (int i = 0; i < 4; ++i) { s1 += fabsf(zv[i]); s2 += zv[i] * zv[i]; }
            continue;
        }
        float av[12], bv[12];
        unpack12(a[3 * g], a[3 * g + 1], a[3 * g + 2], av);
        unpack12(b[3 * g], b[3 * g + 1], b[3 * g + 2], bv);
        float r[12], q[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            float zi = zv[i / 3];
            r[i] = av[i] * (1.f - zi) + zi * bv[i];
            q[i] = bv[i] * (1.f - zi) + zi * av[i];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) mixed[3 * g + j] = make_float4(r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]);
        if (inject) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                mixed[3 * (rep_groups + g) + j] = make_float4(q[4 * j], q[4 * j + 1], q[4 * j + 2], q[4 * j + 3]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { s1 += fabsf(zv[i]); s2 += zv[i] * zv[i]; }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        zpart[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        zpart[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ void __launch_bounds__(256) mix_bwd_kernel(int groups, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                      const float4* __restrict__ z, const float4* __restrict__ dmixed,
                                                      int inject, float l1s, float l2s, float4* __restrict__ dzpre,
                                                      const float* __restrict__ vf_pred, int groups_per_img) {
    int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    // -staticnorm '' (main.py:415-418): the regulariser of A-image i is weighted by valuefak = 1 - pred[i] (L1) and its square (L2)
    const float vf = vf_pred ? 1.f - vf_pred[g / groups_per_img] : 1.f;
    float av[12], bv[12];
    unpack12(a[3 * g], a[3 * g + 1], a[3 * g + 2], av);
    unpack12(b[3 * g], b[3 * g + 1], b[3 * g + 2], bv);
    float dr[12], di[12];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float4 v = dmixed[3 * g + j];
        dr[4 * j] = v.x; dr[4 * j + 1] = v.y; dr[4 * j + 2] = v.z; dr[4 * j + 3] = v.w;
    }
    if (inject) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float4 v = dmixed[3 * (groups + g) + j];
            di[4 * j] = v.x; di[4 * j + 1] = v.y; di[4 * j + 2] = v.z; di[4 * j + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) di[i] = 0.f;
    }
    float4 zz = z[g];
    float zv[4] = {zz.x, zz.y, zz.z, zz.w}, o[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) d = fmaf(bv[3 * p + c] - av[3 * p + c], dr[3 * p + c] - di[3 * p + c], d);
        float zi = zv[p];
        float sg = zi > 0.f ? 1.f : (zi < 0.f ? -1.f : 0.f);
        d += l1s * vf * sg + 2.f * l2s * vf * vf * zi;
        o[p] = d * zi * (1.f - zi);
    }
    dzpre[g] = make_float4(o[0], o[1], o[2], o[3]);
}

// single workgroup; pred slots [B | A | replaced | injected]
__global__ void __launch_bounds__(256) phase2_losses_kernel(int n, const float* __restrict__ pred, const float* __restrict__ y,
                                                            const float* __restrict__ zpart, int nzpart, float lfak,
                                                            float l1, float l2, int flags, float inv_nz,
                                                            float* __restrict__ losses, float* __restrict__ dpred) {
    __shared__ float red[5][4];
    const bool live = flags & 1, inject = flags & 2, bce = flags & 4;
    const float inv_n = 1.f / (float)n;
    float sc = 0.f, sr = 0.f, si = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        float pb = pred[i], pa = pred[n + i], pr = pred[2 * n + i], pi = inject ? pred[3 * n + i] : 0.f;
        float yi = y[i];
        float da = 0.f;
        if (live) {
            if (bce) {
                float lp = fmaxf(logf(pa), -100.f), lq = fmaxf(logf(1.f - pa), -100.f);
                sc += -(yi * lp + (1.f - yi) * lq);
                da = lfak * (pa - yi) / fmaxf((1.f - pa) * pa, 1e-12f) * inv_n;
            } else {
                float d = pa - yi;
                sc += d * d;
                da = lfak * 2.f * d * inv_n;
            }
        }
        float drp = pr - pb;
        sr += drp * drp;
        dpred[i] = 0.f;
        dpred[n + i] = da;
        dpred[2 * n + i] = 2.f * drp * inv_n;
        if (inject) {
            float dip = pi - pa;
            si += dip * dip;
            dpred[3 * n + i] = 2.f * dip * inv_n;
        }
    }
    float z1 = 0.f, z2 = 0.f;
    const int per_img = nzpart / n;      // flag bit 8 (-staticnorm ''): image i's partial sums weighted by 1 - pred_A[i] (and its square)
    for (int i = threadIdx.x; i < nzpart; i += 256) {
        const float vf = (flags & 8) ? 1.f - pred[n + i / per_img] : 1.f;
        z1 += vf * zpart[2 * i]; z2 += vf * vf * zpart[2 * i + 1];
    }
    sc = wave_sum(sc); sr = wave_sum(sr); si = wave_sum(si); z1 = wave_sum(z1); z2 = wave_sum(z2);
    if ((threadIdx.x & 63) == 0) {
        int w = threadIdx.x >> 6;
        red[0][w] = sc; red[1][w] = sr; red[2][w] = si; red[3][w] = z1; red[4][w] = z2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float c = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) * inv_n;
        float r = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) * inv_n;
        float i = (red[2][0] + red[2][1] + red[2][2] + red[2][3]) * inv_n;
        float zs1 = (red[3][0] + red[3][1]) + (red[3][2] + red[3][3]), zs2 = (red[4][0] + red[4][1]) + (red[4][2] + red[4][3]);
        float n1 = l1 * zs1 * inv_nz, n2 = l2 * zs2 * inv_nz;
        losses[0] = c; losses[1] = r; losses[2] = i; losses[3] = n1; losses[4] = n2;
        losses[5] = (live ? lfak * c : 0.f) + r + i + n1 + n2;
        losses[6] = 0.f; losses[7] = 0.f;
    }
}

__global__ void __launch_bounds__(256) phase1_loss_kernel(int n, const float* __restrict__ pred, const float* __restrict__ y,
                                                          int bce, float* __restrict__ losses, float* __restrict__ dpred) {
    __shared__ float red[4];
    const float inv_n = 1.f / (float)n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        float p = pred[i], yi = y[i];
        if (bce) {
            float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
            s += -(yi * lp + (1.f - yi) * lq);
            dpred[i] = (p - yi) / fmaxf((1.f - p) * p, 1e-12f) * inv_n;
        } else {
            float d = p - yi;
            s += d * d;
            dpred[i] = 2.f * d * inv_n;
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) losses[0] = (red[0] + red[1] + red[2] + red[3]) * inv_n;
}

// grid = (ceil(max_count/128), njobs); block = 32 column lanes x 32 slab lanes, up to 16 independent partial sums per column in flight.  A column lane
// covers FOUR consecutive columns (one 16-byte load per slab row: 512 bytes per row and wave-half) when the job's rows are 16-byte aligned (round 6: the
// shape-generic layers' 130 MB of slab rows went through 128-byte row segments at 1.6 TB/s), else -- or for the last columns of a row -- one column per
// pass, four passes.  Every column's sum is formed in the same order either way (bitwise equal to the one-column form).  The longest job (2048 slabs)
// sets the kernel's duration: 64 rows per thread = 4 rounds of loads.
template <int V>
__device__ __forceinline__ void reduce_slabs_cols(const cgs_reduce_job& j, int i0, int sl, float (&out)[V]) {
    constexpr int SL = 32;
    typedef float vec_t __attribute__((ext_vector_type(V)));
    vec_t s[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) s[u] = vec_t(0.f);
    const float* p = j.slab + i0;
    auto ld = [&](int row) { return *(const vec_t*)(p + (size_t)row * j.stride); };
    int b = sl;
    for (; b + 15 * SL < j.nslab; b += 16 * SL) {
#pragma unroll
        for (int u = 0; u < 16; ++u) s[u] += ld(b + u * SL);
    }
    for (; b + 3 * SL < j.nslab; b += 4 * SL) {
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] += ld(b + u * SL);
    }
    for (; b < j.nslab; b += SL) s[0] += ld(b);
    const vec_t t = (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]))) +
                    (((s[8] + s[9]) + (s[10] + s[11])) + ((s[12] + s[13]) + (s[14] + s[15])));
#pragma unroll
    for (int c = 0; c < V; ++c) out[c] = t[c];
}

__global__ void __launch_bounds__(1024) reduce_slabs_kernel(const cgs_reduce_job* __restrict__ jobs, uint64_t* step) {
    constexpr int SL = 32;
    __shared__ float red[4][SL][33];
    const cgs_reduce_job j = jobs[blockIdx.y];
    if (step && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *step += 1ull;
    if (blockIdx.x * 128 >= j.count) return;
    const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const bool v4 = (((size_t)j.slab & 15) == 0) && (j.stride & 3) == 0;          // (uniform per workgroup)
    const int i4 = blockIdx.x * 128 + 4 * col;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (v4 && i4 + 3 < j.count) {
        reduce_slabs_cols<4>(j, i4, sl, v);
    } else {
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            float one[1] = {0.f};
            if (i4 + c < j.count) reduce_slabs_cols<1>(j, i4 + c, sl, one);
            v[c] = one[0];
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) red[c][sl][col] = v[c];
    __syncthreads();
    if (sl < 4) {                      // slab lane c sums column 4 col + c over the 32 slab lanes
        const int c = sl, i = i4 + c;
        if (i < j.count) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < SL; ++k) t += red[c][k][col];
            j.dst[i] = j.accumulate ? j.dst[i] + t : t;
        }
    }
}

__global__ void __launch_bounds__(256) adam_kernel(long count, float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   const uint64_t* __restrict__ step, float lr, float b1, float b2, float eps, float gscale) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const double t = (double)(*step);
    const float c1 = (float)(1.0 - pow((double)b1, t));
    const float c2s = (float)sqrt(1.0 - pow((double)b2, t));
    float gi = g[i] * gscale;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float denom = sqrtf(vi) / c2s + eps;
    p[i] -= (lr / c1) * (mi / denom);
}

__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(long total, int c, int hw, const float* __restrict__ src, float* __restrict__ dst) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;  // index into dst (NHWC)
    if (i >= total) return;
    int ch = i % c;
    long pix = i / c;
    long n = pix / hw, s = pix % hw;
    dst[i] = src[(n * c + ch) * hw + s];
}

__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(long total, int c, int hw, const float* __restrict__ src, float* __restrict__ dst) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;  // index into dst (NCHW)
    if (i >= total) return;
    long s = i % hw;
    long nc = i / hw;
    long n = nc / c;
    int ch = nc % c;
    dst[i] = src[(n * hw + s) * c + ch];
}

__global__ void __launch_bounds__(256) dropout_mask_kernel(long count4, cgs_dropout d, float4* __restrict__ out) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count4) return;
    DropCtx dc = drop_ctx(d);
    out[i] = dc.on ? drop_mult4(dc, (uint32_t)i) : make_float4(1.f, 1.f, 1.f, 1.f);
}

// ------------------------------------------------------------------------------------------------
static int mix_blocks(int groups) { int b = (groups + 255) / 256; return b < 1024 ? b : 1024; }

extern "C" int cgs_mix_fwd_partials(int32_t n, int32_t hw) {
    if (n < 0 || hw <= 0 || (hw & 3)) return CGS_ERR_BADARG;
    return mix_blocks(n * (hw / 4));
}

extern "C" int cgs_mix_fwd(int32_t n, int32_t hw, const uint8_t* a, const uint8_t* b, const float* z, int32_t inject,
                           float* mixed, float* zpart, cgs_stream_t stream) {
    if (n < 0 || hw <= 0 || (hw & 3) || !z || !zpart || (mixed && (!a || !b))) return CGS_ERR_BADARG;      // mixed = NULL: only the partial sums of |Z|, Z^2
    int groups = n * (hw / 4);
    if (groups == 0) return CGS_OK;
    hipLaunchKernelGGL(mix_fwd_kernel, dim3(mix_blocks(groups)), dim3(256), 0, (hipStream_t)stream, groups, groups,
                       (const uint32_t*)a, (const uint32_t*)b, (const float4*)z, inject, (float4*)mixed, zpart);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_mix_bwd(int32_t n, int32_t hw, const uint8_t* a, const uint8_t* b, const float* z, const float* dmixed,
                           int32_t inject, float l1_scale, float l2_scale, float* dzpre, cgs_stream_t stream) {
    if (n < 0 || hw <= 0 || (hw & 3) || !a || !b || !z || !dmixed || !dzpre) return CGS_ERR_BADARG;
    int groups = n * (hw / 4);
    if (groups == 0) return CGS_OK;
    hipLaunchKernelGGL(mix_bwd_kernel, dim3((groups + 255) / 256), dim3(256), 0, (hipStream_t)stream, groups,
                       (const uint32_t*)a, (const uint32_t*)b, (const float4*)z, (const float4*)dmixed, inject, l1_scale,
                       l2_scale, (float4*)dzpre, (const float*)nullptr, hw / 4);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_mix_bwd_weighted(int32_t n, int32_t hw, const uint8_t* a, const uint8_t* b, const float* z, const float* dmixed,
                                    int32_t inject, float l1_scale, float l2_scale, const float* valuefak_pred, float* dzpre,
                                    cgs_stream_t stream) {
    if (n < 0 || hw <= 0 || (hw & 3) || !a || !b || !z || !dmixed || !dzpre) return CGS_ERR_BADARG;
    int groups = n * (hw / 4);
    if (groups == 0) return CGS_OK;
    hipLaunchKernelGGL(mix_bwd_kernel, dim3((groups + 255) / 256), dim3(256), 0, (hipStream_t)stream, groups,
                       (const uint32_t*)a, (const uint32_t*)b, (const float4*)z, (const float4*)dmixed, inject, l1_scale,
                       l2_scale, (float4*)dzpre, valuefak_pred, hw / 4);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_phase2_losses(int32_t n, const float* pred, const float* y, const float* zpart, int32_t nzpart,
                                 float lfak, float l1, float l2, int32_t flags, int64_t nz, float* losses, float* dpred,
                                 cgs_stream_t stream) {
    if (n <= 0 || nz <= 0 || nzpart < 0 || !pred || !y || !zpart || !losses || !dpred) return CGS_ERR_BADARG;
    hipLaunchKernelGGL(phase2_losses_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, n, pred, y, zpart, nzpart, lfak, l1,
                       l2, flags, 1.f / (float)nz, losses, dpred);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_phase1_loss(int32_t n, const float* pred, const float* y, int32_t bce, float* losses, float* dpred,
                               cgs_stream_t stream) {
    if (n <= 0 || !pred || !y || !losses || !dpred) return CGS_ERR_BADARG;
    hipLaunchKernelGGL(phase1_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, n, pred, y, bce, losses, dpred);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_reduce_slabs(const cgs_reduce_job* jobs, int32_t njobs, int32_t max_count, uint64_t* step,
                                cgs_stream_t stream) {
    if (!jobs || njobs <= 0 || max_count <= 0) return CGS_ERR_BADARG;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((max_count + 127) / 128, njobs), dim3(1024), 0, (hipStream_t)stream, jobs, step);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_adam_flat(int64_t count, float* param, const float* grad, float* m, float* v, const uint64_t* step,
                             float lr, float beta1, float beta2, float eps, float grad_scale, cgs_stream_t stream) {
    if (count < 0 || !param || !grad || !m || !v || !step) return CGS_ERR_BADARG;
    if (count == 0) return CGS_OK;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)count,
                       param, grad, m, v, step, lr, beta1, beta2, eps, grad_scale);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_nchw_to_nhwc(int32_t n, int32_t c, int32_t hw, const float* src, float* dst, cgs_stream_t stream) {
    if (n < 0 || c <= 0 || hw <= 0 || !src || !dst) return CGS_ERR_BADARG;
    long total = (long)n * c * hw;
    if (total == 0) return CGS_OK;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, total, c, hw, src, dst);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_nhwc_to_nchw(int32_t n, int32_t c, int32_t hw, const float* src, float* dst, cgs_stream_t stream) {
    if (n < 0 || c <= 0 || hw <= 0 || !src || !dst) return CGS_ERR_BADARG;
    long total = (long)n * c * hw;
    if (total == 0) return CGS_OK;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, total, c, hw, src, dst);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_dropout_mask(cgs_dropout d, int64_t count, float* out, cgs_stream_t stream) {
    if (count < 0 || (count & 3) || !out) return CGS_ERR_BADARG;
    if (count == 0) return CGS_OK;
    long c4 = count / 4;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((c4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c4, d, (float4*)out);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}


// ------------------------------------------------------------------------------------------------
// Contrastive batch assembly on the device (main.py:344-356): dst[i] = frame src[idx[i]] rolled along the width by `shift`
// pixels (dst[y][x] = src[y][(x + shift) mod 64]; shift_batch's left roll by s is shift = s, its right roll shift = 64 - s),
// and the matching gather of the fp32 targets.  One workgroup per frame, one thread per 4 output dwords of a 192-byte row.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gather_roll_u8_kernel(const uint8_t* __restrict__ src, const int64_t* __restrict__ idx,
                                                             int shift, uint32_t* __restrict__ dst) {
    const uint8_t* s = src + (size_t)idx[blockIdx.x] * 12288;
    uint32_t* d = dst + (size_t)blockIdx.x * 3072;
    const int sb = 3 * shift;
    for (int w = threadIdx.x; w < 3072; w += 256) {
        const int y = w / 48, b0 = (w % 48) * 4;
        const uint8_t* row = s + y * 192;
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int b = b0 + k + sb;
            b = b >= 192 ? b - 192 : b;
            v |= (uint32_t)row[b] << (8 * k);
        }
        d[w] = v;
    }
}

__global__ void __launch_bounds__(256) gather_f32_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, int n,
                                                         float* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// The whole batch assembly of a phase-2 step (main.py:344-356) in ONE launch (round 5; it was three cgs_gather_roll_u8 and two cgs_gather_f32
// launches per step of the N = 64 loop): workgroup b < n writes A[b] = roll(b < h ? Xpos[idx[b]] : Xneg[idx[b]]), n <= b < 2n writes
// B[b - n] = Xneg[idx[b]] (not rolled), workgroup 2n gathers the n targets of A.
__global__ void __launch_bounds__(256) gather_contrastive_kernel(const uint8_t* __restrict__ xpos, const uint8_t* __restrict__ xneg,
                                                                 const float* __restrict__ ypos, const float* __restrict__ yneg,
                                                                 const int64_t* __restrict__ idx, int n, int h, int shift,
                                                                 uint32_t* __restrict__ a, uint32_t* __restrict__ bdst, float* __restrict__ y) {
    const int b = blockIdx.x;
    if (b == 2 * n) {
        for (int i = threadIdx.x; i < n; i += 256) y[i] = i < h ? ypos[idx[i]] : yneg[idx[i]];
        return;
    }
    const uint8_t* s = (b < h ? xpos : xneg) + (size_t)idx[b] * 12288;
    uint32_t* d = b < n ? a + (size_t)b * 3072 : bdst + (size_t)(b - n) * 3072;
    const int sb = b < n ? 3 * shift : 0;
    for (int w = threadIdx.x; w < 3072; w += 256) {
        const int yy = w / 48, b0 = (w % 48) * 4;
        const uint8_t* row = s + yy * 192;
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int bb = b0 + k + sb;
            bb = bb >= 192 ? bb - 192 : bb;
            v |= (uint32_t)row[bb] << (8 * k);
        }
        d[w] = v;
    }
}


// ------------------------------------------------------------------------------------------------
// Single-GPU step tail in ONE launch: slab reduction -> gradient -> Adam update of the same element (the value never leaves
// the thread), plus, in one extra workgroup, the step's loss VALUES (they are logged, nothing on the device waits for them).
// Every workgroup reads the step counter s at its start and uses t = s + 1 for Adam's bias correction; the workgroup that
// finishes LAST (ticket counter) stores s + 1: nobody can see the new value early.  Saves the Adam and the losses launches.
// ------------------------------------------------------------------------------------------------
struct AdamArgs {
    float* param; const float* grad_base; float* m; float* v;
    float lr, b1, b2, eps;
    unsigned int* ticket;
};
struct LossArgs {       // phase-2 loss values (see phase2_losses_kernel); n == 0: none
    int n, nzpart, flags;
    const float* pred; const float* y; const float* zpart;
    float lfak, l1, l2, inv_nz;
    float* losses;
};

template <int THREADS>
__device__ __forceinline__ void phase2_loss_values(const LossArgs& L) {
    constexpr int NW = THREADS / 64;
    __shared__ float lred[5][16];
    const int tid = threadIdx.x, n = L.n;
    const bool live = L.flags & 1, inject = L.flags & 2, bce = L.flags & 4;
    float sc = 0.f, sr = 0.f, si = 0.f, z1 = 0.f, z2 = 0.f;
    for (int i = tid; i < n; i += THREADS) {
        const float pb = L.pred[i], pa = L.pred[n + i], pr = L.pred[2 * n + i], yi = L.y[i];
        if (live) {
            if (bce) sc += -(yi * fmaxf(logf(pa), -100.f) + (1.f - yi) * fmaxf(logf(1.f - pa), -100.f));
            else sc += (pa - yi) * (pa - yi);
        }
        sr += (pr - pb) * (pr - pb);
        if (inject) { const float d = L.pred[3 * n + i] - pa; si += d * d; }
    }
    const int per_img = L.nzpart / n;
    for (int i = tid; i < L.nzpart; i += THREADS) {
        const float vf = (L.flags & 8) ? 1.f - L.pred[n + i / per_img] : 1.f;
        z1 += vf * L.zpart[2 * i]; z2 += vf * vf * L.zpart[2 * i + 1];
    }
    sc = wave_sum(sc); sr = wave_sum(sr); si = wave_sum(si); z1 = wave_sum(z1); z2 = wave_sum(z2);
    if ((tid & 63) == 0) { const int w = tid >> 6; lred[0][w] = sc; lred[1][w] = sr; lred[2][w] = si; lred[3][w] = z1; lred[4][w] = z2; }
    __syncthreads();
    if (tid == 0) {
        float t[5];
        for (int k = 0; k < 5; ++k) { float a = 0.f; for (int w = 0; w < NW; ++w) a += lred[k][w]; t[k] = a; }
        const float inv_n = 1.f / (float)n;
        const float c = t[0] * inv_n, r = t[1] * inv_n, i = t[2] * inv_n, n1 = L.l1 * t[3] * L.inv_nz, n2 = L.l2 * t[4] * L.inv_nz;
        L.losses[0] = c; L.losses[1] = r; L.losses[2] = i; L.losses[3] = n1; L.losses[4] = n2;
        L.losses[5] = (live ? L.lfak * c : 0.f) + r + i + n1 + n2;
        L.losses[6] = 0.f; L.losses[7] = 0.f;
    }
}

#ifndef CGS_REDUCE_DEEP
#define CGS_REDUCE_DEEP 0
#endif
#ifndef CGS_REDUCE_COLS
#define CGS_REDUCE_COLS 32      // columns per workgroup (64: a wave reads 256 contiguous bytes of a slab row instead of two rows' 128; A/B round 5)
#endif
template <int SL, int COLS = CGS_REDUCE_COLS>
__global__ void __launch_bounds__(COLS * SL) reduce_adam_kernel(const cgs_reduce_job* __restrict__ jobs, int njobs, uint64_t* step,
                                                              AdamArgs A, LossArgs L) {
    __shared__ float red[SL][COLS + 1];
    __shared__ float bc[2];                  // Adam's bias corrections for t = s + 1 (once per workgroup)
    const bool loss_row = (int)blockIdx.y == njobs;
    int row_blocks = 1;                      // workgroups of this grid row that have work (and therefore read *step)
    cgs_reduce_job j{};
    if (!loss_row) {
        j = jobs[blockIdx.y];
        row_blocks = (j.count + COLS - 1) / COLS;
    }
    if ((int)blockIdx.x >= row_blocks) return;          // nothing to do: never reads the counter, not part of the ticket
    const uint64_t s_old = *step;
    if (loss_row) {
        if (L.n > 0) phase2_loss_values<COLS * SL>(L);
    } else {
        if (threadIdx.x == 0) {
            // 1 - b^t = -expm1(t ln b): no cancellation at small t, and no double-precision pow (its registers halve the
            // occupancy of this 1024-thread workgroup: measured 46 vs 22 us for the whole kernel)
            const float t = (float)(s_old + 1);
            bc[0] = -expm1f(t * logf(A.b1));
            bc[1] = sqrtf(-expm1f(t * logf(A.b2)));
        }
        const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
        const int i = blockIdx.x * COLS + col;
        float s[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) s[u] = 0.f;
        if (i < j.count) {
            const float* p = j.slab + i;
            int b = sl;
#if CGS_REDUCE_DEEP
            // (A/B round 5) 32 loads in flight per thread for the 512- and 1024-row jobs: half as many dependent memory round trips
            for (; b + 31 * SL < j.nslab; b += 32 * SL) {
                float t[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) t[u] = p[(size_t)(b + u * SL) * j.stride];
#pragma unroll
                for (int u = 0; u < 16; ++u) s[u] += t[u];
#pragma unroll
                for (int u = 0; u < 16; ++u) s[u] += t[16 + u];
            }
#endif
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
    // ---- the last working workgroup to finish publishes the new step value (two-level ticket: per row, then over the rows) ----
    // Only an ORDER is needed: this workgroup's read of *step (done: its value was consumed above / is pinned here) before its
    // ticket increment; the relaxed device-scope atomics give that without a release fence (a __threadfence here writes the
    // whole L2 back once per workgroup: measured +30 us).
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("" :: "s"(s_old));
        unsigned int* row = A.ticket + 1 + blockIdx.y;
        if (atomicAdd(row, 1u) == (unsigned int)row_blocks - 1) {
            *row = 0u;
            if (atomicAdd(A.ticket, 1u) == gridDim.y - 1) {
                *A.ticket = 0u;
                *step = s_old + 1;
            }
        }
    }
}

extern "C" int cgs_reduce_adam(const cgs_reduce_job* jobs, int32_t njobs, int32_t max_count, uint64_t* step, float* param,
                               const float* grad_base, float* m, float* v, float lr, float beta1, float beta2, float eps,
                               uint32_t* ticket, int32_t n, const float* pred, const float* y, const float* zpart, int32_t nzpart,
                               float lfak, float l1, float l2, int32_t flags, int64_t nz, float* losses, cgs_stream_t stream) {
    // param == NULL: reduction (+ loss values, + step tick) only -- the data-parallel form, followed by the all-reduce and cgs_adam_flat
    if (!jobs || njobs <= 0 || max_count <= 0 || !step || !grad_base || !ticket || (param && (!m || !v))) return CGS_ERR_BADARG;
    if (n > 0 && (!pred || !y || !zpart || !losses || nz <= 0)) return CGS_ERR_BADARG;
    AdamArgs A{param, grad_base, m, v, lr, beta1, beta2, eps, ticket};
    LossArgs L{n, nzpart, flags, pred, y, zpart, lfak, l1, l2, nz > 0 ? 1.f / (float)nz : 0.f, losses};
    // 16 slab lanes x 32 columns per workgroup: the ~800 working workgroups of a phase-2 step are all resident at once
    // (1024-thread workgroups needed 1.6 rounds of the chip's 2 x 256 slots)
    constexpr int SL = 16;
    hipLaunchKernelGGL(reduce_adam_kernel<SL>, dim3((max_count + CGS_REDUCE_COLS - 1) / CGS_REDUCE_COLS, njobs + 1), dim3(CGS_REDUCE_COLS * SL), 0, (hipStream_t)stream, jobs, njobs,
                       step, A, L);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_gather_roll_u8(const uint8_t* src, const int64_t* idx, int32_t n, int32_t shift_px, uint8_t* dst,
                                  cgs_stream_t stream) {
    if (!src || !idx || !dst || n < 0 || shift_px < 0 || shift_px >= 64) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(gather_roll_u8_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, src, idx, shift_px, (uint32_t*)dst);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_gather_contrastive(const uint8_t* xpos, const uint8_t* xneg, const float* ypos, const float* yneg, const int64_t* idx,
                                      int32_t n, int32_t h, int32_t shift_px, uint8_t* a, uint8_t* b, float* y, cgs_stream_t stream) {
    if (!xpos || !xneg || !ypos || !yneg || !idx || !a || !b || !y || n < 0 || h < 0 || h > n || shift_px < 0 || shift_px >= 64) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(gather_contrastive_kernel, dim3(2 * n + 1), dim3(256), 0, (hipStream_t)stream, xpos, xneg, ypos, yneg, idx, n, h, shift_px,
                       (uint32_t*)a, (uint32_t*)b, y);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_gather_f32(const float* src, const int64_t* idx, int32_t n, float* dst, cgs_stream_t stream) {
    if (!src || !idx || !dst || n < 0) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(gather_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, idx, n, dst);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" const char* cgs_build_arch(void) { return "gfx950"; }
extern "C" int cgs_abi_version(void) { return 1; }

// Optional BatchNorm2d (+ ReLU / LeakyReLU) epilogue, training and eval mode, forward and backward, on NHWC fp32 maps [pixels][c]
// ("an optional BN epilogue"; BASELINE.json north_star: "BatchNorm+ReLU fusions ... wavefront shuffles for the BN
// reductions ... BN stats kept per-GPU").  The reference has no BatchNorm anywhere (keep it OFF in every parity run), so this op is
// NOT on any parity path and nothing of the reference pins it: tests compare it with torch.nn.BatchNorm2d + autograd in float64.
//
//   statistics   one pass over x: every thread owns one float4 channel group and a strided set of pixels (plain sum / sum of squares over its
//                <= 64 elements), turns them into (count, mean, M2) and the workgroup merges those with Chan's pairwise update -- across the
//                lanes of a wave with shuffles (__shfl_xor over the lanes that hold the same channel group: the group count is a power of two
//                for c = 4 .. 64), across the waves through LDS -- into one partial row per workgroup; a one-workgroup kernel merges the rows
//                in a fixed order (bitwise reproducible, no atomics, no E[x^2] - E[x]^2 cancellation) and writes mean, 1 / sqrt(var + eps),
//                the fused scale / shift, and the running statistics (unbiased variance, momentum) exactly as torch does;
//   apply        y = act(x * scale[c] + shift[c]), float4 per thread;
//   backward     dy' = dy * act'(y); per-channel sums of dy' and dy' * xhat (same wave-shuffle + LDS + fixed-order scheme, plain sums), then
//                dx = gamma * rstd * (dy' - sum(dy') / N - xhat * sum(dy' xhat) / N); dgamma = sum(dy' xhat), dbeta = sum(dy').
// Data parallel: statistics are per GPU (no collective), as the north_star prescribes.
#include "cgs_common.h"

namespace {

constexpr int kBnBlocks = 1024;        // partial rows at most (persistent workgroups, grid-stride over the pixels)

struct Wf { float n, mean, m2; };      // Welford / Chan triple

__device__ __forceinline__ Wf wf_merge(Wf a, Wf b) {
    const float n = a.n + b.n;
    if (n == 0.f) return Wf{0.f, 0.f, 0.f};
    const float d = b.mean - a.mean, f = b.n / n;
    return Wf{n, a.mean + d * f, a.m2 + b.m2 + d * d * a.n * f};
}

// BN statistics partials: ws[block][c][3].  G = c / 4 channel groups; thread t owns group t % G and pixels t / G + k * (threads / G).
__global__ void __launch_bounds__(256) bn_stats_kernel(const float4* __restrict__ x, long pixels, int c, float* __restrict__ ws) {
    const int G = c >> 2, tid = threadIdx.x, g = tid % G, lane = tid & 63;
    const int tpb = 256 / G * G;                        // threads that own a (group, pixel slot) pair; the rest idle (G not a divisor of 256)
    const long slots = (long)gridDim.x * (256 / G);
    // sums of (x - K), K = the thread's first element: the short per-thread sums carry no mean^2 term to cancel against
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f}, K[4] = {0.f, 0.f, 0.f, 0.f};
    float cnt = 0.f;
    if (tid < tpb)
        for (long p = (long)blockIdx.x * (256 / G) + tid / G; p < pixels; p += slots) {
            const float4 v = x[p * G + g];
            const float va[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (cnt == 0.f) K[j] = va[j];
                const float d = va[j] - K[j];
                s[j] += d; q[j] += d * d;
            }
            cnt += 1.f;                                 // (<= 64 pixels per thread up to kBnBlocks rows, more beyond)
        }
    Wf w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float m = cnt > 0.f ? s[j] / cnt : 0.f;
        w[j] = Wf{cnt, K[j] + m, cnt > 0.f ? fmaxf(q[j] - s[j] * m, 0.f) : 0.f};
    }
    __shared__ float red[4][16][4][3];                  // [wave][group (pow2 path: <= 16)][channel][n, mean, M2]
    __shared__ float redg[256][4][3];                   // general path: every thread's triple
    const bool pow2 = (G & (G - 1)) == 0 && G <= 16;    // 64 % G == 0: lanes with equal lane % G hold the same group
    if (pow2) {
        for (int m = 32; m >= G; m >>= 1)               // wavefront shuffles: merge the lanes that hold the same channel group
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const Wf o{__shfl_xor(w[j].n, m, 64), __shfl_xor(w[j].mean, m, 64), __shfl_xor(w[j].m2, m, 64)};
                w[j] = wf_merge(w[j], o);
            }
        if (lane < G)
#pragma unroll
            for (int j = 0; j < 4; ++j) { red[tid >> 6][lane][j][0] = w[j].n; red[tid >> 6][lane][j][1] = w[j].mean; red[tid >> 6][lane][j][2] = w[j].m2; }
        __syncthreads();
        if (tid < c) {
            const int gg = tid >> 2, j = tid & 3;
            Wf a{red[0][gg][j][0], red[0][gg][j][1], red[0][gg][j][2]};
            for (int wv = 1; wv < 4; ++wv) a = wf_merge(a, Wf{red[wv][gg][j][0], red[wv][gg][j][1], red[wv][gg][j][2]});
            float* o = ws + ((size_t)blockIdx.x * c + tid) * 3;
            o[0] = a.n; o[1] = a.mean; o[2] = a.m2;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { redg[tid][j][0] = w[j].n; redg[tid][j][1] = w[j].mean; redg[tid][j][2] = w[j].m2; }
        __syncthreads();
        if (tid < c) {                                  // fixed order over the threads of the same group
            const int gg = tid >> 2, j = tid & 3;
            Wf a{0.f, 0.f, 0.f};
            for (int t = gg; t < tpb; t += G) a = wf_merge(a, Wf{redg[t][j][0], redg[t][j][1], redg[t][j][2]});
            float* o = ws + ((size_t)blockIdx.x * c + tid) * 3;
            o[0] = a.n; o[1] = a.mean; o[2] = a.m2;
        }
    }
}

// one workgroup: merge the partial rows in order; stats[c][4] = mean, rstd, scale, shift; running statistics as torch.nn.BatchNorm2d
__global__ void __launch_bounds__(256) bn_finish_kernel(const float* __restrict__ ws, int rows, int c, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, float* __restrict__ stats,
                                                        float* __restrict__ run_mean, float* __restrict__ run_var, float momentum) {
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        Wf a{0.f, 0.f, 0.f};
        for (int r = 0; r < rows; ++r) {
            const float* o = ws + ((size_t)r * c + ch) * 3;
            a = wf_merge(a, Wf{o[0], o[1], o[2]});
        }
        const float var = a.n > 0.f ? a.m2 / a.n : 0.f, rstd = 1.f / sqrtf(var + eps);
        const float sc = (gamma ? gamma[ch] : 1.f) * rstd;
        stats[4 * ch] = a.mean; stats[4 * ch + 1] = rstd; stats[4 * ch + 2] = sc; stats[4 * ch + 3] = (beta ? beta[ch] : 0.f) - a.mean * sc;
        if (run_mean) run_mean[ch] = (1.f - momentum) * run_mean[ch] + momentum * a.mean;
        if (run_var) run_var[ch] = (1.f - momentum) * run_var[ch] + momentum * (a.n > 1.f ? a.m2 / (a.n - 1.f) : var);
    }
}

// eval mode: scale / shift from the running statistics
__global__ void __launch_bounds__(256) bn_eval_stats_kernel(int c, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                                            float* __restrict__ stats) {
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        const float rstd = 1.f / sqrtf(run_var[ch] + eps), sc = (gamma ? gamma[ch] : 1.f) * rstd;
        stats[4 * ch] = run_mean[ch]; stats[4 * ch + 1] = rstd; stats[4 * ch + 2] = sc; stats[4 * ch + 3] = (beta ? beta[ch] : 0.f) - run_mean[ch] * sc;
    }
}

__device__ __forceinline__ float bn_act(float v, int act, float slope) {
    if (act == CGS_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == CGS_ACT_LRELU) return v > 0.f ? v : slope * v;
    return v;
}

__global__ void __launch_bounds__(256) bn_apply_kernel(const float4* __restrict__ x, long quads, int c, const float* __restrict__ stats, int act,
                                                       float slope, float4* __restrict__ y) {
    const int G = c >> 2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < quads; e += (long)gridDim.x * 256) {
        const int g = (int)(e % G);
        const float4 v = x[e];
        const float* st = stats + 16 * g;
        y[e] = make_float4(bn_act(v.x * st[2] + st[3], act, slope), bn_act(v.y * st[6] + st[7], act, slope),
                           bn_act(v.z * st[10] + st[11], act, slope), bn_act(v.w * st[14] + st[15], act, slope));
    }
}

// backward partials: ws[block][c][2] = sum dy', sum dy' xhat  (dy' = dy * act'(y))
__global__ void __launch_bounds__(256) bn_bwd_sums_kernel(const float4* __restrict__ x, const float4* __restrict__ y, const float4* __restrict__ dy,
                                                          long pixels, int c, const float* __restrict__ stats, int act, float slope,
                                                          float* __restrict__ ws) {
    const int G = c >> 2, tid = threadIdx.x, g = tid % G, lane = tid & 63;
    const int tpb = 256 / G * G;
    const long slots = (long)gridDim.x * (256 / G);
    float sd[4] = {0.f, 0.f, 0.f, 0.f}, sx[4] = {0.f, 0.f, 0.f, 0.f};
    const float* st = stats + 16 * g;
    if (tid < tpb)
        for (long p = (long)blockIdx.x * (256 / G) + tid / G; p < pixels; p += slots) {
            const float4 xv = x[p * G + g], dv = dy[p * G + g];
            float d[4] = {dv.x, dv.y, dv.z, dv.w};
            if (act != CGS_ACT_NONE) {
                const float4 yv = y[p * G + g];
                const float ya[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j] = ya[j] > 0.f ? d[j] : (act == CGS_ACT_LRELU ? slope * d[j] : 0.f);
            }
            const float xa[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { sd[j] += d[j]; sx[j] += d[j] * (xa[j] - st[4 * j]) * st[4 * j + 1]; }
        }
    __shared__ float red[256][8];
    const bool pow2 = (G & (G - 1)) == 0 && G <= 16;
    if (pow2) {
        for (int m = 32; m >= G; m >>= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) { sd[j] += __shfl_xor(sd[j], m, 64); sx[j] += __shfl_xor(sx[j], m, 64); }
        if (lane < G)
#pragma unroll
            for (int j = 0; j < 4; ++j) { red[(tid >> 6) * 16 + lane][j] = sd[j]; red[(tid >> 6) * 16 + lane][4 + j] = sx[j]; }
        __syncthreads();
        if (tid < c) {
            const int gg = tid >> 2, j = tid & 3;
            float* o = ws + ((size_t)blockIdx.x * c + tid) * 2;
            o[0] = (red[gg][j] + red[16 + gg][j]) + (red[32 + gg][j] + red[48 + gg][j]);
            o[1] = (red[gg][4 + j] + red[16 + gg][4 + j]) + (red[32 + gg][4 + j] + red[48 + gg][4 + j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[tid][j] = sd[j]; red[tid][4 + j] = sx[j]; }
        __syncthreads();
        if (tid < c) {
            const int gg = tid >> 2, j = tid & 3;
            float a = 0.f, b = 0.f;
            for (int t = gg; t < tpb; t += G) { a += red[t][j]; b += red[t][4 + j]; }
            float* o = ws + ((size_t)blockIdx.x * c + tid) * 2;
            o[0] = a; o[1] = b;
        }
    }
}

// one workgroup: dbeta = sum dy', dgamma = sum dy' xhat (fixed order); coef[c][2] = dbeta / N, dgamma / N
__global__ void __launch_bounds__(256) bn_bwd_finish_kernel(const float* __restrict__ ws, int rows, int c, long pixels, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ coef) {
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < rows; ++r) { a += ws[((size_t)r * c + ch) * 2]; b += ws[((size_t)r * c + ch) * 2 + 1]; }
        if (dbeta) dbeta[ch] = a;
        if (dgamma) dgamma[ch] = b;
        coef[2 * ch] = a / (float)pixels; coef[2 * ch + 1] = b / (float)pixels;
    }
}

__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float4* __restrict__ x, const float4* __restrict__ y, const float4* __restrict__ dy,
                                                           long quads, int c, const float* __restrict__ stats, const float* __restrict__ coef,
                                                           int act, float slope, int train, float4* __restrict__ dx) {
    const int G = c >> 2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < quads; e += (long)gridDim.x * 256) {
        const int g = (int)(e % G);
        const float4 xv = x[e], dv = dy[e];
        float d[4] = {dv.x, dv.y, dv.z, dv.w};
        if (act != CGS_ACT_NONE) {
            const float4 yv = y[e];
            const float ya[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = ya[j] > 0.f ? d[j] : (act == CGS_ACT_LRELU ? slope * d[j] : 0.f);
        }
        const float xa[4] = {xv.x, xv.y, xv.z, xv.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* st = stats + 16 * g + 4 * j;      // mean, rstd, scale (= gamma rstd), shift
            const float xh = (xa[j] - st[0]) * st[1];
            // eval mode: the statistics are constants, dx = dy' * gamma * rstd
            o[j] = train ? st[2] * (d[j] - coef[2 * (4 * g + j)] - xh * coef[2 * (4 * g + j) + 1]) : st[2] * d[j];
        }
        dx[e] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

int bn_rows(long pixels, int c) {
    const int per_block = 256 / (c >> 2);               // pixel slots per workgroup
    long need = (pixels + (long)per_block * 64 - 1) / ((long)per_block * 64);      // a thread sees at most 64 pixels ...
    if (need < 1) need = 1;
    return (int)(need < kBnBlocks ? need : kBnBlocks);  // ... up to kBnBlocks rows (beyond: more pixels per thread, still exact enough: Chan merge)
}

}  // namespace

// rows of the workspace: ws holds (3 rows + 2) c floats ([rows][c][3] statistics partials; the backward uses [rows][c][2] + [c][2] behind them)
extern "C" int cgs_bn_rows(int64_t pixels, int32_t c) {
    if (pixels < 0 || c < 4 || c > 64 || (c & 3)) return CGS_ERR_BADARG;
    return bn_rows(pixels, c);
}

// y = act(BatchNorm(x)) on NHWC fp32 [pixels][c] (c a multiple of 4, <= 64).  train != 0: batch statistics (biased variance for the
// normalisation, unbiased for running_var, torch semantics; running_* may be NULL); train == 0: the running statistics.
// stats [c][4] (mean, 1 / sqrt(var + eps), scale, shift) is written for the backward; ws: (3 cgs_bn_rows(pixels, c) + 2) c floats.
extern "C" int cgs_bn_act_fwd(int64_t pixels, int32_t c, const float* x, const float* gamma, const float* beta, float eps, int32_t act, float slope,
                              int32_t train, float* y, float* stats, float* ws, float* running_mean, float* running_var, float momentum,
                              cgs_stream_t stream) {
    if (pixels < 0 || c < 4 || c > 64 || (c & 3) || !x || !y || !stats || act < CGS_ACT_NONE || act > CGS_ACT_LRELU) return CGS_ERR_BADARG;
    if (train ? !ws : (!running_mean || !running_var)) return CGS_ERR_BADARG;
    if (pixels == 0) return CGS_OK;
    hipStream_t st = (hipStream_t)stream;
    if (train) {
        const int rows = bn_rows(pixels, c);
        hipLaunchKernelGGL(bn_stats_kernel, dim3(rows), dim3(256), 0, st, (const float4*)x, (long)pixels, c, ws);
        hipLaunchKernelGGL(bn_finish_kernel, dim3(1), dim3(256), 0, st, ws, rows, c, gamma, beta, eps, stats, running_mean, running_var, momentum);
    } else {
        hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(1), dim3(256), 0, st, c, gamma, beta, eps, running_mean, running_var, stats);
    }
    const long quads = pixels * (c >> 2);
    const int blocks = (int)((quads + 255) / 256 < 8192 ? (quads + 255) / 256 : 8192);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)x, quads, c, stats, act, slope, (float4*)y);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// Backward of cgs_bn_act_fwd: dx (and dgamma / dbeta [c], may be NULL) from dy, the saved x, y (read when act != none) and stats.
extern "C" int cgs_bn_act_bwd(int64_t pixels, int32_t c, const float* x, const float* y, const float* dy, const float* stats, int32_t act, float slope,
                              int32_t train, float* dx, float* dgamma, float* dbeta, float* ws, cgs_stream_t stream) {
    if (pixels < 0 || c < 4 || c > 64 || (c & 3) || !x || !dy || !stats || !dx || !ws || act < CGS_ACT_NONE || act > CGS_ACT_LRELU) return CGS_ERR_BADARG;
    if (act != CGS_ACT_NONE && !y) return CGS_ERR_BADARG;
    if (pixels == 0) return CGS_OK;
    hipStream_t st = (hipStream_t)stream;
    const int rows = bn_rows(pixels, c);
    float* coef = ws + (size_t)rows * c * 3;            // [c][2] behind the rows (ws holds (3 rows + 2) c floats)
    hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(rows), dim3(256), 0, st, (const float4*)x, (const float4*)y, (const float4*)dy, (long)pixels, c, stats,
                       act, slope, ws);
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(1), dim3(256), 0, st, ws, rows, c, (long)pixels, dgamma, dbeta, coef);
    const long quads = pixels * (c >> 2);
    const int blocks = (int)((quads + 255) / 256 < 8192 ? (quads + 255) / 256 : 8192);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)x, (const float4*)y, (const float4*)dy, quads, c, stats,
                       coef, act, slope, train, (float4*)dx);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

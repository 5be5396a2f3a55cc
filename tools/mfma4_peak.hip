// v_mfma_f32_4x4x1_16B_f32 with the A operand broadcast from one block (cbsz = 4, abid = block): semantics check and
// sustained rate on gfx950.  With cbsz = 4 every one of the 16 blocks multiplies the SAME four A values (lanes 4*abid .. +3 of
// the A register) with its own four B values, so   D[lane][r] = A[4*abid + r] * B[lane] + C[lane][r]:
// lane = pixel, r = output channel: a direct convolution's inner step (one tap, one input channel, 4 output channels, 64 pixels)
// in ONE instruction, no padding for 8-channel layers, and a register holds 16 (tap, channel, 4-channel group) weight quads.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma4_peak.hip -o tools/mfma4_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void sem(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f4 acc = f4{0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 4, 5, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

template <int CHAINS>
__global__ void __launch_bounds__(256) k4(float* out, int iters) {
    f4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                // abid must be an immediate: vary it with the unrolled index
                switch (r & 3) {
                case 0: acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 4, 0, 0); break;
                case 1: acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 4, 5, 0); break;
                case 2: acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 4, 10, 0); break;
                default: acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 4, 15, 0); break;
                }
            }
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same FLOPs on the vector ALU (v_fma_f32, CHAINS x 4 independent accumulators)
template <int CHAINS>
__global__ void __launch_bounds__(256) kv(float* out, int iters) {
    float acc[CHAINS][4];
    for (int c = 0; c < CHAINS; ++c) for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[c][j] = fmaf(a, b, acc[c][j]);
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int j = 0; j < 4; ++j) s += acc[c][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class F>
static void run(const char* name, F launch, int per_iter, int blocks, int threads) {
    float* out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    launch(out, 100, blocks);
    hipDeviceSynchronize();
    hipEventRecord(e0); launch(out, iters, blocks); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)blocks * (threads / 64) * iters * per_iter;      // wave-level 4x4x1-equivalents (512 FLOP each)
    printf("%-34s blocks=%5d  %8.3f ms  %7.1f TFLOP/s  (%.2f ns per 512-FLOP step per wave)\n", name, blocks, ms, n * 512 / ms / 1e9,
           ms * 1e6 / ((double)iters * per_iter));
    hipFree(out);
}

int main() {
    {   // semantics
        float ha[64], hb[64], hd[256];
        for (int i = 0; i < 64; ++i) { ha[i] = 1.f + i; hb[i] = 100.f + 3 * i; }
        float *a, *b, *d;
        hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
        hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, a, b, d);
        hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hd[l * 4 + r] != ha[4 * 5 + r] * hb[l]) ++bad;
        printf("semantics D[lane][r] == A[4*abid + r] * B[lane] (cbsz 4, abid 5): %s (%d mismatches)  e.g. D[7] = %g %g %g %g\n",
               bad ? "NO" : "yes", bad, hd[28], hd[29], hd[30], hd[31]);
    }
    for (int blocks : {256, 512, 1024, 2048}) {
        run("4x4x1 f32 bcast, 1 chain", [](float* o, int it, int b) { hipLaunchKernelGGL(k4<1>, dim3(b), dim3(256), 0, 0, o, it); }, 16, blocks, 256);
        run("4x4x1 f32 bcast, 2 chains", [](float* o, int it, int b) { hipLaunchKernelGGL(k4<2>, dim3(b), dim3(256), 0, 0, o, it); }, 32, blocks, 256);
        run("4x4x1 f32 bcast, 4 chains", [](float* o, int it, int b) { hipLaunchKernelGGL(k4<4>, dim3(b), dim3(256), 0, 0, o, it); }, 64, blocks, 256);
        run("4x4x1 f32 bcast, 8 chains", [](float* o, int it, int b) { hipLaunchKernelGGL(k4<8>, dim3(b), dim3(256), 0, 0, o, it); }, 128, blocks, 256);
        run("v_fma_f32, 8x4 accumulators", [](float* o, int it, int b) { hipLaunchKernelGGL(kv<8>, dim3(b), dim3(256), 0, 0, o, it); }, 128, blocks, 256);
    }
    return 0;
}

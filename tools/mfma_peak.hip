// Measures the sustained rate of the fp32 MFMA instructions on gfx950 (register operands only), to price the
// implicit-GEMM kernels against what the matrix pipe can actually do.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(256) k16(float* out, int iters) {
    f4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS>
__global__ void __launch_bounds__(256) k32(float* out, int iters) {
    f16v acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int j = 0; j < 16; ++j) acc[c][j] = 0;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int j = 0; j < 16; ++j) s += acc[c][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class F>
static void run(const char* name, F launch, double flop_per_mfma, int mfma_per_iter, int blocks) {
    float* out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    launch(out, 100, blocks);
    hipDeviceSynchronize();
    hipEventRecord(e0); launch(out, iters, blocks); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)blocks * 4 * iters * mfma_per_iter;
    printf("%-28s blocks=%5d  %8.3f ms  %7.1f TFLOP/s  (%.1f ns per MFMA per wave)\n", name, blocks, ms, n * flop_per_mfma / ms / 1e9,
           ms * 1e6 / ((double)iters * mfma_per_iter));
    hipFree(out);
}

int main() {
    for (int blocks : {256, 512, 1024}) {
        run("16x16x4 f32, 1 chain", [](float* o, int it, int b) { hipLaunchKernelGGL(k16<1>, dim3(b), dim3(256), 0, 0, o, it); }, 2048, 8, blocks);
        run("16x16x4 f32, 4 chains", [](float* o, int it, int b) { hipLaunchKernelGGL(k16<4>, dim3(b), dim3(256), 0, 0, o, it); }, 2048, 32, blocks);
        run("32x32x2 f32, 1 chain", [](float* o, int it, int b) { hipLaunchKernelGGL(k32<1>, dim3(b), dim3(256), 0, 0, o, it); }, 4096, 8, blocks);
        run("32x32x2 f32, 2 chains", [](float* o, int it, int b) { hipLaunchKernelGGL(k32<2>, dim3(b), dim3(256), 0, 0, o, it); }, 4096, 16, blocks);
    }
    return 0;
}

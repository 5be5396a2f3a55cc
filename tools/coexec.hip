// Do VALU FMAs of one wave and MFMAs of another wave on the SAME SIMD overlap?  512-thread workgroups, one per CU:
// waves 0-3 run v_fma_f32 chains, waves 4-7 run v_mfma_f32_16x16x4_f32 chains (mode 0), or only one kind runs (1: VALU, 2: MFMA).
// Build: hipcc -O3 --offload-arch=gfx950 tools/coexec.hip -o tools/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) k(float* out, int iters, int mode) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x) >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode != 2) {
            float a[16];
            for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
            float x = 1.0001f, y = 0.5f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int rep = 0; rep < 16; ++rep)      // 16 x 16 = 256 FMAs = as many issue cycles as 32 MFMAs (4 vs 32 cycles)
#pragma unroll
                    for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], x, y);
            }
            for (int i = 0; i < 16; ++i) r += a[i];
        }
    } else if (mode != 1) {
        f4 acc[4];
        for (int c = 0; c < 4; ++c) acc[c] = f4{0, 0, 0, 0};
        float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)           // 32 MFMAs x 32 cycles = 1024 cycles
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
        }
        for (int c = 0; c < 4; ++c) r += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[3] = {"VALU waves + MFMA waves", "VALU waves only", "MFMA waves only"};
    for (int round = 0; round < 2; ++round)
        for (int mode = 0; mode < 3; ++mode) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 100, mode);
            hipDeviceSynchronize();
            hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode); hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-26s %8.3f ms  (%.0f ns per iteration: 256 v_fma_f32 per VALU wave | 32 MFMA per matrix wave)\n", names[mode], ms,
                   ms * 1e6 / iters);
        }
    return 0;
}

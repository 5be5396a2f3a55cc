// Micro-benchmark: cost of executing straight-line code for the first time (instruction cache cold) vs the second time.
// One 256-thread workgroup per CU runs a block of N independent v_fma_f32 (8 bytes each) twice; s_memtime around each pass.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define F1 asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y)); \
           asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(x), "v"(y));
#define F4 F1 F1 F1 F1
#define F16 F4 F4 F4 F4
#define F64 F16 F16 F16 F16
#define F256 F64 F64 F64 F64

template <int KB>   // code size of the measured block: KB kilobytes = KB * 128 instructions
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* t) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, x = out[threadIdx.x], y = out[threadIdx.x + 1];
    unsigned long long s[3];
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        s[pass] = __builtin_amdgcn_s_memtime();
        if constexpr (KB >= 2) { F64 }        // 256 instr = 2 KB
        if constexpr (KB >= 4) { F64 }
        if constexpr (KB >= 8) { F64 F64 }
        if constexpr (KB >= 16) { F256 }
        if constexpr (KB >= 32) { F256 F256 }
        asm volatile("s_nop 0" ::: "memory");
    }
    s[2] = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = s[1] - s[0]; t[blockIdx.x * 2 + 1] = s[2] - s[1]; }
}

template <int KB>
void run(float* out, unsigned long long* t, int blocks) {
    // a different kernel first, so this kernel's code is not resident
    hipLaunchKernelGGL(k<KB>, dim3(blocks), dim3(256), 0, 0, out, t);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 2);
    CHECK(hipMemcpy(h.data(), t, blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double c = 0, w = 0; for (int i = 0; i < blocks; ++i) { c += h[2 * i]; w += h[2 * i + 1]; }
    printf("%2d KB of code, %4d workgroups: first pass %8.0f ticks, second pass %8.0f ticks  -> cold cost %.2f ticks/byte\n", KB, blocks,
           c / blocks, w / blocks, (c - w) / blocks / (KB * 1024.0));
}

int main() {
    float* out; unsigned long long* t;
    CHECK(hipMalloc(&out, 2048 * 256 * 4)); CHECK(hipMemset(out, 0, 2048 * 256 * 4)); CHECK(hipMalloc(&t, 2048 * 16));
    for (int blocks : {256, 1024}) {
        run<2>(out, t, blocks); run<4>(out, t, blocks); run<8>(out, t, blocks); run<16>(out, t, blocks); run<32>(out, t, blocks);
    }
    return 0;
}

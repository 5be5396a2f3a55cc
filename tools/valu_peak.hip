// Micro-benchmark: sustained fp32 FMA rate on gfx950 for the instruction forms the conv kernels can use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, const float* wsrc, int iters) {
    float a[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = threadIdx.x * 1e-3f + i;
    float x0 = out[threadIdx.x], x1 = out[threadIdx.x + 1];
    typedef const float __attribute__((address_space(4)))* cptr;
    cptr wc = (cptr)wsrc;
    float w0 = wc[0], w1 = wc[1], w2 = wc[2], w3 = wc[3];
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {        // v_fma_f32, all VGPR
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x0), "v"(x1));
        } else if (MODE == 1) { // v_fmac_f32 with SGPR operand
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(w0), "v"(x0));
        } else if (MODE == 2) { // v_pk_fma_f32, all VGPR pairs
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 acc = {a[i], a[i + 1]}; f2 xx = {x0, x1}; f2 yy = {x1, x0};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(xx), "v"(yy));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(xx), "v"(yy));
                a[i] = acc.x; a[i + 1] = acc.y;
            }
        } else if (MODE == 3) { // v_pk_fma_f32 with SGPR pair + broadcast x (what the conv loop emits)
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 acc = {a[i], a[i + 1]}; f2 xx = {x0, x1}; f2 ww = {w0, w1};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(xx), "s"(ww));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(xx), "s"(ww));
                a[i] = acc.x; a[i + 1] = acc.y;
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + w2 + w3;
}

template <int MODE>
void run(const char* name, int blocks_per_cu, float* out, float* w) {
    int iters = 2000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, w, iters);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, w, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double fma = (double)blocks * 256 * iters * 32;      // scalar FMAs (pk counts 2 per instr, 16 instr x2)
    printf("%-28s blocks/CU=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, ms, 2 * fma / ms / 1e9);
}

int main() {
    float *out, *w;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 4 * 2)); CHECK(hipMalloc(&w, 64));
    CHECK(hipMemset(out, 0, 256 * 8 * 256 * 4 * 2)); CHECK(hipMemset(w, 0, 64));
    for (int b : {1, 2, 4, 8}) {
        run<0>("v_fma_f32 vgpr", b, out, w);
        run<1>("v_fmac_f32 sgpr", b, out, w);
        run<2>("v_pk_fma_f32 vgpr", b, out, w);
        run<3>("v_pk_fma_f32 sgpr-pair", b, out, w);
    }
    return 0;
}

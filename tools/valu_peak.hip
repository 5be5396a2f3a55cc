// Micro-benchmark: sustained fp32 FMA rate on gfx950 for the instruction forms the conv kernels can use.
// Reports BOTH the wall-clock rate (TFLOP/s) and the in-kernel issue cost (shader cycles per wave-instruction per SIMD,
// from s_memtime of one wave per workgroup), so a clock drop under load and an issue limit can be told apart.
//   mode 0  v_fma_f32  acc, x0, x1, acc          (VOP3, 64-bit encoding, three VGPR sources)
//   mode 1  v_fmac_f32 acc, x0, x1               (VOP2, 32-bit encoding)
//   mode 2  v_fmac_f32 acc, s, x                 (VOP2 with an SGPR operand)
//   mode 3  v_pk_fma_f32 acc2, xx, yy, acc2      (two FMAs per lane per instruction)
//   mode 4  v_pk_fma_f32 with an SGPR pair, op_sel broadcast (the form the first conv core emitted)
//   mode 5  v_fmac_f32 with operands spread over four VGPR banks (acc i, x[i & 3], w[(i >> 2) & 3])
// FMAs per loop trip: modes 0,1,2,5: 32 instructions x 1;  modes 3,4: 32 instructions x 2 = 64 (the round-1 tool credited 32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, const float* wsrc, int iters, unsigned long long* cyc) {
    float a[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = threadIdx.x * 1e-3f + i;
    float x[4], w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[i] = out[threadIdx.x + i]; w[i] = out[threadIdx.x + 4 + i]; }
    typedef const float __attribute__((address_space(4)))* cptr;
    cptr wc = (cptr)wsrc;
    float s0 = wc[0], s1 = wc[1];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x[0]), "v"(x[1]));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x[0]), "v"(x[1]));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(s0), "v"(x[0]));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                f2 acc = {a[i], a[i + 1]}; f2 xx = {x[0], x[1]}; f2 yy = {x[2], x[3]};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(xx), "v"(yy));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(xx), "v"(yy));
                a[i] = acc.x; a[i + 1] = acc.y;
            }
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                f2 acc = {a[i], a[i + 1]}; f2 xx = {x[0], x[1]}; f2 ww = {s0, s1};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(xx), "s"(ww));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(xx), "s"(ww));
                a[i] = acc.x; a[i + 1] = acc.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x[i & 3]), "v"(w[(i >> 2) & 3]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + s0 + s1;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks_per_cu, float* out, float* w, unsigned long long* cyc) {
    int iters = 4000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, w, iters, cyc);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, w, iters, cyc);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int per_instr = (MODE == 3 || MODE == 4) ? 2 : 1;
    double instr = (double)iters * 32;                              // wave-instructions per wave
    double fma = (double)blocks * 256 * instr * per_instr;          // scalar FMAs over the launch
    std::vector<unsigned long long> h(blocks);
    CHECK(hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0; for (auto c : h) mean += (double)c; mean /= blocks;
    // waves per SIMD = blocks_per_cu (a 256-thread block puts one wave on each SIMD); s_memtime ticks at 100 MHz on gfx9
    // (constant clock), so convert through the wall time instead: cycles per instruction per SIMD = time / (instr * waves/SIMD)
    double ns_per_instr_simd = (double)ms * 1e6 / (instr * blocks_per_cu);
    printf("%-34s waves/SIMD=%d  %.3f ms  %7.1f TFLOP/s  %.3f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)  memtime/instr %.3f\n", name,
           blocks_per_cu, ms, 2 * fma / ms / 1e9, ns_per_instr_simd, ns_per_instr_simd * 2.4, mean / instr);
}

int main() {
    float *out, *w; unsigned long long* cyc;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 4 * 2)); CHECK(hipMalloc(&w, 64)); CHECK(hipMalloc(&cyc, 256 * 8 * 8));
    CHECK(hipMemset(out, 0, 256 * 8 * 256 * 4 * 2)); CHECK(hipMemset(w, 0, 64));
    for (int b : {1, 2, 4, 8}) {
        run<0>("v_fma_f32 vgpr (VOP3)", b, out, w, cyc);
        run<1>("v_fmac_f32 vgpr (VOP2)", b, out, w, cyc);
        run<2>("v_fmac_f32 sgpr", b, out, w, cyc);
        run<3>("v_pk_fma_f32 vgpr", b, out, w, cyc);
        run<4>("v_pk_fma_f32 sgpr-pair", b, out, w, cyc);
        run<5>("v_fmac_f32 vgpr, spread banks", b, out, w, cyc);
    }
    return 0;
}

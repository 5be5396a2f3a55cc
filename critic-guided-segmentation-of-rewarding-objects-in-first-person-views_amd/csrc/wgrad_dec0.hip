// Weight gradient of dec_model.0 (cat(e0, Upsample(o1)): 16 -> 8 channels at 32x32, nets.py:480,516-517) as OUTER PRODUCTS on
// v_mfma_f32_4x4x1_16B_f32 (round 3).  Without broadcast the instruction is 16 independent 4x4 outer products: block b = pixel b
// of a 16-pixel row segment, A = four input channels of X at a tap, B = four channels of dY:
//     acc[tap][cog] (lane 4b + j, register r) += X[p_b + tap][4w + r] * dY[p_b][4 cog + j]
// Every multiply is useful: the 16x16x4 implicit GEMM this replaces (wgrad_body.h) runs with half of each 16-wide tile empty at 8
// output channels -- 80 MFMA cycles per pixel against 36 here.  Wave w owns input channels 4w .. 4w+3 (18 accumulators), all four
// waves walk every pixel of the tile; the 16 block sums are added once per persistent workgroup (DPP rotates + two shuffles).
// LDS: X tile with a 20-float pixel slot and dY tile with a 12-float slot: the 8 pixels x 4 dwords a 32-lane group reads fall on
// 32 different banks.  The next tile's global loads are issued right after the commit and fly during the MFMAs.
#include "wgrad_dec0.h"

__global__ void __launch_bounds__(256) wgrad_dec0_kernel(WDec0Params P) {
    __shared__ __attribute__((aligned(16))) float smem[kWD0LdsFloats];
    wgrad_dec0_body(P, blockIdx.x, gridDim.x, smem);
}

int wgrad_dec0_slabs(int n) {
    const int tiles = n * kStrips;
    return tiles < 512 ? tiles : 512;
}

int wgrad_dec0_launch(int n, const float* e0, const float* o1, const float* dy, float* slab, hipStream_t st) {
    if (n <= 0) return CGS_OK;
    WDec0Params P{e0, o1, dy, slab, n, n * kStrips};
    hipLaunchKernelGGL(wgrad_dec0_kernel, dim3(wgrad_dec0_slabs(n)), dim3(256), 0, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

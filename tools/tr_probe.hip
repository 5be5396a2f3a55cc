// What does ds_read_b64_tr_b16 deliver?  LDS image [row][16 columns] of u16 = row * 100 + column; every lane of a 16-lane group g
// supplies the address of row (4 g' + q), columns 4 p (q = (lane & 15) >> 2, p = lane & 3) and prints the four values it receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4_t lds_short4_t;
__global__ void probe(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 16];
    for (int i = threadIdx.x; i < 64 * 16; i += 64) lds[i] = (short)((i / 16) * 100 + (i % 16));
    __syncthreads();
    const int lane = threadIdx.x, l15 = lane & 15, g = lane >> 4, q = l15 >> 2, p = l15 & 3;
    short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)(lds + (8 * g + q) * 16 + 4 * p));
    out[lane * 4 + 0] = v.x; out[lane * 4 + 1] = v.y; out[lane * 4 + 2] = v.z; out[lane * 4 + 3] = v.w;
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    return 0;
}

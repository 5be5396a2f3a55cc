// Cycles per ds_read_b128 (one wave alone on its CU) for a given lane -> float4-slot pattern: which address patterns the LDS of
// gfx950 serves without bank conflicts.  Patterns: the 4x4 patch reads of conv_tile.h (lane = 2x2 quad: slots 2 qx + dx of tile
// row 2 qy + dy, padded row length PWA, one pad slot per 16 columns) and candidates.
// Build: hipcc -O3 --offload-arch=gfx950 tools/lds_pattern.hip -o tools/lds_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

__global__ void __launch_bounds__(64) probe(const int* slots, unsigned long long* cyc, float* sink) {
    __shared__ float4 lds[4096];
    const int l = threadIdx.x;
    for (int e = l; e < 4096; e += 64) lds[e] = make_float4(e, 1, 2, 3);
    __syncthreads();
    const int s = slots[blockIdx.x * 64 + l] & 2047;
    const float4* p = lds + s;
    float4 acc = make_float4(0, 0, 0, 0);
    // warm
    acc.x += p[0].x;
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned addr = (unsigned)(size_t)p;      // LDS byte address of this lane's slot
#pragma unroll 1
    for (int it = 0; it < 32; ++it) {
        // 16 reads that nothing consumes inside the loop (the vector ALU stays out of the measurement); offsets k * 1024 B = same banks
        float4 v0, v1, v2, v3;
        asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
                     "ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168\n"
                     "ds_read_b128 %0, %4 offset:8192\n ds_read_b128 %1, %4 offset:9216\n ds_read_b128 %2, %4 offset:10240\n ds_read_b128 %3, %4 offset:11264\n"
                     "ds_read_b128 %0, %4 offset:12288\n ds_read_b128 %1, %4 offset:13312\n ds_read_b128 %2, %4 offset:14336\n ds_read_b128 %3, %4 offset:15360\n"
                     "s_waitcnt lgkmcnt(0)\n"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(addr));
        acc.x += v0.x + v1.x + v2.x + v3.x;
    }
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + l] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    std::vector<std::vector<int>> pats;
    std::vector<std::string> names;
    auto add = [&](const std::string& n, auto f) { std::vector<int> v(64); for (int l = 0; l < 64; ++l) v[l] = f(l); pats.push_back(v); names.push_back(n); };
    add("contiguous slot = l", [](int l) { return l; });
    add("stride 2", [](int l) { return 2 * l; });
    add("stride 2, lanes 8-15 +1 per 8", [](int l) { return 2 * l + (l >> 3); });
    add("stride 4", [](int l) { return 4 * l; });
    add("lanes i and i+8 same slot row (256 B apart)", [](int l) { return (l & 7) * 2 + (l >> 3) * 16; });
    add("lanes i and i+16 256 B apart, i+8 +1", [](int l) { return (l & 7) * 2 + ((l >> 3) & 1) + (l >> 4) * 16; });
    add("all lanes same slot (broadcast)", [](int) { return 5; });
    add("lane pairs same slot", [](int l) { return l >> 1; });
    auto pc = [](int c) { return c + (c >> 4); };
    // the patch reads of conv_tile.h: lane = quad (qx = l % (W/2), quad row l / (W/2)), column 2 qx + dx of tile row 2 qy + dy
    for (int W : {16, 32, 64})
        for (int pwa = W + 3; pwa <= W + 14; ++pwa)
            for (int dx = 0; dx < 4; ++dx)
                add("W=" + std::to_string(W) + " PWA=" + std::to_string(pwa) + " dx=" + std::to_string(dx),
                    [=](int l) { return 2 * pwa * (l / (W / 2)) + pc(2 * (l % (W / 2)) + dx); });
    // W = 32 at the LDS budget of four workgroups per CU (PWA <= 37): where the pad slot sits, pc_k(c) = c + ((c + k) >> 4)
    for (int pwa = 35; pwa <= 37; ++pwa)
        for (int k = 0; k < 16; ++k)
            for (int dx = 0; dx < 4; ++dx)
                add("W=32 PWA=" + std::to_string(pwa) + " k=" + std::to_string(k) + " dx=" + std::to_string(dx),
                    [=](int l) { int c = 2 * (l % 16) + dx; return 2 * pwa * (l / 16) + c + ((c + k) >> 4); });
    // one pad slot at column P (pc(c) = c + (c >= P)): 35 slots per row hold the 34 tile columns
    for (int pwa = 35; pwa <= 36; ++pwa)
        for (int P = 2; P <= 32; P += 2)
            for (int dx = 0; dx < 4; ++dx)
                add("W=32 PWA=" + std::to_string(pwa) + " P=" + std::to_string(P) + " dx=" + std::to_string(dx),
                    [=](int l) { int c = 2 * (l % 16) + dx; return 2 * pwa * (l / 16) + c + (c >= P ? 1 : 0); });
    // two pads at P1 < P2, 36 slots
    for (int P1 = 6; P1 <= 14; P1 += 2)
        for (int P2 = 20; P2 <= 30; P2 += 2)
            for (int dx = 0; dx < 4; ++dx)
                add("W=32 PWA=36 P1=" + std::to_string(P1) + ",P2=" + std::to_string(P2) + " dx=" + std::to_string(dx),
                    [=](int l) { int c = 2 * (l % 16) + dx; return 2 * 36 * (l / 16) + c + (c >= P1 ? 1 : 0) + (c >= P2 ? 1 : 0); });
    const int np = (int)pats.size();
    int* d_s; unsigned long long* d_c; float* d_k;
    hipMalloc(&d_s, np * 64 * 4); hipMalloc(&d_c, np * 8); hipMalloc(&d_k, np * 64 * 4);
    std::vector<int> flat;
    for (auto& v : pats) flat.insert(flat.end(), v.begin(), v.end());
    hipMemcpy(d_s, flat.data(), np * 64 * 4, hipMemcpyHostToDevice);
    // at most 128 patterns per launch: one wave alone on its CU (waves that share a CU's LDS would disturb each other)
    for (int rep = 0; rep < 2; ++rep)
        for (int b0 = 0; b0 < np; b0 += 128) {
            hipLaunchKernelGGL(probe, dim3(np - b0 < 128 ? np - b0 : 128), dim3(64), 0, 0, d_s + b0 * 64, d_c + b0, d_k + b0 * 64);
            hipDeviceSynchronize();
        }
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(np);
    hipMemcpy(c.data(), d_c, np * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < np; ++i) {
        if (names[i].rfind("W=", 0) == 0 && names[i].find("dx=0") != std::string::npos && i + 3 < np) {
            printf("%-28s dx 0..3: %6.2f %6.2f %6.2f %6.2f\n", names[i].substr(0, names[i].find(" dx")).c_str(), (double)c[i] / 512, (double)c[i + 1] / 512,
                   (double)c[i + 2] / 512, (double)c[i + 3] / 512);
            i += 3;
        } else printf("%-52s %7.2f ticks per ds_read_b128\n", names[i].c_str(), (double)c[i] / (32 * 16));
    }
    return 0;
}

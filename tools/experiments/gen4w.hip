// NOT PART OF THE PRODUCT (round 3 experiment, kept for the record): measured slower than the 16x16x4 weight gradient of
// gen_train.hip -- chfak 5, n = 512: 3461 vs 2820 us over the nine layer shapes in isolation (tools/time_genw.py at that commit),
// 3.93 vs 3.37 ms inside the phase-2 step.  Both forms sit at ~45 % of their own MFMA issue floor: a staged 256-pixel tile feeds
// too few matrix instructions per barrier pair whichever way the accumulators are laid out.
// Shape-generic weight gradient of a 3x3 layer on v_mfma_f32_4x4x1_16B_f32 as OUTER PRODUCTS (round 3).  Replaces the 16x16x4
// GEMM of gen_train.hip behind cgs_gen_conv3x3_bwd_weight: backward of nets.py:170-183, 480-489 for any chfak / neck.
//
// Without broadcast the instruction is 16 independent 4x4 outer products.  Here (the "tap-block" form of tail4.h, run-time sizes)
// block c of register set G = combination 16 G + c = (tap, group of 4 input channels) of a 16-channel input chunk, one PIXEL per
// step:   acc[G][cog] (lane 4 c + j, register r) += X[p + tap][4 cig + r] * dY[p][4 cog + j].
// A block accumulates ITS weights over the pixels: nothing is summed across blocks, and 16 input x 40 output channels x 9 taps
// are 3 x 10 accumulators = 120 registers (the pixel-block form of wgrad_dec0.hip needs 9 x 10 x 4 = 360 for the same block,
// measured 786 us on the 40 -> 40 layer at 32x32 with the output channels split in two passes: a staged tile then feeds too few
// instructions).  36 of the 48 block slots of a full chunk are used.
// Workgroup = 4 waves, each walks 64 of the tile's 256 pixels for ALL combinations of the chunk and <= 10 output groups; grid =
// image shares x input chunks x output passes; one slab row per share (cgs_reduce_slabs sums them).
// LDS: X tile [rows + halo][W + 2][20 floats], dY tile [256 pixels][4 COG + 4 floats]; the A operand is one dword per lane at a
// per-lane constant offset from the pixel, the B operand the same four floats in every block (broadcast reads).
#include "gen4_common.h"

namespace {

struct Gen4WParams {
    GenSrc in;          // the layer's input cat(A, up(B))
    GenSrc dy;          // gradient at the pre-activation output: GEN_SRC_F32 [n,hw,hw,co] or GEN_SRC_POOLEXP (dE + argmax); ca = co
    float* slab;        // [G][9 * ci_total * co + co]
    int n, hw, lw, imgs, th, G, ncib, ncop;
};

constexpr int G4W_SLOT = 20;           // floats per pixel slot of both tiles

__device__ __forceinline__ float g4w_block_sum16(float v) {      // sum over the 16 blocks (lanes with equal lane & 3)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false));     // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));     // row_ror:8
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

template <int COG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(COG <= 5 ? 3 : 2, COG <= 5 ? 3 : 2))) gen4_wgrad_kernel(Gen4WParams P) {
    extern __shared__ __attribute__((aligned(16))) float4 g4wsm[];
    constexpr int YS = 4 * COG + 4;                                  // floats per pixel of the dY tile
    constexpr int NCALL = (COG + 3) / 4;                             // staging calls of <= 4 planes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = lane >> 2, li = lane & 3;
    const int H = P.hw, W = P.hw, lw = P.lw, PWX = W + 2;
    const int cop = blockIdx.x % P.ncop, cib = (blockIdx.x / P.ncop) % P.ncib, share = blockIdx.x / (P.ncop * P.ncib);
    const int co = P.dy.ca, ci_total = P.in.ca + P.in.cb;
    const int rowsx = P.imgs * (P.th + 2);
    float* xt = (float*)g4wsm;                                       // [rowsx][PWX][20]
    float* yt = xt + rowsx * PWX * G4W_SLOT;                         // [256 pixels][YS]
    const G4Geo geo{P.n, P.hw, P.lw, P.imgs, P.th};

    // this chunk's 4-channel groups of the input (padded cat space) and the lane's (tap, group) combination per register set
    const int cpx = gen_pa4(P.in) + P.in.cb;
    const int remx = cpx - cib * GEN_KC, ncig = remx >= GEN_KC ? 4 : (remx + 3) >> 2;
    const int ncomb = 9 * ncig, nset = (ncomb + 15) >> 4;            // 1 .. 3 register sets
    const int cog0 = cop * COG;                                      // first output group of this pass
    const int ngt = (co + 3) >> 2;
    int offa[3];
#pragma unroll
    for (int G = 0; G < 3; ++G) {
        const int c = 16 * G + blk, cc = c < ncomb ? c : 0, tap = cc / ncig, cg = cc - tap * ncig;
        offa[G] = (((tap / 3) * PWX + tap % 3) * G4W_SLOT + 4 * cg + li) * 4;      // bytes
    }

    // zero halo columns of the X tile (the staging writes interior columns only), once
    for (int e = tid; e < rowsx * 2 * 5; e += 256) {
        const int q = e % 5, side = (e / 5) & 1, r = e / 10;
        *(float4*)(xt + (r * PWX + (side ? W + 1 : 0)) * G4W_SLOT + 4 * q) = f4zero();
    }

    frag4 acc[3][COG];
#pragma unroll
    for (int G = 0; G < 3; ++G)
#pragma unroll
        for (int g = 0; g < COG; ++g) acc[G][g] = frag4{0.f, 0.f, 0.f, 0.f};
    float4 bs[NCALL];                            // bias gradient: this thread's dY float4s, per staging call
#pragma unroll
    for (int c = 0; c < NCALL; ++c) bs[c] = f4zero();

    const int strips = P.imgs == 1 ? H / P.th : 1;
    const int ntiles = P.imgs == 1 ? P.n * strips : (P.n + P.imgs - 1) / P.imgs;
    const int RL = W < 16 ? W : 16;              // pixels of a run (consecutive in a row)
    const int lth = P.th == 4 ? 2 : (P.th == 8 ? 3 : 4);

    for (int tl = share; tl < ntiles; tl += P.G) {
        const int img0 = P.imgs == 1 ? tl / strips : tl * P.imgs;
        const int row0 = P.imgs == 1 ? (tl % strips) * P.th : 0;
        int ltid = tid;
        asm volatile("" : "+v"(ltid));
        gen4_stage_any<3>(G4Dst{(float4*)xt, 1, PWX * 5, 5, 1}, P.in, geo, 1, img0, row0, cib * GEN_KC, 4, ltid);
#pragma unroll
        for (int c = 0; c < NCALL; ++c)
            gen4_stage_sum<3>(G4Dst{(float4*)yt + 4 * c, 1, W * (YS / 4), YS / 4, 0}, P.dy, geo, 0, img0, row0, 4 * cog0 + 16 * c,
                              COG - 4 * c < 4 ? COG - 4 * c : 4, ltid, bs[c]);
        __syncthreads();
        // this wave's 64 pixels [64 wave, 64 wave + 64) of the tile (tile-linear: image parts one after the other, row-major)
        auto walk = [&](auto NSC) {
            constexpr int NS = decltype(NSC)::value;
            int lo[3] = {offa[0], offa[1], offa[2]};
            asm volatile("" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]));      // (keeps the per-run addresses out of registers held across the tile loop)
            const int lj = li * 4;
#pragma unroll 1
            for (int p0 = 64 * wave; p0 < 64 * wave + 64; p0 += RL) {
                const int pin = p0 & ((P.th << lw) - 1), il = p0 >> (lw + lth), y = pin >> lw, x = pin & (W - 1);
                const int xb = ((il * (P.th + 2) + y) * PWX + x) * G4W_SLOT * 4;      // bytes: top-left of the 3x3 window of the run's first pixel
                const char* xp = (const char*)xt + xb;
                const char* yp = (const char*)yt + p0 * YS * 4 + lj;
                float a0[NS], a1[NS], b0[COG], b1[COG];
                auto rd = [&](float (&a)[NS], float (&bv)[COG], int k) {
#pragma unroll
                    for (int G = 0; G < NS; ++G) a[G] = *(const float*)(xp + lo[G] + k * (G4W_SLOT * 4));
#pragma unroll
                    for (int g = 0; g < COG; ++g) bv[g] = *(const float*)(yp + k * (YS * 4) + 16 * g);
                };
                auto mm = [&](const float (&a)[NS], const float (&bv)[COG]) {
#pragma unroll
                    for (int G = 0; G < NS; ++G)
#pragma unroll
                        for (int g = 0; g < COG; ++g) acc[G][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[G], bv[g], acc[G][g], 0, 0, 0);
                };
                rd(a0, b0, 0);
#pragma unroll 1
                for (int k = 0; k < RL; k += 2) {
                    rd(a1, b1, k + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (k + 2 < RL) rd(a0, b0, k + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(a1, b1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (nset == 3) walk(std::integral_constant<int, 3>{});
        else if (nset == 2) walk(std::integral_constant<int, 2>{});
        else walk(std::integral_constant<int, 1>{});
        __syncthreads();
    }

    // ---- slab row of this share: rows tap * ci_total + ci, columns co.  The four waves' partial blocks are added through LDS, one
    //      register set at a time: [wave][cog][r][lane] ----
    float* row = P.slab + (size_t)share * ((size_t)9 * ci_total * co + co);
    float* red = (float*)g4wsm;
    __syncthreads();                             // (a share without tiles: the halo zeroes above are still in flight)
    for (int G = 0; G < 3; ++G) {
        if (G) __syncthreads();
        if (G < nset) {
#pragma unroll
            for (int g = 0; g < COG; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (constant register indices: a set picked by a run-time G would move the accumulators to scratch)
                    const float v = G == 0 ? acc[0][g][r] : (G == 1 ? acc[1][g][r] : acc[2][g][r]);
                    red[((wave * COG + g) * 4 + r) * 64 + lane] = v;
                }
        }
        __syncthreads();
        if (G < nset) {
            for (int e = tid; e < COG * 4 * 64; e += 256) {
                const int ln = e & 63, r = (e >> 6) & 3, g = e >> 8;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += red[((w * COG + g) * 4 + r) * 64 + ln];
                const int c = 16 * G + (ln >> 2);
                if (c < ncomb) {
                    const int tap = c / ncig, cg = c - tap * ncig;
                    const int ci = gen_real_channel(P.in, cib * GEN_KC + 4 * cg + r), col = 4 * (cog0 + g) + (ln & 3);
                    if (ci >= 0 && col < co) row[((size_t)tap * ci_total + ci) * co + col] = v;
                }
            }
        }
    }
    if (cib == 0) {
        // bias: call c staged planes 4 c .. of the pass; plane g of a call was summed by the threads with (tid & (2^lp - 1)) == g
        __syncthreads();
        float4* bred = (float4*)red;
#pragma unroll
        for (int c = 0; c < NCALL; ++c) bred[c * 256 + tid] = bs[c];
        __syncthreads();
        if (tid < 4 * COG) {
            const int g = tid >> 2, ch = tid & 3, call = g >> 2, gl = g & 3, col = 4 * (cog0 + g) + ch;
            int npa = ngt - (cog0 + 4 * call);                       // planes that call staged
            const int cap = COG - 4 * call < 4 ? COG - 4 * call : 4;
            npa = npa < cap ? npa : cap;
            const int lp = npa <= 1 ? 0 : (npa == 2 ? 1 : 2);
            if (gl < npa && col < co) {
                float v = 0.f;
                for (int k = gl; k < 256; k += (1 << lp)) v += f4get(bred[call * 256 + k], ch);
                row[(size_t)9 * ci_total * co + col] = v;
            }
        }
    }
}

}  // namespace

struct Gen4WLaunch {
    GenSrc in, dy; float* slab; int n, hw;
};

static void gen4w_split(int ca, int cb, int co, int& ncib, int& ncop, int& cog) {
    const int cp = ((ca + 3) & ~3) + cb, ngt = (co + 3) / 4;
    ncib = (cp + GEN_KC - 1) / GEN_KC;
    ncop = (ngt + 7) / 8;
    const int per = (ngt + ncop - 1) / ncop;
    cog = per <= 1 ? 1 : per <= 2 ? 2 : per <= 4 ? 4 : per <= 5 ? 5 : per <= 6 ? 6 : 8;
}

int gen4_wgrad_shares(int n, int ca, int cb, int co) {      // image shares (= slab rows): two workgroups per CU in flight
    int ncib, ncop, cog;
    gen4w_split(ca, cb, co, ncib, ncop, cog);
    int g = (768 + ncib * ncop - 1) / (ncib * ncop);
    return g < n ? g : n;
}

int gen4_wgrad_launch(const Gen4WLaunch& L, hipStream_t st) {
    Gen4WParams P{};
    P.in = L.in; P.dy = L.dy; P.slab = L.slab; P.n = L.n; P.hw = L.hw;
    const int hw = L.hw;
    P.lw = hw == 64 ? 6 : hw == 32 ? 5 : hw == 16 ? 4 : hw == 8 ? 3 : 2;
    P.imgs = hw >= 16 ? 1 : (hw == 8 ? 4 : 16);
    P.th = hw >= 16 ? 256 / hw : hw;
    int cog;
    gen4w_split(L.in.ca, L.in.cb, L.dy.ca, P.ncib, P.ncop, cog);
    P.G = gen4_wgrad_shares(L.n, L.in.ca, L.in.cb, L.dy.ca);
    const size_t tiles = ((size_t)P.imgs * (P.th + 2) * (hw + 2) * G4W_SLOT + (size_t)256 * (4 * cog + 4)) * sizeof(float);
    const size_t red = (size_t)4 * cog * 4 * 64 * sizeof(float);
    const size_t lds = tiles > red ? tiles : red;
    const dim3 grid(P.G * P.ncib * P.ncop);
#define G4W_LAUNCH(C_)                                                                                                   \
    do {                                                                                                                 \
        if (lds > 64 * 1024) {                                                                                           \
            static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gen4_wgrad_kernel<C_>),          \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);         \
            if (attr != hipSuccess) return (int)attr;                                                                    \
        }                                                                                                                \
        hipLaunchKernelGGL(gen4_wgrad_kernel<C_>, grid, dim3(256), lds, st, P);                                          \
    } while (0)
    switch (cog) {
        case 1: G4W_LAUNCH(1); break;
        case 2: G4W_LAUNCH(2); break;
        case 4: G4W_LAUNCH(4); break;
        case 5: G4W_LAUNCH(5); break;
        case 6: G4W_LAUNCH(6); break;
        default: G4W_LAUNCH(8); break;
    }
#undef G4W_LAUNCH
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

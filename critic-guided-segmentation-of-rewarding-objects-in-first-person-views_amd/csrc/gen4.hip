// Shape-generic 3x3 convolution (forward and data gradient) on v_mfma_f32_4x4x1_16B_f32 with lane = PIXEL (round 3).
//
// Replaces the 16x16x4 implicit GEMM of gen.hip (round 2) behind the same C entry points (cgs_gen_conv3x3_fwd,
// cgs_gen_conv3x3_bwd_data): NewCritic / UnetDecoder at chfak != 1 or neck != 32 (nets.py:161-212, 453-523) and the legacy Unet.
//
// The instruction with its A operand broadcast from one block (cbsz = 4, abid = block) computes
//     D[lane][r] += A[4 * abid + r] * B[lane]                 (64 lanes x 4 results x 1 k-step, 512 FLOP in 8 cycles),
// so with lane = pixel, B = the pixel's input value of one (tap, channel) step and A = the four weights of one output-channel
// group the GEMM's granularity is 4 output channels x 1 k-step x 64 pixels: nothing is padded at 40 channels (the 16-wide tiles
// of the 16x16x4 form run 48 columns for 40), at 3 input channels (27 k-steps, not 36) or at 3 output channels (the data gradient
// of the image layer: one group of 4, not a 16-wide tile).  A weight REGISTER holds 16 steps (lane 4 * cin + i = channel cin of
// the staged 16-channel chunk, output channel 4 g + i): one conflict-free ds_read_b32 per (tap, group); one ds_read_b128 of the
// input tile (channel-planar float4 slots) feeds 4 steps x NG groups = up to 40 instructions.
//
// Workgroup = 4 waves = 256 pixels in quad order (lane = 4 * quad + 2 * dy + dx: MaxPool2d(2) is two DPP quad permutes):
//   hw = 64: 4 rows of one image;  32: 8 rows;  16: one image;  8: four images;  4: sixteen images.
// Per 16-channel chunk of the input: stage the tile -> barrier -> 9 taps x <= 16 channels x NG groups of MFMAs -> barrier.
// The weights never pass through LDS: cgs_gen_conv_pack_weights lays them out as register images ([chunk][tap][group][64 lanes],
// zero for padding channels / columns) and a wave loads the NG registers of the NEXT tap with one coalesced 256-byte load each
// while the current tap multiplies (the images are L2-resident and shared by every wave of the launch).
// Output channels beyond 4 * NG (NG <= 10) are further passes = further workgroups over the same tile (blockIdx picks the pass).
// Epilogue: bias + activation (+ MaxPool2d(2) and the argmax byte), then through LDS so that a workgroup's output -- one
// contiguous block of NHWC memory -- leaves in full lines (the lane = pixel registers would store 16-byte pieces 4 * co bytes apart).
#include "gen4_common.h"

namespace {

struct Gen4Params {
    GenSrc src; const float* wp; const float* bias;     // wp: packed weights (gen4_pack_kernel); bias may be NULL (data gradient)
    float* out; uint8_t* argmax;
    const float* addend; int n_addend;                  // pool = 0: out += addend for images < n_addend (same shape as out)
    int n, hw, co, act, pool;
    float slope;
    int imgs, th, pw, rows, lw, npass, ngt;             // ngt: output-channel groups of 4 of the whole layer
    int dbuf;                                           // 1: two tile buffers (chunk c + 1 is staged while chunk c multiplies)
    // split (data gradient of a layer over cat(A [split_ca], nearest-up(B))): passes below split_ca write d_a = out with row stride
    // split_ca, passes above it write the cell sums (the upsample's backward) to out2 [n, hw >> split_ush, hw >> split_ush, co - split_ca]
    float* out2; int split_ca, split_ush;
    unsigned long long* dbg;                            // debug: per-workgroup phase stamps (tools/gen4_stamps.py), NULL in the product path
};
unsigned long long* g_gen4_stamps = nullptr;
#define G4_STAMP(k) do { if (CGS_STAMP_PTR(P.dbg) && tid == 0 && blockIdx.x < 4096) CGS_STAMP_PTR(P.dbg)[(size_t)blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)

// ---- weights as register images: wp[((chunk * 9 + tap) * ngp + g) * 64 + 4 * cin + i] = w[tap][channel cin of the chunk][4 g + i],
// ngp = passes x groups per pass (gen4_groups), zero for the groups past the layer's last ----
// transposed = 0: w = HWIO [9][ca + cb][co] (forward).  transposed = 1 (cb = 0): w = HWIO [9][ci_layer][ca] of the LAYER whose data
// gradient this is (ca = its output channels = dY's), read with the taps reversed; the operand covers the layer's input channels
// [ci_off, ci_off + co) (the whole layer: ci_off = 0, co = ci_layer).
struct Gen4PackParams { const float* w; float* wp; int ca, cb, co, transposed, total, ngp, ci_layer, ci_off; };
// transposed = 2: FOLDED forward operand of a layer over cat(A, nearest-up_2(B)) (gen4_conv3x3_kernel<NG, true>).  A pixel of row / column
// parity (py, px) sees B only through the 2 x 2 neighbourhood (a, b) of its low-resolution cell: the 3 x 3 taps over the upsampled map
// collapse to   Wf[py][px][a][b] = sum_{ky in K(py,a)} sum_{kx in K(px,b)} w[ky][kx],   K(0,0) = {0}, K(0,1) = {1,2}, K(1,0) = {0,1}, K(1,1) = {2}
// (a whole folded tap is inside the map or outside it, like the taps it sums: the zero padding commutes with the fold) -- 4 instead of
// 9 tap steps for every B channel.  Layout: A's chunks first, 9 slots each; then B's chunks, 16 slots each (slot = 4 * (2 py + px) + 2 a + b).
__device__ __forceinline__ float gen4_pack_elem(const Gen4PackParams& P, int e) {
    const int pa4 = (P.ca + 3) & ~3, ngt = P.ngp, ci_total = P.ca + P.cb;
    const int lane = e & 63, r = e >> 6, g = r % ngt, slot = r / ngt, cin = lane >> 2, col = 4 * g + (lane & 3);
    if (P.transposed == 2) {
        const int ncha = (pa4 + GEN_KC - 1) / GEN_KC;
        if (col >= P.co) return 0.f;
        if (slot < 9 * ncha) {
            const int tap = slot % 9, k = (slot / 9) * GEN_KC + cin;
            return k < P.ca ? P.w[((size_t)tap * ci_total + k) * P.co + col] : 0.f;
        }
        const int idx = slot - 9 * ncha, kb = (idx >> 4) * GEN_KC + cin, py = (idx >> 3) & 1, px = (idx >> 2) & 1, a = (idx >> 1) & 1, b = idx & 1;
        if (kb >= P.cb) return 0.f;
        const int ky0 = py ? (a ? 2 : 0) : (a ? 1 : 0), ky1 = py ? (a ? 2 : 1) : (a ? 2 : 0);
        const int kx0 = px ? (b ? 2 : 0) : (b ? 1 : 0), kx1 = px ? (b ? 2 : 1) : (b ? 2 : 0);
        float v = 0.f;
        for (int ky = ky0; ky <= ky1; ++ky)
            for (int kx = kx0; kx <= kx1; ++kx) v += P.w[((size_t)(ky * 3 + kx) * ci_total + P.ca + kb) * P.co + col];
        return v;
    }
    if (P.transposed == 3) {
        // the data gradient towards the nearest-upsampled source [ci_off, ci_off + co) of a layer w = HWIO [9][ci_layer][ca] (ca = dY's channels):
        // gen4_conv3x3_kernel<NG, 2> over the space-to-depth view of dY (gen4_stage_s2d).  Chunk c lies in parity block (py, px); the cell
        // (Y + ty, X + tx) of that block, ty = (ab >> 1) - py, reaches output cell (Y, X) through the taps ky with ((2 (Y + ty) + py) + ky - 1) >> 1 == Y:
        // ty = -1: {2}, ty = +1: {0}, ty = 0: py ? {0, 1} : {1, 2} (columns likewise) -- 16 (block, cell) pairs instead of 4 x 9 pixel taps per cell.
        const int pbw = (P.ca + GEN_KC - 1) / GEN_KC * GEN_KC, c = slot >> 2, ab = slot & 3, par = (c * GEN_KC) / pbw;
        const int kk = c * GEN_KC - par * pbw + cin, py = par >> 1, px = par & 1, ty = (ab >> 1) - py, tx = (ab & 1) - px;
        if (kk >= P.ca || col >= P.co) return 0.f;
        const int ky0 = ty < 0 ? 2 : (ty > 0 ? 0 : (py ? 0 : 1)), ky1 = ty < 0 ? 2 : (ty > 0 ? 0 : (py ? 1 : 2));
        const int kx0 = tx < 0 ? 2 : (tx > 0 ? 0 : (px ? 0 : 1)), kx1 = tx < 0 ? 2 : (tx > 0 ? 0 : (px ? 1 : 2));
        float v = 0.f;
        for (int ky = ky0; ky <= ky1; ++ky)
            for (int kx = kx0; kx <= kx1; ++kx) v += P.w[((size_t)(ky * 3 + kx) * P.ci_layer + P.ci_off + col) * P.ca + kk];
        return v;
    }
    const int tap = slot % 9, k = (slot / 9) * GEN_KC + cin;
    const int real = k < pa4 ? (k < P.ca ? k : -1) : (k - pa4 < P.cb ? P.ca + (k - pa4) : -1);
    float v = 0.f;
    if (real >= 0 && col < P.co)
        v = P.transposed ? P.w[((size_t)(8 - tap) * P.ci_layer + P.ci_off + col) * P.ca + real] : P.w[((size_t)tap * ci_total + real) * P.co + col];
    return v;
}
__global__ void __launch_bounds__(256) gen4_pack_kernel(Gen4PackParams P) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < P.total; e += gridDim.x * 256) P.wp[e] = gen4_pack_elem(P, e);
}

// every layer's operand of one training step in ONE launch (25 launches of ~4.7 us each otherwise): blockIdx.y = job
constexpr int G4_PACK_BATCH = 32;
struct Gen4PackBatch { Gen4PackParams job[G4_PACK_BATCH]; };
__global__ void __launch_bounds__(256) gen4_pack_batch_kernel(Gen4PackBatch B) {
    const Gen4PackParams& P = B.job[blockIdx.y];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < P.total; e += gridDim.x * 256) P.wp[e] = gen4_pack_elem(P, e);
}

// FOLD = 1 (a layer over cat(A, nearest-up_2(B)), no pooling, no split): A's channels run as usual; B's are staged at B's OWN resolution (a
// quarter of the elements, no upsampling in the loader) and multiplied through the folded taps (gen4_pack_elem): 4 instead of 9 tap steps per
// channel.  The folded weights depend on the pixel's parity, the A operand of the instruction is one register for the whole wave: wave w
// takes position w of every 2 x 2 cell of the tile (lane = cell), not 64 consecutive pixels.
// FOLD = 2: the DATA gradient towards the upsampled source, computed at the source's resolution: input = the space-to-depth view of dY
// (gen4_stage_s2d: 4 parity blocks of S.cb >= S.ca channels), output = one low-resolution cell per lane; a 16-channel chunk lies in one parity
// block and meets 4 of the 9 cell offsets (gen4_pack_elem, transposed = 3): 16 (block, offset) steps per channel and cell where the full-resolution
// form (9 taps per pixel, then the 2 x 2 cell sum in the epilogue) runs 36.
// workgroups (= waves per SIMD) a CU is meant to hold: NG >= 8 (32 / 40 accumulator registers) and the narrower instances
#ifndef G4_WPE_BIG
#define G4_WPE_BIG 3
#endif
#ifndef G4_WPE_SMALL
#define G4_WPE_SMALL 4
#endif
// VEC = 1 (round 6; launcher: FOLD = 0, NG >= 8, a vector source -- fp32 with ca % 4 == 0, or the pooled map + argmax): only those loaders are
// instantiated and the chunk loop is peeled (every chunk but the last has 16 channels: ONE tap nest in the loop body, the accumulators stay in one
// register set) -- the instance fits 128 registers, FOUR workgroups per CU with one tile buffer.
#ifndef G4_VEC
#define G4_VEC 1
#endif
#ifndef G4_VEC_BATCH
#define G4_VEC_BATCH 3
#endif
#ifndef G4_VEC_S2D
#define G4_VEC_S2D 1
#endif
#ifndef G4_WPE_VEC
#define G4_WPE_VEC 4
#endif
#ifndef G4_WPE_TINY
#define G4_WPE_TINY 5
#endif
// (the narrow folded instances fit 96 registers: five workgroups per CU)
constexpr bool g4_tiny(int ng, int fold) { return ng <= 4 && fold >= 1; }
template <int NG, int FOLD, bool VEC> constexpr int g4_wpe() { return VEC ? G4_WPE_VEC : (NG >= 8 ? G4_WPE_BIG : (g4_tiny(NG, FOLD) ? G4_WPE_TINY : G4_WPE_SMALL)); }
template <int NG, int FOLD, bool VEC = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(g4_wpe<NG, FOLD, VEC>(), g4_wpe<NG, FOLD, VEC>()))) gen4_conv3x3_kernel(Gen4Params P) {
    if (CGS_KARG_PREFETCH) cgs_kernarg_prefetch<sizeof(Gen4Params)>();
    extern __shared__ __attribute__((aligned(16))) float4 g4sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const GenSrc& S = P.src;
    const int H = P.hw, W = P.hw, lw = P.lw;
    const int pass = blockIdx.x % P.npass, tileid = blockIdx.x / P.npass;
    const int strips = P.imgs == 1 ? H / P.th : 1;
    const int img0 = P.imgs == 1 ? tileid / strips : tileid * P.imgs;
    const int row0 = P.imgs == 1 ? (tileid % strips) * P.th : 0;
    const int pstride = P.rows * P.pw;
    float4* tile = g4sm;
    const G4Geo geo{P.n, P.hw, P.lw, P.imgs, P.th};
    const int pa4 = gen_pa4(S), cp = FOLD == 2 ? 0 : (FOLD ? pa4 : pa4 + S.cb);            // channels of the full-resolution chunks
    const int ncha = (cp + GEN_KC - 1) / GEN_KC, nchb = FOLD == 2 ? (4 * S.cb) / GEN_KC : (FOLD ? (S.cb + GEN_KC - 1) / GEN_KC : 0), nchunk = ncha + nchb;
    // FOLD: B's tile = (th / 2 + 2) x (W / 2 + 2) low-resolution pixels per image part, in the same buffers (A's chunks are done by then)
    const int pwb = (P.hw >> 1) + 2, rowsb = P.imgs * ((P.th >> 1) + 2), pstrideb = rowsb * pwb;
    const G4Geo geob{P.n, P.hw >> 1, P.lw - 1, P.imgs, P.th >> 1};
    auto stage = [&](float4* dst, int ch, int ltid) __attribute__((always_inline)) {      // channel-planar tile: dst[(plane * rows + row) * pw + col], col 0 = left halo
        if constexpr (FOLD == 2) {
            gen4_stage_s2d(G4Dst{dst, pstride, P.pw, 1, 1}, (const float*)S.a, S.ca, S.cb, geo, img0, row0, ch * GEN_KC, ltid);
        } else if constexpr (FOLD == 1) {
            if (ch < ncha) {
                const GenSrc SA{S.a, nullptr, nullptr, S.mode, S.ca, 0, 1};
                gen4_stage_any(G4Dst{dst, pstride, P.pw, 1, 1}, SA, geo, 1, img0, row0, ch * GEN_KC, 4, ltid);
            } else {
                const GenSrc SB{S.b, nullptr, nullptr, GEN_SRC_F32, S.cb, 0, 1};
                gen4_stage<GEN_K_F32V4, false>(G4Dst{dst, pstrideb, pwb, 1, 1}, SB, geob, 1, img0, row0 >> 1, (ch - ncha) * GEN_KC, 4, ltid);
                // the halo columns of THIS layout (the buffer held full-resolution chunks before)
                for (int e = ltid; e < 4 * rowsb * 2; e += 256) dst[(e >> 1) * pwb + ((e & 1) ? pwb - 1 : 0)] = f4zero();
            }
        } else if constexpr (VEC) {
            const G4Dst D{dst, pstride, P.pw, 1, 1};
            if (S.mode == GEN_SRC_POOLEXP) gen4_stage<GEN_K_POOLEXP, false, false, G4_VEC_BATCH>(D, S, geo, 1, img0, row0, ch * GEN_KC, 4, ltid);
            else if (S.cb > 0) gen4_stage<GEN_K_F32V4, true, false, G4_VEC_BATCH>(D, S, geo, 1, img0, row0, ch * GEN_KC, 4, ltid);
            else gen4_stage<GEN_K_F32V4, false, false, G4_VEC_BATCH>(D, S, geo, 1, img0, row0, ch * GEN_KC, 4, ltid);
        } else {
            gen4_stage_any(G4Dst{dst, pstride, P.pw, 1, 1}, S, geo, 1, img0, row0, ch * GEN_KC, 4, ltid);
        }
    };
    const int g0 = pass * NG;

    // this wave's weight registers: register image (chunk, tap, group g0 + g) = 64 consecutive floats; the packed buffer holds
    // npass * NG groups per (chunk, tap) (zero past the layer's last), so group g is an immediate offset of 256 g bytes
    const int ngp = P.npass * NG;
    const float* wlane = P.wp + (size_t)g0 * 64 + lane;
    auto wload = [&](float (&dst)[NG], int ct) __attribute__((always_inline)) {          // ct = chunk * 9 + tap
        const float* q = wlane + (size_t)(ct * ngp) * 64;
#pragma unroll
        for (int g = 0; g < NG; ++g) dst[g] = q[g * 64];
    };
    float w0[NG], w1[NG];
    wload(w0, 0);
    G4_STAMP(0);

    // this lane's pixel
    const int p = FOLD == 1 ? 4 * lane + wave : wave * 64 + lane, q = p >> 2, pos = p & 3;
    const int lqi = (lw - 1) + (P.th == 4 ? 1 : (P.th == 8 ? 2 : 3));        // log2(quads per image part) = log2((th / 2) * (hw / 2))
    const int il = q >> lqi, qi = q & ((1 << lqi) - 1), qy = qi >> (lw - 1), qx = qi & ((W >> 1) - 1);
    const int y = 2 * qy + (pos >> 1), x = 2 * qx + (pos & 1);
    const int base = (il * (P.th + 2) + y + 1) * P.pw + x + 1;
    // FOLD: low-resolution tile slot of fold (a, b) = baseb + a * pwb + b  (tile row (y >> 1) + a + py, column (x >> 1) + b + px)
    const int baseb = (il * ((P.th >> 1) + 2) + (y >> 1) + (pos >> 1)) * pwb + (x >> 1) + (pos & 1);
    // first weight slot of chunk c (FOLD: a B chunk starts at this wave's parity)
    auto ctfirst = [&](int c) { return FOLD == 2 ? 4 * c : ((!FOLD || c < ncha) ? c * 9 : ncha * 9 + (c - ncha) * 16 + 4 * pos); };

    // zero halo columns (col 0 and col W + 1) of all four planes (of both buffers), once
    const int nbuf = P.dbuf ? 2 : 1;
    for (int e = tid; e < nbuf * 4 * P.rows * 2; e += 256) {
        const int side = e & 1, r = e >> 1;          // r over 4 * rows (plane-major rows are contiguous; the buffers follow each other)
        tile[r * P.pw + (side ? W + 1 : 0)] = f4zero();
    }

    // Two tile buffers (dbuf): chunk c + 1 is staged BEFORE chunk c multiplies, one barrier per chunk -- a wave's staging (global
    // latency, address arithmetic) then overlaps the other waves' matrix instructions instead of standing between two barriers.
    float4* const tile0 = g4sm;
    if (P.dbuf) {
        int ltid = tid;
        asm volatile("" : "+v"(ltid));
        stage(tile0, 0, ltid);
        __syncthreads();
    }
    frag4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = frag4{0.f, 0.f, 0.f, 0.f};
    auto chunk = [&](int ch, auto FULLC) __attribute__((always_inline)) {      // FULLC: the chunk has 16 channels by construction
        int ltid = tid;                         // opaque per chunk: keeps the staging addresses from being hoisted out of the loop
        asm volatile("" : "+v"(ltid));
        if (P.dbuf) {
            tile = tile0 + (ch & 1) * 4 * pstride;
            if (ch + 1 < nchunk) stage(tile0 + ((ch + 1) & 1) * 4 * pstride, ch + 1, ltid);
            if (ch < 3) G4_STAMP(1 + 5 * ch);
        } else {
            stage(tile, ch, ltid);
            if (ch < 3) G4_STAMP(1 + 5 * ch);
            __syncthreads();
        }
        if (ch < 3) G4_STAMP(3 + 5 * ch);
        const bool bch = FOLD && ch >= ncha;
        const int rem = FOLD == 2 ? GEN_KC : (bch ? S.cb - (ch - ncha) * GEN_KC : cp - ch * GEN_KC), np = rem >= GEN_KC ? 4 : (rem + 3) >> 2;
        // FOLD = 2: the chunk's parity block (py, px) reads the cells at row offsets {-py, 1 - py}, column offsets {-px, 1 - px}
        const int parc = FOLD == 2 ? (ch * GEN_KC) / S.cb : 0;
        const int bpw = FOLD == 2 ? P.pw : pwb, bstride = FOLD == 2 ? pstride : pstrideb;
        int lbase = FOLD == 2 ? base - (parc >> 1) * P.pw - (parc & 1) : (bch ? baseb : base);
        asm volatile("" : "+v"(lbase));
        const int ct0 = ctfirst(ch);
        // The tap loop per plane count NP (compile-time: straight-line matrix code -- with the plane / channel tests as run-time
        // branches inside the loop the compiler copies all accumulators at every merge).  Padding channels (A's tail when ca is not
        // a multiple of 4) multiply zeros by zero weights.
        // Taps as a REAL loop, two per trip (static rotation of the weight registers): the unrolled 9-tap body is 1440 matrix
        // instructions -- more code than the instruction cache holds with four workgroups at different places in it.  A tap's
        // weights are requested one tap ahead; its plane reads (one LDS round trip per tap) hide behind the other waves.
        auto taps = [&](auto NPC) __attribute__((always_inline)) {
            constexpr int NP = decltype(NPC)::value;
            auto readx = [&](float4 (&xr)[NP], int tap) __attribute__((always_inline)) {
                const int a0 = lbase + (tap / 3 - 1) * P.pw + (tap % 3 - 1);
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) xr[pl] = tile[pl * pstride + a0];
            };
            auto mac = [&](const float (&w)[NG], const float4 (&xr)[NP]) __attribute__((always_inline)) {
                t4_static_for<NP>([&](auto PL) {
                    constexpr int pl = decltype(PL)::value;
                    t4_static_for<4>([&](auto CC) {
                        constexpr int c = decltype(CC)::value;
                        const float xv = f4get(xr[pl], c);
#pragma unroll
                        for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[g], xv, acc[g], 4, 4 * pl + c, 0);
                    });
                });
            };
            float4 xr[NP];
            if (!bch) {
#pragma unroll 1
                for (int tap = 0; tap < 8; tap += 2) {
                    wload(w1, ct0 + tap + 1);
                    readx(xr, tap);
                    __builtin_amdgcn_sched_barrier(0);
                    mac(w0, xr);
                    __builtin_amdgcn_sched_barrier(0);
                    wload(w0, ct0 + tap + 2);
                    readx(xr, tap + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mac(w1, xr);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (ch + 1 < nchunk) wload(w1, ctfirst(ch + 1));    // the next chunk's first slot (moved to w0 below, behind the staging)
                readx(xr, 8);
                __builtin_amdgcn_sched_barrier(0);
                mac(w0, xr);
            } else if constexpr (FOLD) {
                // the four folds (a, b) of this wave's parity: slots ct0 .. ct0 + 3; the next chunk's first slot lands in w0 behind the last
                auto readb = [&](float4 (&xr)[NP], int ab) __attribute__((always_inline)) {
                    const int a0 = lbase + (ab >> 1) * bpw + (ab & 1);
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) xr[pl] = tile[pl * bstride + a0];
                };
#pragma unroll 1
                for (int ab = 0; ab < 4; ab += 2) {
                    wload(w1, ct0 + ab + 1);
                    readb(xr, ab);
                    __builtin_amdgcn_sched_barrier(0);
                    mac(w0, xr);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ab == 0) wload(w0, ct0 + 2);
                    else if (ch + 1 < nchunk) wload(w0, ctfirst(ch + 1));
                    readb(xr, ab + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mac(w1, xr);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if constexpr (decltype(FULLC)::value) taps(std::integral_constant<int, 4>{});
        else {
            if (np == 4) taps(std::integral_constant<int, 4>{});
            else if (np == 3) taps(std::integral_constant<int, 3>{});
            else if (np == 2) taps(std::integral_constant<int, 2>{});
            else taps(std::integral_constant<int, 1>{});
        }
        if (!bch && ch + 1 < nchunk) {
#pragma unroll
            for (int g = 0; g < NG; ++g) w0[g] = w1[g];
        }
        if (ch < 3) G4_STAMP(4 + 5 * ch);
        __syncthreads();
        if (ch < 3) G4_STAMP(5 + 5 * ch);
    };
    if constexpr (VEC && FOLD == 2) {
        for (int ch = 0; ch < nchunk; ++ch) chunk(ch, std::true_type{});
    } else if constexpr (VEC) {
        for (int ch = 0; ch + 1 < nchunk; ++ch) chunk(ch, std::true_type{});
        chunk(nchunk - 1, std::false_type{});
    } else {
        for (int ch = 0; ch < nchunk; ++ch) chunk(ch, std::false_type{});
    }

    // ---- epilogue: activation, (max-pool + argmax byte | addend), NHWC stores through LDS ----
    const int img = img0 + il;
    const bool live = img < P.n;
    const int gy = row0 + y;
    int ngv = P.ngt - g0;                        // valid groups of this pass
    ngv = ngv < NG ? ngv : NG;
    const bool vec = !(P.co & 3);
    float* ot = (float*)g4sm;                    // the tile area is free now (barrier above)
    constexpr int pitch = VEC && NG == 10 ? 40 : 4 * NG + 4;      // (VEC, 10 groups: 256 x 40 floats = the 40 KB a workgroup has at four per CU)
    const float slope = P.slope;
    auto epilogue = [&](auto ACT) {
        constexpr int act = decltype(ACT)::value;
        // (the bias is added AFTER the products, as every other kernel of this library does)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = 4 * (g0 + g) + r;
                acc[g][r] += (P.bias && col < P.co) ? cgs_to_const(P.bias)[col] : 0.f;
            }
        }
        auto fact = [&](float v) -> float {
            if constexpr (act == CGS_ACT_RELU) return fmaxf(v, 0.f);
            else if constexpr (act == CGS_ACT_LRELU) return v > 0.f ? v : slope * v;
            else if constexpr (act == CGS_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
            else return v;
        };
        // copy-out mapping: 16 lanes per pixel (lanes g < ngv carry one 4-channel group each), 16 pixels per round
        const int cg = tid & 15, cpx = tid >> 4;
        if (vec && P.pool) {
            // The tile's pixel p = 4 q + pos is position pos of 2x2 cell q, and cell q is pooled pixel q of the tile (64 pixels,
            // contiguous in memory, image-major / row-major).  Every lane writes its activations to LDS; the copy-out threads
            // (16 per pooled pixel, one 4-channel group each) take the maximum and the argmax code of the cell's four values --
            // once per output, where a quad of lanes used to do it four times over with DPP (11 vector instructions per channel
            // and lane: as much issue time as the layer's matrix instructions at 3 input channels).
#pragma unroll
            for (int g = 0; g < NG; ++g)
                if (g < ngv)
                    *(float4*)(ot + p * pitch + 4 * g) = make_float4(fact(acc[g][0]), fact(acc[g][1]), fact(acc[g][2]), fact(acc[g][3]));
            __syncthreads();
            const size_t pix0 = P.imgs == 1 ? ((size_t)img0 * (H >> 1) + (row0 >> 1)) * (W >> 1) : (size_t)img0 * (H >> 1) * (W >> 1);
            const size_t pixend = (size_t)P.n * (H >> 1) * (W >> 1);
            // items (pooled pixel, group) in memory order, 256 per round: every thread works (16 fixed lanes per pixel left 6 of them idle at
            // 10 groups); item / ngv by a multiply (exact for item < 1024, ngv <= 10)
            const uint32_t rngv = 65536u / (uint32_t)ngv + 1u;
            for (int item = tid; item < 64 * ngv; item += 256) {
                const int px = (int)(((uint32_t)item * rngv) >> 16), ig = item - px * ngv;
                if (pix0 + px < pixend) {
                    const float* cell = ot + 4 * px * pitch + 4 * ig;
                    const float4 v0 = *(const float4*)cell, v1 = *(const float4*)(cell + pitch), v2 = *(const float4*)(cell + 2 * pitch),
                                 v3 = *(const float4*)(cell + 3 * pitch);
                    auto one = [&](float a, float b, float c, float d, float& mm) -> uint32_t {
                        mm = fmaxf(fmaxf(a, b), fmaxf(c, d));
                        const uint32_t cd = a == mm ? 0u : (b == mm ? 1u : (c == mm ? 2u : (d == mm ? 3u : 4u)));      // first position holding the maximum
                        return (cd & 3u) | ((act == CGS_ACT_RELU && !(mm > 0.f)) ? 4u : 0u);
                    };
                    float4 m;
                    uint32_t word = one(v0.x, v1.x, v2.x, v3.x, m.x);
                    word |= one(v0.y, v1.y, v2.y, v3.y, m.y) << 8;
                    word |= one(v0.z, v1.z, v2.z, v3.z, m.z) << 16;
                    word |= one(v0.w, v1.w, v2.w, v3.w, m.w) << 24;
                    const size_t o = (pix0 + px) * P.co + 4 * (g0 + ig);
                    *(float4*)(P.out + o) = m;
                    if (P.argmax) *(uint32_t*)(P.argmax + o) = word;
                }
            }
        } else if (P.out2 && 4 * g0 >= P.split_ca) {
            // second source of a cat: the gradient of nearest-up(B) is the sum over each ups x ups cell -- a quad of lanes (ups = 2) or the
            // 16 lanes of a 4x4 image (ups = 4).  The tile's cells are contiguous in memory like a pooled map.
            const int ush = P.split_ush, cbn = P.co - P.split_ca;
            const int wq = W >> ush, cells_img = (P.th >> ush) * wq;
            const int lin = il * cells_img + (y >> ush) * wq + (x >> ush);
            const bool writer = ush == 1 ? pos == 0 : (lane & 15) == 0;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g < ngv) {
                    float c4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = quad_sum(acc[g][r]);
                        if (ush == 2) { v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); }
                        c4[r] = v;
                    }
                    if (writer) *(float4*)(ot + lin * pitch + 4 * g) = make_float4(c4[0], c4[1], c4[2], c4[3]);
                }
            }
            __syncthreads();
            const int ncell = 256 >> (2 * ush);
            const size_t pix0 = P.imgs == 1 ? ((size_t)img0 * (H >> ush) + (row0 >> ush)) * wq : (size_t)img0 * (H >> ush) * wq;
            const size_t pixend = (size_t)P.n * (H >> ush) * wq;
            if (cg < ngv) {
                for (int px = cpx; px < ncell; px += 16) {
                    if (pix0 + px < pixend)
                        *(float4*)(P.out2 + (pix0 + px) * cbn + (4 * (g0 + cg) - P.split_ca)) = *(const float4*)(ot + px * pitch + 4 * cg);
                }
            }
        } else if (!P.pool) {
            // the tile's 256 pixels are contiguous in memory
            const int hwp = P.th * W;                                             // pixels per image part of the tile
            const int lin = il * hwp + y * W + x;
            const size_t pix0 = P.imgs == 1 ? ((size_t)img0 * H + row0) * W : (size_t)img0 * H * W;
            const size_t pixend = (size_t)P.n * H * W;
            const size_t addend_end = P.addend ? (size_t)P.n_addend * H * W : 0;
            {
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    if (g < ngv)
                        *(float4*)(ot + lin * pitch + 4 * g) = make_float4(fact(acc[g][0]), fact(acc[g][1]), fact(acc[g][2]), fact(acc[g][3]));
                __syncthreads();
                const size_t hp0 = pix0;
                if (vec) {
                    const int ostride = P.out2 ? P.split_ca : P.co;      // (split: this pass lies below split_ca: d_a's own row stride)
                    if (P.out) {
                        // items (pixel, group) in memory order, 256 per round (see the pooled copy-out)
                        const uint32_t rngv = 65536u / (uint32_t)ngv + 1u;
#pragma unroll 2
                        for (int item = tid; item < 256 * ngv; item += 256) {
                            const int px = (int)(((uint32_t)item * rngv) >> 16), ig = item - px * ngv;
                            const size_t gp = hp0 + px, o = gp * ostride + 4 * (g0 + ig);
                            if (gp < pixend) {
                                float4 t = *(const float4*)(ot + px * pitch + 4 * ig);
                                if (gp < addend_end) { const float4 a = *(const float4*)(P.addend + o); t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w; }
                                *(float4*)(P.out + o) = t;
                            }
                        }
                    }
                } else {      // channel counts that are not multiples of 4 (3: the image layer's data gradient; 3 + c: cat(image, .))
                    int nch = P.co - 4 * g0;
                    nch = nch < 4 * ngv ? nch : 4 * ngv;
                    for (int f = tid; f < 256 * nch; f += 256) {
                        const int px = f / nch, c = f - px * nch;
                        const size_t gp = hp0 + px;
                        if (gp < pixend) {
                            const size_t o = gp * P.co + 4 * g0 + c;
                            float t = ot[px * pitch + c];
                            if (gp < addend_end) t += P.addend[o];
                            P.out[o] = t;
                        }
                    }
                }
            }
        } else {
            // max-pooled layers whose channel count is not a multiple of 4 (none in the Hourglass): scalar stores
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int col = 4 * (g0 + g);
                if (col >= P.co) break;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = fact(acc[g][r]);
                    float mm = fmaxf(v, dppf_xor1(v));
                    mm = fmaxf(mm, dppf_xor2(mm));
                    uint32_t cd = (v == mm) ? (uint32_t)pos : 4u;
                    cd = min(cd, dpp_xor1(cd));
                    cd = min(cd, dpp_xor2(cd));
                    const uint32_t code = (cd & 3u) | ((act == CGS_ACT_RELU && !(mm > 0.f)) ? 4u : 0u);
                    if (live && pos == 0 && col + r < P.co) {
                        const size_t pp = (((size_t)img * (H >> 1) + (gy >> 1)) * (W >> 1) + (x >> 1)) * P.co + col + r;
                        P.out[pp] = mm;
                        if (P.argmax) P.argmax[pp] = (uint8_t)code;
                    }
                }
            }
        }
    };
    if (P.act == CGS_ACT_RELU) epilogue(std::integral_constant<int, CGS_ACT_RELU>{});
    else if (P.act == CGS_ACT_LRELU) epilogue(std::integral_constant<int, CGS_ACT_LRELU>{});
    else if (P.act == CGS_ACT_SIGMOID) epilogue(std::integral_constant<int, CGS_ACT_SIGMOID>{});
    else epilogue(std::integral_constant<int, CGS_ACT_NONE>{});
    G4_STAMP(31);
}

}  // namespace
#ifdef CGS_DEBUG_STAMPS
extern "C" int dbg_gen4_stamps(unsigned long long* stamps) { g_gen4_stamps = stamps; return CGS_OK; }
#endif

// ---- launchers (gen.hip's C entry points call these) ----
struct Gen4Launch {
    GenSrc src; const float* wp; const float* bias; float* out; uint8_t* argmax; const float* addend;
    int n_addend, n, hw, co, act, pool;
    float slope;
    float* out2; int split_ca, split_ups;      // (optional: data gradient of a cat layer written as d_a / cell-summed d_b)
    int fold;                                  // wp is the FOLDED operand (cgs_gen_conv_pack_weights, transposed = 2): cb > 0, ups = 2, hw >= 16, no pooling / split
};

// output-channel groups of 4: passes (workgroups over the same tile) x groups per pass (the kernel's NG)
static void gen4_groups(int co, int& npass, int& ng) {
    const int ngt = (co + 3) / 4;
    npass = (ngt + 9) / 10;
    const int per = (ngt + npass - 1) / npass;
    ng = per <= 1 ? 1 : per <= 2 ? 2 : per <= 4 ? 4 : per <= 6 ? 6 : per <= 8 ? 8 : 10;
}

long gen4_packed_floats(int ca, int cb, int co, int fold) {      // fold = the pack mode's folded forms: 1 / 2 = forward fold, 3 = the s2d data gradient
    int npass, ng;
    gen4_groups(co, npass, ng);
    const int pa4 = (ca + 3) & ~3;
    if (fold == 3) return (long)(4 * ((ca + GEN_KC - 1) / GEN_KC)) * 4 * npass * ng * 64;
    if (fold) return (long)(((pa4 + GEN_KC - 1) / GEN_KC) * 9 + ((cb + GEN_KC - 1) / GEN_KC) * 16) * npass * ng * 64;
    const int cp = pa4 + cb, nchunk = (cp + GEN_KC - 1) / GEN_KC;
    return (long)nchunk * 9 * npass * ng * 64;
}

int gen4_pack_launch(int ca, int cb, int co, int transposed, const float* w, float* wp, int ci_layer, int ci_off, hipStream_t st) {
    const long total = gen4_packed_floats(ca, cb, co, transposed >= 2 ? transposed : 0);
    int npass, ng;
    gen4_groups(co, npass, ng);
    Gen4PackParams P{w, wp, ca, cb, co, transposed, (int)total, npass * ng, ci_layer, ci_off};
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(gen4_pack_kernel, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

struct Gen4PackJob { const float* w; float* wp; int ca, cb, co, transposed, ci_layer, ci_off; };      // (= cgs_gen_pack_job of cgs_hip.h)
int gen4_pack_batch_launch(const Gen4PackJob* jobs, int njobs, hipStream_t st) {
    for (int j0 = 0; j0 < njobs; j0 += G4_PACK_BATCH) {
        Gen4PackBatch B{};
        const int nb = njobs - j0 < G4_PACK_BATCH ? njobs - j0 : G4_PACK_BATCH;
        long most = 0;
        for (int j = 0; j < nb; ++j) {
            const Gen4PackJob& J = jobs[j0 + j];
            int npass, ng;
            gen4_groups(J.co, npass, ng);
            const long total = gen4_packed_floats(J.ca, J.cb, J.co, J.transposed >= 2 ? J.transposed : 0);
            B.job[j] = Gen4PackParams{J.w, J.wp, J.ca, J.cb, J.co, J.transposed, (int)total, npass * ng, J.ci_layer, J.ci_off};
            most = total > most ? total : most;
        }
        int blocks = (int)((most + 255) / 256);
        blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
        hipLaunchKernelGGL(gen4_pack_batch_kernel, dim3(blocks, nb), dim3(256), 0, st, B);
        CGS_HIP_CHECK_LAUNCH();
    }
    return CGS_OK;
}

int gen4_conv_launch(const Gen4Launch& L, hipStream_t st) {
    Gen4Params P{};
    P.src = L.src; P.wp = L.wp; P.bias = L.bias; P.out = L.out; P.argmax = L.argmax; P.addend = L.addend; P.n_addend = L.n_addend;
    P.n = L.n; P.hw = L.hw; P.co = L.co; P.act = L.act; P.pool = L.pool; P.slope = L.slope;
    const int hw = L.hw;
    P.lw = hw == 64 ? 6 : hw == 32 ? 5 : hw == 16 ? 4 : hw == 8 ? 3 : 2;
    P.imgs = hw >= 16 ? 1 : (hw == 8 ? 4 : 16);
    P.th = hw >= 16 ? 256 / hw : hw;
    // padded width in float4 slots: 8 (mod 16) where the tile is wide enough, so the two rows a 16-lane group reads sit 128 bytes apart
    P.pw = hw >= 16 ? ((hw + 2 + 7) / 16) * 16 + 8 : hw + 2;
    P.rows = P.imgs * (P.th + 2);
    P.dbg = g_gen4_stamps;
    P.ngt = (L.co + 3) / 4;
    int ng;
    gen4_groups(L.co, P.npass, ng);
    if (L.out2) {
        // every output pass must lie on one side of the split (split_ca = 0: everything is cell-summed), whole 4-channel groups on both
        // sides, no pooling / addend / activation
        const int ush = L.split_ups == 4 ? 2 : (L.split_ups == 2 ? 1 : -1);
        if (ush < 0 || (L.split_ca % (4 * ng)) || ((L.co - L.split_ca) & 3) || L.co <= L.split_ca || L.pool || L.addend ||
            (ush == 2 && hw != 4) || hw < (1 << ush))
            return CGS_ERR_UNSUPPORTED;
        P.out2 = L.out2; P.split_ca = L.split_ca; P.split_ush = ush;
    }
    const int tiles = P.imgs == 1 ? L.n * (hw / P.th) : (L.n + P.imgs - 1) / P.imgs;
    // Few tiles (the 4x4 and 8x8 maps: 16 / 4 images per tile -- dec_model.3 at n = 512 is 32 tiles x 2 passes on 256 CUs): more passes
    // of fewer groups over the same tile until every CU has a workgroup.  The packed weights are [chunk][tap][all groups][64], so any
    // split of the padded group count into passes reads the same buffer.
    {
        const int ngp = P.npass * ng;
        static const int cand[] = {8, 6, 4, 2, 1};
        for (int i = 0; i < 5 && tiles * P.npass < 256; ++i) {
            const int c = cand[i];
            if (c >= ng || ngp % c) continue;
            if (L.out2 && (L.split_ca % (4 * c))) continue;
            ng = c; P.npass = ngp / c;
        }
    }
    // LDS: the input tile -- two buffers when there is more than one chunk and three (NG = 10) / four workgroups still fit a CU;
    // the epilogue reuses the area for 256 pixels x (4 ng + 4) floats (pooling included: the cells' maxima are taken by the copy-out threads)
    const size_t tile_bytes = (size_t)4 * P.rows * P.pw * sizeof(float4);
    const int cp = ((L.src.ca + 3) & ~3) + L.src.cb;
    const bool vec = G4_VEC && ng >= 8 &&
                     ((!L.fold && (L.src.mode == GEN_SRC_POOLEXP || (L.src.mode == GEN_SRC_F32 && !(L.src.ca & 3)))) || (L.fold == 2 && G4_VEC_S2D));
    const int wpe = vec ? G4_WPE_VEC : (ng >= 8 ? G4_WPE_BIG : (g4_tiny(ng, L.fold) ? G4_WPE_TINY : G4_WPE_SMALL));
    const size_t budget = vec && wpe == 4 ? (size_t)40960 : (size_t)(160 * 1024) / wpe - 512;
    P.dbuf = ((cp > GEN_KC || L.fold) && 2 * tile_bytes <= budget) ? 1 : 0;
    size_t lds = tile_bytes * (P.dbuf ? 2 : 1);
    const size_t epi = (size_t)256 * (vec && ng == 10 ? 40 : 4 * ng + 4) * sizeof(float);
    lds = lds > epi ? lds : epi;
    const dim3 grid(tiles * P.npass);
    if (L.fold == 2) {
        // src = the s2d view: a = dY [n, 2 hw, 2 hw, ca], cb = the parity block width (ca rounded up to 16)
        if (L.pool || L.out2 || L.src.mode != GEN_SRC_F32 || (L.src.ca & 3) || (L.src.cb & 15) || L.src.cb < L.src.ca || hw > 32) return CGS_ERR_UNSUPPORTED;
        if (vec) {
            if (ng == 8) hipLaunchKernelGGL((gen4_conv3x3_kernel<8, 2, true>), grid, dim3(256), lds, st, P);
            else hipLaunchKernelGGL((gen4_conv3x3_kernel<10, 2, true>), grid, dim3(256), lds, st, P);
            CGS_HIP_CHECK_LAUNCH();
            return CGS_OK;
        }
#define G4_LAUNCH_S(NG_) hipLaunchKernelGGL((gen4_conv3x3_kernel<NG_, 2>), grid, dim3(256), lds, st, P)
        switch (ng) {
            case 1: G4_LAUNCH_S(1); break;
            case 2: G4_LAUNCH_S(2); break;
            case 4: G4_LAUNCH_S(4); break;
            case 6: G4_LAUNCH_S(6); break;
            case 8: G4_LAUNCH_S(8); break;
            default: G4_LAUNCH_S(10); break;
        }
#undef G4_LAUNCH_S
        CGS_HIP_CHECK_LAUNCH();
        return CGS_OK;
    }
    if (L.fold) {
        // (the operand was packed for the layer's own pass split: gen4_groups(co) -- the few-tiles re-split above reads the same buffer)
        if (L.src.cb <= 0 || L.src.ups != 2 || hw < 16 || L.pool || L.out2 || L.src.mode == GEN_SRC_POOLEXP) return CGS_ERR_UNSUPPORTED;
#define G4_LAUNCH_F(NG_) hipLaunchKernelGGL((gen4_conv3x3_kernel<NG_, 1>), grid, dim3(256), lds, st, P)
        switch (ng) {
            case 1: G4_LAUNCH_F(1); break;
            case 2: G4_LAUNCH_F(2); break;
            case 4: G4_LAUNCH_F(4); break;
            case 6: G4_LAUNCH_F(6); break;
            case 8: G4_LAUNCH_F(8); break;
            default: G4_LAUNCH_F(10); break;
        }
#undef G4_LAUNCH_F
        CGS_HIP_CHECK_LAUNCH();
        return CGS_OK;
    }
    if (vec) {
        if (ng == 8) hipLaunchKernelGGL((gen4_conv3x3_kernel<8, 0, true>), grid, dim3(256), lds, st, P);
        else hipLaunchKernelGGL((gen4_conv3x3_kernel<10, 0, true>), grid, dim3(256), lds, st, P);
        CGS_HIP_CHECK_LAUNCH();
        return CGS_OK;
    }
#define G4_LAUNCH(NG_) hipLaunchKernelGGL((gen4_conv3x3_kernel<NG_, 0>), grid, dim3(256), lds, st, P)
    switch (ng) {
        case 1: G4_LAUNCH(1); break;
        case 2: G4_LAUNCH(2); break;
        case 4: G4_LAUNCH(4); break;
        case 6: G4_LAUNCH(6); break;
        case 8: G4_LAUNCH(8); break;
        default: G4_LAUNCH(10); break;
    }
#undef G4_LAUNCH
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

// Body of the 3x3 convolution kernels (forward and data gradient) + the instance table.  Included by
// conv_fwd.hip (one kernel per layer) and fused_small.hip (several small layers per workgroup).
#pragma once
#include "conv_tile.h"

// 1: the FMA block of every 3x3 kernel runs on the matrix cores (mfma_plane, conv_tile.h); 0: on the vector ALU (fma_plane)
#ifndef CGS_WHATIF_NO_POOL_EPI
#define CGS_WHATIF_NO_POOL_EPI 0
#endif
#ifndef CGS_POOL_EPI_MAX3
#define CGS_POOL_EPI_MAX3 2      // 0 = the running `v > m` scan; 2 = maximum first, argmax from equality (r05 A/B: -2.7 us per step); 1 = the same with the nibble
                                 // packing as inline-asm v_lshl_or_b32 (+2.2 us: fewer instructions, slower -- the asm statements pin the schedule); 3 = opaque idx
#endif
#ifndef CGS_CONV_MFMA4
#define CGS_CONV_MFMA4 1
#endif

struct ConvParams {
    const void* src_a;
    const float* src_b;
    const uint32_t* amask_in;
    const float* w;
    const float* bias;
    float* out;        // fwd: output;  dgrad: d_a
    float* out2;       // dgrad: d_b
    uint32_t* amask_out;
    const float* addend;
    const float* a_post;
    const float* w2;       // SRC_DH: masker.2 weights [9][16]
    float* dh_out;         // SRC_DH: gradient w.r.t. the masker.0 output, written for the weight-gradient kernel
    // MIX_EPI configs, mix backward (conv_bwd_both.hip): the source tile is dY(replaced) - dY(injected), so d_a is
    // d_rep - d_inj of A-image q.n; the epilogue turns it into d(pre-sigmoid mask) and nothing else is written
    const uint8_t* mix_a; const uint8_t* mix_b; const float* mix_z; float* mix_dz;
    float mix_l1s, mix_l2s;
    const float* mix_vf_pred;   // -staticnorm '' (main.py:415-418): the mask regulariser of A-image n is weighted by 1 - pred[n] (NULL: 1)
    int mix_inject;
    int mix_n_a;           // SRC_MIXC3: number of A-images (mix image n >= mix_n_a is the injected one of n - mix_n_a)
    float* zpart;          // masker.2 forward: optional per-workgroup partial sums (sum |z|, sum z^2) for the L1 / L2 mask losses
    int n, n_addend;
    cgs_dropout drop;
};

enum { EPI_POOL = 0, EPI_PLAIN = 1, EPI_DGRAD = 2 };

// optional second destination of a pooled epilogue: an NHWC LDS tile of the workgroup's image (pixel (y, x) at base + (y + 1) pitch + (x + 1) ps):
// the fused features.3 + encoder-tail kernel (tail.hip) hands the pooled map to the next layers without a trip through memory
struct PoolLds { float* base; int pitch, ps; };

// optional config member: static constexpr bool MIX_EPI = true -> the data gradient w.r.t. source A stays in registers and
// the epilogue writes the mix backward's d(pre-sigmoid mask) instead (conv_bwd_both.hip: features.0 + mix backward)
template <class C, class = void> struct mix_epi_of { static constexpr bool value = false; };
template <class C> struct mix_epi_of<C, decltype((void)C::MIX_EPI)> { static constexpr bool value = C::MIX_EPI; };

// Cfg members: H,W,THREADS, SRC, CA, CB, UPS, WT (0 normal / 1 transposed), WCI, WCO (HWIO dims),
// OC0, OC (logical output channel window), OCB (register block), EPI, ACT, OUT_A (dgrad: channels that
// belong to source A), POST_ACT (dgrad: activation whose derivative multiplies d_a).
// FUSED = true: the body is one stage of a multi-stage workgroup (fused_small.hip): `bid` selects the image
// group, nobody returns early (barriers of later stages follow) and the LDS region is handed in.
// LDS carve-up of a conv3x3 workgroup: source-A tile (float4 planes), one spare slot for redirected stores, source-B tile, the layer's
// weights in HWIO order (wave-uniform ds_reads broadcast them into VGPRs / the MFMA weight registers), the SRC_DH dzpre tile.
template <class C>
struct ConvLds {
    using G = Geo<C::H, C::W, C::THREADS, C::CW>;
    static constexpr int PA = (C::SRC == SRC_SCALAR) ? 1 : (C::CA + 3) / 4;
    static constexpr int PB = C::CB / 4;
    static constexpr int A_ELEMS = PA * G::IMGS * G::TRA * G::PWA;  // float4 slots (floats for SRC_SCALAR)
    static constexpr int DUMP = (C::SRC == SRC_SCALAR) ? (A_ELEMS + 3) / 4 : A_ELEMS;   // one spare slot for redirected stores
    static constexpr int B_ELEMS = PB == 0 ? 0 : (C::UPS == 2 ? PB * G::IMGS * G::TRB * G::PWB : PB * G::IMGS);
    static constexpr int W_FLOATS = 9 * C::WCI * C::WCO;
    static constexpr int DZW = G::W + 4, DZ_FLOATS = (C::SRC == SRC_DH) ? (G::TRA + 2) * DZW : 0;
};

// stage 1 of a conv3x3 workgroup: the input tile(s) of tile `bid` into LDS (global loads and LDS stores of all elements, unrolled)
template <class C>
__device__ __forceinline__ void conv_stage_inputs(const ConvParams& P, const int bid, const int tid, float4* smem) {
    using G = Geo<C::H, C::W, C::THREADS, C::CW>;
    using L = ConvLds<C>;
    constexpr int PA = L::PA, PB = L::PB, DUMP = L::DUMP, W_FLOATS = L::W_FLOATS, DZW = L::DZW, DZ_FLOATS = L::DZ_FLOATS;
    float4* const ldsA = smem;
    float4* const ldsB = smem + L::DUMP + 1;
    float* const ldsW = (float*)(ldsB + L::B_ELEMS);
    [[maybe_unused]] float* const ldsDz = ldsW + ((W_FLOATS + 3) / 4) * 4;     // SRC_DH: dzpre tile with a 2-pixel halo
    const int qtid = tid % C::THREADS;           // quad handled by this thread
    [[maybe_unused]] const int cw = tid / C::THREADS;             // wave-uniform: which output-channel chunks this wave computes
    const QuadPos q = quad_pos<G>(qtid, bid);
    [[maybe_unused]] const int n0 = (G::IMGS == 1) ? q.n : bid * G::IMGS;
    [[maybe_unused]] const int N = P.n;
    [[maybe_unused]] const DropCtx dc = drop_ctx(P.drop);
    // ---- stage inputs ----
    if constexpr (C::SRC == SRC_F32) {
        DropCtx dl = dc;
        if constexpr (C::EPI == EPI_DGRAD || !C::DROP) dl.on = false;  // dgrad applies the mask in the epilogue
        load_a_f32<G, PA>(ldsA, (const float4*)P.src_a, n0, q.row0, N, tid, dl);
    } else if constexpr (C::SRC == SRC_U8C3) {
        load_a_u8c3<G>(ldsA, (const uint32_t*)P.src_a, n0, q.row0, N, tid);
    } else if constexpr (C::SRC == SRC_F32C3) {
        load_a_f32c3<G>(ldsA, (const float4*)P.src_a, n0, q.row0, N, tid);
    } else if constexpr (C::SRC == SRC_MIXC3) {
        load_a_mix<G>(ldsA, (const uint32_t*)P.mix_a, (const uint32_t*)P.mix_b, (const float4*)P.mix_z, P.mix_n_a, n0, q.row0, N, tid);
    } else if constexpr (C::SRC == SRC_POOLEXP) {
        load_poolexp<G, PA, 1>(ldsA, (const float4*)P.src_a, P.amask_in, n0, q.row0, N, tid,
                               [](int p, int img, int r, int x) { return ldsA_idx<G, PA>(p, img, r, x + 1); }, DUMP);
        zero_halo_cols<G, PA>(ldsA, tid);
    } else if constexpr (C::SRC == SRC_POOLEXP_DIFF) {
        // gradient tile of (replaced image q.n) - (injected image q.n + mix_n_a): one data-gradient pass serves the mix backward
        load_poolexp_diff<G, PA, 1>(ldsA, (const float4*)P.src_a, P.amask_in, n0, P.mix_n_a, P.mix_inject != 0, q.row0, N, tid,
                                    [](int p, int img, int r, int x) { return ldsA_idx<G, PA>(p, img, r, x + 1); }, DUMP);
        zero_halo_cols<G, PA>(ldsA, tid);
    } else if constexpr (C::SRC == SRC_DH) {
        // d(masker.0 output) is never read from memory here: it is rebuilt from dzpre (1 channel) and the LeakyReLU
        // mask of the saved activation, dH = LeakyReLU'(h) * conv_bwd(dzpre; masker.2), used as this kernel's input
        // tile AND written out once (interior rows) for the masker.0 weight-gradient kernel.
        static_assert(G::IMGS == 1 && C::CA == 16, "mask-head tile");
        const float* dzp = (const float*)P.src_a;
        for_elems<DZ_FLOATS, G::LT>(tid, [&](int e) {
            int c2 = e % DZW, r2 = e / DZW;
            int y = q.row0 + r2 - 2, x = c2 - 2;
            bool in = n0 < N && y >= 0 && y < G::H && x >= 0 && x < G::W;
            float v = dzp[in ? (n0 * G::H + y) * G::W + x : 0];
            ldsDz[e] = in ? v : 0.f;
        });
        const int pl = tid & 3;                       // the 4-channel plane this thread builds (LT % 4 == 0)
        float w2r[9][4];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) w2r[t][c] = P.w2[t * 16 + 4 * pl + c];
        __syncthreads();
        constexpr int E = G::TRA * G::W * 4;
        for_elems<E, G::LT>(tid, [&](int e) {
            int x = (e / 4) % G::W, r = e / (4 * G::W);
            int y = q.row0 + r - 1;
            bool in = n0 < N && y >= 0 && y < G::H;
            int gi = in ? ((n0 * G::H + y) * G::W + x) * 4 + pl : 0;
            float4 hv = ((const float4*)P.a_post)[gi];
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    float d = ldsDz[(r + 2 - ky) * DZW + x + 3 - kx];
                    a0 = fmaf(d, w2r[ky * 3 + kx][0], a0); a1 = fmaf(d, w2r[ky * 3 + kx][1], a1);
                    a2 = fmaf(d, w2r[ky * 3 + kx][2], a2); a3 = fmaf(d, w2r[ky * 3 + kx][3], a3);
                }
            float4 v = make_float4(a0 * (hv.x > 0.f ? 1.f : 0.01f), a1 * (hv.y > 0.f ? 1.f : 0.01f),
                                   a2 * (hv.z > 0.f ? 1.f : 0.01f), a3 * (hv.w > 0.f ? 1.f : 0.01f));
            v = in ? v : f4zero();
            ldsA[ldsA_idx<G, PA>(pl, 0, r, x + 1)] = v;
            if (in && r >= 1 && r <= G::TH) ((float4*)P.dh_out)[gi] = v;
        });
        zero_halo_cols<G, PA>(ldsA, tid);
    } else {  // SRC_SCALAR: one fp32 channel, tile of floats [img][TRA][W+2]
        float* t = (float*)ldsA;
        const float* s = (const float*)P.src_a;
        constexpr int E = G::IMGS * G::TRA * (G::W + 2);
        for_elems<E, G::LT>(tid, [&](int e) {
            int c = e % (G::W + 2), r = (e / (G::W + 2)) % G::TRA, img = e / ((G::W + 2) * G::TRA);
            int n = n0 + img, y = q.row0 + r - 1, x = c - 1;
            bool in = n < N && y >= 0 && y < G::H && x >= 0 && x < G::W;
            float v = s[in ? (n * G::H + y) * G::W + x : 0];
            t[e] = in ? v : 0.f;
        });
    }
    if constexpr (PB > 0) {
        if constexpr (C::UPS == 2) load_b_half<G, PB>(ldsB, (const float4*)P.src_b, n0, q.row0, N, tid);
        else load_b_pix<G, PB>(ldsB, (const float4*)P.src_b, n0, N, tid);
    }
}

template <class C>
__device__ __forceinline__ void conv_stage_weights(const ConvParams& P, const int tid, float4* smem) {
    using G = Geo<C::H, C::W, C::THREADS, C::CW>;
    using L = ConvLds<C>;
    constexpr int PA = L::PA, PB = L::PB, DUMP = L::DUMP, W_FLOATS = L::W_FLOATS, DZW = L::DZW, DZ_FLOATS = L::DZ_FLOATS;
    float4* const ldsA = smem;
    float4* const ldsB = smem + L::DUMP + 1;
    float* const ldsW = (float*)(ldsB + L::B_ELEMS);
    [[maybe_unused]] float* const ldsDz = ldsW + ((W_FLOATS + 3) / 4) * 4;     // SRC_DH: dzpre tile with a 2-pixel halo
    [[maybe_unused]] const int cw = tid / C::THREADS;             // wave-uniform: which output-channel chunks this wave computes
    [[maybe_unused]] const int N = P.n;
    [[maybe_unused]] const DropCtx dc = drop_ctx(P.drop);
    for_elems<(W_FLOATS + 3) / 4, G::LT>(tid, [&](int e) {
        int i = 4 * e;
        float4 v;
        v.x = P.w[i < W_FLOATS ? i : 0]; v.y = P.w[i + 1 < W_FLOATS ? i + 1 : 0];
        v.z = P.w[i + 2 < W_FLOATS ? i + 2 : 0]; v.w = P.w[i + 3 < W_FLOATS ? i + 3 : 0];
        ((float4*)ldsW)[e] = v;
    });
}

// stage 2: the FMA block and the epilogue of tile `bid` from the staged LDS tiles
template <class C, bool FUSED>
__device__ __forceinline__ void conv_compute(const ConvParams& P, const int bid, const int tid, float4* smem, const PoolLds pool_lds = PoolLds{nullptr, 0, 0}) {
    using G = Geo<C::H, C::W, C::THREADS, C::CW>;
    using L = ConvLds<C>;
    constexpr int PA = L::PA, PB = L::PB, DUMP = L::DUMP, W_FLOATS = L::W_FLOATS, DZW = L::DZW, DZ_FLOATS = L::DZ_FLOATS;
    float4* const ldsA = smem;
    float4* const ldsB = smem + L::DUMP + 1;
    float* const ldsW = (float*)(ldsB + L::B_ELEMS);
    [[maybe_unused]] float* const ldsDz = ldsW + ((W_FLOATS + 3) / 4) * 4;     // SRC_DH: dzpre tile with a 2-pixel halo
    const int qtid = tid % C::THREADS;           // quad handled by this thread
    [[maybe_unused]] const int cw = tid / C::THREADS;             // wave-uniform: which output-channel chunks this wave computes
    const QuadPos q = quad_pos<G>(qtid, bid);
    [[maybe_unused]] const int n0 = (G::IMGS == 1) ? q.n : bid * G::IMGS;
    [[maybe_unused]] const int N = P.n;
    [[maybe_unused]] const DropCtx dc = drop_ctx(P.drop);
    auto wf = [&](int tap, int ci, int oc) -> float {
        if constexpr (C::WT == 0) return ldsW[(tap * C::WCI + ci) * C::WCO + oc];
        else return ldsW[((8 - tap) * C::WCI + oc) * C::WCO + ci];
    };
    int pcx[4];
#pragma unroll
    for (int dx = 0; dx < 4; ++dx) pcx[dx] = G::pc(2 * q.qx + dx);

    const int y0 = q.row0 + 2 * q.qy_l, x0 = 2 * q.qx;  // top-left conv output of the quad
    const bool live = q.n < N;
    // Retire padding threads now (no barrier follows).  Besides saving work this keeps the compiler from
    // sinking the whole FMA block into the `if (live)` store region, which would leave every weight read
    // of the kernel live at once.  (The x4-upsample gradient sums across lanes, so it keeps all lanes.)
    // (matrix-core path: v_mfma ignores EXEC and broadcasts its weight operand from the lanes of one block, so every lane
    //  must keep running -- and hold valid weight registers -- until the last MFMA; the stores below are guarded by `live`)
    constexpr bool MFMA4 = CGS_CONV_MFMA4 && C::SRC != SRC_SCALAR;
    constexpr bool ALL_LANES = (C::EPI == EPI_DGRAD && C::UPS == 4) || MFMA4;
    if constexpr (!ALL_LANES && !FUSED) {
        if (!live) return;
    }
    if (ALL_LANES || !FUSED || live) {
    constexpr int NCHUNK = C::OC / C::OCB;
    static_assert(C::OC % C::OCB == 0, "chunking");

    static_assert(NCHUNK % C::CW == 0, "chunks split evenly over the chunk waves");
#pragma unroll 1
    for (int ch = cw; ch < NCHUNK; ch += C::CW) {
        const int oc0 = C::OC0 + ch * C::OCB;
        if constexpr (C::EPI == EPI_DGRAD) {  // skip gradients nobody asked for (uniform branch)
            const bool is_a = oc0 < C::OUT_A;
            if (is_a ? (P.out == nullptr && !mix_epi_of<C>::value) : (P.out2 == nullptr)) continue;
        }
        float acc[4][C::OCB];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int o = 0; o < C::OCB; ++o) acc[i][o] = 0.f;

        if constexpr (MFMA4) {
            constexpr int OCG = (C::OCB + 3) / 4, CIN = C::CA + C::CB, NREG = (9 * CIN + 15) / 16;
            constexpr int PA_FULL = C::CA / 4, REM = C::CA % 4;
            float wreg[OCG][NREG];       // lane 4*b + i of register k: weight of step 16*k + b (= tap * CIN + ci), channel oc0 + 4*g + i
            {
                const int lb = (tid & 63) >> 2, li = tid & 3;
#pragma unroll
                for (int g = 0; g < OCG; ++g)
#pragma unroll
                    for (int k = 0; k < NREG; ++k) {
                        const int step = 16 * k + lb, o = 4 * g + li;
                        const bool ok = step < 9 * CIN && o < C::OCB;
                        const int st = ok ? step : 0;
                        const float wv = wf(st / CIN, st % CIN, oc0 + (ok ? o : 0));
                        wreg[g][k] = ok ? wv : 0.f;
                    }
            }
            frag4 a4[4][OCG];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < OCG; ++g) a4[i][g] = frag4{0.f, 0.f, 0.f, 0.f};
            static_for<PA_FULL>([&](auto PAI) {
                constexpr int pa = decltype(PAI)::value;
                float4 pt[4][4];
                read_patch_a<G>(pt, ldsA, pa, q, pcx);
                mfma_plane<OCG, 4, CIN, 4 * pa, NREG>(a4, pt, wreg);
            });
            if constexpr (REM > 0) {
                float4 pt[4][4];
                read_patch_a<G>(pt, ldsA, PA_FULL, q, pcx);
                mfma_plane<OCG, REM, CIN, 4 * PA_FULL, NREG>(a4, pt, wreg);
            }
            static_for<PB>([&](auto PBI) {
                constexpr int pb = decltype(PBI)::value;
                float4 pt[4][4];
                if constexpr (C::UPS == 2) read_patch_b2<G>(pt, ldsB, pb, q);
                else read_patch_b4<G>(pt, ldsB, pb, q);
                mfma_plane<OCG, 4, CIN, C::CA + 4 * pb, NREG>(a4, pt, wreg);
            });
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int o = 0; o < C::OCB; ++o) acc[i][o] = a4[i][o / 4][o % 4];
        } else {
        if constexpr (C::SRC == SRC_SCALAR) {
            const float* t = (const float*)ldsA;
            float pt[4][4];
            int base = (q.img_l * G::TRA + 2 * q.qy_l) * (G::W + 2) + 2 * q.qx;
#pragma unroll
            for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                for (int dx = 0; dx < 4; ++dx) pt[dy][dx] = t[base + dy * (G::W + 2) + dx];
            fma_scalar<C::OCB>(acc, pt, wf, oc0);
        } else {
            constexpr int PA_FULL = C::CA / 4, REM = C::CA % 4;
#pragma unroll 1
            for (int pa = 0; pa < PA_FULL; ++pa) {
                float4 pt[4][4];
                read_patch_a<G>(pt, ldsA, pa, q, pcx);
                fma_plane<C::OCB, 4>(acc, pt, wf, 4 * pa, oc0);
            }
            if constexpr (REM > 0) {
                float4 pt[4][4];
                read_patch_a<G>(pt, ldsA, PA_FULL, q, pcx);
                fma_plane<C::OCB, REM>(acc, pt, wf, 4 * PA_FULL, oc0);
            }
        }
        if constexpr (PB > 0) {
#pragma unroll 1
            for (int pb = 0; pb < PB; ++pb) {
                float4 pt[4][4];
                if constexpr (C::UPS == 2) read_patch_b2<G>(pt, ldsB, pb, q);
                else read_patch_b4<G>(pt, ldsB, pb, q);
                fma_plane<C::OCB, 4>(acc, pt, wf, C::CA + 4 * pb, oc0);
            }
        }
        }

        // ---------------- epilogues ----------------
        if constexpr (C::EPI == EPI_POOL) {
            constexpr int HP = G::H / 2, WP = G::W / 2;
            float pooled[C::OCB];
            static_assert(C::OCB % 4 == 0, "pooled chunks are multiples of 4 channels");
            uint32_t nib[(C::OCB + 7) / 8];
#pragma unroll
            for (int i = 0; i < (C::OCB + 7) / 8; ++i) nib[i] = 0;
#if CGS_WHATIF_NO_POOL_EPI      // (experiment only: wrong results) how much of the kernel is the pooled epilogue's arithmetic?
            if constexpr (true) {
#pragma unroll
                for (int o = 0; o < C::OCB; ++o) pooled[o] = (acc[0][o] + acc[1][o]) + (acc[2][o] + acc[3][o]);
            } else
#endif
            if constexpr (C::ACT == CGS_ACT_RELU && CGS_POOL_EPI_MAX3) {
                // max over the window FIRST (ReLU is monotone: max_i relu(s_i) = relu(max_i s_i)), the argmax from equality with it:
                // the first position whose pre-activation equals a POSITIVE maximum is the first maximum of the activations (what the
                // running `v > m` scan below picks); a window whose maximum is <= 0 is 0xF either way.  ~14 instead of 21 VALU
                // instructions per channel -- the forward tail kernels are instruction-issue bound (round 5, tools/isa_lines.py).
                typedef float f2_t __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int o2 = C::OCB - 2; o2 >= 0; o2 -= 2) {
                    const f2_t b2 = {cgs_to_const(P.bias)[oc0 + o2], cgs_to_const(P.bias)[oc0 + o2 + 1]};
                    f2_t s[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) s[i] = f2_t{acc[i][o2], acc[i][o2 + 1]} + b2;
#pragma unroll
                    for (int h = 1; h >= 0; --h) {
                        const int o = o2 + h;
                        const float mr = fmaxf(fmaxf(s[0][h], s[1][h]), fmaxf(s[2][h], s[3][h]));
                        pooled[o] = fmaxf(mr, 0.f);
                        uint32_t idx = s[2][h] == mr ? 2u : 3u;
                        idx = s[1][h] == mr ? 1u : idx;
                        idx = s[0][h] == mr ? 0u : idx;
                        if (!(mr > 0.f)) idx = 15u;
                        // channels from the top down: channel o ends at bits 4 (o % 8) ..  (as an instruction: left to the compiler the shift is
                        // folded into the selects' constants -- 0x20000000, 0x30000000, ... -- which it then has to keep moving into registers)
                        if constexpr (CGS_POOL_EPI_MAX3 == 1) asm("v_lshl_or_b32 %0, %1, 4, %2" : "=v"(nib[o / 8]) : "v"(nib[o / 8]), "v"(idx));
                        else if constexpr (CGS_POOL_EPI_MAX3 == 3) { asm("" : "+v"(idx)); nib[o / 8] = (nib[o / 8] << 4) | idx; }
                        else nib[o / 8] = (nib[o / 8] << 4) | idx;
                    }
                }
            } else {
#pragma unroll
                for (int o = 0; o < C::OCB; ++o) {
                    float b = cgs_to_const(P.bias)[oc0 + o];
                    float m = act_fwd<C::ACT>(acc[0][o] + b);
                    uint32_t idx = 0;
#pragma unroll
                    for (int i = 1; i < 4; ++i) {
                        float v = act_fwd<C::ACT>(acc[i][o] + b);
                        if (v > m) { m = v; idx = i; }
                    }
                    if (!(m > 0.f)) idx = 15u;
                    pooled[o] = m;
                    nib[o / 8] |= idx << (4 * (o % 8));
                }
            }
            if (live) {
                int pi = (q.n * HP + (y0 >> 1)) * WP + q.qx;
                float4* o4 = (float4*)(P.out + (size_t)pi * C::WCO + oc0);
#pragma unroll
                for (int i = 0; i < C::OCB / 4; ++i)
                    o4[i] = make_float4(pooled[4 * i], pooled[4 * i + 1], pooled[4 * i + 2], pooled[4 * i + 3]);
                if (pool_lds.base) {
                    float4* l4 = (float4*)(pool_lds.base + ((y0 >> 1) + 1) * pool_lds.pitch + (q.qx + 1) * pool_lds.ps + oc0);
#pragma unroll
                    for (int i = 0; i < C::OCB / 4; ++i)
                        l4[i] = make_float4(pooled[4 * i], pooled[4 * i + 1], pooled[4 * i + 2], pooled[4 * i + 3]);
                }
                if (P.amask_out) {
                    if constexpr (C::OCB % 8 == 0) {
#pragma unroll
                        for (int i = 0; i < C::OCB / 8; ++i) P.amask_out[pi * (C::WCO / 8) + oc0 / 8 + i] = nib[i];
                    } else {   // a 4-channel chunk owns one 16-bit half of the uint32
                        ((uint16_t*)P.amask_out)[pi * (C::WCO / 4) + oc0 / 4] = (uint16_t)nib[0];
                    }
                }
            }
        } else if constexpr (C::EPI == EPI_PLAIN) {
            [[maybe_unused]] float zs1 = 0.f, zs2 = 0.f;
            if (live) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int y = y0 + (i >> 1), x = x0 + (i & 1);
                    float v[C::OCB];
#pragma unroll
                    for (int o = 0; o < C::OCB; ++o) v[o] = act_fwd<C::ACT>(acc[i][o] + cgs_to_const(P.bias)[oc0 + o]);
                    if constexpr (C::ACT == CGS_ACT_SIGMOID && C::WCO == 1) { zs1 += fabsf(v[0]); zs2 += v[0] * v[0]; }
                    float* dst = P.out + ((size_t)(q.n * G::H + y) * G::W + x) * C::WCO + oc0;
                    if constexpr (C::OCB % 4 == 0) {
#pragma unroll
                        for (int j = 0; j < C::OCB / 4; ++j)
                            ((float4*)dst)[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
                    } else {
#pragma unroll
                        for (int o = 0; o < C::OCB; ++o) dst[o] = v[o];
                    }
                }
            }
            if constexpr (C::ACT == CGS_ACT_SIGMOID && C::WCO == 1 && !FUSED) {
                // the mask layer: per-workgroup (sum |z|, sum z^2) for the L1 / L2 mask losses (main.py:421-429), no atomics
                static_assert(G::IMGS == 1 && C::CW == 1 && NCHUNK == 1, "every thread of the workgroup is live and passes here once");
                if (P.zpart) {
                    float* zred = (float*)smem;          // the tiles are consumed: reuse (after a barrier)
                    zs1 = wave_sum(zs1); zs2 = wave_sum(zs2);
                    __syncthreads();
                    if ((tid & 63) == 0) { zred[2 * (tid >> 6)] = zs1; zred[2 * (tid >> 6) + 1] = zs2; }
                    __syncthreads();
                    if (tid == 0) {
                        float t1 = 0.f, t2 = 0.f;
#pragma unroll
                        for (int wv = 0; wv < G::LT / 64; ++wv) { t1 += zred[2 * wv]; t2 += zred[2 * wv + 1]; }
                        P.zpart[2 * bid] = t1; P.zpart[2 * bid + 1] = t2;
                    }
                }
            }
        } else {  // EPI_DGRAD: logical output channel = input channel of the layer
            const bool is_a = oc0 < C::OUT_A;
            if (is_a) {
                constexpr int CAO = C::OUT_A;  // channels per pixel of d_a
                if (live) {
                    // MIX_EPI: the quad's 2 x 6 frame bytes per row as one dword + one ushort (2-byte aligned), z per pixel
                    uint64_t a6[2] = {0, 0}, b6[2] = {0, 0};
                    if constexpr (mix_epi_of<C>::value) {
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const size_t off = ((size_t)(q.n * G::H + y0 + r) * G::W + x0) * 3;
                            a6[r] = (uint64_t)*(const uint32_t*)(P.mix_a + off) | ((uint64_t)*(const uint16_t*)(P.mix_a + off + 4) << 32);
                            b6[r] = (uint64_t)*(const uint32_t*)(P.mix_b + off) | ((uint64_t)*(const uint16_t*)(P.mix_b + off + 4) << 32);
                        }
                    }
                    // ... and the quad's four z (+ the image's valuefak prediction) requested with them: read inside the position loop below, each
                    // was a load -> s_waitcnt vmcnt(0) round trip of its own (nine dependent ones per strip in the ISA)
                    [[maybe_unused]] float zq[4] = {0.f, 0.f, 0.f, 0.f};
                    [[maybe_unused]] float vfp = 0.f;
                    if constexpr (mix_epi_of<C>::value) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) zq[i] = P.mix_z[(size_t)(q.n * G::H + y0 + (i >> 1)) * G::W + x0 + (i & 1)];
                        if (P.mix_vf_pred) vfp = P.mix_vf_pred[q.n];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int y = y0 + (i >> 1), x = x0 + (i & 1);
                        size_t pix = (size_t)(q.n * G::H + y) * G::W + x;
                        float v[C::OCB];
#pragma unroll
                        for (int o = 0; o < C::OCB; ++o) v[o] = acc[i][o];
                        if constexpr (C::OCB % 4 == 0 && C::DROP) {
                            if (dc.on) {
#pragma unroll
                                for (int j = 0; j < C::OCB / 4; ++j) {
                                    float4 m = drop_mult4(dc, (uint32_t)((pix * CAO + oc0) / 4 + j));
                                    v[4 * j] *= m.x; v[4 * j + 1] *= m.y; v[4 * j + 2] *= m.z; v[4 * j + 3] *= m.w;
                                }
                            }
                        }
                        if constexpr (C::POST_ACT == CGS_ACT_LRELU) {
                            const float* ap = P.a_post + pix * CAO + oc0;
#pragma unroll
                            for (int o = 0; o < C::OCB; ++o) v[o] *= (ap[o] > 0.f) ? 1.f : 0.01f;
                        }
                        if (P.addend && q.n < P.n_addend) {
                            const float* ad = P.addend + pix * CAO + oc0;
#pragma unroll
                            for (int o = 0; o < C::OCB; ++o) v[o] += ad[o];
                        }
                        if constexpr (mix_epi_of<C>::value) {
                            static_assert(NCHUNK == 1 && C::CW == 1 && C::OCB == 3 && C::SRC == SRC_POOLEXP_DIFF, "one 3-channel chunk per thread");
                            // dzpre = [ sum_c (B - A)(d_rep - d_inj) + l1s sign(z) + 2 l2s z ] z (1 - z)   (= cgs_mix_bwd)
                            const float s255 = 1.f / 255.f;
                            float dsum = 0.f;
#pragma unroll
                            for (int o = 0; o < 3; ++o) {
                                // byte k of the row's six, from the 32-bit halves: `(float)((a6 >> sh) & 255)` on the 64-bit value is a 64-bit
                                // integer -> float conversion (shift pair, leading-zero count, ldexp: 30 instructions per byte, a fifth of this
                                // kernel's vector instructions; round 5, tools/isa_lines.py) where one v_cvt_f32_ubyteN does it.  Same values.
                                const int kb = 3 * (i & 1) + o;
                                const uint32_t aw = kb < 4 ? (uint32_t)a6[i >> 1] : (uint32_t)(a6[i >> 1] >> 32);
                                const uint32_t bw = kb < 4 ? (uint32_t)b6[i >> 1] : (uint32_t)(b6[i >> 1] >> 32);
                                const int sh = 8 * (kb & 3);
                                const float av = (float)((aw >> sh) & 255u) * s255, bv = (float)((bw >> sh) & 255u) * s255;
                                dsum = fmaf(bv - av, v[o], dsum);
                            }
                            const float zi = zq[i];
                            const float sg = zi > 0.f ? 1.f : (zi < 0.f ? -1.f : 0.f);
                            if (P.mix_vf_pred) {      // valuefak = 1 - pred (>= 0) weights |z| and, squared, z^2
                                const float vf = 1.f - vfp;
                                dsum += P.mix_l1s * vf * sg + 2.f * P.mix_l2s * vf * vf * zi;
                            } else {                  // (the same expression as cgs_mix_bwd)
                                dsum += P.mix_l1s * sg + 2.f * P.mix_l2s * zi;
                            }
                            P.mix_dz[pix] = dsum * zi * (1.f - zi);
                        } else {
                            float* dst = P.out + pix * CAO + oc0;
                            if constexpr (C::OCB % 4 == 0) {
#pragma unroll
                                for (int j = 0; j < C::OCB / 4; ++j)
                                    ((float4*)dst)[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
                            } else {
#pragma unroll
                                for (int o = 0; o < C::OCB; ++o) dst[o] = v[o];
                            }
                        }
                    }
                }
            } else {  // source-B gradient: nearest-upsample backward = sum over the cell
                constexpr int CBO = C::WCI - C::OUT_A;
                float s[C::OCB];
#pragma unroll
                for (int o = 0; o < C::OCB; ++o) s[o] = (acc[0][o] + acc[1][o]) + (acc[2][o] + acc[3][o]);
                if constexpr (C::UPS == 4) {  // 4 quads (consecutive lanes) of an image sum to one pixel
#pragma unroll
                    for (int o = 0; o < C::OCB; ++o) {
                        s[o] += __shfl_xor(s[o], 1, 64);
                        s[o] += __shfl_xor(s[o], 2, 64);
                    }
                    if (live && (qtid & 3) == 0) {
                        float* dst = P.out2 + (size_t)q.n * CBO + (oc0 - C::OUT_A);
#pragma unroll
                        for (int o = 0; o < C::OCB; ++o) dst[o] = s[o];
                    }
                } else if (live) {
                    size_t pi = (size_t)(q.n * G::QH + (y0 >> 1)) * G::QW + q.qx;
                    float* dst = P.out2 + pi * CBO + (oc0 - C::OUT_A);
                    if constexpr (C::OCB % 4 == 0) {
#pragma unroll
                        for (int j = 0; j < C::OCB / 4; ++j)
                            ((float4*)dst)[j] = make_float4(s[4 * j], s[4 * j + 1], s[4 * j + 2], s[4 * j + 3]);
                    } else {
#pragma unroll
                        for (int o = 0; o < C::OCB; ++o) dst[o] = s[o];
                    }
                }
            }
        }
    }
    }
}


// Cfg members: see conv_compute.  FUSED = true: the body is one stage of a multi-stage workgroup: `bid` selects the image
// group, nobody returns early (barriers of later stages follow) and the LDS region is handed in.
template <class C, bool FUSED>
__device__ __forceinline__ void conv3x3_body(const ConvParams& P, const int bid, float4* smem) {
    const int tid = threadIdx.x;                 // all threads take part in the loads
    conv_stage_inputs<C>(P, bid, tid, smem);
    conv_stage_weights<C>(P, tid, smem);
    __syncthreads();
    conv_compute<C, FUSED>(P, bid, tid, smem);
}

// optional config member: static constexpr int TPW = k -> a workgroup runs k consecutive tiles (strips of an image) as a software
// pipeline: the global loads of tile t+1 are issued into registers before the matrix instructions of tile t and written to the LDS
// tile after them (conv_tile.h fetch_* / commit_*); weights and halo columns are staged once per workgroup.
template <class C, class = void> struct tpw_of { static constexpr int value = 1; };
template <class C> struct tpw_of<C, decltype((void)C::TPW)> { static constexpr int value = C::TPW; };

template <class C>
__device__ __forceinline__ void conv3x3_body_pipe(const ConvParams& P, const int bid0, float4* smem, const PoolLds pool_lds = PoolLds{nullptr, 0, 0}) {
    using G = Geo<C::H, C::W, C::THREADS, C::CW>;
    using L = ConvLds<C>;
    constexpr int PA = L::PA, PB = L::PB, TPW = tpw_of<C>::value;
    static_assert(G::IMGS == 1, "pipelined tiles are strips of one image");
    static_assert(CGS_CONV_MFMA4 && C::SRC != SRC_SCALAR && C::SRC != SRC_DH && C::SRC != SRC_F32C3, "matrix-core path, split loaders");
    static_assert(PB == 0 || C::UPS == 2, "source B at half resolution");
    static_assert(C::EPI == EPI_DGRAD || !C::DROP, "no Dropout in the split fp32 loader");
    float4* const ldsA = smem;
    [[maybe_unused]] float4* const ldsB = smem + L::DUMP + 1;
    const int tid = threadIdx.x;
    const int N = P.n;
    conv_stage_weights<C>(P, tid, smem);
    zero_halo_cols<G, PA>(ldsA, tid);
    auto pos = [&](int t, int& n0, int& row0) { const int b = bid0 + t; n0 = b / G::STRIPS; row0 = (b % G::STRIPS) * G::TH; };
    using RA = std::conditional_t<C::SRC == SRC_F32, FetchF32<G, PA>,
               std::conditional_t<C::SRC == SRC_U8C3, FetchU8<G>,
               std::conditional_t<C::SRC == SRC_MIXC3, FetchMix<G>,
               std::conditional_t<C::SRC == SRC_POOLEXP, FetchPool<G, PA, false>, FetchPool<G, PA, true>>>>>;
    RA ra;
    [[maybe_unused]] FetchBHalf<G, PB == 0 ? 1 : PB> rb;
    auto fetch = [&](int t) {
        int n0, row0;
        pos(t, n0, row0);
        if constexpr (C::SRC == SRC_F32) fetch_a_f32<G, PA>(ra, (const float4*)P.src_a, n0, row0, N, tid);
        else if constexpr (C::SRC == SRC_U8C3) fetch_a_u8c3<G>(ra, (const uint32_t*)P.src_a, n0, row0, N, tid);
        else if constexpr (C::SRC == SRC_MIXC3) fetch_a_mix<G>(ra, (const uint32_t*)P.mix_a, (const uint32_t*)P.mix_b, (const float4*)P.mix_z, P.mix_n_a, n0, row0, N, tid);
        else if constexpr (C::SRC == SRC_POOLEXP) fetch_poolexp<G, PA, false>(ra, (const float4*)P.src_a, P.amask_in, n0, 0, false, row0, N, tid);
        else fetch_poolexp<G, PA, true>(ra, (const float4*)P.src_a, P.amask_in, n0, P.mix_n_a, P.mix_inject != 0, row0, N, tid);
        if constexpr (PB > 0) fetch_b_half<G, PB>(rb, (const float4*)P.src_b, n0, row0, N, tid);
    };
    auto commit = [&](int t) {
        int n0, row0;
        pos(t, n0, row0);
        auto idx = [](int p, int img, int r, int x) { return ldsA_idx<G, PA>(p, img, r, x + 1); };
        if constexpr (C::SRC == SRC_F32) commit_a_f32<G, PA>(ra, ldsA, n0, row0, N, tid);
        else if constexpr (C::SRC == SRC_U8C3) commit_a_u8c3<G>(ra, ldsA, n0, row0, N, tid);
        else if constexpr (C::SRC == SRC_MIXC3) commit_a_mix<G>(ra, ldsA, P.mix_n_a, n0, row0, N, tid);
        else if constexpr (C::SRC == SRC_POOLEXP) commit_poolexp<G, PA, false>(ra, ldsA, n0, false, row0, N, tid, idx, L::DUMP);
        else commit_poolexp<G, PA, true>(ra, ldsA, n0, P.mix_inject != 0, row0, N, tid, idx, L::DUMP);
        if constexpr (PB > 0) commit_b_half<G, PB>(rb, ldsB, n0, row0, N, tid);
    };
    fetch(0);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        commit(t);
        __syncthreads();
        if (t + 1 < TPW) fetch(t + 1);
        conv_compute<C, true>(P, bid0 + t, tid, smem, pool_lds);
        if (t + 1 < TPW) __syncthreads();      // every wave is done with the tile before the next one is written
    }
}

template <class C>
__global__ void __launch_bounds__(C::THREADS * C::CW) conv3x3_pipe_kernel(ConvParams P) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    conv3x3_body_pipe<C>(P, blockIdx.x * tpw_of<C>::value, smem);
}

template <class C>
__global__ void __launch_bounds__(C::THREADS * C::CW) conv3x3_kernel(ConvParams P) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    conv3x3_body<C, false>(P, blockIdx.x, smem);
}

template <class C>
static size_t conv_lds_bytes() {
    using G = Geo<C::H, C::W, C::THREADS, C::CW>;
    constexpr int PA = (C::SRC == SRC_SCALAR) ? 1 : (C::CA + 3) / 4;
    constexpr int PB = C::CB / 4;
    size_t a = (C::SRC == SRC_SCALAR) ? (size_t)((G::IMGS * G::TRA * (G::W + 2) + 3) / 4) : (size_t)PA * G::IMGS * G::TRA * G::PWA;
    size_t b = PB == 0 ? 0 : (C::UPS == 2 ? (size_t)PB * G::IMGS * G::TRB * G::PWB : (size_t)PB * G::IMGS);
    size_t w = (size_t)(9 * C::WCI * C::WCO + 3) / 4;
    size_t dz = (C::SRC == SRC_DH) ? (size_t)((G::TRA + 2) * (G::W + 4) + 3) / 4 : 0;
    return (a + b + 1 + w + dz) * sizeof(float4);
}

// ------------------------------------------------------------------------------------------------
// Instance table (chfak = 1 Hourglass).  FWD(name, H, THREADS, SRC, CA, CB, UPS, CO, EPI, ACT)
// ------------------------------------------------------------------------------------------------
#define CGS_FWD_CFG(NAME, HW, THR, SRC_, CA_, CB_, UPS_, CO_, EPI_, ACT_, OCB_, CW_)                   \
    struct NAME {                                                                                      \
        static constexpr int H = HW, W = HW, THREADS = THR, SRC = SRC_, CA = CA_, CB = CB_, UPS = UPS_; \
        static constexpr int WT = 0, WCI = CA_ + CB_, WCO = CO_, OC0 = 0, OC = CO_, OCB = OCB_, CW = CW_; \
        static constexpr int EPI = EPI_, ACT = ACT_, OUT_A = 0, POST_ACT = CGS_ACT_NONE;                \
        static constexpr bool DROP = (HW == 8 && CB_ == 0);  /* only features.10 reads a dropped-out input */ \
    };
// data gradient: dY has DYC channels; weights are the layer's HWIO [.,.,LCI,LCO]
#define CGS_DG_CFG(NAME, HW, THR, SRC_, DYC, LCI, LCO, UPS_, OC0_, OC_, OCB_, OUTA, PACT, CW_)         \
    struct NAME {                                                                                      \
        static constexpr int H = HW, W = HW, THREADS = THR, SRC = SRC_, CA = DYC, CB = 0, UPS = UPS_;   \
        static constexpr int WT = 1, WCI = LCI, WCO = LCO, OC0 = OC0_, OC = OC_, OCB = OCB_, CW = CW_;  \
        static constexpr int EPI = EPI_DGRAD, ACT = CGS_ACT_NONE, OUT_A = OUTA, POST_ACT = PACT;        \
        static constexpr bool DROP = (HW == 8 && SRC_ == SRC_POOLEXP);                                   \
    };

CGS_FWD_CFG(FEnc0U8, 64, 256, SRC_U8C3, 3, 0, 2, 8, EPI_POOL, CGS_ACT_RELU, 8, 1)
CGS_FWD_CFG(FEnc0F32, 64, 256, SRC_F32C3, 3, 0, 2, 8, EPI_POOL, CGS_ACT_RELU, 8, 1)
CGS_FWD_CFG(FEnc0Mix, 64, 256, SRC_MIXC3, 3, 0, 2, 8, EPI_POOL, CGS_ACT_RELU, 8, 1)
CGS_FWD_CFG(FEnc1, 32, 128, SRC_F32, 8, 0, 2, 8, EPI_POOL, CGS_ACT_RELU, 4, 2)
CGS_FWD_CFG(FEnc2, 16, 128, SRC_F32, 8, 0, 2, 8, EPI_POOL, CGS_ACT_RELU, 4, 2)
CGS_FWD_CFG(FEnc3, 8, 64, SRC_F32, 8, 0, 2, 16, EPI_POOL, CGS_ACT_RELU, 4, 4)
CGS_FWD_CFG(FDec3, 4, 64, SRC_F32, 16, 32, 4, 16, EPI_PLAIN, CGS_ACT_NONE, 4, 4)
CGS_FWD_CFG(FDec2, 8, 64, SRC_F32, 8, 16, 2, 8, EPI_PLAIN, CGS_ACT_NONE, 2, 4)
CGS_FWD_CFG(FDec1, 16, 64, SRC_F32, 8, 8, 2, 8, EPI_PLAIN, CGS_ACT_NONE, 4, 2)
CGS_FWD_CFG(FDec0, 32, 128, SRC_F32, 8, 8, 2, 8, EPI_PLAIN, CGS_ACT_NONE, 4, 2)
CGS_FWD_CFG(FMask0U8, 64, 128, SRC_U8C3, 3, 8, 2, 16, EPI_PLAIN, CGS_ACT_LRELU, 8, 2)
CGS_FWD_CFG(FMask0F32, 64, 128, SRC_F32C3, 3, 8, 2, 16, EPI_PLAIN, CGS_ACT_LRELU, 8, 2)
CGS_FWD_CFG(FMask2, 64, 256, SRC_F32, 16, 0, 2, 1, EPI_PLAIN, CGS_ACT_SIGMOID, 1, 1)

//          name     HW  THR  SRC          DYC LCI LCO UPS OC0 OC OCB OUT_A post-act       CW
CGS_DG_CFG(DEnc0, 64, 256, SRC_POOLEXP, 8, 3, 8, 2, 0, 3, 3, 3, CGS_ACT_NONE, 1)
CGS_DG_CFG(DEnc1, 32, 128, SRC_POOLEXP, 8, 8, 8, 2, 0, 8, 4, 8, CGS_ACT_NONE, 2)
CGS_DG_CFG(DEnc2, 16, 128, SRC_POOLEXP, 8, 8, 8, 2, 0, 8, 4, 8, CGS_ACT_NONE, 2)
CGS_DG_CFG(DEnc3, 8, 64, SRC_POOLEXP, 16, 8, 16, 2, 0, 8, 4, 8, CGS_ACT_NONE, 2)
CGS_DG_CFG(DDec3, 4, 64, SRC_F32, 16, 48, 16, 4, 0, 48, 8, 16, CGS_ACT_NONE, 6)
CGS_DG_CFG(DDec2, 8, 64, SRC_F32, 8, 24, 8, 2, 0, 24, 8, 8, CGS_ACT_NONE, 3)
CGS_DG_CFG(DDec1, 16, 64, SRC_F32, 8, 16, 8, 2, 0, 16, 4, 8, CGS_ACT_NONE, 4)
CGS_DG_CFG(DDec0, 32, 128, SRC_F32, 8, 16, 8, 2, 0, 16, 8, 8, CGS_ACT_NONE, 2)
CGS_DG_CFG(DMask0, 64, 128, SRC_F32, 16, 11, 16, 2, 3, 8, 4, 3, CGS_ACT_NONE, 2)
CGS_DG_CFG(DMask2, 64, 128, SRC_SCALAR, 1, 16, 1, 2, 0, 16, 8, 16, CGS_ACT_LRELU, 2)
// masker.2 + masker.0 data gradients in one pass (SRC_DH rebuilds d(masker.0 output) in the loader)
CGS_DG_CFG(DMaskHead, 64, 128, SRC_DH, 16, 11, 16, 2, 3, 8, 4, 3, CGS_ACT_NONE, 2)

// software-pipelined forms (conv3x3_body_pipe: a workgroup runs all strips of an image), kept where they measured faster (r4b A/B:
// features.0 on the mixes 32.0 -> 30.8 us, features.0 + mix backward 54.1 -> 50.0, features.3 backward 30.2 -> 29.7, features.0 on the
// frames 28.7 -> 28.4; features.3 forward, dec_model.0 forward / data gradient were neutral to slower: these kernels are bound by
// instruction issue, their load phases already hide behind the other resident workgroups)
#ifndef CGS_CONV_PIPE
#define CGS_CONV_PIPE 1
#endif
struct FEnc0U8P : FEnc0U8 { static constexpr int TPW = 4; };
struct FEnc0MixP : FEnc0Mix { static constexpr int TPW = 4; };
struct DEnc1P : DEnc1 { static constexpr int TPW = 2; };
struct FEnc1P : FEnc1 { static constexpr int TPW = 2; };
struct DDec0P : DDec0 { static constexpr int TPW = 2; };
struct FDec0P : FDec0 { static constexpr int TPW = 2; };      // used by the fused decoder-tail forward + dec_model.0 kernel (tail.hip)      // used by the fused dec_model.0 data gradient + decoder-tail backward kernel (tail.hip)      // used by the fused features.3 + encoder-tail kernel (tail.hip)

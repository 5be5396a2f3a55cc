// Shared device-side helpers for the gfx950 kernels of libcgs_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cgs_hip.h"

#define CGS_WAVE 64

// Weights are read through the constant address space: uniform addresses there always lower to
// s_load_dword* (scalar cache -> SGPRs), so an FMA takes its weight as an SGPR operand and the
// vector memory path stays free for activations.
#define CGS_CONSTANT __attribute__((address_space(4)))
typedef const float CGS_CONSTANT* cgs_cptr;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
__device__ __forceinline__ cgs_cptr cgs_to_const(const float* p) { return (cgs_cptr)p; }
#pragma clang diagnostic pop

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float f4get(const float4& v, int c) {
    return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11), counter = (idx, site, step_lo, step_hi), key = seed.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair: integer multiplies are quarter rate,
        // and Dropout's four tail kernels cost the step 4.8 us (r05: --dropout 0.0 vs 0.3)
        const uint64_t p0 = (uint64_t)M0 * c.x, p1 = (uint64_t)M1 * c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += W0;
        k.y += W1;
    }
    return c;
}

struct DropCtx {
    float p, scale;
    uint32_t site;
    uint2 key;
    uint32_t step_lo, step_hi, base;
    bool on;
};

#ifndef CGS_KARG_PREFETCH
#define CGS_KARG_PREFETCH 1
#endif
// One scalar load per 64-byte line of the kernel arguments, all in one batch, at the top of a kernel whose set-up reads its (several hundred bytes of)
// arguments piecemeal: those reads are scalar loads from a segment the command processor has just written -- cold in the scalar cache -- and the compiler
// waits for each small group before the next (SMEM returns out of order: any use is an s_waitcnt lgkmcnt(0)): ~20 dependent round trips to L2 at the start
// of every workgroup of a launch, all workgroups in lockstep, nothing to hide them behind.  After this the set-up's loads hit the scalar cache.
template <int BYTES>
__device__ __forceinline__ void cgs_kernarg_prefetch() {
    typedef __attribute__((address_space(4))) const uint32_t karg_t;
    karg_t* ka = (karg_t*)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < (BYTES + 63) / 64; ++i) acc |= ka[16 * i];
    asm volatile("" ::"s"(acc));
}

__device__ __forceinline__ DropCtx drop_ctx(const cgs_dropout& d) {
    DropCtx c;
    c.on = d.p > 0.f;
    c.p = d.p;
    c.scale = c.on ? 1.f / (1.f - d.p) : 1.f;
    c.site = d.site;
    c.base = d.base;
    c.key = make_uint2((uint32_t)d.seed, (uint32_t)(d.seed >> 32));
    uint64_t s = (c.on && d.step) ? *d.step : 0ull;
    c.step_lo = (uint32_t)s;
    c.step_hi = (uint32_t)(s >> 32);
    return c;
}

// The same, branch-free: the step counter is read unconditionally (from `safe`, any readable 8-byte-aligned device address, when the
// layer has no Dropout), so several contexts' loads are in flight together instead of one round trip per conditional block.
__device__ __forceinline__ DropCtx drop_ctx(const cgs_dropout& d, const void* safe) {
    DropCtx c;
    c.on = d.p > 0.f;
    c.p = d.p;
    c.scale = c.on ? 1.f / (1.f - d.p) : 1.f;
    c.site = d.site;
    c.base = d.base;
    c.key = make_uint2((uint32_t)d.seed, (uint32_t)(d.seed >> 32));
    const bool have = c.on && d.step;
    const uint64_t* sp = have ? (const uint64_t*)d.step : (const uint64_t*)safe;
    const uint64_t s = have ? *sp : (*sp & 0ull);
    c.step_lo = (uint32_t)s;
    c.step_hi = (uint32_t)(s >> 32);
    return c;
}

#ifndef CGS_DROPCTX3
#define CGS_DROPCTX3 1
#endif
// Three contexts at once: the three step-counter loads (each a scalar round trip to L2 -- the counter was written by the previous step's optimiser
// kernel) are requested back to back BEFORE anything uses one of them; written as three drop_ctx calls the loads of the second and third context
// sit behind the waits of the first one's argument reads.
// The same in two halves: the three step counters REQUESTED (raw values) / the contexts FILLED from them.  A kernel whose first image's
// global loads follow calls _load before them and _fill after: the counters' round trip then flies together with the image's loads instead
// of in front of them (tail_enc_bwd: global_load x 3 -> s_waitcnt vmcnt(0) stood at the top of every workgroup, 2.3 us before its first
// image load by the stage stamps; round 5).
__device__ __forceinline__ void drop_ctx3_load(const cgs_dropout& da, const cgs_dropout& db, const cgs_dropout& dc, const void* safe,
                                               uint64_t& va, uint64_t& vb, uint64_t& vc) {
    const bool ha = da.p > 0.f && da.step, hb = db.p > 0.f && db.step, hc = dc.p > 0.f && dc.step;
    va = *(ha ? (const uint64_t*)da.step : (const uint64_t*)safe);
    vb = *(hb ? (const uint64_t*)db.step : (const uint64_t*)safe);
    vc = *(hc ? (const uint64_t*)dc.step : (const uint64_t*)safe);
}
__device__ __forceinline__ void drop_ctx3_fill(const cgs_dropout& da, const cgs_dropout& db, const cgs_dropout& dc, uint64_t va, uint64_t vb,
                                               uint64_t vc, DropCtx& ca, DropCtx& cb, DropCtx& cc) {
    auto fill = [](DropCtx& c, const cgs_dropout& d, uint64_t v) {
        const bool have = d.p > 0.f && d.step;
        c.on = d.p > 0.f;
        c.p = d.p;
        c.scale = c.on ? 1.f / (1.f - d.p) : 1.f;
        c.site = d.site;
        c.base = d.base;
        c.key = make_uint2((uint32_t)d.seed, (uint32_t)(d.seed >> 32));
        const uint64_t s = have ? v : 0ull;
        c.step_lo = (uint32_t)s;
        c.step_hi = (uint32_t)(s >> 32);
    };
    fill(ca, da, va); fill(cb, db, vb); fill(cc, dc, vc);
}

__device__ __forceinline__ void drop_ctx3(const cgs_dropout& da, const cgs_dropout& db, const cgs_dropout& dc, const void* safe,
                                          DropCtx& ca, DropCtx& cb, DropCtx& cc) {
    const bool ha = da.p > 0.f && da.step, hb = db.p > 0.f && db.step, hc = dc.p > 0.f && dc.step;
    const uint64_t* pa = ha ? (const uint64_t*)da.step : (const uint64_t*)safe;
    const uint64_t* pb = hb ? (const uint64_t*)db.step : (const uint64_t*)safe;
    const uint64_t* pc = hc ? (const uint64_t*)dc.step : (const uint64_t*)safe;
    const uint64_t va = *pa, vb = *pb, vc = *pc;
    auto fill = [](DropCtx& c, const cgs_dropout& d, bool have, uint64_t v) {
        c.on = d.p > 0.f;
        c.p = d.p;
        c.scale = c.on ? 1.f / (1.f - d.p) : 1.f;
        c.site = d.site;
        c.base = d.base;
        c.key = make_uint2((uint32_t)d.seed, (uint32_t)(d.seed >> 32));
        const uint64_t s = have ? v : 0ull;
        c.step_lo = (uint32_t)s;
        c.step_hi = (uint32_t)(s >> 32);
    };
    fill(ca, da, ha, va); fill(cb, db, hb, vb); fill(cc, dc, hc, vc);
}

// Multipliers (0 or 1/(1-p)) for the four consecutive floats whose float4 index is idx4.
__device__ __forceinline__ float4 drop_mult4(const DropCtx& c, uint32_t idx4) {
    uint4 r = philox4x32_10(make_uint4(idx4 + c.base, c.site, c.step_lo, c.step_hi), c.key);
    const float u = 1.f / 16777216.f;
    float4 m;
    m.x = ((r.x >> 8) * u >= c.p) ? c.scale : 0.f;
    m.y = ((r.y >> 8) * u >= c.p) ? c.scale : 0.f;
    m.z = ((r.z >> 8) * u >= c.p) ? c.scale : 0.f;
    m.w = ((r.w >> 8) * u >= c.p) ? c.scale : 0.f;
    return m;
}

__device__ __forceinline__ float4 operator*(const float4& a, const float4& b) {
    return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}

// Wave-level sum over 64 lanes (result in every lane).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <int ACT>
__device__ __forceinline__ float act_fwd(float v) {
    if constexpr (ACT == CGS_ACT_RELU) return v > 0.f ? v : 0.f;
    else if constexpr (ACT == CGS_ACT_LRELU) return v > 0.f ? v : 0.01f * v;
    else if constexpr (ACT == CGS_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    else return v;
}

#define CGS_HIP_CHECK_LAUNCH()                         \
    do {                                               \
        hipError_t e__ = hipGetLastError();            \
        if (e__ != hipSuccess) return (int)e__;        \
    } while (0)

// In-kernel s_memtime stamps (tools/*_stamps.py) exist only in a debug build (tools/build_variant.py stamps -DCGS_DEBUG_STAMPS): in the
// product library the stamp pointer is a compile-time NULL, the stamping code folds away and the dbg_* hooks are not exported.
#ifdef CGS_DEBUG_STAMPS
#define CGS_STAMP_PTR(p) (p)
#else
#define CGS_STAMP_PTR(p) ((unsigned long long*)nullptr)
#endif

// Phase staggering (round 4; used by the fused features.3 + encoder-tail kernel -- in every other kernel of the step it measured neutral to
// +1 us, r4p): the workgroups of such a launch are all resident at once and walk through the same sequence of
// issue-bound and latency-bound phases in lockstep; delaying every other co-resident workgroup by a few microseconds at its start lets
// the latency-bound phases of one half run under the matrix instructions of the other.  Workgroups 256 apart share a CU (8 XCDs x 32
// CUs, round-robin dispatch), hence bit 8.  s_sleep SLEEP = SLEEP x 64 cycles.
// ASSUMES the MI355X SPX partition (one device = 8 XCDs x 32 CUs = 256 CUs, workgroups dealt round-robin): in CPX / DPX modes or on another SKU the
// workgroups bit BIT separates do not share a CU and the sleep would be plain added latency -- so it only happens in launches of MORE than 2^BIT
// workgroups (a smaller launch has no second co-resident workgroup per CU to offset), and never changes results.  cgs_xcd_contiguous below makes the
// same assumption (8 XCDs); under a different dispatch it is merely a different, still bijective, order.
template <int BIT, int SLEEP>
__device__ __forceinline__ void cgs_stagger() {
    if constexpr (SLEEP > 0) {
        if (gridDim.x > (1u << BIT) && ((blockIdx.x >> BIT) & 1)) __builtin_amdgcn_s_sleep(SLEEP);
    }
}

// Virtual workgroup id under which XCD x (= blockIdx.x % 8: workgroups are dealt to the 8 XCDs round-robin, each XCD has its own L2) owns a
// CONTIGUOUS range of ids: neighbouring ids -- neighbouring strips of an image, whose halo rows overlap -- then run on the same XCD at the
// same time and the shared rows come out of that XCD's L2 instead of crossing the fabric once per XCD.  A bijection on [0, g).
__device__ __forceinline__ int cgs_xcd_contiguous(int b, int g) {
    const int per = g >> 3, rem = g & 7, x = b & 7;
    return x * per + (x < rem ? x : rem) + (b >> 3);
}


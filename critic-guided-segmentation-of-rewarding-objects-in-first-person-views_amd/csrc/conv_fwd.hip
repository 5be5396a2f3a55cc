// Forward and data-gradient 3x3 convolutions of the Hourglass (one kernel template, see conv_tile.h).
//   forward : NewCritic.features convs + ReLU + MaxPool2d(2) (+Dropout on the input)   nets.py:170-183
//             UnetDecoder.dec / masker convs with fused Upsample + cat + activation     nets.py:480-521
//   backward: the same core with transposed/flipped weights; pool/ReLU gradient re-expansion in the
//             loader; dropout, LeakyReLU', skip-gradient add and upsample-backward sums in the epilogue.
#include <cstdlib>
#include "conv_body.h"

// layers routed to the MFMA implicit-GEMM kernels (mconv.hip); CGS_MCONV=0 keeps them on the VALU kernels (A/B)
int mconv_fwd_dispatch(int which, int n, const void* src_a, const float* src_b, const float* w, const float* bias, float* out,
                       hipStream_t st);
int mask_head_launch(int n, int img_kind, const void* img, const float* o0, const float* dzpre, const float* h,
                     const float* w2, const float* w0, float* dh, float* d_o0, float* slab2, float* slab0, hipStream_t st);
int mask_head_slabs(int n);
int mask_infer_launch(int n, int img_kind, const void* img, const float* o0, const float* w0, const float* b0, const float* w2,
                      const float* b2, float* z, hipStream_t st);
int mask_train_partials(int n);
int mask_train_launch(int n, int img_kind, const void* img, const float* o0, const float* w0, const float* b0, const float* w2,
                      const float* b2, float* h, float* z, float* zpart, const float* w0_pack, hipStream_t st);
int mask_infer_f16_launch(int n, int img_kind, const void* img, const float* o0, const float* w0, const float* b0,
                          const float* w2, const float* b2, float* z, hipStream_t st, int o0_f16);
// (round 2's opt-in two-pixels-per-MFMA-row form of features.0 / features.3 -- measured slower -- lives in tools/experiments/pconv.hip,
//  outside the product library; the 3x3 layers now run on v_mfma_f32_4x4x1 inside conv3x3_body)
static constexpr bool use_mconv() { return true; }      // the layers with a matrix-core implicit-GEMM kernel always use it

// software-pipelined form: TPW consecutive strips per workgroup (conv3x3_body_pipe)
template <class C>
static int launch_conv_pipe(const ConvParams& P, hipStream_t st) {
    using G = Geo<C::H, C::W, C::THREADS, C::CW>;
    static_assert(G::IMGS == 1 && G::STRIPS % tpw_of<C>::value == 0, "a workgroup's strips belong to one image");
    if (P.n <= 0) return CGS_OK;
    const int blocks = P.n * G::STRIPS / tpw_of<C>::value;
    const size_t lds = conv_lds_bytes<C>();
    static_assert(sizeof(float4) > 0, "");
    hipLaunchKernelGGL(conv3x3_pipe_kernel<C>, dim3(blocks), dim3(C::THREADS * C::CW), lds, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

template <class C>
static int launch_conv(const ConvParams& P, hipStream_t st) {
    using G = Geo<C::H, C::W, C::THREADS, C::CW>;
    if (P.n <= 0) return CGS_OK;
    int blocks = (G::IMGS == 1) ? P.n * G::STRIPS : (P.n + G::IMGS - 1) / G::IMGS;
    const size_t lds = conv_lds_bytes<C>();
    if (lds > 64 * 1024) {   // more than the default dynamic-LDS limit: raise it once for this instance
        static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<C>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (attr != hipSuccess) return (int)attr;
    }
    hipLaunchKernelGGL(conv3x3_kernel<C>, dim3(blocks), dim3(C::THREADS * C::CW), lds, st, P);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

static bool desc_is(const cgs_conv_desc* d, int hw, int ca, int cb, int co, int src, int ups, int act, int pool) {
    return d->h == hw && d->w == hw && d->ca == ca && d->cb == cb && d->co == co && d->src_a == src &&
           (cb == 0 || d->ups == ups) && d->act == act && d->pool == pool;
}

extern "C" int cgs_conv3x3_fwd(const cgs_conv_desc* d, const void* src_a, const float* src_b, const float* w,
                               const float* bias, float* out, uint32_t* amask, cgs_stream_t stream) {
    if (!d || !src_a || !w || !bias || !out || d->n < 0) return CGS_ERR_BADARG;
    if (d->cb > 0 && !src_b) return CGS_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    ConvParams P{};
    P.src_a = src_a; P.src_b = src_b; P.w = w; P.bias = bias; P.out = out; P.amask_out = amask;
    P.n = d->n; P.drop = d->drop_a;
    if (d->src_a == CGS_SRC_MIX) {     // features.0 on the replaced | injected mixes, computed in the loader
        const cgs_mix_src* m = (const cgs_mix_src*)src_a;       // HOST struct
        if (!m->a || !m->b || !m->z || m->n_a <= 0 || d->n > 2 * m->n_a) return CGS_ERR_BADARG;
        if (!(d->h == 64 && d->w == 64 && d->ca == 3 && d->cb == 0 && d->co == 8 && d->act == CGS_ACT_RELU && d->pool == 1 &&
              d->drop_a.p == 0.f))
            return CGS_ERR_UNSUPPORTED;
        P.src_a = nullptr; P.mix_a = m->a; P.mix_b = m->b; P.mix_z = m->z; P.mix_n_a = m->n_a;
        return CGS_CONV_PIPE ? launch_conv_pipe<FEnc0MixP>(P, st) : launch_conv<FEnc0Mix>(P, st);
    }
    const int R = CGS_ACT_RELU, L = CGS_ACT_LRELU, S = CGS_ACT_SIGMOID, NO = CGS_ACT_NONE;
    if (d->drop_a.p > 0.f && !(desc_is(d, 8, 8, 0, 16, CGS_SRC_F32, 2, R, 1))) return CGS_ERR_UNSUPPORTED;
    if (desc_is(d, 64, 3, 0, 8, CGS_SRC_U8, 2, R, 1))
        return CGS_CONV_PIPE ? launch_conv_pipe<FEnc0U8P>(P, st) : launch_conv<FEnc0U8>(P, st);
    if (desc_is(d, 64, 3, 0, 8, CGS_SRC_F32, 2, R, 1)) return launch_conv<FEnc0F32>(P, st);
    if (desc_is(d, 32, 8, 0, 8, CGS_SRC_F32, 2, R, 1))
        return launch_conv<FEnc1>(P, st);      // (pipelined form: 17.9 vs 17.7 us, r4b)
    if (desc_is(d, 16, 8, 0, 8, CGS_SRC_F32, 2, R, 1)) return launch_conv<FEnc2>(P, st);
    if (desc_is(d, 8, 8, 0, 16, CGS_SRC_F32, 2, R, 1)) return launch_conv<FEnc3>(P, st);
    if (desc_is(d, 4, 16, 32, 16, CGS_SRC_F32, 4, NO, 0))
        return use_mconv() ? mconv_fwd_dispatch(2, d->n, src_a, src_b, w, bias, out, st) : launch_conv<FDec3>(P, st);
    if (desc_is(d, 8, 8, 16, 8, CGS_SRC_F32, 2, NO, 0)) return launch_conv<FDec2>(P, st);
    if (desc_is(d, 16, 8, 8, 8, CGS_SRC_F32, 2, NO, 0)) return launch_conv<FDec1>(P, st);
    if (desc_is(d, 32, 8, 8, 8, CGS_SRC_F32, 2, NO, 0)) return launch_conv<FDec0>(P, st);      // (pipelined form: 19.2 vs 17.7 us)
    if (desc_is(d, 64, 3, 8, 16, CGS_SRC_U8, 2, L, 0))
        return use_mconv() ? mconv_fwd_dispatch(0, d->n, src_a, src_b, w, bias, out, st) : launch_conv<FMask0U8>(P, st);
    if (desc_is(d, 64, 3, 8, 16, CGS_SRC_F32, 2, L, 0))
        return use_mconv() ? mconv_fwd_dispatch(1, d->n, src_a, src_b, w, bias, out, st) : launch_conv<FMask0F32>(P, st);
    if (desc_is(d, 64, 16, 0, 1, CGS_SRC_F32, 2, S, 0)) {
        P.amask_out = nullptr; P.zpart = (float*)amask;      // the mask layer: `amask` carries the optional z partial sums
        return launch_conv<FMask2>(P, st);
    }
    return CGS_ERR_UNSUPPORTED;
}

extern "C" int cgs_conv3x3_bwd_data(const cgs_conv_desc* d, const float* dy, const uint32_t* amask, const float* w,
                                    const float* src_a_post, int32_t src_a_act, const float* addend,
                                    int32_t n_addend, float* d_a, float* d_b, cgs_stream_t stream) {
    if (!d || !dy || !w || d->n < 0) return CGS_ERR_BADARG;
    if (d->pool && !amask) return CGS_ERR_BADARG;
    if (!d_a && !d_b) return CGS_OK;
    hipStream_t st = (hipStream_t)stream;
    ConvParams P{};
    P.src_a = dy; P.amask_in = amask; P.w = w; P.out = d_a; P.out2 = d_b; P.addend = addend;
    P.n_addend = n_addend; P.a_post = src_a_post; P.n = d->n; P.drop = d->drop_a;
    const int R = CGS_ACT_RELU, L = CGS_ACT_LRELU, S = CGS_ACT_SIGMOID, NO = CGS_ACT_NONE;
    if (d_b && d->cb == 0) return CGS_ERR_BADARG;
    if (d->drop_a.p > 0.f && !(desc_is(d, 8, 8, 0, 16, CGS_SRC_F32, 2, R, 1))) return CGS_ERR_UNSUPPORTED;
    if (src_a_act != CGS_ACT_NONE && !(src_a_act == L && src_a_post && desc_is(d, 64, 16, 0, 1, CGS_SRC_F32, 2, S, 0)))
        return CGS_ERR_UNSUPPORTED;
    if (desc_is(d, 64, 3, 0, 8, CGS_SRC_U8, 2, R, 1) || desc_is(d, 64, 3, 0, 8, CGS_SRC_F32, 2, R, 1))
        return launch_conv<DEnc0>(P, st);
    if (desc_is(d, 32, 8, 0, 8, CGS_SRC_F32, 2, R, 1)) return launch_conv<DEnc1>(P, st);
    if (desc_is(d, 16, 8, 0, 8, CGS_SRC_F32, 2, R, 1)) return launch_conv<DEnc2>(P, st);
    if (desc_is(d, 8, 8, 0, 16, CGS_SRC_F32, 2, R, 1)) return launch_conv<DEnc3>(P, st);
    if (desc_is(d, 4, 16, 32, 16, CGS_SRC_F32, 4, NO, 0)) return launch_conv<DDec3>(P, st);
    if (desc_is(d, 8, 8, 16, 8, CGS_SRC_F32, 2, NO, 0)) return launch_conv<DDec2>(P, st);
    if (desc_is(d, 16, 8, 8, 8, CGS_SRC_F32, 2, NO, 0)) return launch_conv<DDec1>(P, st);
    if (desc_is(d, 32, 8, 8, 8, CGS_SRC_F32, 2, NO, 0)) return launch_conv<DDec0>(P, st);      // (pipelined form: 20.9 vs 18.7 us)
    if (desc_is(d, 64, 3, 8, 16, CGS_SRC_U8, 2, L, 0) || desc_is(d, 64, 3, 8, 16, CGS_SRC_F32, 2, L, 0)) {
        if (d_a) return CGS_ERR_UNSUPPORTED;  // the image needs no gradient on this path
        return launch_conv<DMask0>(P, st);
    }
    if (desc_is(d, 64, 16, 0, 1, CGS_SRC_F32, 2, S, 0)) {
        if (src_a_act != L) return CGS_ERR_UNSUPPORTED;
        return launch_conv<DMask2>(P, st);
    }
    return CGS_ERR_UNSUPPORTED;
}

extern "C" int cgs_mask_head_bwd_slabs(int32_t n) {
    if (n < 0) return CGS_ERR_BADARG;
    return use_mconv() ? mask_head_slabs(n) : 0;
}

extern "C" int cgs_mask_head_bwd(int32_t n, int32_t src_a, const void* x, const float* o0, const float* dzpre, const float* h,
                                 const float* w_m2, const float* w_m0, float* d_h, float* d_o0, float* slab_m2,
                                 float* slab_m0, cgs_stream_t stream) {
    if (n < 0 || !dzpre || !h || !w_m2 || !w_m0 || !d_o0) return CGS_ERR_BADARG;
    if (slab_m0 && (!slab_m2 || !x || !o0 || (src_a != CGS_SRC_U8 && src_a != CGS_SRC_F32))) return CGS_ERR_BADARG;
    if (use_mconv())
        return mask_head_launch(n, src_a, x, o0, dzpre, h, w_m2, w_m0, d_h, d_o0, slab_m2, slab_m0, (hipStream_t)stream);
    // VALU build (CGS_MCONV=0): data gradients only; cgs_mask_head_bwd_slabs() said 0, the caller runs the separate wgrads
    if (slab_m2 || slab_m0) return CGS_ERR_UNSUPPORTED;
    if (!d_h) return CGS_ERR_BADARG;
    ConvParams P{};
    P.src_a = dzpre; P.a_post = h; P.w2 = w_m2; P.w = w_m0; P.dh_out = d_h; P.out2 = d_o0; P.n = n;
    return launch_conv<DMaskHead>(P, (hipStream_t)stream);
}

extern "C" int cgs_mask_infer_fwd_packed(int32_t n, int32_t src_a, const void* x, const float* o0, const float* w_m0, const float* b_m0,
                                         const float* w_m2, const float* b_m2, float* z, const float* w_m0_pack, cgs_stream_t stream);
extern "C" int cgs_mask_infer_fwd(int32_t n, int32_t src_a, const void* x, const float* o0, const float* w_m0, const float* b_m0,
                                  const float* w_m2, const float* b_m2, float* z, cgs_stream_t stream) {
    return cgs_mask_infer_fwd_packed(n, src_a, x, o0, w_m0, b_m0, w_m2, b_m2, z, nullptr, stream);
}

extern "C" int cgs_mask_infer_fwd_packed(int32_t n, int32_t src_a, const void* x, const float* o0, const float* w_m0, const float* b_m0,
                                         const float* w_m2, const float* b_m2, float* z, const float* w_m0_pack, cgs_stream_t stream) {
    if (n < 0 || !x || !o0 || !w_m0 || !b_m0 || !w_m2 || !b_m2 || !z) return CGS_ERR_BADARG;
    if (src_a != CGS_SRC_U8 && src_a != CGS_SRC_F32) return CGS_ERR_BADARG;
    if (!use_mconv()) return CGS_ERR_UNSUPPORTED;     // VALU build: the caller runs masker.0 and masker.2 as two convolutions
    // (round 5) the TRAINING forward's kernel with nothing stored but Z: masker.2 on the matrix cores from registers instead of 144 multiply-adds and 36
    // 16-byte LDS reads per pixel on an h tile -- 2048 frames: 317 -> see DESIGN 8.8 us; Z is then bit-identical to the training forward's.
    // CGS_MASK_INFER_KERNEL=tile selects the tile kernel (mask_infer_kernel) again.
    static const bool tile = [] { const char* e = getenv("CGS_MASK_INFER_KERNEL"); return e && e[0] == 't'; }();
    if (!tile) return mask_train_launch(n, src_a, x, o0, w_m0, b_m0, w_m2, b_m2, nullptr, z, nullptr, w_m0_pack, (hipStream_t)stream);
    return mask_infer_launch(n, src_a, x, o0, w_m0, b_m0, w_m2, b_m2, z, (hipStream_t)stream);
}

// the tile kernel (mask_infer_kernel: masker.0 into an LDS tile, masker.2 + sigmoid on that tile), whatever the environment says
extern "C" int cgs_mask_infer_fwd_tile(int32_t n, int32_t src_a, const void* x, const float* o0, const float* w_m0, const float* b_m0,
                                       const float* w_m2, const float* b_m2, float* z, cgs_stream_t stream) {
    if (n < 0 || !x || !o0 || !w_m0 || !b_m0 || !w_m2 || !b_m2 || !z) return CGS_ERR_BADARG;
    if (src_a != CGS_SRC_U8 && src_a != CGS_SRC_F32) return CGS_ERR_BADARG;
    if (!use_mconv()) return CGS_ERR_UNSUPPORTED;
    return mask_infer_launch(n, src_a, x, o0, w_m0, b_m0, w_m2, b_m2, z, (hipStream_t)stream);
}

extern "C" int cgs_mask_train_fwd_partials(int32_t n) { return n < 0 ? CGS_ERR_BADARG : mask_train_partials(n); }

extern "C" int cgs_mask_train_fwd_packed(int32_t n, int32_t src_a, const void* x, const float* o0, const float* w_m0, const float* b_m0,
                                         const float* w_m2, const float* b_m2, float* h, float* z, float* zpart, const float* w_m0_pack,
                                         cgs_stream_t stream) {
    if (n < 0 || !x || !o0 || !w_m0 || !b_m0 || !w_m2 || !b_m2 || !h || !z || !zpart) return CGS_ERR_BADARG;
    if (src_a != CGS_SRC_U8 && src_a != CGS_SRC_F32) return CGS_ERR_BADARG;
    return mask_train_launch(n, src_a, x, o0, w_m0, b_m0, w_m2, b_m2, h, z, zpart, w_m0_pack, (hipStream_t)stream);
}

extern "C" int cgs_mask_train_fwd(int32_t n, int32_t src_a, const void* x, const float* o0, const float* w_m0, const float* b_m0,
                                  const float* w_m2, const float* b_m2, float* h, float* z, float* zpart, cgs_stream_t stream) {
    return cgs_mask_train_fwd_packed(n, src_a, x, o0, w_m0, b_m0, w_m2, b_m2, h, z, zpart, nullptr, stream);
}

extern "C" int cgs_mask_infer_fwd_f16(int32_t n, int32_t src_a, const void* x, const float* o0, const float* w_m0,
                                      const float* b_m0, const float* w_m2, const float* b_m2, float* z, cgs_stream_t stream) {
    if (n < 0 || !x || !o0 || !w_m0 || !b_m0 || !w_m2 || !b_m2 || !z) return CGS_ERR_BADARG;
    if (src_a != CGS_SRC_U8 && src_a != CGS_SRC_F32) return CGS_ERR_BADARG;
    return mask_infer_f16_launch(n, src_a, x, o0, w_m0, b_m0, w_m2, b_m2, z, (hipStream_t)stream, 0);
}

// the same with o0 stored as fp16 NHWC (cgs_f16_dec0_fwd's output): the mask head of the fused fp16 inference path
extern "C" int cgs_mask_infer_fwd_f16o(int32_t n, int32_t src_a, const void* x, const void* o0_f16, const float* w_m0,
                                       const float* b_m0, const float* w_m2, const float* b_m2, float* z, cgs_stream_t stream) {
    if (n < 0 || !x || !o0_f16 || !w_m0 || !b_m0 || !w_m2 || !b_m2 || !z) return CGS_ERR_BADARG;
    if (src_a != CGS_SRC_U8 && src_a != CGS_SRC_F32) return CGS_ERR_BADARG;
    return mask_infer_f16_launch(n, src_a, x, (const float*)o0_f16, w_m0, b_m0, w_m2, b_m2, z, (hipStream_t)stream, 1);
}

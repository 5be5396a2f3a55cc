// Critic head (4x4 valid conv = [n,256]x[256,32] GEMM, Linear 32->32, Linear 32->1, sigmoid; nets.py:184-194)
// and the decoder's 1x1 bottleneck conv (nets.py:484,501) -- the true GEMMs of the path.
#include "head_body.h"

// ------------------------------------------------------------------------------------------------
extern "C" int cgs_head_fwd(int32_t n, const float* e3, const float* w4, const float* b4, const float* w1,
                            const float* b1, const float* w2, const float* b2, cgs_dropout drop_in,
                            cgs_dropout drop_h, float* e4, float* h1, float* pred, const float* w_pw, const float* b_pw,
                            float* o4, cgs_stream_t stream) {
    if (n < 0 || !e3 || !w4 || !b4 || !w1 || !b1 || !w2 || !b2 || !e4 || !h1 || !pred) return CGS_ERR_BADARG;
    if (o4 && (!w_pw || !b_pw)) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(head_fwd_kernel, dim3((n + 7) / 8), dim3(256), HEAD_FWD_LDS, (hipStream_t)stream, n, e3, w4, b4, w1, b1, w2,
                       b2, drop_in, drop_h, e4, h1, pred, w_pw, b_pw, o4);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_head_bwd_slabs(int32_t n) { return n < 0 ? CGS_ERR_BADARG : (n + HB_IPB - 1) / HB_IPB; }

extern "C" int cgs_head_bwd(int32_t n, const float* e3, const float* e4, const float* h1, const float* pred,
                            const float* dpred, const float* d_e4_extra, const float* d_e3_extra, int32_t n_extra,
                            const float* w4, const float* w1, const float* w2, cgs_dropout drop_in, cgs_dropout drop_h,
                            float* d_e3, float* slab, const float* d_o4, const float* w_pw, float* slab_pw,
                            cgs_stream_t stream) {
    if (n < 0 || !e3 || !e4 || !h1 || !pred || !dpred || !w4 || !w1 || !w2 || !d_e3 || !slab) return CGS_ERR_BADARG;
    if (d_o4 && (!w_pw || !slab_pw)) return CGS_ERR_BADARG;
    if (n == 0) return CGS_OK;
    static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&head_bwd_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, HEAD_BWD_LDS);
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL(head_bwd_kernel, dim3((n + HB_IPB - 1) / HB_IPB), dim3(256), HEAD_BWD_LDS, (hipStream_t)stream, n, e3, e4, h1,
                       pred, dpred, d_e4_extra, d_e3_extra, n_extra, w4, w1, w2, drop_in, drop_h, d_e3, slab, d_o4, w_pw, slab_pw);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_pointwise_fwd(int32_t n, int32_t ci, int32_t co, const float* x, const float* w, const float* b,
                                 float* y, cgs_stream_t stream) {
    if (n < 0 || !x || !w || !b || !y) return CGS_ERR_BADARG;
    if (ci != 32 || co != 32) return CGS_ERR_UNSUPPORTED;
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(pointwise32_fwd_kernel, dim3((n + 31) / 32), dim3(64), 0, (hipStream_t)stream, n, x, w, b, y);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

extern "C" int cgs_pointwise_bwd_slabs(int32_t n) { return n < 0 ? CGS_ERR_BADARG : (n + 31) / 32; }

extern "C" int cgs_pointwise_bwd(int32_t n, int32_t ci, int32_t co, const float* x, const float* dy, const float* w,
                                 float* dx, float* slab, cgs_stream_t stream) {
    if (n < 0 || !x || !dy || !w || !slab) return CGS_ERR_BADARG;
    if (ci != 32 || co != 32) return CGS_ERR_UNSUPPORTED;
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(pointwise32_bwd_kernel, dim3((n + 31) / 32), dim3(64), 0, (hipStream_t)stream, n, x, dy, w, dx, slab);
    CGS_HIP_CHECK_LAUNCH();
    return CGS_OK;
}

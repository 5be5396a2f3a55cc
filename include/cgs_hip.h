/*
 * cgs_hip.h -- C ABI of libcgs_hip.so: the MI355X (gfx950) kernels behind the Hourglass
 * (encoder + critic head + decoder/mask head) training and inference path.
 *
 * The reference (ndrwmlnk/critic-guided-segmentation-...) has no FFI/plugin boundary: its hot
 * path is ordinary PyTorch module calls.  Each entry point below therefore names the reference
 * lines whose ATen dispatches it replaces.  Conventions:
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless marked "host";
 *   - the caller owns every buffer (inputs, outputs, saved-for-backward side buffers, slabs);
 *     the library never allocates, frees or retains pointers across calls;
 *   - every call enqueues on the given hipStream_t and never synchronises it (graph-capturable);
 *   - return value: 0 = CGS_OK, CGS_ERR_* (< 0) for argument errors, a positive hipError_t otherwise.
 *   - activations are NHWC fp32; images are NHWC uint8 or NHWC fp32 (3 channels);
 *   - conv weights are HWIO fp32  w[ky][kx][ci][co]  (checkpoints stay OIHW; the host permutes);
 *     input-channel order of a two-source conv is (source A channels, then source B channels),
 *     i.e. the reference's torch.cat((skip, upsampled), 1) / torch.cat((X, upsampled), 1).
 */
#ifndef CGS_HIP_H
#define CGS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cgs_stream_t; /* hipStream_t */

enum { CGS_OK = 0, CGS_ERR_UNSUPPORTED = -1, CGS_ERR_BADARG = -2 };
enum { CGS_SRC_F32 = 0, CGS_SRC_U8 = 1, CGS_SRC_MIX = 2 };

/* CGS_SRC_MIX (features.0 only): `src_a` points to this HOST struct; the source image n of the launch is computed in the
 * tile loader and never stored: n < n_a: A[n](1-Z[n]) + Z[n] B[n] (replaced), n >= n_a: B(1-Z) + Z A of A-image n - n_a
 * (injected) -- main.py:395,406.  a, b: uint8 [n_a,64,64,3]; z: fp32 [n_a,64,64] (all device pointers).            */
typedef struct {
    const uint8_t* a;
    const uint8_t* b;
    const float* z;
    int32_t n_a;
    int32_t reserved;
} cgs_mix_src;
enum { CGS_ACT_NONE = 0, CGS_ACT_RELU = 1, CGS_ACT_LRELU = 2, CGS_ACT_SIGMOID = 3 };

/* Dropout on a tensor = Philox4x32-10 keep-mask keyed by (seed; base + element/4, site, *step).
 * p == 0 disables it.  `step` points to a device-resident counter so that a captured graph
 * draws fresh masks on every replay; forward and backward of one step read the same value.
 * `base` (in units of 4 floats) is added to the element index: a call that works on a slice of a
 * larger batch buffer passes the slice's offset so that forward and backward launches that slice
 * the batch differently still draw identical masks for identical images. */
typedef struct {
    float p;
    uint32_t site;
    uint64_t seed;
    const uint64_t* step; /* device; may be NULL when p == 0 */
    uint32_t base;
    uint32_t reserved;
} cgs_dropout;

/* One 3x3, stride-1, pad-1 convolution of the Hourglass (nets.py:170-183, 480-490). */
typedef struct {
    int32_t n;      /* images */
    int32_t h, w;   /* conv output size before pooling (= size of source A) */
    int32_t ca;     /* channels of source A (direct / skip input) */
    int32_t cb;     /* channels of source B (nearest-upsampled low-res input); 0 = none */
    int32_t co;     /* output channels */
    int32_t src_a;  /* CGS_SRC_F32 | CGS_SRC_U8 (uint8 image, /255 fused into the loader) | CGS_SRC_MIX (see cgs_mix_src) */
    int32_t ups;    /* source B scale: 2 (source is h/2 x w/2) or 4 (source is 1x1, h = w = 4) */
    int32_t act;    /* CGS_ACT_* applied after bias */
    int32_t pool;   /* 1: fused 2x2/2 max-pool after the activation */
    cgs_dropout drop_a; /* dropout applied to source A while loading (nets.py:179,183) */
} cgs_conv_desc;

/* ---- convolution forward ------------------------------------------------------------
 * Replaces Conv2d + ReLU/LeakyReLU/Sigmoid + MaxPool2d + Upsample + cat + Dropout of
 * NewCritic.forward (nets.py:197-205) and UnetDecoder.forward (nets.py:500-521).
 *   out   : [n, h, w, co]  (pool=0)  or [n, h/2, w/2, co] (pool=1)
 *   amask : pool=1 only, may be NULL.  uint32 [n, h/2, w/2, co/8]: one nibble per channel =
 *           index (0..3, row-major in the 2x2 window, first maximum wins) of the pooled
 *           element, or 0xF when the pooled value is <= 0 (ReLU gradient is zero there).
 *           For the mask layer (h=w=64, ca=16, co=1, sigmoid, pool=0) the same slot optionally receives, as floats
 *           [n*4][2], the per-workgroup partial sums (sum |z|, sum z^2) that cgs_phase2_losses consumes.         */
int cgs_conv3x3_fwd(const cgs_conv_desc* d, const void* src_a, const float* src_b,
                    const float* w_hwio, const float* bias, float* out, uint32_t* amask,
                    cgs_stream_t stream);

/* ---- convolution backward, data -----------------------------------------------------
 * Replaces the convolution_backward(input) + max_pool2d/threshold/upsample/cat/dropout backward
 * nodes autograd builds for the layer (loss.backward(), main.py:198,462).
 *   dy      : gradient w.r.t. the layer output: [n,h,w,co], or [n,h/2,w/2,co] + amask when pool=1
 *             (the pre-pool gradient is re-expanded on the fly, ReLU mask included).
 *   d_a     : [n,h,w,ca] gradient w.r.t. source A, or NULL to skip.  Fused epilogue:
 *             d_a = conv_bwd * dropout_mask(drop_a) * act'(src_a_post) + addend
 *               src_a_post/src_a_act: optional (may be NULL/NONE) post-activation values of source A
 *               and the activation that produced them (LeakyReLU for masker.0 -> masker.2);
 *               addend: optional [n_addend,h,w,ca] gradient already accumulated for source A (skip
 *               connection), added for images < n_addend; may alias d_a.
 *   d_b     : gradient w.r.t. source B at ITS resolution ([n,h/2,w/2,cb] or [n,cb] for ups=4), i.e. the
 *             nearest-upsample backward (2x2 / 4x4 sum) is fused; NULL to skip.            */
int cgs_conv3x3_bwd_data(const cgs_conv_desc* d, const float* dy, const uint32_t* amask,
                         const float* w_hwio, const float* src_a_post, int32_t src_a_act,
                         const float* addend, int32_t n_addend, float* d_a, float* d_b,
                         cgs_stream_t stream);

/* ---- mask head backward in one pass (nets.py:488-491 backward: masker.2 and masker.0) ----------------
 * dzpre [n,64,64] (gradient w.r.t. the pre-sigmoid mask), h [n,64,64,16] (saved masker.0 output, post-LeakyReLU).
 *   d_o0 [n,32,32,8]  = upsample-backward of conv_bwd_data(d_h; w_m0) restricted to the 8 decoder channels, where
 *                       d_h = LeakyReLU'(h) * conv_bwd_data(dzpre; w_m2) is rebuilt on the fly from the 1-channel
 *                       dzpre (never read from memory) and the nearest-upsample in front of masker.0 is folded into
 *                       its weights (one stride-2 4x4 convolution on the matrix cores);
 *   d_h  [n,64,64,16] : optional copy of that intermediate (NULL: never stored);
 *   slab_m2 : optional [cgs_mask_head_bwd_slabs(n)][145]  partial masker.2 weight+bias gradients,
 *   slab_m0 : optional [cgs_mask_head_bwd_slabs(n)][1600] partial masker.0 weight+bias gradients (needs slab_m2, x =
 *             the masker.0 image input [n,64,64,3] of kind src_a, and o0 [n,32,32,8]); both in the slab layout of
 *             cgs_conv3x3_bwd_weight for those layers, to be summed by cgs_reduce_slabs().
 * cgs_mask_head_bwd_slabs() returns 0 when this build cannot produce the slabs (the caller then passes NULL, asks for
 * d_h and runs cgs_conv3x3_bwd_weight for the two layers).                                                      */
int cgs_mask_head_bwd_slabs(int32_t n);
int cgs_mask_head_bwd(int32_t n, int32_t src_a, const void* x, const float* o0, const float* dzpre, const float* h,
                      const float* w_m2_hwio, const float* w_m0_hwio, float* d_h, float* d_o0, float* slab_m2,
                      float* slab_m0, cgs_stream_t stream);

/* ---- mask head forward for inference (nets.py:488-491 in eval mode) in one pass -----------------------
 * z [n,64,64] = sigmoid(conv3x3(LeakyReLU(conv3x3(cat(x, up(o0)); w_m0, b_m0)); w_m2, b_m2)) without ever storing the
 * 16-channel intermediate (it is needed only by the backward pass).  x [n,64,64,3] of kind src_a (CGS_SRC_U8 / CGS_SRC_F32), o0 [n,32,32,8].
 * Since round 5 this is the training forward's kernel (cgs_mask_train_fwd) storing nothing but z: masker.0 and masker.2 on the
 * matrix cores from registers; z is bit-identical to the training forward's.  (cgs_mask_infer_fwd_tile, or CGS_MASK_INFER_KERNEL=tile in the
 * environment: the earlier tile kernel -- masker.0 into an LDS tile, masker.2 + sigmoid on that tile.)  cgs_mask_infer_fwd_packed: the same with
 * masker.0's per-lane weight registers w_m0_pack [40][64] as cgs_tail_dec_fwd_pack / cgs_tail_dec_fwd_dec0 build them (NULL: gathered
 * per workgroup).  Returns CGS_ERR_UNSUPPORTED in the VALU fallback build (use two cgs_conv3x3_fwd calls).          */
int cgs_mask_infer_fwd(int32_t n, int32_t src_a, const void* x, const float* o0, const float* w_m0_hwio,
                       const float* b_m0, const float* w_m2_hwio, const float* b_m2, float* z,
                       cgs_stream_t stream);
int cgs_mask_infer_fwd_packed(int32_t n, int32_t src_a, const void* x, const float* o0, const float* w_m0_hwio,
                       const float* b_m0, const float* w_m2_hwio, const float* b_m2, float* z, const float* w_m0_pack,
                       cgs_stream_t stream);
int cgs_mask_infer_fwd_tile(int32_t n, int32_t src_a, const void* x, const float* o0, const float* w_m0_hwio,
                            const float* b_m0, const float* w_m2_hwio, const float* b_m2, float* z, cgs_stream_t stream);

/* Training form of the same pass (nets.py:488-491 in train mode; replaces masker.0's and masker.2's separate forwards): also
 * stores h [n,64,64,16] = LeakyReLU(masker.0) ONCE from the on-chip tile (the backward pass needs it) and leaves, per tile,
 * (sum |z|, sum z^2) for the L1 / L2 mask losses of main.py:421-429 in zpart [2 * cgs_mask_train_fwd_partials(n)].
 * h is never re-read to compute z (the stand-alone masker.2 forward read all 134 MB of it at n = 512).              */
int cgs_mask_train_fwd_partials(int32_t n);
int cgs_mask_train_fwd(int32_t n, int32_t src_a, const void* x, const float* o0, const float* w_m0_hwio,
                       const float* b_m0, const float* w_m2_hwio, const float* b_m2, float* h, float* z,
                       float* zpart, cgs_stream_t stream);
/* The same with masker.0's per-lane weight registers (40 x 64 floats) taken from w_m0_pack instead of rebuilt by every workgroup;
 * cgs_tail_dec_fwd_pack -- the launch before it on the training path -- builds them in one spare workgroup (w_m0_pack = NULL: as above). */
int cgs_mask_train_fwd_packed(int32_t n, int32_t src_a, const void* x, const float* o0, const float* w_m0_hwio, const float* b_m0,
                              const float* w_m2, const float* b_m2, float* h, float* z, float* zpart, const float* w_m0_pack,
                              cgs_stream_t stream);

/* Same contract with fp16 OPERANDS for the masker.0 GEMM (v_mfma_f32_16x16x16_f16, fp32 accumulate; masker.2 stays
 * fp32): BASELINE config 4 ("-process inference-only, fp16 conv kernels").  Opt-in: z differs from the fp32 result by
 * about 1e-3 absolute; never used by training.                                                                  */
int cgs_mask_infer_fwd_f16(int32_t n, int32_t src_a, const void* x, const float* o0, const float* w_m0_hwio,
                           const float* b_m0, const float* w_m2_hwio, const float* b_m2, float* z,
                           cgs_stream_t stream);

/* ---- fused fp16 inference path (BASELINE config 4; csrc/hconv.hip): the three 64x64 / 32x32 convolutions of the eval-mode forward
 * (nets.py:170-176, 516-517) with fp16 activations / weights and fp32 accumulation on v_mfma_f32_16x16x32_f16.  cgs_f16_enc0_fwd:
 * uint8 frames [n,64,64,3] -> e0 fp16 [n,32,32,8] (conv + ReLU + MaxPool2d(2)); cgs_f16_enc1_fwd: e0 -> e1 fp32 [n,16,16,8] (the
 * 16x16-and-smaller layers stay on the fp32 tail kernels); cgs_f16_dec0_fwd: cat(e0, nearest-up(o1 fp32 [n,16,16,8])) -> o0 fp16
 * [n,32,32,8]; cgs_mask_infer_fwd_f16o: cgs_mask_infer_fwd_f16 reading that fp16 o0.  Weights / bias: the layer's HWIO fp32 parameters.
 * Opt-in precision (engine.infer(fp16=True)); never used by training.                                                              */
int cgs_f16_enc0_fwd(int32_t n, const uint8_t* x_u8, const float* w_hwio, const float* bias, void* e0_f16, cgs_stream_t stream);
int cgs_f16_enc1_fwd(int32_t n, const void* e0_f16, const float* w_hwio, const float* bias, float* e1_f32, cgs_stream_t stream);
int cgs_f16_dec0_fwd(int32_t n, const void* e0_f16, const float* o1_f32, const float* w_hwio, const float* bias, void* o0_f16,
                     cgs_stream_t stream);
int cgs_mask_infer_fwd_f16o(int32_t n, int32_t src_a, const void* x, const void* o0_f16, const float* w_m0_hwio, const float* b_m0,
                            const float* w_m2_hwio, const float* b_m2, float* z, cgs_stream_t stream);
/* config 5 (build-defined 128x128 variant, hourglass128.py; chfak 1): features.0 on the same kernel in bfloat16 -- x: uint8 frames
 * (x_is_f32 = 0) or fp32 frames [n,128,128,3] (the replaced / injected mixes, main.py:642-670) -> e0 bf16 [n,64,64,8]; codes (optional):
 * the MaxPool2d argmax bytes [n,64,64,8] of the training forward (cgs_bf16_pool_expand reads them).                                    */
int cgs_bf16_enc0_fwd(int32_t n, const void* x, int32_t x_is_f32, const float* w_hwio, const float* bias, void* e0_bf16, uint8_t* codes,
                      cgs_stream_t stream);
/* The 128x128 layers of config 5's training step at chfak 1 on the whole-strip kernel in bfloat16 (csrc/hconv.hip, h5conv_kernel), with the
 * step's element-wise neighbours fused; w_hwio / bias = the layer's fp32 master parameters [9][ci][co] / [co]:
 *   cgs_bf16_mask0_fwd      hm bf16 [n,128,128,16] = LeakyReLU(conv3x3(cat(frames uint8 / 255, nearest-up2(o0 bf16 [n,64,64,8]))))
 *   cgs_bf16_mask2_fwd      Z fp32 [n,128,128]     = Sigmoid(conv3x3(hm))
 *   cgs_bf16_enc0_bwd_data  d frames fp32 [n,128,128,3] from d features.0 bf16 [n,128,128,8]
 *   cgs_bf16_mask2_bwd_data d (masker.0 pre-activation) bf16 [n,128,128,16] = conv3x3^T(dz fp32 [n,128,128]) x LeakyReLU'(hm)
 *   cgs_bf16_mask0_bwd_data d o0 bf16 [n,64,64,8] = 2x2 cell sums of conv3x3^T(dhm) over the upsampled source's channels             */
int cgs_bf16_mask0_fwd(int32_t n, const uint8_t* x_u8, const void* o0_bf16, const float* w_hwio, const float* bias, void* hm_bf16,
                       cgs_stream_t stream);
int cgs_bf16_mask2_fwd(int32_t n, const void* hm_bf16, const float* w_hwio, const float* bias, float* z, cgs_stream_t stream);
int cgs_bf16_enc0_bwd_data(int32_t n, const void* dy_bf16, const float* w_hwio, float* dx, cgs_stream_t stream);
int cgs_bf16_enc0_bwd_data_pooled(int32_t n, const void* dp_bf16, const void* addend_bf16, const uint8_t* codes, const float* w_hwio, float* dx,
                                  cgs_stream_t stream);      /* the same from dP bf16 [n,64,64,8] (+ addend) and the forward argmax bytes */
/* features.0's data gradient on the two mixes AND cgs_mix_bwd in one pass over n A-images: dp / codes = the pooled gradient bf16 [2n,64,64,8] /
 * argmax bytes of the 2 n mix images [replaced | injected], a_u8 / b_u8 the frames [n,128,128,3], z the mask [n,128,128] ->
 * dzpre [n,128,128] (the gradient at the mask's pre-Sigmoid value, regularisers l1s / l2s as cgs_mix_bwd); no d mix tensor in memory.       */
/* features.0 forward / weight gradient on the VIRTUAL mixes (main.py:395,406): the 2 n images [replaced | injected] are formed from the frame pairs
 * a_u8 / b_u8 [n,128,128,3] and the mask z [n,128,128] while a tile is staged (cgs_mix_fwd's formula); the fp32 mixes are
 * never written: cgs_mix_fwd(.., mixed = NULL, zpart) then leaves only the partial sums of |Z| and Z^2 the loss needs.  Outputs as
 * cgs_bf16_enc0_fwd / cgs_bf16_hwgrad_pooled on 2 n materialised images (slab rows: cgs_bf16_hwgrad_slabs(2 n, 128, 3, 0, 8)).                     */
int cgs_bf16_enc0_fwd_mix(int32_t n, const uint8_t* a_u8, const uint8_t* b_u8, const float* z, const float* w_hwio, const float* bias, void* e0_bf16,
                          uint8_t* codes, cgs_stream_t stream);
int cgs_bf16_hwgrad_pooled_mix(int32_t n, const uint8_t* a_u8, const uint8_t* b_u8, const float* z, const void* dp, const uint8_t* codes, float* slab,
                               cgs_stream_t stream);
int cgs_bf16_enc0_bwd_mix(int32_t n, const void* dp_bf16, const uint8_t* codes, const float* w_hwio, const uint8_t* a_u8, const uint8_t* b_u8,
                          const float* z, float l1s, float l2s, float* dzpre, cgs_stream_t stream);
int cgs_bf16_mask2_bwd_data(int32_t n, const float* dz, const void* hm_bf16, const float* w_hwio, void* dhm_bf16, cgs_stream_t stream);
int cgs_bf16_mask0_bwd_data(int32_t n, const void* dhm_bf16, const float* w_hwio, void* do0_bf16, cgs_stream_t stream);
/* The 64x64 layers of the same step: features.3 forward (src_a = e0 bf16 [n,64,64,8] -> e1 bf16 [n,32,32,8] + argmax bytes) and data gradient
 * (src_a = dy bf16 [n,64,64,8] -> d e0); dec_model.0 forward (cat(src_a = e0, nearest-up2(src_b = o1 bf16 [n,32,32,8])) -> o0 [n,64,64,8]) and
 * its data gradients (src_a = d o0 [n,64,64,8] -> skip gradient [n,64,64,8] / cell-summed low-resolution gradient [n,32,32,8]).  bias NULL
 * for the data gradients.                                                                                                            */
enum { CGS_H5_ENC1_FWD = 1, CGS_H5_ENC1_BWD_DATA = 2, CGS_H5_DEC0_FWD = 3, CGS_H5_DEC0_BWD_SKIP = 4, CGS_H5_DEC0_BWD_LOW = 5,
       CGS_H5_ENC1_BWD_DATA_POOLED = 6, /* src_a = dP bf16 [n,32,32,8], src_b = addend or NULL, codes = the forward argmax bytes (read) */
       /* the 32x32 level: features.6 forward (e1 bf16 [n,32,32,8] -> e2 FP32 [n,16,16,8] + argmax bytes: the tail kernels' input), its data
        * gradient from the pooled gradient (as 6, one level down), dec_model.1 forward (cat(e1, nearest-up2(o2 bf16 [n,16,16,8])) -> o1) and
        * its data gradients (skip: bf16 [n,32,32,8]; low: cell sums, FP32 [n,16,16,8])                                                  */
       CGS_H5_ENC2_FWD = 7, CGS_H5_ENC2_BWD_DATA_POOLED = 8, CGS_H5_DEC1_FWD = 9, CGS_H5_DEC1_BWD_SKIP = 10, CGS_H5_DEC1_BWD_LOW = 11,
       CGS_H5_DEC1_FWD_F32B = 12 /* as 9 with src_b = o2 in FP32 [n,16,16,8] (the tail kernel's output: no bf16 copy) */ };
int cgs_bf16_h5conv(int32_t which, int32_t n, const void* src_a, const void* src_b, const float* w_hwio, const float* bias, void* out,
                    uint8_t* codes, cgs_stream_t stream);
/* Weight + bias gradient of the large-map layers of config 5 at chfak 1 (csrc/hwgrad.hip; same arithmetic as cgs_bf16_conv3x3_bwd_weight):
 * (hw, ca, cb, co) = (128,3,0,8) features.0, (128,3,8,16) masker.0, (128,16,0,1) masker.2, (64,8,0,8) features.3, (64,8,8,8) dec_model.0,
 * (32,8,0,8) features.6, (32,8,8,8) dec_model.1.
 * cgs_bf16_hwgrad_slabs: slab rows written for n images (0: not a dedicated shape).  a_kind: 0 bf16 [n,hw,hw,ca], 1 uint8 / 2 fp32 frames
 * [n,hw,hw,3]; src_b bf16 [n,hw/2,hw/2,cb] (nearest-upsampled x2); dy bf16 [n,hw,hw,co], or fp32 [n,hw,hw] when co == 1.              */
int cgs_bf16_hwgrad_slabs(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co);
int cgs_bf16_hwgrad(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_kind, const void* src_a, const void* src_b,
                    const void* dy, float* slab, cgs_stream_t stream);
/* features.0 (hw 128, ca 3) / features.3 (hw 64, ca 8) / features.6 (hw 32, ca 8) with dY given as the pooled gradient dp bf16 [n,hw/2,hw/2,8] (+ addend or NULL) and the
 * forward pass's argmax bytes (what cgs_bf16_pool_expand would re-expand); slab rows = cgs_bf16_hwgrad_slabs(n, hw, ca, 0, 8).          */
int cgs_bf16_hwgrad_pooled(int32_t n, int32_t hw, int32_t ca, int32_t a_kind, const void* src_a, const void* dp, const void* addend,
                           const uint8_t* codes, float* slab, cgs_stream_t stream);

/* ---- the 16x16-and-smaller layers, image by image inside one workgroup ("tail" kernels, csrc/tail.hip) ------------
 * Replace, for one critic pass / the decoder, the per-layer launches of features.6, features.10 (nets.py:176-183), the
 * critic head features.14 + crit.1 + crit.4 (nets.py:184-194) and dec_model.4/.3/.2/.1 (nets.py:501-513): every
 * intermediate stays in LDS, the convolutions run on the matrix cores, the same tensors as the per-layer entry points are
 * written (so either form feeds the other's backward).  Weight pointers are the kernel-layout (HWIO / k-major) tensors of
 * the flat parameter buffer.  Shapes (NHWC): e1 [n,16,16,8], e2 [n,8,8,8], am2 [n,8,8,1], e3 [n,4,4,16], am3 [n,4,4,2],
 * e4/h1/o4 [n,32], pred [n], o3 [n,4,4,16], o2 [n,8,8,8], o1 [n,16,16,8].                                            */
typedef struct {
    const float *w6, *b6;     /* features.6   [3][3][8][8],  [8]  */
    const float *w10, *b10;   /* features.10  [3][3][8][16], [16] */
    const float *w14, *b14;   /* features.14  [256][32] (k = (y*4+x)*16+c), [32] */
    const float *wl1, *bl1;   /* crit.1       [32][32] k-major, [32] */
    const float *wl2, *bl2;   /* crit.4       [32], [1] */
    const float *wpw, *bpw;   /* dec_model.4  [32][32] k-major, [32]; only read when o4 / d_o4 is given */
} cgs_tail_enc_weights;
typedef struct {
    const float *w3, *b3;     /* dec_model.3  [3][3][48][16], [16] */
    const float *w2, *b2;     /* dec_model.2  [3][3][24][8],  [8]  */
    const float *w1, *b1;     /* dec_model.1  [3][3][16][8],  [8]  */
} cgs_tail_dec_weights;
/* drop_e2 / drop_e3 / drop_h1: Dropout in front of features.10, features.14 and crit.4 (base = first image * 128 / 64 / 8).
 * o4 may be NULL (no decoder follows this pass).                                                                      */
int cgs_tail_enc_fwd(int32_t n, const cgs_tail_enc_weights* w, const float* e1, float* e2, uint32_t* am2, float* e3,
                     uint32_t* am3, float* e4, float* h1, float* pred, float* o4, cgs_dropout drop_e2,
                     cgs_dropout drop_e3, cgs_dropout drop_h1, cgs_stream_t stream);
/* features.3 (cgs_conv3x3_fwd of the 8 -> 8 layer at 32x32: e0 [n,32,32,8] -> e1 [n,16,16,8] + am1, nets.py:173-175) and cgs_tail_enc_fwd in ONE
 * launch, one workgroup per image (round 4): e1 / am1 are written for the backward pass and the decoder as before, the tail stages read
 * the pooled map from the workgroup's LDS tile.  w3 / b3: features.3's HWIO weights and bias.                                          */
int cgs_enc1_tail_fwd(int32_t n, const cgs_tail_enc_weights* w, const float* e0, const float* w3, const float* b3, float* e1,
                      uint32_t* am1, float* e2, uint32_t* am2, float* e3, uint32_t* am3, float* e4, float* h1, float* pred, float* o4,
                      cgs_dropout drop_e2, cgs_dropout drop_e3, cgs_dropout drop_h1, cgs_stream_t stream);
/* The whole critic forward (nets.py:170-194) of an image in one workgroup (round 4): features.0 on uint8 frames x [n,64,64,3] or -- x_is_mix --
 * on the virtual mixes (x = the cgs_mix_src descriptor, as cgs_conv3x3_fwd with CGS_SRC_MIX) -> e0 [n,32,32,8] + am0, then cgs_enc1_tail_fwd.           */
int cgs_critic_fwd_fused(int32_t n, const cgs_tail_enc_weights* w, const void* x, int32_t x_is_mix, const float* w0, const float* b0,
                         float* e0, uint32_t* am0, const float* w3, const float* b3, float* e1, uint32_t* am1, float* e2, uint32_t* am2,
                         float* e3, uint32_t* am3, float* e4, float* h1, float* pred, float* o4, cgs_dropout drop_e2, cgs_dropout drop_e3,
                         cgs_dropout drop_h1, cgs_stream_t stream);
int cgs_tail_dec_fwd(int32_t n, const cgs_tail_dec_weights* w, const float* e1, const float* e2, const float* e3,
                     const float* o4, float* o3, float* o2, float* o1, cgs_stream_t stream);
/* (round 6, BASELINE config 4: main.py:1130-1151 with fp16 conv kernels) cgs_tail_enc_fwd in eval mode / cgs_tail_dec_fwd with fp16 OPERANDS in
 * their 3x3 layers (features.6 / .10, nets.py:176-183; dec_model.3 / .2 / .1, nets.py:503-513) on v_mfma_f32_16x16x16_f16: fp32 accumulation,
 * fp32 tensors in and out, the Linear layers and the head in fp32.  Used by the fused fp16 inference path only (engine.infer(fp16=True)).   */
int cgs_tail_enc_fwd_h16(int32_t n, const cgs_tail_enc_weights* w, const float* e1, float* e2, uint32_t* am2, float* e3,
                         uint32_t* am3, float* e4, float* h1, float* pred, float* o4, cgs_stream_t stream);
int cgs_tail_dec_fwd_h16(int32_t n, const cgs_tail_dec_weights* w, const float* e1, const float* e2, const float* e3,
                         const float* o4, float* o3, float* o2, float* o1, cgs_stream_t stream);
/* (round 6) Both of them in ONE launch, one workgroup per image, storing only what the -process path consumes (nets.py:176-194 + 501-513 in eval
 * mode; csrc/tail_infer.hip): e1 [n,16,16,8] -> pred [n] and o1 [n,16,16,8] (both fp32); the tiles between the layers hold halves and never leave
 * the workgroup.  wd = NULL and o1 = NULL: the critic alone.                                                                                  */
int cgs_tail_infer_h16(int32_t n, const cgs_tail_enc_weights* we, const cgs_tail_dec_weights* wd, const float* e1, float* pred, float* o1,
                       cgs_stream_t stream);
/* ... and features.3 in front of it (cgs_f16_enc1_fwd + cgs_tail_infer_h16 in one launch; nets.py:173-175 in eval mode): e0 fp16 [n,32,32,8] as
 * cgs_f16_enc0_fwd writes it, w3 / b3 = features.3's HWIO weights and bias; e1 is never stored.                                              */
int cgs_f16_enc1_tail_infer(int32_t n, const void* e0_f16, const float* w3, const float* b3, const cgs_tail_enc_weights* we,
                            const cgs_tail_dec_weights* wd, float* pred, float* o1, cgs_stream_t stream);
/* cgs_tail_dec_fwd + (m0_pack != NULL) one extra workgroup that packs masker.0's HWIO weights w_m0 [9][11][16] into the mask head
 * forward's weight registers m0_pack [40 * 64] (cgs_mask_train_fwd_packed).                                                     */
int cgs_tail_dec_fwd_pack(int32_t n, const cgs_tail_dec_weights* w, const float* e1, const float* e2, const float* e3,
                          const float* o4, float* o3, float* o2, float* o1, const float* w_m0, float* m0_pack, cgs_stream_t stream);

/* Backward of the same layers.  cgs_tail_enc_bwd: one critic pass, from dpred [n] to de1 [n,16,16,8] (gradient w.r.t. e1,
 * input of features.3's backward).  dE1 / dE2 / dE3 (skip gradients from the decoder, shapes of e1 / e2 / e3) and d_o4 [n,32]
 * (gradient w.r.t. dec_model.4's output: its backward then runs here too) may be NULL; they are read for images < n_add.
 * Every workgroup writes one slab per convolution: cgs_tail_enc_bwd_slabs(n) slabs of slab10 [1152 + 16] (features.10) and
 * slab6 [576 + 8] (features.6); a NULL slab pointer skips that store (data gradient only).  The head's weight gradients are
 * sums of outer products over the batch: the kernel leaves the per-image vectors in hvec [n,384] (may be NULL:
 * [0,256) dropout(e3) | [256,288) dz4 | [288,320) dh1 | [320,352) dz2 h1 mask | [352] dz2) and cgs_tail_head_wgrad forms them
 * for up to two image ranges (the critic passes of a step; d_o4_k [n_o4_k,32] = that range's decoder gradient or NULL) as one
 * GEMM over the images: cgs_tail_head_wgrad_slabs(n0 + n1) slabs of
 *   slab_head [8192 + 32 + 1024 + 32 + 32 + 1]  (features.14 w,b | crit.1 w,b | crit.4 w,b -- contiguous in the flat buffer)
 *   slab_pw [1024 + 32] (dec_model.4; required when a d_o4 is given).
 * The loss gradient at pred is either given (dpred [n]) or, with dpred = NULL, derived in the kernel from target [n]:
 *   loss_scale * 2 (pred - target)   (MSE terms of main.py:380-411: loss_scale = weight / n), or with bce != 0
 *   loss_scale * (pred - target) / (pred (1 - pred))  (binary cross-entropy, --threshrew);  target = NULL: zero.
 * cgs_tail_dec_bwd: dec_model.1/.2/.3 from do1 [n,16,16,8]: skip gradients dE1/dE2/dE3, d_o4 [n,32], and
 * cgs_tail_dec_bwd_slabs(n) slabs of slab1 [1152 + 8], slab2 [1728 + 8], slab3 [6912 + 16].                          */
int cgs_tail_enc_bwd_slabs(int32_t n);
int cgs_tail_enc_bwd(int32_t n, const cgs_tail_enc_weights* w, const float* e1, const float* e2, const uint32_t* am2,
                     const float* e3, const uint32_t* am3, const float* e4, const float* h1, const float* pred,
                     const float* dpred, const float* target, float loss_scale, int32_t bce, const float* dE1, const float* dE2, const float* dE3, const float* d_o4,
                     int32_t n_add, float* de1, float* hvec, float* slab10, float* slab6,
                     cgs_dropout drop_e2, cgs_dropout drop_e3, cgs_dropout drop_h1, cgs_stream_t stream);
/* The same with dec_model.0's weight gradient (cgs_conv3x3_bwd_weight of that layer over n_r images: skip input e0_r [n_r,32,32,8],
 * low-resolution input o1_r [n_r,16,16,8], output gradient dy_r [n_r,32,32,8], slab_r [nslab_r][1160], nslab_r <= 4 n_r) computed by
 * nslab_r spare workgroups of the launch (slab_r = NULL: none).                                                                    */
int cgs_tail_enc_bwd_rider(int32_t n, const cgs_tail_enc_weights* w, const float* e1, const float* e2, const uint32_t* am2,
                           const float* e3, const uint32_t* am3, const float* e4, const float* h1, const float* pred,
                           const float* dpred, const float* target, float loss_scale, int32_t bce, const float* dE1, const float* dE2,
                           const float* dE3, const float* d_o4, int32_t n_add, float* de1, float* hvec, float* slab10, float* slab6,
                           cgs_dropout drop_e2, cgs_dropout drop_e3, cgs_dropout drop_h1,
                           int32_t n_r, const float* e0_r, const float* o1_r, const float* dy_r, float* slab_r, int32_t nslab_r,
                           cgs_stream_t stream);
/* cgs_tail_enc_bwd_rider AND the data-gradient half of features.3's backward (cgs_conv3x3_bwd_both of the 8 -> 8 layer at 32x32 with ReLU +
 * pool, nets.py:173-175 backward: d e1 re-expanded by the argmax nibbles am1 [n,16,16,1], weights w_enc1 (HWIO [3][3][8][8]), + the decoder's
 * skip gradient addend0 [n_addend,32,32,8] (NULL: none) -> de0 [n,32,32,8]) in ONE launch (round 5): every workgroup continues with the
 * convolution of the image(s) whose tail it just ran, so d e1 never crosses a launch boundary.  That layer's WEIGHT gradient: slab1 != NULL
 * (with e0 [n,32,32,8], the layer's input): formed here too, per workgroup over its own images, slab1 [cgs_tail_enc_bwd_slabs(n)][584] (the
 * partition of the sum over the images differs from cgs_conv3x3_bwd_weight's: equal up to fp32 summation order); slab1 = NULL: left to
 * extra workgroups of the features.0 backward launch that follows (cgs_enc0_bwd_mix_enc1 / cgs_enc0_wgrad_u8_with_head_enc1), which
 * reproduce cgs_conv3x3_bwd_both's slabs bit for bit.  de0 is bit-identical to the two-launch form either way.
 * slab0 != NULL: features.0's weight gradient of the same images as well (cgs_conv3x3_bwd_weight of the 3 -> 8 layer at 64x64 with ReLU + pool,
 * nets.py:170-172 backward), from de0, the argmax nibbles am0 [n,32,32,1] and the layer's input x: x_kind = CGS_SRC_U8 (x = uint8 frames
 * [n,64,64,3]) or CGS_SRC_MIX (x = the cgs_mix_src descriptor, n = n_a or 2 n_a); slab0 [cgs_tail_enc_bwd_slabs(n)][224], one row per
 * workgroup over its own images (equal to the stand-alone launch up to fp32 summation order).                                        */
int cgs_tail_enc_bwd_enc1(int32_t n, const cgs_tail_enc_weights* w, const float* e1, const float* e2, const uint32_t* am2,
                          const float* e3, const uint32_t* am3, const float* e4, const float* h1, const float* pred,
                          const float* dpred, const float* target, float loss_scale, int32_t bce, const float* dE1, const float* dE2,
                          const float* dE3, const float* d_o4, int32_t n_add, float* de1, float* hvec, float* slab10, float* slab6,
                          cgs_dropout drop_e2, cgs_dropout drop_e3, cgs_dropout drop_h1,
                          int32_t n_r, const float* e0_r, const float* o1_r, const float* dy_r, float* slab_r, int32_t nslab_r,
                          const uint32_t* am1, const float* w_enc1, const float* addend0, int32_t n_addend, float* de0,
                          const float* e0, float* slab1,
                          int32_t x_kind, const void* x, const uint32_t* am0, float* slab0, cgs_stream_t stream);
int cgs_tail_head_wgrad_slabs(int32_t n_total);
int cgs_tail_head_wgrad(int32_t n0, const float* hvec0, const float* e4_0, const float* d_o4_0, int32_t n_o4_0, int32_t n1,
                        const float* hvec1, const float* e4_1, const float* d_o4_1, int32_t n_o4_1, float* slab_head,
                        float* slab_pw, cgs_stream_t stream);
/* cgs_conv3x3_bwd_weight of features.0 on n uint8 frames (slab: cgs_conv3x3_bwd_weight_slabs rows of [224]) and cgs_tail_head_wgrad
 * (same arguments) as ONE launch: the head's small GEMM rides along as extra workgroups.                                        */
int cgs_enc0_wgrad_u8_with_head(int32_t n, const uint8_t* x_u8, const float* dy, const uint32_t* amask, float* slab,
                                int32_t n0, const float* hvec0, const float* e4_0, const float* d_o4_0, int32_t n_o4_0,
                                int32_t n1, const float* hvec1, const float* e4_1, const float* d_o4_1, int32_t n_o4_1,
                                float* slab_head, float* slab_pw, cgs_stream_t stream);
/* The same + features.3's weight gradient over n1w images (cgs_conv3x3_bwd_weight of the 8 -> 8 layer at 32x32 with ReLU + pool: input e0_1
 * [n1w,32,32,8], pooled output gradient dy1 [n1w,16,16,8], argmax nibbles am1, slab1 [nslab1][584], nslab1 = cgs_enc1_wgrad_rider_slabs(n1w))
 * as nslab1 more workgroups (slab1 = NULL: none) -- the weight-gradient half that cgs_tail_enc_bwd_enc1 leaves behind (round 5).        */
int cgs_enc1_wgrad_rider_slabs(int32_t n);
int cgs_enc0_wgrad_u8_with_head_enc1(int32_t n, const uint8_t* x_u8, const float* dy, const uint32_t* amask, float* slab,
                                     int32_t n0, const float* hvec0, const float* e4_0, const float* d_o4_0, int32_t n_o4_0,
                                     int32_t n1, const float* hvec1, const float* e4_1, const float* d_o4_1, int32_t n_o4_1,
                                     float* slab_head, float* slab_pw,
                                     int32_t n1w, const float* e0_1, const float* dy1, const uint32_t* am1, float* slab1, int32_t nslab1,
                                     cgs_stream_t stream);
/* ... + dec_model.3's weight gradient (cgs_conv3x3_bwd_weight of the 48 -> 16 layer at 4x4 over cat(e3, Upsample(x4)(o4)), nets.py:483,503-505) over
 * n3 images as a GEMM over the images in the launch's last workgroups (round 5): e3 [n3,4,4,16], o4 [n3,32], do3 [n3,4,4,16] = the gradient at that
 * layer's output as cgs_dec0_tail_dec_bwd_do3 leaves it; slab3 [cgs_dec3_wgrad_rider_slabs(n3)][6928]; slab3 = NULL: none.                        */
int cgs_dec3_wgrad_rider_slabs(int32_t n);
int cgs_enc0_wgrad_u8_with_head_riders(int32_t n, const uint8_t* x_u8, const float* dy, const uint32_t* amask, float* slab,
                                       int32_t n0, const float* hvec0, const float* e4_0, const float* d_o4_0, int32_t n_o4_0,
                                       int32_t n1, const float* hvec1, const float* e4_1, const float* d_o4_1, int32_t n_o4_1,
                                       float* slab_head, float* slab_pw,
                                       int32_t n1w, const float* e0_1, const float* dy1, const uint32_t* am1, float* slab1, int32_t nslab1,
                                       int32_t n3, const float* e3, const float* o4, const float* do3, float* slab3, cgs_stream_t stream);
int cgs_tail_dec_bwd_slabs(int32_t n);
int cgs_tail_dec_bwd(int32_t n, const cgs_tail_dec_weights* w, const float* e1, const float* e2, const float* e3,
                     const float* o4, const float* o3, const float* o2, const float* do1, float* dE1, float* dE2,
                     float* dE3, float* d_o4, float* slab3, float* slab2, float* slab1, cgs_stream_t stream);
/* dec_model.0's data gradient (cgs_conv3x3_bwd_data of the 16 -> 8 layer at 32x32: dy_o0 [n,32,32,8] -> dE0 [n,32,32,8] = the skip
 * gradient at e0, do1 [n,16,16,8] = the cell-summed gradient at o1) and cgs_tail_dec_bwd (which consumes that do1) in ONE launch, one
 * workgroup per image (round 4).  n <= cgs_tail_dec_bwd_slabs(n) (one slab row per image), else CGS_ERR_UNSUPPORTED.               */
int cgs_dec0_tail_dec_bwd(int32_t n, const cgs_tail_dec_weights* w, const float* dy_o0, const float* w0_hwio, float* dE0, const float* e1,
                          const float* e2, const float* e3, const float* o4, const float* o3, const float* o2, float* do1, float* dE1,
                          float* dE2, float* dE3, float* d_o4, float* slab3, float* slab2, float* slab1, cgs_stream_t stream);
/* The same; do3 != NULL (then slab3 = NULL): dec_model.3's weight gradient is NOT formed here -- d o3 [n,4,4,16] is written for the rider workgroups
 * of cgs_enc0_wgrad_u8_with_head_riders (64 slab rows for the layer instead of one per image; equal up to fp32 summation order; round 5).         */
int cgs_dec0_tail_dec_bwd_do3(int32_t n, const cgs_tail_dec_weights* w, const float* dy_o0, const float* w0, float* dE0, const float* e1,
                              const float* e2, const float* e3, const float* o4, const float* o3, const float* o2, float* do1,
                              float* dE1, float* dE2, float* dE3, float* d_o4, float* slab3, float* slab2, float* slab1, float* do3,
                              cgs_stream_t stream);
/* cgs_tail_dec_fwd_pack and dec_model.0's forward (cgs_conv3x3_fwd of the 16 -> 8 layer at 32x32: cat(e0, nearest-up(o1)) -> o0, linear;
 * nets.py:516-517) in ONE launch, one workgroup per image (round 4; n <= 1024, else CGS_ERR_UNSUPPORTED).                              */
int cgs_tail_dec_fwd_dec0(int32_t n, const cgs_tail_dec_weights* w, const float* e0, const float* e1, const float* e2, const float* e3,
                          const float* o4, float* o3, float* o2, float* o1, const float* w0_hwio, const float* b0, float* o0,
                          const float* w_m0, float* m0_pack, cgs_stream_t stream);

/* ---- convolution backward, weights --------------------------------------------------
 * Replaces convolution_backward(weight, bias).  Each workgroup writes one partial "slab"
 *   slab[b][0 .. 9*(ca+cb)*co)  = partial dW (HWIO),  slab[b][9*(ca+cb)*co ..][co] = partial dbias
 * and cgs_reduce_slabs() sums the slabs (deterministic, no float atomics).
 * cgs_conv3x3_bwd_weight_slabs() returns the number of slabs the launch will write (>0) or an error. */
int cgs_conv3x3_bwd_weight_slabs(const cgs_conv_desc* d);
int cgs_conv3x3_bwd_weight(const cgs_conv_desc* d, const void* src_a, const float* src_b,
                           const float* dy, const uint32_t* amask, float* slab,
                           cgs_stream_t stream);

/* ---- both backward halves of a SMALL layer in one launch ------------------------------
 * Same results as cgs_conv3x3_bwd_weight + cgs_conv3x3_bwd_data (identical device code), but the wgrad
 * workgroups and the data-gradient workgroups share one grid, so the two latency-bound halves overlap.
 * Supported: features.3/6/10, features.0 on fp32 input, dec_model.0-3, masker.0 (CGS_ERR_UNSUPPORTED otherwise).
 * d_a / d_b as for cgs_conv3x3_bwd_data (either may be NULL); slab as for cgs_conv3x3_bwd_weight with _both_slabs() slabs. */
int cgs_conv3x3_bwd_both_slabs(const cgs_conv_desc* d);
int cgs_conv3x3_bwd_both(const cgs_conv_desc* d, const void* src_a, const float* src_b, const float* dy,
                         const uint32_t* amask, const float* w_hwio, const float* addend, int32_t n_addend,
                         float* d_a, float* d_b, float* slab, cgs_stream_t stream);

/* One reduction job: dst[i] (+)= sum_{b<nslab} slab[b*stride + i], i < count. */
typedef struct {
    const float* slab;
    float* dst;
    int32_t nslab, stride, count;
    int32_t accumulate; /* 0: overwrite dst, 1: add to dst */
} cgs_reduce_job;
/* jobs: DEVICE array of njobs entries.  If step != NULL, *step is incremented by one (by one thread)
 * -- this is the per-iteration tick that Adam and the dropout masks read. */
int cgs_reduce_slabs(const cgs_reduce_job* jobs, int32_t njobs, int32_t max_count,
                     uint64_t* step, cgs_stream_t stream);

/* Single-GPU tail of a step in one launch: cgs_reduce_slabs, the Adam update (torch.optim.Adam defaults semantics as
 * cgs_adam_flat, grad_scale 1) of every element a job writes -- param / m / v are indexed by (job.dst - grad_base) -- and, when
 * n > 0, the loss values of cgs_phase2_losses (losses[8]; no dpred: the tail backward kernels derive it themselves).  *step is
 * read by every workgroup and advanced by one by the LAST workgroup to finish (ticket: device uint32 [njobs + 2], zero before
 * the first call; the kernel leaves it zero).  param == NULL: reduction, loss values and step tick only -- the data-parallel form:
 * the all-reduce of the gradient follows, then cgs_adam_flat (which reads the advanced *step).                                       */
int cgs_reduce_adam(const cgs_reduce_job* jobs, int32_t njobs, int32_t max_count, uint64_t* step, float* param,
                    const float* grad_base, float* m, float* v, float lr, float beta1, float beta2, float eps,
                    uint32_t* ticket, int32_t n, const float* pred, const float* y, const float* zpart, int32_t nzpart,
                    float lfak, float l1, float l2, int32_t flags, int64_t nz, float* losses, cgs_stream_t stream);

/* ---- features.0 backward of the replaced / injected passes + mix backward in one launch ----------------------
 * (main.py:395,406 backward chained onto convolution_backward of features.0).  n_a A-images; the mixes are images
 * [0,n_a) = replaced, [n_a,2 n_a) = injected (inject != 0).  dy [n_mix,32,32,8] + amask: gradient at features.0's pooled
 * output; slab [cgs_enc0_bwd_mix_slabs(n_mix)][224]: weight-gradient partials (NULL: data path only, frozen critic);
 * mixed [n_mix,64,64,3]: the weight gradient's input if the mixes were materialised, NULL: they are recomputed from a, b, z
 * in the tile loader (as cgs_conv3x3_fwd with CGS_SRC_MIX does).  a, b: uint8 frames [n_a,64,64,3]; z [n_a,64,64]; l1/l2_scale as cgs_mix_bwd.
 * Writes dzpre [n_a,64,64] = what cgs_conv3x3_bwd_data -> cgs_mix_bwd would (bit-identical); the image gradients are
 * never stored.                                                                                                  */
int cgs_enc0_bwd_mix_slabs(int32_t n_mix);
int cgs_enc0_bwd_mix(int32_t n_a, int32_t inject, const float* mixed, const float* dy, const uint32_t* amask,
                     const float* w_hwio, const uint8_t* a, const uint8_t* b, const float* z, float l1_scale,
                     float l2_scale, const float* valuefak_pred, float* dzpre, float* slab, cgs_stream_t stream);
/* The same + features.3's weight gradient over the n_mix mixes (e0_1 [n_mix,32,32,8], dy1 [n_mix,16,16,8], am1, slab1
 * [cgs_enc1_wgrad_rider_slabs(n_mix)][584]; all NULL: none) as extra workgroups of the launch (round 5, see cgs_tail_enc_bwd_enc1). */
int cgs_enc0_bwd_mix_enc1(int32_t n_a, int32_t inject, const float* mixed, const float* dy, const uint32_t* amask,
                          const float* w_hwio, const uint8_t* a, const uint8_t* b, const float* z, float l1_scale,
                          float l2_scale, const float* valuefak_pred, float* dzpre, float* slab,
                          const float* e0_1, const float* dy1, const uint32_t* am1, float* slab1, cgs_stream_t stream);
/* valuefak_pred (may be NULL): -staticnorm '' of main.py:415-418 -- the regulariser terms of A-image i are weighted by
 * (1 - valuefak_pred[i]) (L1) and its square (L2); with cgs_phase2_losses / cgs_reduce_adam the same weighting of the loss
 * VALUES is selected by flag bit 8 (weights from pred[n + i], zpart holding nzpart / n partial pairs per image).        */

/* ---- critic head: 4x4 valid conv + Linear + Linear (nets.py:184-194) -----------------
 *   e3     : [n,4,4,16] pooled embed (dropout `drop_in` fused into the load)
 *   w4     : [256][32]  (k = (y*4+x)*16 + c, HWIO of features.14), b4 [32]
 *   w1     : [32][32]   (k-major: crit.1.weight transposed), b1 [32];  w2 [32], b2 [1]
 *   e4     : [n,32] post-ReLU bottleneck (embeds[4]);  h1 : [n,32] post-ReLU hidden (pre-dropout)
 *   pred   : [n] sigmoid output.
 *   o4     : optional [n,32] = the DECODER's bottleneck 1x1 conv of e4 (nets.py:501, weights w_pw [32][32] k-major and
 *            b_pw [32] of dec_model.4) computed from the e4 row while it is on chip; NULL: not computed
 *            (cgs_pointwise_fwd is the stand-alone form).                                     */
int cgs_head_fwd(int32_t n, const float* e3, const float* w4, const float* b4, const float* w1,
                 const float* b1, const float* w2, const float* b2, cgs_dropout drop_in,
                 cgs_dropout drop_h, float* e4, float* h1, float* pred, const float* w_pw,
                 const float* b_pw, float* o4, cgs_stream_t stream);
/* Backward of the head.  dpred [n]; d_e4_extra / d_e3_extra: optional [n_extra,32] / [n_extra,4,4,16]
 * gradients arriving at e4 / e3 from the decoder (images < n_extra; d_e3_extra may alias d_e3).
 * Writes d_e3 [n,4,4,16] = head gradient (dropout mask applied) + d_e3_extra, and one slab per
 * workgroup: [w4 8192 | b4 32 | w1 1024 | b1 32 | w2 32 | b2 1] = 9313 floats.
 * d_o4 (optional, [n_extra,32]): gradient w.r.t. the decoder bottleneck output o4 instead of (or besides)
 * d_e4_extra: the 1x1 conv's backward runs here too (d e4 += W_pw d_o4) and its weight/bias gradient goes to
 * slab_pw [cgs_head_bwd_slabs(n)][1056] (layout of cgs_pointwise_bwd's slab).                 */
int cgs_head_bwd_slabs(int32_t n);
int cgs_head_bwd(int32_t n, const float* e3, const float* e4, const float* h1, const float* pred,
                 const float* dpred, const float* d_e4_extra, const float* d_e3_extra, int32_t n_extra,
                 const float* w4,
                 const float* w1, const float* w2, cgs_dropout drop_in, cgs_dropout drop_h,
                 float* d_e3, float* slab, const float* d_o4, const float* w_pw, float* slab_pw,
                 cgs_stream_t stream);

/* ---- decoder 1x1 conv on the bottleneck (nets.py:484,501) --------------------------
 *   y[n][o] = sum_k x[n][k] * w[k][o] + b[o],  32 -> 32 (MFMA v_mfma_f32_32x32x2_f32).      */
int cgs_pointwise_fwd(int32_t n, int32_t ci, int32_t co, const float* x, const float* w,
                      const float* b, float* y, cgs_stream_t stream);
int cgs_pointwise_bwd_slabs(int32_t n);
/* dx[n][k] = sum_o dy[n][o] w[k][o];  slab[b] = [dW ci*co | db co] partials. */
int cgs_pointwise_bwd(int32_t n, int32_t ci, int32_t co, const float* x, const float* dy,
                      const float* w, float* dx, float* slab, cgs_stream_t stream);

/* ---- mask replace / inject mix (main.py:395,406) --------------------------------------
 *   rep = A(1-Z) + Z B,  inj = B(1-Z) + Z A  with A,B uint8 NHWC (/255 fused), Z [n,64,64].
 *   mixed : [2n,h,w,3] fp32 (rep images then inj images; inj skipped when inject == 0)
 *   zpart : device float[2 * cgs_mix_fwd_partials(n,hw)]: per-workgroup partial (sum |Z|, sum Z^2),
 *           summed by cgs_phase2_losses (fixed order: no float atomics).                    */
int cgs_mix_fwd_partials(int32_t n, int32_t hw);
int cgs_mix_fwd(int32_t n, int32_t hw, const uint8_t* a, const uint8_t* b, const float* z,
                int32_t inject, float* mixed, float* zpart, cgs_stream_t stream);
/* dzpre = (sum_c (B-A)_c (dRep_c - dInj_c) + l1*sign(Z) + 2*l2*Z) * Z(1-Z): the gradient w.r.t. the
 * mask head's pre-sigmoid output, regularisers (main.py:421-429) included. */
int cgs_mix_bwd(int32_t n, int32_t hw, const uint8_t* a, const uint8_t* b, const float* z,
                const float* dmixed, int32_t inject, float l1_scale, float l2_scale,
                float* dzpre, cgs_stream_t stream);
/* The same with -staticnorm '' (main.py:415-418): the regulariser terms of A-image i are weighted by valuefak = 1 - valuefak_pred[i]
 * (L1) and its square (L2); valuefak_pred [n] (the detached critic values of A) may be NULL (= cgs_mix_bwd).  hw = pixels per image. */
int cgs_mix_bwd_weighted(int32_t n, int32_t hw, const uint8_t* a, const uint8_t* b, const float* z, const float* dmixed,
                         int32_t inject, float l1_scale, float l2_scale, const float* valuefak_pred, float* dzpre,
                         cgs_stream_t stream);

/* ---- losses (main.py:380-384,400,411,421-429 and main.py:192-195) --------------------
 * pred layout [4n]: slots [B | A | replaced | injected].  y [n].  zpart/nzpart from cgs_mix_fwd.
 * losses[8] = {critic, replace, inject, l1, l2, total, 0, 0};  dpred [4n] = d total / d pred.
 * flags: bit0 live, bit1 inject, bit2 bce (--threshrew).  nz = n*h*w (mask elements).        */
int cgs_phase2_losses(int32_t n, const float* pred, const float* y, const float* zpart,
                      int32_t nzpart, float lfak, float l1, float l2, int32_t flags, int64_t nz,
                      float* losses, float* dpred, cgs_stream_t stream);
/* phase 1: loss = mse(pred, y) or bce; losses[0] = loss; dpred [n]. */
int cgs_phase1_loss(int32_t n, const float* pred, const float* y, int32_t bce, float* losses,
                    float* dpred, cgs_stream_t stream);

/* ---- Adam, flat (torch.optim.Adam defaults semantics; main.py:178,330-334,463) --------
 * t = *step (already ticked by cgs_reduce_slabs).  grad_scale multiplies the gradient as it is read
 * (1/world_size after a sum all-reduce; 1 otherwise). */
int cgs_adam_flat(int64_t count, float* param, const float* grad, float* m, float* v,
                  const uint64_t* step, float lr, float beta1, float beta2, float eps,
                  float grad_scale, cgs_stream_t stream);

/* ---- layout helpers (module API boundary: the reference hands NCHW fp32 tensors) ------ */
int cgs_nchw_to_nhwc(int32_t n, int32_t c, int32_t hw, const float* src, float* dst, cgs_stream_t stream);
int cgs_nhwc_to_nchw(int32_t n, int32_t c, int32_t hw, const float* src, float* dst, cgs_stream_t stream);
/* Writes the keep-mask multipliers (0 or 1/(1-p)) a dropout descriptor generates for `count` floats
 * (count % 4 == 0): test hook used to feed the SAME masks to the CPU oracle. */
int cgs_dropout_mask(cgs_dropout d, int64_t count, float* out, cgs_stream_t stream);

/* Library / device info: returns the gfx arch string the library was built for. */
/* ---- shape-generic kernels (csrc/gen.hip): model sizes outside the specialised set ---------------------------------
 * NewCritic / UnetDecoder with chfak != 1 (nets.py:166,184,190; the paper's model is chfak = 5) and the legacy `Unet` with its
 * ConvTranspose2d(4,2,1) decoder and LeakyReLU(0.2) (nets.py:356-449).  NHWC fp32 (source A optionally uint8, /255 fused),
 * runtime channel counts, weights in kernel layout (HWIO; [k][n] for GEMMs; [ky][kx][ci][co] for the transposed conv).
 * cgs_gen_conv_pack_weights: HWIO w [9][ca+cb][co] -> the convolution's weight operand wp (cgs_gen_conv_packed_floats floats:
 *   register images [16-channel chunk][tap][4-channel output group][64 lanes], zero for padding, csrc/gen4.hip).  transposed = 1
 *   (cb = 0): w is the HWIO weight [9][co][ca] of the LAYER whose data gradient is wanted (ca = its output channels = dY's, co =
 *   its input channels): wp is then the operand of cgs_gen_conv3x3_bwd_data (taps reversed, channels transposed).
 * cgs_gen_conv3x3_fwd: out = act(conv3x3(cat(A [ca], nearest-up_ups(B [cb])), wp) + bias), hw in {4,8,16,32,64}; pool = 1:
 *   MaxPool2d(2) of it, out [n,hw/2,hw/2,co] and argmax [same] = position 0..3 of the first maximum in bits 0-1 (may be NULL).
 *   transposed = 2 + cgs_gen_conv3x3_fwd_folded: the same layer over cat(A, nearest-up_2(B)) (cb > 0, hw >= 16, no pooling) with B read at
 *   its own resolution: a pixel of parity (py, px) sees B through the 2 x 2 cells around it, its nine taps over the upsampled map sum to four
 *   (weights added per parity by the pack, cgs_gen_conv_packed_floats_folded floats) -- 4 / 9 of the matrix work for B's channels.
 * cgs_gen_gemm: out [m,n] = act(x [m,k] w [k,n] + bias [n] (may be NULL)).
 * cgs_gen_convt4s2_*: ConvTranspose2d(4,2,1) over cat(A, B) [n,h,h,*] -> [n,2h,2h,co]: forward (+bias, act), data gradient
 *   (dy = gradient at the PRE-activation output; da / db may be NULL), weight + bias gradient (dw [16*(ca+cb)*co], dbias [co]). */
int64_t cgs_gen_conv_packed_floats(int32_t ca, int32_t cb, int32_t co);
int64_t cgs_gen_conv_packed_floats_folded(int32_t ca, int32_t cb, int32_t co);
int cgs_gen_conv_pack_weights(int32_t ca, int32_t cb, int32_t co, int32_t transposed, const float* w, float* wp,
                              cgs_stream_t stream);
/* The data gradient's operand for a window of the layer's input channels: w = HWIO [9][ci_layer][co_layer]; wp
 * (cgs_gen_conv_packed_floats(co_layer, 0, ci_n) floats) maps dY to d(input channels [ci_off, ci_off + ci_n)).                   */
int cgs_gen_conv_pack_weights_window(int32_t co_layer, int32_t ci_layer, int32_t ci_off, int32_t ci_n, const float* w, float* wp,
                                     cgs_stream_t stream);

/* All 3x3 layers' operands of one training step in ONE launch (the step packs ~25 of them: weights change at every Adam step).
 * Job i is cgs_gen_conv_pack_weights(ca, cb, co, transposed, w, wp) when ci_layer == 0 and
 * cgs_gen_conv_pack_weights_window(ca, ci_layer, ci_off, co, w, wp) when ci_layer > 0 (transposed must be 1).  jobs: HOST array. */
typedef struct cgs_gen_pack_job {
    const float* w; float* wp;
    int32_t ca, cb, co, transposed, ci_layer, ci_off;
} cgs_gen_pack_job;
int cgs_gen_conv_pack_batch(const cgs_gen_pack_job* jobs, int32_t njobs, cgs_stream_t stream);
int cgs_gen_conv3x3_fwd(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_is_u8, int32_t ups,
                        int32_t act, float slope, int32_t pool, const void* src_a, const float* src_b, const float* wp,
                        const float* bias, float* out, uint8_t* argmax, cgs_stream_t stream);
int cgs_gen_conv3x3_fwd_folded(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_is_u8, int32_t act, float slope,
                               const void* src_a, const float* src_b, const float* wp, const float* bias, float* out, cgs_stream_t stream);
/* features.0 of NewCritic at chfak 2 / 3 / 4 / 5 (co = 16 / 24 / 32 / 40; nets.py:170-172) on kernels of their own (csrc/gen_enc0.hip): same
 * tensors as cgs_gen_conv3x3_fwd(hw 64, ca 3, ReLU, pool) -- out [n,32,32,co] + argmax bytes (am may be NULL) -- from x = uint8 (x_is_u8) or
 * fp32 frames [n,64,64,3] and the layer's HWIO weights [9][3][co] (not packed).  CGS_ERR_UNSUPPORTED for other channel counts.          */
int cgs_gen_enc0_fwd(int32_t n, int32_t co, int32_t x_is_u8, const void* x, const float* w_hwio, const float* bias, float* out, uint8_t* am,
                     cgs_stream_t stream);
/* ... and the layer's weight + bias gradient: slab rows [cgs_gen_enc0_bwd_weight_slabs(n, co)][27 co + co] (-> cgs_reduce_slabs) from the frames,
 * the pooled gradient de [n,32,32,co] and the forward pass's argmax bytes (the tensors of cgs_gen_conv3x3_bwd_weight(hw 64, ca 3, cb 0)).   */
int cgs_gen_enc0_bwd_weight_slabs(int32_t n, int32_t co);
int cgs_gen_enc0_bwd_weight(int32_t n, int32_t co, int32_t x_is_u8, const void* x, const float* de, const uint8_t* am, float* slab,
                            cgs_stream_t stream);
/* ... and its data gradient (the image gradient of the mixes): d x [n,64,64,3] from de, am and the layer's raw HWIO weights (the tensors of
 * cgs_gen_conv3x3_bwd_data(hw 64, co, ci 3, dy_argmax)).                                                                                 */
int cgs_gen_enc0_bwd_data(int32_t n, int32_t co, const float* de, const uint8_t* am, const float* w_hwio, float* dx, cgs_stream_t stream);
int cgs_gen_gemm(int32_t m, int32_t k, int32_t n, int32_t act, float slope, const float* x, const float* w,
                 const float* bias, float* out, cgs_stream_t stream);

/* Training pass at chfak != 1 (csrc/gen_train.hip; backward of nets.py:197-212, 494-523 for any channel count).
 * cgs_gen_conv3x3_fwd's argmax byte carries bit 2 (value >= 4) where a ReLU'd pooled value is <= 0: no gradient there.
 * cgs_gen_flip_weights: HWIO w [9][ci][co] -> wflip [9][co][ci] with the taps reversed (the data gradient's kernel in HWIO form:
 *   cgs_gen_conv_pack_weights(co, 0, ci, 0, wflip) == cgs_gen_conv_pack_weights(co, 0, ci, 1, w)).
 * cgs_gen_conv3x3_bwd_data: d_cat [n,hw,hw,ci] = conv3x3(dY, wp) (+ addend [n_addend,hw,hw,ci] for images < n_addend), wp =
 *   cgs_gen_conv_pack_weights(co, 0, ci, transposed = 1, the layer's HWIO weights).  dY is the
 *   gradient at the pre-activation output [n,hw,hw,co], or -- dy_argmax != NULL, a max-pooled ReLU layer -- the gradient dE
 *   [n,hw/2,hw/2,co] at the pooled output with the forward pass's argmax bytes (re-expanded in the tile loader).
 * cgs_gen_conv3x3_bwd_weight: one slab row [9*(ca+cb)*co | co] (HWIO dW, dbias) per image share, cgs_gen_conv3x3_bwd_weight_slabs
 *   rows, to be summed by cgs_reduce_slabs; the layer input is cat(A, nearest-up_ups(B)) as in the forward call.
 * cgs_gen_cat_split: d_cat [n,hw,hw,ca+cb] -> d_a [n,hw,hw,ca] and d_b [n,hw/ups,hw/ups,cb] (sum over each ups x ups cell); either
 *   output may be NULL.
 * cgs_gen_grad_fix: d[i] = (d[i] * dropout multiplier + (i < addend_count ? addend[i] : 0)) * act'(saved[i]) in place; saved (the
 *   layer's stored OUTPUT; NULL: no activation factor), addend and the dropout (p = 0) are optional.
 * cgs_gen_dropout_fwd: out = x * keep-mask / (1 - p) (count % 4 == 0; the mask stream of cgs_dropout_mask).
 * cgs_gen_gemm_ex: out [m,n] (+)= act(sum_k x[m sxm + k sxk] w[k swk + n swn] + bias[n]) -- Linear layers and their gradients.
 * cgs_gen_u8_to_f32: out = x / 255.                                                                                         */
int cgs_gen_flip_weights(int32_t ci, int32_t co, const float* w, float* wflip, cgs_stream_t stream);
int cgs_gen_conv3x3_bwd_data(int32_t n, int32_t hw, int32_t co, int32_t ci, const float* dy, const uint8_t* dy_argmax,
                             const float* wp, const float* addend, int32_t n_addend, float* d_cat, cgs_stream_t stream);
/* d_cat of a layer over cat(A [ca], nearest-up_ups(B [cb])) written straight as d_a [n,hw,hw,ca] (NULL: not wanted) and d_b
 * [n,hw/ups,hw/ups,cb] (sum over each ups x ups cell): cgs_gen_conv3x3_bwd_data + cgs_gen_cat_split without the d_cat tensor; wp as for
 * cgs_gen_conv3x3_bwd_data with ci = ca + cb (ca = 0, d_a = NULL: the operand of a channel window that lies entirely in B,
 * cgs_gen_conv_pack_weights_window).  CGS_ERR_UNSUPPORTED when ca is not a multiple of the kernel's output pass width.          */
int cgs_gen_conv3x3_bwd_data_split(int32_t n, int32_t hw, int32_t co, int32_t ca, int32_t cb, int32_t ups, const float* dy,
                                   const float* wp, float* d_a, float* d_b, cgs_stream_t stream);
/* The second output of cgs_gen_conv3x3_bwd_data_split for ups = 2 alone, computed at the source's resolution: d_b [n,hw/2,hw/2,ci_n] = the gradient of
 * the x2-upsampled input channels [ci_off, ci_off + ci_n) of a layer (HWIO w [9][ci_layer][co_layer]) summed over each 2 x 2 cell, from dy [n,hw,hw,co_layer]
 * read as its space-to-depth view -- 16 (parity block, cell offset) steps per channel and cell instead of 9 taps per pixel + the cell sum (36).
 * hw 16 / 32 / 64, co_layer % 4 == 0; operand: cgs_gen_conv_pack_weights_up2 (cgs_gen_conv_packed_floats_up2(co_layer, ci_n) floats; in
 * cgs_gen_conv_pack_batch: transposed = 3 with ca = co_layer, co = ci_n, ci_layer, ci_off).                                                     */
int64_t cgs_gen_conv_packed_floats_up2(int32_t co_layer, int32_t ci_n);
int cgs_gen_conv_pack_weights_up2(int32_t co_layer, int32_t ci_layer, int32_t ci_off, int32_t ci_n, const float* w, float* wp, cgs_stream_t stream);
int cgs_gen_conv3x3_bwd_data_up2(int32_t n, int32_t hw, int32_t co_layer, int32_t ci_n, const float* dy, const float* wp, float* d_b,
                                 cgs_stream_t stream);
int cgs_gen_conv3x3_bwd_weight_slabs(int32_t n, int32_t ca, int32_t cb, int32_t co);
int cgs_gen_conv3x3_bwd_weight(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_is_u8, int32_t ups,
                               const void* src_a, const float* src_b, const float* dy, const uint8_t* dy_argmax, float* slab,
                               cgs_stream_t stream);
/* The same slab rows for a layer over cat(A, nearest-up_2(B)) (hw 16 / 32 / 64, cb 16 / 24 / 32 / 40, dY not pooled) with B read at its own
 * resolution: per pixel parity the nine taps over the upsampled map are four folds of B's 2 x 2 cells -- 4 cb instead of 9 cb GEMM rows
 * (csrc/gen_wgrad_fold.h); A's rows and the bias row come from the row-block kernel.  _slabs: rows per layer (<= 0: not supported);
 * CGS_ERR_UNSUPPORTED is returned before anything is launched.                                                                    */
int cgs_gen_conv3x3_bwd_weight_folded_slabs(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co);
int cgs_gen_conv3x3_bwd_weight_folded(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_is_u8, const void* src_a,
                                      const float* src_b, const float* dy, float* slab, cgs_stream_t stream);
int cgs_gen_cat_split(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t ups, const float* d_cat, float* d_a, float* d_b,
                      cgs_stream_t stream);
int cgs_gen_grad_fix(int64_t count, float* d, const float* saved, int32_t act, float slope, const float* addend,
                     int64_t addend_count, cgs_dropout drop, cgs_stream_t stream);
int cgs_gen_dropout_fwd(int64_t count, const float* x, float* out, cgs_dropout drop, cgs_stream_t stream);
int cgs_gen_gemm_ex(int32_t m, int32_t k, int32_t n, const float* x, int64_t sxm, int64_t sxk, const float* w, int64_t swk,
                    int64_t swn, const float* bias, int32_t act, float slope, int32_t accumulate, float* out, cgs_stream_t stream);
/* Split-K form for the long reductions over the batch (X^T dY of the Linear layers, column sums): K share y of nsplit writes its
 * partial product to slab[y][m*n]; the rows are summed in order by cgs_reduce_slabs with the 3x3 layers' slabs
 * (reference: the autograd of nn.Linear, nets.py:187-194, 462-464). */
int cgs_gen_gemm_ex_splitk(int32_t m, int32_t k, int32_t n, const float* x, int64_t sxm, int64_t sxk, const float* w,
                           int64_t swk, int64_t swn, int32_t nsplit, float* slab, cgs_stream_t stream);
int cgs_gen_u8_to_f32(int64_t count, const uint8_t* x, float* out, cgs_stream_t stream);

/* ---- fp16 inference family (csrc/gen_f16.hip): BASELINE config 4, "-process inference-only mask path ... fp16 conv kernels" --------
 * Every layer of the eval-mode forward (nets.py:197-212, 494-523; main.py:1130-1151) with fp16 activations (NHWC) and fp16 weights,
 * fp32 accumulation on v_mfma_f32_16x16x16_f16, runtime channel counts (any chfak).  Opt-in: ~1e-3 absolute in the masks.
 * cgs_gen16_pack_weights: HWIO fp32 w [9][ca+cb][co] -> the kernel's fp16 operand layout (cgs_gen16_packed_weight_halves halves).
 * cgs_gen16_conv3x3_fwd: out = act(conv3x3(cat(A, nearest-up_ups(B))) + bias) (+ MaxPool2d(2)); A uint8 (/255 fused) or fp16
 *   (ca % 4 == 0), B fp16; out fp16, or fp32 when out_is_f32.
 * cgs_gen16_gemm: out [m,n] = act(x [m,k] w [k,n] + bias), x fp16 or fp32, w / bias fp32, out fp16 or fp32.                         */
int64_t cgs_gen16_packed_weight_halves(int32_t ca, int32_t cb, int32_t co);
int cgs_gen16_pack_weights(int32_t ca, int32_t cb, int32_t co, const float* w, void* w16, cgs_stream_t stream);
int cgs_gen16_conv3x3_fwd(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_is_u8, int32_t ups, int32_t act,
                          float slope, int32_t pool, int32_t out_is_f32, const void* src_a, const void* src_b, const void* w16,
                          const float* bias, void* out, cgs_stream_t stream);
int cgs_gen16_gemm(int32_t m, int32_t k, int32_t n, int32_t act, float slope, int32_t x_is_f16, int32_t out_is_f16, const void* x,
                   const float* w, const float* bias, void* out, cgs_stream_t stream);

/* ---- bfloat16 form of the same family (BASELINE config 5: "128x128x3 frames ... bf16 with MFMA 1x1 pointwise") -----------------------
 * The same kernels with bf16 activations / weights (fp32 accumulation on v_mfma_f32_16x16x16_bf16), additionally at map size 128:
 * the build-defined six-stage 128x128 Hourglass (hourglass128.py; the reference cannot run 128x128 inputs -- nets.py:184,189-190 --
 * so this variant has no reference counterpart: parity unpinned, checked against the build's own fp32 restatement in oracle/).
 * Operand layout and packed size as cgs_gen16_* (cgs_gen16_packed_weight_halves 16-bit words).                                       */
int cgs_genbf16_pack_weights(int32_t ca, int32_t cb, int32_t co, const float* w, void* w16, cgs_stream_t stream);
int cgs_genbf16_conv3x3_fwd(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_is_u8, int32_t ups, int32_t act,
                            float slope, int32_t pool, int32_t out_is_f32, const void* src_a, const void* src_b, const void* w16,
                            const float* bias, void* out, cgs_stream_t stream);
int cgs_genbf16_gemm(int32_t m, int32_t k, int32_t n, int32_t act, float slope, int32_t x_is_bf16, int32_t out_is_bf16, const void* x,
                     const float* w, const float* bias, void* out, cgs_stream_t stream);

/* ---- bf16 TRAINING of the build-defined 128x128 variant (BASELINE config 5; hourglass128.py; csrc/gen_bf16_train.hip) -------------------
 * No reference counterpart (the reference's 4x4 valid convolution nets.py:184 cannot take 128x128 frames): parity unpinned.
 * cgs_genbf16_conv3x3_fwd_train: cgs_genbf16_conv3x3_fwd with source A bf16 (a_kind 0), uint8 (1) or fp32 (2: the replaced / injected
 *   mixes) and, for pooled layers, codes [n,hw/2,hw/2,co] = the argmax position 0..3 of every pooled element (4: value <= 0).
 * cgs_genbf16_pack_weights_t: the data gradient's operand -- cgs_genbf16_conv3x3_fwd(ca = co_layer, cb = 0, co = ci_layer) on it computes
 *   d cat(A, up(B)) [n,hw,hw,ci_layer] from dY [n,hw,hw,co_layer] (packed size cgs_gen16_packed_weight_halves(co_layer, 0, ci_layer)).
 * cgs_bf16_conv3x3_bwd_weight: slab [cgs_bf16_conv3x3_bwd_weight_slabs(n,hw,ca,cb)][9 (ca+cb) co + co] = per-workgroup partial
 *   (dW in HWIO order | db) of conv3x3(cat(A, up_ups(B))) from dY [n,hw,hw,dy_channels] (bf16; columns >= co zero padding), on
 *   v_mfma_f32_16x16x32_bf16 with K = 32 pixels (operands through ds_read_b64_tr_b16); summed by cgs_reduce_slabs.
 * cgs_bf16_pool_expand: out [n,2hp,2hp,c] = (dp [n,hp,hp,c] (+ addend)) at the argmax position, 0 elsewhere (backward of ReLU + MaxPool2d(2)).
 * cgs_bf16_cat_split: d cat [n,hw,hw,ca+cb] -> dskip [n,hw,hw,ca] (may be NULL) and dlow [n,hw/ups,hw/ups,cb] (cell sums; fp32 when low_is_f32).
 * cgs_bf16_lrelu_bwd: d *= LeakyReLU'(h) in place.  cgs_bf16_convert: rows x c_src -> rows x c_dst (zero pad), fp32 -> bf16 or back.      */
int cgs_genbf16_conv3x3_fwd_train(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t a_kind, int32_t ups, int32_t act,
                                  float slope, int32_t pool, int32_t out_is_f32, const void* src_a, const void* src_b, const void* w16,
                                  const float* bias, void* out, uint8_t* codes, cgs_stream_t stream);
int cgs_genbf16_pack_weights_t(int32_t ci_layer, int32_t co_layer, const float* w_hwio, void* w16, cgs_stream_t stream);
/* All bf16 operand copies of a step in one launch: jobs = DEVICE array of njobs entries; transposed = 0: cgs_genbf16_pack_weights(ca, cb, co),
 * 1: cgs_genbf16_pack_weights_t(ca + cb, co) of the layer whose HWIO weights are w.                                                       */
typedef struct cgs_gen16_pack_job {
    const float* w;
    void* out;
    int32_t ca, cb, co, transposed;
} cgs_gen16_pack_job;
int cgs_genbf16_pack_batch(const cgs_gen16_pack_job* jobs, int32_t njobs, cgs_stream_t stream);
int cgs_bf16_conv3x3_bwd_weight_slabs(int32_t n, int32_t hw, int32_t ca, int32_t cb);
int cgs_bf16_conv3x3_bwd_weight(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t co, int32_t dy_channels, int32_t a_kind, int32_t ups,
                                const void* src_a, const void* src_b, const void* dy, float* slab, cgs_stream_t stream);
int cgs_bf16_pool_expand(int32_t n, int32_t hp, int32_t c, const void* dp, const void* addend, const uint8_t* codes, void* out,
                         cgs_stream_t stream);
int cgs_bf16_cat_split(int32_t n, int32_t hw, int32_t ca, int32_t cb, int32_t ups, const void* dcat, void* dskip, void* dlow,
                       int32_t low_is_f32, cgs_stream_t stream);
int cgs_bf16_lrelu_bwd(int64_t count, void* d, const void* h, float slope, cgs_stream_t stream);
int cgs_bf16_convert(int64_t rows, int32_t c_src, int32_t c_dst, int32_t to_f32, const void* src, void* dst, cgs_stream_t stream);
int cgs_gen_convt4s2_fwd(int32_t n, int32_t h, int32_t ca, int32_t cb, int32_t co, int32_t act, float slope,
                         const float* a, const float* b, const float* w, const float* bias, float* out,
                         cgs_stream_t stream);
int cgs_gen_convt4s2_bwd_data(int32_t n, int32_t h, int32_t ca, int32_t cb, int32_t co, const float* dy,
                              const float* w, float* da, float* db, cgs_stream_t stream);
int cgs_gen_convt4s2_bwd_weight(int32_t n, int32_t h, int32_t ca, int32_t cb, int32_t co, const float* a,
                                const float* b, const float* dy, float* dw, float* dbias, cgs_stream_t stream);

/* ---- contrastive batch assembly on the device (main.py:344-356, 584-591) ------------------------------------------
 * dst [n,64,64,3] uint8: frame src[idx[i]] of a device-resident uint8 frame set, rolled circularly along the width:
 * dst[i][y][x] = src[idx[i]][y][(x + shift_px) mod 64].  Handler.shift_batch's "left" roll by s pixels is shift_px = s, its
 * "right" roll shift_px = (64 - s) mod 64.  idx: device int64 [n].  cgs_gather_f32: dst[i] = src[idx[i]] (the targets). */
int cgs_gather_roll_u8(const uint8_t* src, const int64_t* idx, int32_t n, int32_t shift_px, uint8_t* dst,
                       cgs_stream_t stream);
int cgs_gather_f32(const float* src, const int64_t* idx, int32_t n, float* dst, cgs_stream_t stream);
/* The contrastive batch assembly of a phase-2 step in one launch (main.py:344-356 + shift_batch main.py:584-591): a [n,64,64,3] =
 * [xpos[idx[0..h)] | xneg[idx[h..n)]] rolled along the width by shift_px pixels (0 <= shift_px < 64), b [n,64,64,3] = xneg[idx[n..2n)] (not
 * rolled), y [n] = [ypos[idx[0..h)] | yneg[idx[h..n)]].  Equal to three cgs_gather_roll_u8 and two cgs_gather_f32 calls.                 */
int cgs_gather_contrastive(const uint8_t* xpos, const uint8_t* xneg, const float* ypos, const float* yneg, const int64_t* idx, int32_t n,
                           int32_t h, int32_t shift_px, uint8_t* a, uint8_t* b, float* y, cgs_stream_t stream);

const char* cgs_build_arch(void);
int cgs_abi_version(void);

/* ---- optional BatchNorm2d (+ ReLU / LeakyReLU) epilogue (csrc/bn.hip) ------------------------------------------------------------------
 * SURVEY section 8 row f4 ("an optional BN epilogue") / the north_star's BN wording.  The reference has NO BatchNorm: this op is on
 * no parity path and nothing of the reference pins it (tests: torch.nn.BatchNorm2d + autograd in float64).  NHWC fp32 x / y / dy / dx
 * [pixels][c], c a multiple of 4, <= 64; act: CGS_ACT_NONE / RELU / LRELU.  train != 0: batch statistics (biased variance for the normalisation,
 * unbiased into running_var, momentum as torch); train == 0: running statistics.  stats [c][4] = mean, 1/sqrt(var+eps), scale, shift.
 * ws: (3 cgs_bn_rows(pixels, c) + 2) * c floats.  Statistics are per GPU under data parallelism (no collective).                          */
int cgs_bn_rows(int64_t pixels, int32_t c);
int cgs_bn_act_fwd(int64_t pixels, int32_t c, const float* x, const float* gamma, const float* beta, float eps, int32_t act, float slope,
                   int32_t train, float* y, float* stats, float* ws, float* running_mean, float* running_var, float momentum,
                   cgs_stream_t stream);
int cgs_bn_act_bwd(int64_t pixels, int32_t c, const float* x, const float* y, const float* dy, const float* stats, int32_t act, float slope,
                   int32_t train, float* dx, float* dgamma, float* dbeta, float* ws, cgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CGS_HIP_H */

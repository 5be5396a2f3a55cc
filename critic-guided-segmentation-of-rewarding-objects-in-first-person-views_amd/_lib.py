"""ctypes binding of libcgs_hip.so (C ABI declared in include/cgs_hip.h).

The product path has NO fallback: if the library is missing or was not built, importing a kernel
raises.  PyTorch only supplies device memory (``tensor.data_ptr()``) and the current HIP stream."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CGS_LIB_PATH") or os.path.join(HERE, "libcgs_hip.so")   # override: A/B builds

OK, ERR_UNSUPPORTED, ERR_BADARG = 0, -1, -2
SRC_F32, SRC_U8, SRC_MIX = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SIGMOID = 0, 1, 2, 3
H5_ENC1_FWD, H5_ENC1_BWD_DATA, H5_DEC0_FWD, H5_DEC0_BWD_SKIP, H5_DEC0_BWD_LOW = 1, 2, 3, 4, 5
H5_ENC1_BWD_DATA_POOLED = 6
H5_ENC2_FWD, H5_ENC2_BWD_DATA_POOLED, H5_DEC1_FWD, H5_DEC1_BWD_SKIP, H5_DEC1_BWD_LOW, H5_DEC1_FWD_F32B = 7, 8, 9, 10, 11, 12      # cgs_bf16_h5conv (cgs_hip.h)

vp = C.c_void_p
i32 = C.c_int32
i64 = C.c_int64
f32 = C.c_float


class Dropout(C.Structure):
    _fields_ = [("p", C.c_float), ("site", C.c_uint32), ("seed", C.c_uint64), ("step", C.c_void_p),
                ("base", C.c_uint32), ("reserved", C.c_uint32)]


class ConvDesc(C.Structure):
    _fields_ = [("n", i32), ("h", i32), ("w", i32), ("ca", i32), ("cb", i32), ("co", i32), ("src_a", i32),
                ("ups", i32), ("act", i32), ("pool", i32), ("drop_a", Dropout)]


class MixSrc(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("z", C.c_void_p), ("n_a", i32), ("reserved", i32)]


class TailEncWeights(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("w6", "b6", "w10", "b10", "w14", "b14", "wl1", "bl1", "wl2", "bl2", "wpw", "bpw")]


class TailDecWeights(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("w3", "b3", "w2", "b2", "w1", "b1")]


class Pack16Job(C.Structure):        # = cgs_gen16_pack_job (include/cgs_hip.h)
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p), ("ca", i32), ("cb", i32), ("co", i32), ("transposed", i32)]


class ReduceJob(C.Structure):
    _fields_ = [("slab", C.c_void_p), ("dst", C.c_void_p), ("nslab", i32), ("stride", i32), ("count", i32),
                ("accumulate", i32)]


# name -> (restype, argtypes); mirrors include/cgs_hip.h one to one
SIGNATURES = {
    "cgs_conv3x3_fwd": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp]),
    "cgs_conv3x3_bwd_data": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, i32, vp, i32, vp, vp, vp]),
    "cgs_mask_infer_fwd": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_mask_infer_fwd_packed": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_mask_infer_fwd_tile": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_mask_infer_fwd_f16": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_mask_infer_fwd_f16o": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_f16_enc0_fwd": (i32, [i32, vp, vp, vp, vp, vp]),
    "cgs_f16_enc1_fwd": (i32, [i32, vp, vp, vp, vp, vp]),
    "cgs_f16_dec0_fwd": (i32, [i32, vp, vp, vp, vp, vp, vp]),
    "cgs_bf16_enc0_fwd": (i32, [i32, vp, i32, vp, vp, vp, vp, vp]),
    "cgs_bf16_mask0_fwd": (i32, [i32, vp, vp, vp, vp, vp, vp]),
    "cgs_bf16_mask2_fwd": (i32, [i32, vp, vp, vp, vp, vp]),
    "cgs_bf16_enc0_bwd_data": (i32, [i32, vp, vp, vp, vp]),
    "cgs_bf16_mask2_bwd_data": (i32, [i32, vp, vp, vp, vp, vp]),
    "cgs_bf16_enc0_bwd_data_pooled": (i32, [i32, vp, vp, vp, vp, vp, vp]),
    "cgs_bf16_enc0_bwd_mix": (i32, [i32, vp, vp, vp, vp, vp, vp, f32, f32, vp, vp]),
    "cgs_bf16_enc0_fwd_mix": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_bf16_hwgrad_pooled_mix": (i32, [i32, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_bf16_mask0_bwd_data": (i32, [i32, vp, vp, vp, vp]),
    "cgs_bf16_h5conv": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_gen_enc0_fwd": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "cgs_gen_enc0_bwd_data": (i32, [i32, i32, vp, vp, vp, vp, vp]),
    "cgs_gen_enc0_bwd_weight_slabs": (i32, [i32, i32]),
    "cgs_gen_enc0_bwd_weight": (i32, [i32, i32, i32, vp, vp, vp, vp, vp]),
    "cgs_bn_rows": (i32, [i64, i32]),
    "cgs_bn_act_fwd": (i32, [i64, i32, vp, vp, vp, f32, i32, f32, i32, vp, vp, vp, vp, vp, f32, vp]),
    "cgs_bn_act_bwd": (i32, [i64, i32, vp, vp, vp, vp, i32, f32, i32, vp, vp, vp, vp, vp]),
    "cgs_bf16_hwgrad_slabs": (i32, [i32, i32, i32, i32, i32]),
    "cgs_bf16_hwgrad": (i32, [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "cgs_bf16_hwgrad_pooled": (i32, [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "cgs_mask_train_fwd_partials": (i32, [i32]),
    "cgs_mask_train_fwd": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_mask_train_fwd_packed": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_mask_head_bwd_slabs": (i32, [i32]),
    "cgs_mask_head_bwd": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_tail_enc_fwd": (i32, [i32, C.POINTER(TailEncWeights), vp, vp, vp, vp, vp, vp, vp, vp, vp, Dropout, Dropout, Dropout, vp]),
    "cgs_enc1_tail_fwd": (i32, [i32, C.POINTER(TailEncWeights), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, Dropout, Dropout, Dropout, vp]),
    "cgs_critic_fwd_fused": (i32, [i32, C.POINTER(TailEncWeights), vp, i32] + [vp] * 16 + [Dropout, Dropout, Dropout, vp]),
    "cgs_tail_dec_fwd": (i32, [i32, C.POINTER(TailDecWeights), vp, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_tail_enc_fwd_h16": (i32, [i32, C.POINTER(TailEncWeights), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_tail_dec_fwd_h16": (i32, [i32, C.POINTER(TailDecWeights), vp, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_tail_infer_h16": (i32, [i32, C.POINTER(TailEncWeights), C.POINTER(TailDecWeights), vp, vp, vp, vp]),
    "cgs_f16_enc1_tail_infer": (i32, [i32, vp, vp, vp, C.POINTER(TailEncWeights), C.POINTER(TailDecWeights), vp, vp, vp]),
    "cgs_tail_dec_fwd_pack": (i32, [i32, C.POINTER(TailDecWeights), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_tail_enc_bwd_slabs": (i32, [i32]),
    "cgs_tail_enc_bwd": (i32, [i32, C.POINTER(TailEncWeights)] + [vp] * 10 + [f32, i32] + [vp] * 4 + [i32] + [vp] * 4 + [Dropout, Dropout, Dropout, vp]),
    "cgs_tail_enc_bwd_rider": (i32, [i32, C.POINTER(TailEncWeights)] + [vp] * 10 + [f32, i32] + [vp] * 4 + [i32] + [vp] * 4 + [Dropout, Dropout, Dropout, i32, vp, vp, vp, vp, i32, vp]),
    "cgs_tail_enc_bwd_enc1": (i32, [i32, C.POINTER(TailEncWeights)] + [vp] * 10 + [f32, i32] + [vp] * 4 + [i32] + [vp] * 4 + [Dropout, Dropout, Dropout, i32, vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp]),
    "cgs_enc1_wgrad_rider_slabs": (i32, [i32]),
    "cgs_enc0_wgrad_u8_with_head_enc1": (i32, [i32, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, i32, vp]),
    "cgs_enc0_bwd_mix_enc1": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_tail_head_wgrad_slabs": (i32, [i32]),
    "cgs_tail_head_wgrad": (i32, [i32, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp]),
    "cgs_enc0_wgrad_u8_with_head": (i32, [i32, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp]),
    "cgs_reduce_adam": (i32, [vp, i32, i32, vp, vp, vp, vp, vp, f32, f32, f32, f32, vp, i32, vp, vp, vp, i32, f32, f32, f32, i32, i64, vp, vp]),
    "cgs_tail_dec_bwd_slabs": (i32, [i32]),
    "cgs_tail_dec_bwd": (i32, [i32, C.POINTER(TailDecWeights)] + [vp] * 14 + [vp]),
    "cgs_dec0_tail_dec_bwd": (i32, [i32, C.POINTER(TailDecWeights)] + [vp] * 17 + [vp]),
    "cgs_dec0_tail_dec_bwd_do3": (i32, [i32, C.POINTER(TailDecWeights)] + [vp] * 18 + [vp]),
    "cgs_dec3_wgrad_rider_slabs": (i32, [i32]),
    "cgs_enc0_wgrad_u8_with_head_riders": (i32, [i32, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    "cgs_tail_dec_fwd_dec0": (i32, [i32, C.POINTER(TailDecWeights)] + [vp] * 13 + [vp]),
    "cgs_conv3x3_bwd_weight_slabs": (i32, [C.POINTER(ConvDesc)]),
    "cgs_conv3x3_bwd_weight": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp]),
    "cgs_conv3x3_bwd_both_slabs": (i32, [C.POINTER(ConvDesc)]),
    "cgs_conv3x3_bwd_both": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]),
    "cgs_reduce_slabs": (i32, [vp, i32, i32, vp, vp]),
    "cgs_head_fwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, Dropout, Dropout, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_enc0_bwd_mix_slabs": (i32, [i32]),
    "cgs_enc0_bwd_mix": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp]),
    "cgs_head_bwd_slabs": (i32, [i32]),
    "cgs_head_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, Dropout, Dropout, vp, vp, vp, vp, vp, vp]),
    "cgs_pointwise_fwd": (i32, [i32, i32, i32, vp, vp, vp, vp, vp]),
    "cgs_pointwise_bwd_slabs": (i32, [i32]),
    "cgs_pointwise_bwd": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "cgs_mix_fwd": (i32, [i32, i32, vp, vp, vp, i32, vp, vp, vp]),
    "cgs_mix_bwd": (i32, [i32, i32, vp, vp, vp, vp, i32, f32, f32, vp, vp]),
    "cgs_mix_bwd_weighted": (i32, [i32, i32, vp, vp, vp, vp, i32, f32, f32, vp, vp, vp]),
    "cgs_mix_fwd_partials": (i32, [i32, i32]),
    "cgs_phase2_losses": (i32, [i32, vp, vp, vp, i32, f32, f32, f32, i32, i64, vp, vp, vp]),
    "cgs_phase1_loss": (i32, [i32, vp, vp, i32, vp, vp, vp]),
    "cgs_adam_flat": (i32, [i64, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, vp]),
    "cgs_nchw_to_nhwc": (i32, [i32, i32, i32, vp, vp, vp]),
    "cgs_nhwc_to_nchw": (i32, [i32, i32, i32, vp, vp, vp]),
    "cgs_gen_conv_packed_floats": (i64, [i32, i32, i32]),
    "cgs_gen_conv_packed_floats_folded": (i64, [i32, i32, i32]),
    "cgs_gen_conv_packed_floats_up2": (i64, [i32, i32]),
    "cgs_gen_conv_pack_weights_up2": (i32, [i32, i32, i32, i32, vp, vp, vp]),
    "cgs_gen_conv3x3_bwd_data_up2": (i32, [i32, i32, i32, i32, vp, vp, vp, vp]),
    "cgs_gen_conv_pack_weights": (i32, [i32, i32, i32, i32, vp, vp, vp]),
    "cgs_gen_conv_pack_weights_window": (i32, [i32, i32, i32, i32, vp, vp, vp]),
    "cgs_gen_conv_pack_batch": (i32, [vp, i32, vp]),
    "cgs_gen_conv3x3_fwd": (i32, [i32, i32, i32, i32, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_gen_conv3x3_fwd_folded": (i32, [i32, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp]),
    "cgs_gen_gemm": (i32, [i32, i32, i32, i32, f32, vp, vp, vp, vp, vp]),
    "cgs_gen_flip_weights": (i32, [i32, i32, vp, vp, vp]),
    "cgs_gen_conv3x3_bwd_data": (i32, [i32, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp]),
    "cgs_gen_conv3x3_bwd_data_split": (i32, [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "cgs_gen_conv3x3_bwd_weight_slabs": (i32, [i32, i32, i32, i32]),
    "cgs_gen_conv3x3_bwd_weight_folded_slabs": (i32, [i32, i32, i32, i32, i32]),
    "cgs_gen_conv3x3_bwd_weight_folded": (i32, [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "cgs_gen_conv3x3_bwd_weight": (i32, [i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "cgs_gen_cat_split": (i32, [i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "cgs_gen_grad_fix": (i32, [i64, vp, vp, i32, f32, vp, i64, Dropout, vp]),
    "cgs_gen_dropout_fwd": (i32, [i64, vp, vp, Dropout, vp]),
    "cgs_gen_gemm_ex": (i32, [i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, i32, f32, i32, vp, vp]),
    "cgs_gen_gemm_ex_splitk": (i32, [i32, i32, i32, vp, i64, i64, vp, i64, i64, i32, vp, vp]),
    "cgs_gen_u8_to_f32": (i32, [i64, vp, vp, vp]),
    "cgs_gen16_packed_weight_halves": (i64, [i32, i32, i32]),
    "cgs_gen16_pack_weights": (i32, [i32, i32, i32, vp, vp, vp]),
    "cgs_gen16_conv3x3_fwd": (i32, [i32, i32, i32, i32, i32, i32, i32, i32, f32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "cgs_gen16_gemm": (i32, [i32, i32, i32, i32, f32, i32, i32, vp, vp, vp, vp, vp]),
    "cgs_genbf16_pack_weights": (i32, [i32, i32, i32, vp, vp, vp]),
    "cgs_genbf16_conv3x3_fwd": (i32, [i32, i32, i32, i32, i32, i32, i32, i32, f32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "cgs_genbf16_gemm": (i32, [i32, i32, i32, i32, f32, i32, i32, vp, vp, vp, vp, vp]),
    "cgs_genbf16_conv3x3_fwd_train": (i32, [i32, i32, i32, i32, i32, i32, i32, i32, f32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "cgs_genbf16_pack_weights_t": (i32, [i32, i32, vp, vp, vp]),
    "cgs_genbf16_pack_batch": (i32, [vp, i32, vp]),
    "cgs_bf16_conv3x3_bwd_weight_slabs": (i32, [i32, i32, i32, i32]),
    "cgs_bf16_conv3x3_bwd_weight": (i32, [i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "cgs_bf16_pool_expand": (i32, [i32, i32, i32, vp, vp, vp, vp, vp]),
    "cgs_bf16_cat_split": (i32, [i32, i32, i32, i32, i32, vp, vp, vp, i32, vp]),
    "cgs_bf16_lrelu_bwd": (i32, [i64, vp, vp, f32, vp]),
    "cgs_bf16_convert": (i32, [i64, i32, i32, i32, vp, vp, vp]),
    "cgs_gen_convt4s2_fwd": (i32, [i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp]),
    "cgs_gen_convt4s2_bwd_data": (i32, [i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "cgs_gen_convt4s2_bwd_weight": (i32, [i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "cgs_gather_roll_u8": (i32, [vp, vp, i32, i32, vp, vp]),
    "cgs_gather_f32": (i32, [vp, vp, i32, vp, vp]),
    "cgs_gather_contrastive": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "cgs_dropout_mask": (i32, [Dropout, i64, vp, vp]),
    "cgs_build_arch": (C.c_char_p, []),
    "cgs_abi_version": (i32, []),
}

_lib = None


class CgsError(RuntimeError):
    pass


def load():
    """Loads the HIP library (once).  Fails loudly when it is absent -- there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CgsError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what):
    if code == OK:
        return
    if code == ERR_UNSUPPORTED:
        raise CgsError(f"{what}: shape/configuration not supported by the HIP kernels (CGS_ERR_UNSUPPORTED)")
    if code == ERR_BADARG:
        raise CgsError(f"{what}: bad argument (CGS_ERR_BADARG)")
    raise CgsError(f"{what}: hipError {code}")


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)

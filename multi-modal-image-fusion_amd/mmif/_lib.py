"""ctypes binding of libmmif_hip.so (C ABI declared in include/mmif.h).

The library is the product: if it is missing or fails to load, importing this module raises --
there is no CPU or torch fallback for the hot path."""
import ctypes as C
import os

import torch  # noqa: F401  MUST precede the CDLL below: the process-wide HIP runtime has to be the one torch
#                            ships (libamdhip64 in torch/lib); loading ours first leaves two runtimes that do not
#                            share devices ("no ROCm-capable device is detected" on the first launch)

_HERE = os.path.dirname(os.path.abspath(__file__))
# $MMIF_LIB: load another build of the library (A/B timing of kernel variants inside one GPU session)
LIB_PATH = os.environ.get("MMIF_LIB") or os.path.join(os.path.dirname(_HERE), "libmmif_hip.so")

F32, BF16 = 0, 1
IMPL_AUTO, IMPL_VALU, IMPL_MFMA, IMPL_X3 = 0, 1, 2, 3
PACK_BF16, PACK_X3 = 0, 1
FUSE_SUM, FUSE_MEAN, FUSE_MAX = 0, 1, 2
T_FOLDED = 1


class MmifTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("dtype", C.c_int32), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("halo", C.c_int32), ("cb_total", C.c_int32), ("cb_off", C.c_int32), ("cb", C.c_int32), ("flags", C.c_int32)]


class MmifPackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("cout", C.c_int32), ("cin", C.c_int32), ("ksize", C.c_int32), ("format", C.c_int32),
                ("packed_fwd", C.c_void_p), ("packed_dgrad", C.c_void_p)]


class MmifDenseEncoder(C.Structure):
    _fields_ = [("img", C.c_void_p), ("w0", C.c_void_p), ("b0", C.c_void_p), ("packed", C.c_void_p * 3), ("bias", C.c_void_p * 3)]


class MmifDenseChain(C.Structure):
    _fields_ = [("g3", C.POINTER(MmifTensor)), ("glow", C.POINTER(MmifTensor)), ("x", C.POINTER(MmifTensor)), ("packed", C.c_void_p * 3),
                ("out", C.POINTER(MmifTensor))]


class MmifError(RuntimeError):
    pass


def _load():
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"mmif: {LIB_PATH} not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C multi-modal-image-fusion_amd/csrc`).  The HIP library is required; there is no fallback.")
    return C.CDLL(LIB_PATH)


lib = _load()

_TP = C.POINTER(MmifTensor)
_vp, _i32, _u64, _f32, _sz, _i64, _f64 = C.c_void_p, C.c_int32, C.c_uint64, C.c_float, C.c_size_t, C.c_int64, C.c_double

# name -> (restype, argtypes); mirrors include/mmif.h one for one
SIGNATURES = {
    "mmif_version": (C.c_char_p, []),
    "mmif_last_error": (C.c_char_p, []),
    "mmif_nchw_to_blocked": (_i32, [_vp, _i32, _TP, _vp]),
    "mmif_blocked_to_nchw": (_i32, [_TP, _vp, _i32, _vp]),
    "mmif_zero": (_i32, [_TP, _vp]),
    "mmif_fold_halo": (_i32, [_TP, _vp]),
    "mmif_packed_weight_bytes": (_sz, [_i32, _i32, _i32]),
    "mmif_pack_weights": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mmif_pack_weights_multi": (_i32, [C.POINTER(MmifPackJob), _i32, _vp]),
    "mmif_set_x3_forward_pieces": (None, [_i32]),
    "mmif_get_x3_forward_pieces": (_i32, []),
    "mmif_get_x3_enabled": (_i32, []),
    "mmif_x3_pack_saturations": (_i32, [_i32]),
    "mmif_packed_weight_bytes_x3": (_sz, [_i32, _i32, _i32]),
    "mmif_pack_weights_x3": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mmif_gconv_fwd": (_i32, [_vp, _vp, _vp, _vp] + [_i32] * 10 + [_vp]),
    "mmif_gconv_dgrad_workspace": (_sz, [_i32] * 6),
    "mmif_gconv_dgrad": (_i32, [_vp, _vp, _vp] + [_i32] * 9 + [_vp, _sz, _vp]),
    "mmif_gconv_wgrad_workspace": (_sz, [_i32] * 3),
    "mmif_gconv_wgrad": (_i32, [_vp, _vp, _vp, _vp] + [_i32] * 9 + [_vp, _sz, _vp]),
    "mmif_gconvt_fwd": (_i32, [_vp, _vp, _vp, _vp] + [_i32] * 10 + [_vp]),
    "mmif_gconvt_dgrad": (_i32, [_vp, _vp, _vp] + [_i32] * 9 + [_vp]),
    "mmif_gconvt_wgrad": (_i32, [_vp, _vp, _vp, _vp] + [_i32] * 9 + [_vp, _sz, _vp]),
    "mmif_relu_bwd": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "mmif_dwconv_fwd": (_i32, [_vp, _vp, _vp, _vp] + [_i32] * 6 + [_vp]),
    "mmif_dwconv_dgrad": (_i32, [_vp, _vp, _vp] + [_i32] * 6 + [_vp]),
    "mmif_dwconv_wgrad": (_i32, [_vp, _vp, _vp, _vp] + [_i32] * 6 + [_vp]),
    "mmif_channel_sum": (_i32, [_vp, _vp, _i32, _i32, _i64, _vp]),
    "mmif_bilinear_up_fwd": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mmif_bilinear_up_bwd": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mmif_maxpool_nchw_fwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "mmif_maxpool_nchw_bwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "mmif_nearest_up_fwd": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "mmif_nearest_up_bwd": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "mmif_reflect_pad_fwd": (_i32, [_vp, _vp, _i64] + [_i32] * 6 + [_vp]),
    "mmif_reflect_pad_bwd": (_i32, [_vp, _vp, _i64] + [_i32] * 6 + [_vp]),
    "mmif_norm_workspace": (_sz, [_i32, _i32]),
    "mmif_norm_act_fwd": (_i32, [_vp] * 7 + [_i32, _i32, _i64, _i32, _f32, _f32, _i32, _f32, _vp, _sz, _vp]),
    "mmif_norm_act_bwd": (_i32, [_vp] * 8 + [_i32, _i32, _i64, _i32, _i32, _f32, _vp, _sz, _vp]),
    "mmif_bn_moments": (_i32, [_vp, _vp, _i32, _i32, _i64, _vp, _sz, _vp]),
    "mmif_bn_apply_fwd": (_i32, [_vp] * 8 + [_i32, _i32, _i64, _f32, _f32, _i32, _f32, _vp]),
    "mmif_bn_bwd_sums": (_i32, [_vp] * 7 + [_i32, _i32, _i64, _i32, _f32, _vp, _sz, _vp]),
    "mmif_bn_apply_bwd": (_i32, [_vp] * 8 + [_i32, _i32, _i64, _i32, _f32, _vp]),
    "mmif_dense_encoder_fwd": (_i32, [C.POINTER(MmifDenseEncoder), _TP, C.POINTER(MmifDenseEncoder), _TP, _vp]),
    "mmif_conv2d_dgrad_dup_supported": (_i32, [_TP, _TP, _i32, _i32, _i32]),
    "mmif_conv2d_reflect_dgrad_folded_dup": (_i32, [_TP, _vp, _TP, _i32, _i32, _i32, _TP, _TP, _i32, _vp]),
    "mmif_dense_encoder_bwd_fits": (_i32, [_i32, _i32, _i32, _i32]),
    "mmif_dense_encoder_fwd_sum_supported": (_i32, [C.POINTER(MmifDenseEncoder), C.POINTER(MmifDenseEncoder), _i32, _i32, _i32]),
    "mmif_dense_encoder_fwd_sum": (_i32, [C.POINTER(MmifDenseEncoder), _TP, C.POINTER(MmifDenseEncoder), _TP, _TP, _vp]),
    "mmif_conv2d_bwd_pair_supported": (_i32, [_i32, _i32, _i32]),
    "mmif_conv2d_reflect_bwd_pair": (_i32, [_TP, _vp, _TP, _TP, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "mmif_conv2d_bwd_wide_supported": (_i32, [_i32, _i32, _i32]),
    "mmif_conv2d_bwd_wide_supported_f32": (_i32, [_i32, _i32, _i32]),
    "mmif_conv2d_bwd_wide_signs_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "mmif_conv2d_reflect_bwd_wide": (_i32, [_TP, _vp, _TP, _TP, _vp, _vp, _i32, _i32, _i32, _u64, _i32, _vp, _sz, _vp, _sz, _vp]),
    "mmif_pack_dense_chain": (_i32, [_vp] * 7),
    "mmif_pack_dense_chain_pair": (_i32, [_vp] * 5),
    "mmif_reduce_defer_begin": (_i32, [_vp, _sz]),
    "mmif_reduce_defer_flush": (_i32, [_i32, _vp]),
    "mmif_reduce_defer_pending": (_i32, []),
    "mmif_pack_dense_chain_x3": (_i32, [_vp] * 7),
    "mmif_dense_encoder_chain": (_i32, [C.POINTER(MmifDenseChain), C.POINTER(MmifDenseChain), _vp]),
    "mmif_dense_encoder_bwd_workspace": (_sz, []),
    "mmif_dense_encoder_bwd": (_i32, [C.POINTER(MmifDenseChain), _vp, C.POINTER(C.c_void_p), _i32, C.POINTER(MmifDenseChain), _vp, C.POINTER(C.c_void_p), _i32,
                                      _vp, _sz, _vp]),
    "mmif_dense_encoder_wgrad_workspace": (_sz, []),
    "mmif_dense_encoder_wgrad": (_i32, [_vp, _TP, _TP] + [_vp] * 8 + [_i32, _vp, _sz, _vp]),
    "mmif_act_fwd": (_i32, [_vp, _vp, _i64, _i32, _f32, _vp]),
    "mmif_act_bwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "mmif_conv2d_reflect_fwd": (_i32, [_TP, _vp, _vp, _vp, _TP, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mmif_conv2d_reflect_dgrad": (_i32, [_TP, _vp, _vp, _TP, _TP, _i32, _i32, _i32, _u64, _u64, _i32, _vp]),
    "mmif_conv2d_dgrad_onto_supported": (_i32, [_TP, _TP, _i32, _i32, _i32]),
    "mmif_conv2d_reflect_dgrad_folded_onto": (_i32, [_TP, _vp, _TP, _TP, _TP, _i32, _i32, _i32, _u64, _u64, _vp]),
    "mmif_conv2d_reflect_dgrad_folded": (_i32, [_TP, _vp, _vp, _TP, _TP, _i32, _i32, _i32, _u64, _u64, _i32, _vp]),
    "mmif_conv2d_wgrad_workspace": (_sz, [_i32, _i32, _i32]),
    "mmif_conv2d_reflect_wgrad": (_i32, [_TP, _TP, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _sz, _i32, _vp]),
    "mmif_conv2d_image_in_fwd": (_i32, [_vp, _vp, _vp, _TP, _i32, _i32, _i32, _vp]),
    "mmif_conv2d_image_in_wgrad": (_i32, [_vp, _TP, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "mmif_conv2d_image_out_fwd": (_i32, [_TP, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "mmif_conv2d_image_out_dgrad": (_i32, [_vp, _vp, _vp, _TP, _TP, _i32, _i32, _u64, _u64, _vp]),
    "mmif_conv2d_image_out_wgrad": (_i32, [_TP, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "mmif_conv2d_image_wgrad_workspace": (_sz, [_i32, _i32]),
    "mmif_conv2d_image_out_bwd_supported": (_i32, [_i32, _i32, _i32, _i32, _i32]),
    "mmif_conv2d_image_out_bwd": (_i32, [_TP, _vp, _vp, _vp, _TP, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "mmif_fuse_elem_fwd": (_i32, [_TP, _TP, _TP, _i32, _vp]),
    "mmif_fuse_elem_bwd": (_i32, [_TP, _TP, _TP, _TP, _TP, _i32, _i32, _vp]),
    "mmif_fuse_attn_workspace": (_sz, [_i32, _i32]),
    "mmif_fuse_attn_fwd": (_i32, [_TP, _TP, _TP, _i32, _vp, _sz, _vp]),
    "mmif_fuse_attn_bwd": (_i32, [_TP, _TP, _TP, _TP, _TP, _i32, _i32, _vp, _sz, _vp]),
    "mmif_fuse_attn_bwd_cached": (_i32, [_TP, _TP, _TP, _TP, _TP, _i32, _i32, _vp, _sz, _vp]),
    "mmif_pairconv_fwd": (_i32, [_TP, _TP, _vp, _vp, _i32, _TP, _TP, _i32, _TP, _TP, _vp]),
    "mmif_pairconv_dgrad": (_i32, [_TP, _TP, _vp, _i32, _TP, _TP, _TP, _TP, _u64, _TP, _vp]),
    "mmif_pairconv_wgrad_workspace": (_sz, []),
    "mmif_pairconv_wgrad": (_i32, [_TP, _TP, _TP, _TP, _i32, _vp, _vp, _i32, _vp, _sz, _vp]),
    "mmif_pairconv_bwd": (_i32, [_TP, _TP, _vp, _i32, _TP, _TP, _TP, _TP, _u64, _TP, _vp, _vp, _i32, _vp, _sz, _vp]),
    "mmif_maxpool2x2_fwd": (_i32, [_TP, _TP, _vp]),
    "mmif_maxpool2x2_bwd": (_i32, [_TP, _TP, _TP, _i32, _vp]),
    "mmif_upsample2x_fwd": (_i32, [_TP, _TP, _vp]),
    "mmif_upsample2x_bwd": (_i32, [_TP, _TP, _i32, _vp]),
    "mmif_relu_mask": (_i32, [_TP, _TP, _vp]),
    "mmif_maxpool2x2_bwd_relu": (_i32, [_TP, _TP, _TP, _i32, _vp]),
    "mmif_upsample2x_bwd_relu": (_i32, [_TP, _TP, _i32, _TP, _vp]),
    "mmif_loss_workspace": (_sz, [_i32, _i32, _i32]),
    "mmif_ssim_loss": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _sz, _vp]),
    "mmif_ssim_loss_mode_workspace": (_sz, [_i32, _i32, _i32, _i32]),
    "mmif_ssim_loss_mode": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "mmif_ssim_terms": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "mmif_tv_loss_workspace": (_sz, []),
    "mmif_tv_loss": (_i32, [_vp, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "mmif_pixel_loss": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "mmif_grad_loss": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "mmif_fusion_loss_workspace": (_sz, [_i32, _i32, _i32]),
    "mmif_fusion_loss": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _f32, _i32, _i32, _f32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "mmif_patch_feed": (_i32, [_vp, _i64, _i32, _vp, _vp, _i32, _i32, _vp, _vp]),
    "mmif_clip_adam_workspace": (_sz, [_i64]),
    "mmif_clip_adam_step": (_i32, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _i32, _f32, _f32, _vp, _vp, _sz, _vp]),
    "mmif_debug_set_trace": (None, [_vp]),
    "mmif_debug_set_conv_dma": (None, [_i32]),
    "mmif_debug_set_conv1x1_stream": (None, [_i32]),
    "mmif_debug_set_bwd_pair_dma": (None, [_i32]),
    "mmif_debug_set_thin_wide": (None, [_i32]),
    "mmif_debug_set_wgrad_dma_blocks": (None, [_i32]),
    "mmif_debug_set_ragged": (None, [_i32]),
    "mmif_debug_set_enc_stream2": (None, [_i32]),
    "mmif_probe_tr16": (_i32, [_vp, _vp, _vp]),
    "mmif_probe_mfma": (_i32, [_vp, _vp, _vp, _vp]),
    "mmif_probe_dma": (_i32, [_vp, _vp, _vp, _vp, _vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header / library mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


def check(rc, what=""):
    if rc != 0:
        raise MmifError(f"{what or 'mmif'} failed (code {rc}): {lib.mmif_last_error().decode()}")


def version():
    return lib.mmif_version().decode()

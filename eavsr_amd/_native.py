"""ctypes binding of libeavsr_hip.so (the C ABI declared in include/eavsr_hip.h).

There is no CPU path and no PyTorch fallback: if the shared object is missing or an entry
point is absent, the first use raises.  Build it with ``python -m eavsr_amd.build`` (or
``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# EAVSR_LIB_PATH: A/B builds of the same ABI (the lab flavour eavsr_amd/lib/libeavsr_lab.so among them; EAVSR_BUILD_LAB=1 picks that one)
LIB_PATH = os.environ.get("EAVSR_LIB_PATH") or os.path.join(
    _HERE, "lib", "libeavsr_lab.so" if os.environ.get("EAVSR_BUILD_LAB", "0") == "1" else "libeavsr_hip.so")

ABI_VERSION = 30

p_f32 = C.c_void_p  # device pointers travel as integers
i32 = C.c_int32
i64 = C.c_int64
f32 = C.c_float
vp = C.c_void_p


class ConvDesc(C.Structure):
    """struct eavsr_conv2d_desc"""
    _fields_ = [
        ("src", vp * 5),
        ("src_c", i32 * 5),
        ("n_src", i32),
        ("ksize", i32),
        ("weight_packed", vp),
        ("bias", vp),
        ("residual", vp),
        ("out", vp),
        ("chan_partial", vp),
        ("n", i32), ("h", i32), ("w", i32), ("cin", i32), ("cout", i32),
        ("act", i32),
        ("slope", f32),
        ("ca_scale", vp),
        ("ca_x", vp),
        ("ca_out", vp),
        ("out_shuffle", i32),
        ("res_scale", vp),
        ("border_pieces", vp),
        ("border_stride", i32),
        ("sum_mul", vp),
    ]


# name -> (restype, argtypes); every symbol include/eavsr_hip.h declares
SIGNATURES = {
    "eavsr_abi_version": (C.c_int, []),
    "eavsr_version": (C.c_char_p, []),
    "eavsr_conv2d_desc_size": (C.c_size_t, []),
    "eavsr_lab_build": (C.c_int, []),
    "eavsr_last_error": (C.c_char_p, []),
    "eavsr_selftest_mfma_f32": (C.c_int, [vp, vp]),
    "eavsr_flow_warp_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "eavsr_flow_warp_pair_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "eavsr_dcnv2_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "eavsr_dcnv2_generic_f32": (C.c_int, [vp] * 6 + [i32] * 15 + [vp]),
    "eavsr_dcn_weight_x9_bytes": (C.c_int64, [i32, i32]),
    "eavsr_pack_dcn_weight_x9": (C.c_int, [vp, vp, i32, i32, vp]),
    "eavsr_nchw_to_il8_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "eavsr_dcnv2_il_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "eavsr_dcn_weight_il2_bytes": (C.c_int64, [i32, i32]),
    "eavsr_pack_dcn_weight_il2": (C.c_int, [vp, vp, i32, i32, vp]),
    "eavsr_dcnv2_il2_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "eavsr_dcn_il16_weight_bytes": (C.c_int64, [i32, i32]),
    "eavsr_pack_dcn_il16_weight": (C.c_int, [vp, vp, i32, i32, i32, vp]),
    "eavsr_nchw_to_il8_h16": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "eavsr_dcnv2_il16": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "eavsr_wino4_schedule": (C.c_int, []),
    "eavsr_conv3x3_wino4_border_pieces": (C.c_int, [i32, i32, C.POINTER(i32), C.POINTER(i32)]),
    "eavsr_ca_scale_pre_pieces": (C.c_int, [vp, vp, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "eavsr_conv2d_f32": (C.c_int, [C.POINTER(ConvDesc), vp]),
    "eavsr_wino4_weight_elems": (C.c_int64, [i32, i32]),
    "eavsr_pack_conv_weight_wino4": (C.c_int, [vp, vp, i32, i32, vp]),
    "eavsr_conv3x3_wino4_tiles": (i32, [i32, i32]),
    "eavsr_conv3x3_wino4_f32": (C.c_int, [vp, vp, vp]),
    "eavsr_wgrad3_mode": (C.c_int, []),
    "eavsr_conv3x3_x6s_tiles": (i32, [i32, i32]),
    "eavsr_conv3x3_f32x6s": (C.c_int, [vp, vp, vp]),
    "eavsr_pack_conv_weight_wino5x5": (C.c_int, [vp, vp, i32, i32, vp]),
    "eavsr_conv5x5_wino_tiles": (i32, [i32, i32]),
    "eavsr_conv5x5_wino_f32": (C.c_int, [vp, vp, vp]),
    "eavsr_conv2d_ck": (i32, [i32]),
    "eavsr_conv2d_tile_rows": (i32, [i32, i32, i32, i32]),
    "eavsr_conv2d_tiles": (i32, [i32, i32, i32, i32]),
    "eavsr_packed_weight_elems": (i64, [i32, i32, i32]),
    "eavsr_pack_conv_weight_f32": (C.c_int, [vp, vp, i32, i32, i32, vp]),
    "eavsr_conv3x3_smallco_f32": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp]),
    "eavsr_smallco_packed_elems": (C.c_int64, [i32, i32]),
    "eavsr_pack_smallco_weight": (C.c_int, [vp, vp, i32, i32, vp]),
    "eavsr_conv3x3_smallco_lite_f32": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp]),
    "eavsr_conv_weight_x6_bytes": (C.c_size_t, [i32, i32, i32]),
    "eavsr_pack_conv_weight_x6": (C.c_int, [vp, vp, i32, i32, i32, vp]),
    "eavsr_pack_conv_weight_x6_dgrad": (C.c_int, [vp, vp, i32, i32, i32, vp]),
    "eavsr_pack_conv_weight_x6_multi": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "eavsr_conv_f32x6": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, i32, vp]),
    "eavsr_ca_scale_f32": (C.c_int, [vp, i32, i32, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "eavsr_ca_scale_mean_f32": (C.c_int, [vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "eavsr_scale_residual_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "eavsr_ca_tail_f32": (C.c_int, [vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "eavsr_ca_tail_stats_f32": (C.c_int, [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "eavsr_adapt_frontend_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "eavsr_affine_offsets_f32": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "eavsr_resize_bilinear_ac_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp]),
    "eavsr_pyramid_f32": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "eavsr_add_f32": (C.c_int, [vp, vp, vp, vp, i64, vp]),
    "eavsr_normalize_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "eavsr_avg_pool2_f32": (C.c_int, [vp, vp, i32, i32, i32, vp]),
    "eavsr_resize_bilinear_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, f32, vp]),
    "eavsr_concat3_f32": (C.c_int, [vp, i32, vp, i32, vp, i32, vp, i32, i32, vp]),
    # backward entry points
    "eavsr_act_bwd_f32": (C.c_int, [vp, vp, vp, i64, i32, f32, vp]),
    "eavsr_plane_sum_f32": (C.c_int, [vp, vp, vp, i32, i32, f32, vp]),
    "eavsr_channel_sum_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "eavsr_channel_sum_multi_f32": (C.c_int, [vp, i32, vp, i32, i32, i32, i32, vp]),
    "eavsr_scale_residual_bwd_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "eavsr_rcab_tail_bwd_f32": (C.c_int, [vp] * 13 + [i32] * 7 + [vp]),
    "eavsr_ca_mlp_bwd_f32": (C.c_int, [vp] * 11 + [i32, i32, i32, vp]),
    "eavsr_flow_warp_bwd_f32": (C.c_int, [vp] * 6 + [i32, i32, i32, i32, vp]),
    "eavsr_resize_bilinear_ac_bwd_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, i32, f32, vp]),
    "eavsr_pyramid_bwd_f32": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "eavsr_affine_offsets_bwd_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "eavsr_conv_wgrad_blocks": (i32, [i32, i32, i32, i32]),
    "eavsr_conv_wgrad_f32": (C.c_int, [vp, vp, vp, vp] + [i32] * 11 + [vp]),
    "eavsr_conv_wgrad_multi_f32": (C.c_int, [vp, vp, i32, vp, vp] + [i32] * 11 + [vp]),
    "eavsr_conv_wgrad_span_f32": (C.c_int, [vp, vp, vp, vp] + [i32] * 9 + [vp]),
    "eavsr_conv_wgrad_bias_multi_f32": (C.c_int, [vp, vp, i32, vp, vp, vp] + [i32] * 11 + [vp]),
    "eavsr_dcnv2_bwd_grid": (i32, [i32, i32, i32]),
    "eavsr_dcnv2_bwd_workspace_floats": (C.c_int64, [i32, i32, i32]),
    "eavsr_dcnv2_bwd_f32": (C.c_int, [vp] * 10 + [i32] * 7 + [vp]),
    "eavsr_il8_to_nchw_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "eavsr_dcnv2_im2col_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "eavsr_dcnv2_col2im_f32": (C.c_int, [vp] * 7 + [i32, i32, i32, i32, i32, vp]),
    "eavsr_gconv3x3_fwd_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp]),
    "eavsr_gconv3x3_bwd_f32": (C.c_int, [vp] * 6 + [i32, i32, i32, i32, i32, vp]),
    "eavsr_gconv3x3_bwd_acc_f32": (C.c_int, [vp] * 6 + [i32, i32, i32, i32, i32, i32, vp]),
    # 16-bit backbone
    "eavsr_conv_h16_tiles": (i32, [i32, i32]),
    "eavsr_pack_conv3x3_c64_h16": (C.c_int, [vp, vp, i32, vp]),
    "eavsr_conv5x5_c64_h16_weight_bytes": (C.c_int64, []),
    "eavsr_pack_conv5x5_c64_h16": (C.c_int, [vp, vp, i32, i32, vp]),
    "eavsr_conv5x5_c64_h16": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "eavsr_conv_h16_partial_rows": (i32, [i32, i32, i32]),
    "eavsr_conv3x3_c64_h16": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "eavsr_conv3x3_c64_h16_res": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "eavsr_ca_scale_pre_f32": (C.c_int, [vp] * 11 + [i32, i32, i32, i32, vp]),
    "eavsr_ca_scale_pre_ws_floats": (C.c_int64, [i32]),
    "eavsr_ca_scale_pre_h16": (C.c_int, [vp] * 11 + [i32, i32, i32, i32, i32, vp]),
    "eavsr_conv_h16_border_pieces": (C.c_int, [i32, i32, C.POINTER(i32), C.POINTER(i32)]),
    "eavsr_conv3x3_c64_h16_b": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "eavsr_conv3x3_c64_h16_act": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, i32, vp]),
    "eavsr_conv3x3_c64to3_h16": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "eavsr_conv3x3_h16g_weight_bytes": (C.c_int64, [i32, i32]),
    "eavsr_pack_conv3x3_h16g": (C.c_int, [vp, vp, i32, i32, i32, vp]),
    "eavsr_conv3x3_h16g_f32": (C.c_int, [vp, i32, vp]),
    "eavsr_conv_weight_h16x1_bytes": (C.c_size_t, [i32, i32, i32]),
    "eavsr_pack_conv_weight_h16x1": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "eavsr_conv_h16x1": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, i32, vp]),
    "eavsr_nchw_f32_to_nhwc_h16": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "eavsr_nhwc_h16_to_nchw_f32": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "eavsr_scale_residual_h16": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
}

# Entry points of the LAB build only (`python -m eavsr_amd.build --lab`; the header's EXPERIMENTAL section): bound when the
# library exports them, absent from a default build (ops raises LabBuildRequired for the modes that need them).
LAB_SIGNATURES = {
    "eavsr_conv3x3_f32x9": (C.c_int, [vp, vp, vp]),
    "eavsr_dcnv2_f32x9": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "eavsr_dcnv2_ws_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "eavsr_wino_weight_elems": (C.c_int64, [i32, i32]),
    "eavsr_pack_conv_weight_wino": (C.c_int, [vp, vp, i32, i32, vp]),
    "eavsr_conv3x3_wino_tiles": (i32, [i32, i32]),
    "eavsr_conv3x3_wino_f32": (C.c_int, [vp, vp, vp]),
    "eavsr_flow_level_f32": (C.c_int, [vp] * 11 + [i32, i32, i32, i32, vp]),
    "eavsr_rcab_h16_partial_rows": (i32, [i32, i32, i32]),
    "eavsr_rcab_convs_h16": (C.c_int, [vp] * 7 + [i32] * 4 + [vp]),
}

_lock = threading.Lock()
_lib = None


class NativeLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library with typed entry points."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -m eavsr_amd.build`). eavsr_amd has no CPU or PyTorch fallback.")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # e.g. libamdhip64 not found
            raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        if lib.eavsr_abi_version() != ABI_VERSION:
            raise NativeLibraryError(
                f"ABI mismatch: library {lib.eavsr_abi_version()} vs binding {ABI_VERSION}; rebuild")
        if lib.eavsr_conv2d_desc_size() != C.sizeof(ConvDesc):      # a descriptor that grew without an ABI bump
            raise NativeLibraryError(
                f"struct eavsr_conv2d_desc: library {lib.eavsr_conv2d_desc_size()} bytes vs binding {C.sizeof(ConvDesc)}; rebuild")
        if lib.eavsr_lab_build():
            for name, (res, args) in LAB_SIGNATURES.items():
                try:
                    fn = getattr(lib, name)
                except AttributeError as e:
                    raise NativeLibraryError(f"{LIB_PATH} says it is a lab build but does not export {name}") from e
                fn.restype = res
                fn.argtypes = args
        _lib = lib
    return _lib


def lab_build() -> bool:
    """True when the loaded library is the lab build (the retired schedules are there)"""
    return bool(load().eavsr_lab_build())


def check(code: int, what: str):
    if code != 0:
        msg = load().eavsr_last_error().decode(errors="replace")
        kind = "argument error" if code < 0 else "hipError"
        raise RuntimeError(f"{what} failed ({kind} {code}): {msg}")

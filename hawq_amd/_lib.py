"""ctypes binding of libhawq_mi355.so (C ABI: include/hawq_mi355.h).

There is NO fallback: if the shared library is missing or an entry point is absent, importing
the product path raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C hawq_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HAWQ_LIB: another build of the same ABI (A/B measurements of kernel variants); default = the in-tree build
LIB_PATH = os.environ.get("HAWQ_LIB") or os.path.join(_HERE, "lib", "libhawq_mi355.so")

EPI_RAW, EPI_REQUANT, EPI_RESIDUAL, EPI_DEQUANT = 0, 1, 2, 3

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class ConvArgs(C.Structure):
    """struct hawq_conv_args (include/hawq_mi355.h)."""
    _fields_ = [
        ("in_", vp), ("wgt", vp), ("bias", vp),
        ("N", i32), ("H", i32), ("W", i32), ("Cin", i32), ("Cout", i32), ("KH", i32), ("KW", i32),
        ("stride", i32), ("pad", i32), ("in_bits", i32), ("w_bits", i32),
        ("in2", vp), ("wgt2", vp), ("bias2", vp),
        ("H2", i32), ("W2", i32), ("Cin2", i32), ("stride2", i32), ("in2_bits", i32), ("w2_bits", i32),
        ("epilogue", i32), ("relu", i32),
        ("m", vp), ("e", vp), ("m_id", vp), ("e_id", vp),
        ("m_id_scalar", i32), ("e_id_scalar", i32),
        ("res_in", vp), ("res_in_bits", i32),
        ("res_out", vp), ("res_out_bits", i32),
        ("out_q", vp), ("out_bits", i32), ("q_lo", i32), ("q_hi", i32), ("mq", i32), ("eq", i32),
        ("out_acc", vp), ("out_f32", vp), ("fscale", vp), ("ldo", i32), ("n_valid", i32),
        ("flags", vp), ("tile", i32), ("ctab", vp), ("ctab_id", vp), ("fast_tables", i32),
        ("in_planar", i32), ("out_planar", i32),
        ("res_no_relu", i32), ("res_clamp16", i32),
        ("in_pitch", i32), ("out_pitch", i32),
        ("wgt_band", vp), ("wgt_k128", vp), ("wgt2_k128", vp),
    ]


class ExpandReduceArgs(C.Structure):
    """struct hawq_expand_reduce_args (include/hawq_mi355.h)."""
    _fields_ = [("expand", ConvArgs), ("reduce", ConvArgs), ("tile", i32)]


class BottleneckArgs(C.Structure):
    """struct hawq_bottleneck_args (include/hawq_mi355.h): one launch per MobileNetV2 linear-bottleneck unit."""
    _fields_ = [("expand", ConvArgs), ("project", ConvArgs), ("dw_wgt9c", vp), ("dw_ctab", vp),
                ("dw_stride", i32), ("dw_q_lo", i32), ("dw_q_hi", i32), ("dw_fast_tables", i32), ("c_mid", i32), ("tile", i32)]


# name -> (argtypes); every function returns int except hawq_last_error
SIGNATURES = {
    "hawq_abi_version": [],
    "hawq_device_ok": [],
    "hawq_conv2d": [C.POINTER(ConvArgs), vp],
    "hawq_conv2d_num_tiles": [],
    "hawq_conv2d_num_band_tiles": [],
    "hawq_conv2d_band_tile": [C.POINTER(ConvArgs)],
    "hawq_conv2d_num_band2_tiles": [],
    "hawq_conv2d_band2_tile": [C.POINTER(ConvArgs)],
    "hawq_pack_w3x3_band": [vp, vp, i32, i32],
    "hawq_conv2d_num_gemm2_tiles": [],
    "hawq_conv2d_gemm2_first": [],
    "hawq_pack_w1x1_k128": [vp, vp, i32, i32],
    "hawq_fc_dequant_ok": [i32, i32, i32],
    "hawq_fc_dequant": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "hawq_conv_expand_reduce": [C.POINTER(ExpandReduceArgs), vp],
    "hawq_conv_expand_reduce_variants": [C.POINTER(ExpandReduceArgs)],
    "hawq_linear_bottleneck": [C.POINTER(BottleneckArgs), vp],
    "hawq_linear_bottleneck_ok": [C.POINTER(BottleneckArgs)],
    "hawq_stem3x3s2": [vp, vp, vp, i32, i32, f32, i32, i32, C.POINTER(ConvArgs), vp],
    "hawq_stem3x3s2_ok": [vp, vp, vp, i32, i32, C.POINTER(ConvArgs)],
    "hawq_quantize_input": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, i32, i32, vp],
    "hawq_stem_conv7": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp],
    "hawq_stem_fused": [vp, i32, i32, i32, i32, f32, i32, i32, vp, vp, vp, vp, i32, i32, vp, vp, i32, i32, i32, i32, i32,
                        i32, vp],
    "hawq_stem_fused_u8": [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "hawq_maxpool3s2_requant": [vp, i32, i32, i32, i32, vp, vp, i32, i32, i32, i32, i32, vp],
    "hawq_requant_residual": [vp, i32, i64, vp, i32, i32, i32, i32, i32, vp],
    "hawq_avgpool_requant": [vp, i32, i32, i32, i32, vp, vp, i32, i32, i32, i32, vp],
    "hawq_f32_nchw_to_q_nhwc": [vp, vp, i32, i32, i32, i32, i32, i32, f32, vp],
    "hawq_acc_nhwc_to_f32_nchw": [vp, vp, i32, i32, i32, i32, i32, vp, vp],
    "hawq_fixedpoint_f32": [vp, vp, i32, i32, i32, f32, vp, vp, vp, i32, vp, f32, vp, vp, vp, i32, f32, i32, i32,
                            i32, vp],
    "hawq_fakequant_f32": [vp, vp, i64, f32, f32, i32, i32, vp],
    "hawq_avgpool_f32": [vp, vp, i32, i32, f32, vp],
    "hawq_conv2d_grouped": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp],
    "hawq_depthwise3x3": [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp],
    "hawq_depthwise3x3_requant": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp],
    "hawq_depthwise3x3_requant_fast": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp],
    "hawq_quantize_im2col3x3s2": [vp, vp, i32, i32, i32, i32, f32, i32, i32, vp],
    "hawq_quantize_im2col3x3s2_u8": [vp, vp, vp, i32, i32, i32, i32, vp],
    "hawq_resample_u8": [vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp],
    "hawq_minmax_f32": [vp, i64, vp, vp, vp],
    "hawq_kthvalue_f32": [vp, i64, i64, i32, vp, vp, vp],
    "hawq_graph_begin": [vp],
    "hawq_graph_end": [vp, C.POINTER(vp)],
    "hawq_graph_launch": [vp, vp],
    "hawq_graph_destroy": [vp],
    "hawq_event_create": [C.POINTER(vp)],
    "hawq_event_record": [vp, vp],
    "hawq_event_elapsed_ms": [vp, vp, C.POINTER(f32)],
    "hawq_event_destroy": [vp],
}

_lib = None


class HawqLibraryError(RuntimeError):
    pass


def library_path() -> str:
    """Path of the shared library this process loads (the in-tree build unless HAWQ_LIB names another one)."""
    return LIB_PATH


def load():
    """Load the HIP library or raise - never degrade to a CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HawqLibraryError(
            f"{LIB_PATH} not found: hawq_amd has no CPU fallback. Build the gfx950 library first "
            "(`python -c 'import __graft_entry__ as g; g.build()'` or `make -C hawq_amd/csrc`).")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise HawqLibraryError(f"{LIB_PATH} does not export {name}") from exc
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.hawq_last_error.restype = C.c_char_p
    lib.hawq_last_error.argtypes = []
    if lib.hawq_abi_version() != 5:
        raise HawqLibraryError("libhawq_mi355.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise RuntimeError("libhawq_mi355: " + load().hawq_last_error().decode())


def call(name: str, *args):
    check(getattr(load(), name)(*args))

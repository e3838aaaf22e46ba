"""ctypes binding of ``liberd_hip.so`` (the C ABI declared in ``include/erd_hip.h``).

There is NO fallback: if the library is missing or a call fails, the product
raises.  (The CPU restatement under ``oracle/`` is test infrastructure and is
never imported from here.)"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ERD_HIP_LIB") or os.path.join(_HERE, "lib", "liberd_hip.so")   # override: A/B builds

ERD_MAX_SEG = 5
ERD_MAX_TAPS = 9

c_float_p = C.c_void_p  # raw device pointers travel as integers
i64 = C.c_int64
i32 = C.c_int
f32 = C.c_float


class ConvSeg(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("res", C.c_void_p), ("alpha", C.c_void_p),
                ("mask", C.c_void_p), ("N", i32), ("IH", i32), ("IW", i32), ("GH", i32), ("GW", i32), ("OH", i32), ("OW", i32),
                ("in_nstride", i64), ("out_nstride", i64), ("res_nstride", i64),
                ("tap0", i32), ("ntaps", i32), ("oy", i32), ("ox", i32)]


class ConvDesc(C.Structure):
    _fields_ = [("nseg", i32), ("seg", ConvSeg * ERD_MAX_SEG), ("w", C.c_void_p),
                ("Cin", i32), ("Cout", i32), ("wrow", i32), ("ntaps", i32),
                ("dy", i32 * ERD_MAX_TAPS), ("dx", i32 * ERD_MAX_TAPS), ("wk", i32 * ERD_MAX_TAPS),
                ("in_stride", i32), ("out_stride", i32), ("oy", i32), ("ox", i32),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("relu", i32), ("colsum", C.c_void_p),
                ("sk_ws", C.c_void_p), ("sk_ws_bytes", C.c_size_t), ("w_bf16", C.c_void_p), ("colsum_copies", i32),
                ("in_bf16", i32), ("out_bf16", i32), ("w_x3", C.c_void_p)]


class WgradSeg(C.Structure):
    _fields_ = [("x_off", i64), ("dz_off", i64),
                ("N", i32), ("IH", i32), ("IW", i32), ("GH", i32), ("GW", i32), ("OH", i32), ("OW", i32),
                ("x_nstride", i64), ("dz_nstride", i64)]


class WgradDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("dz", C.c_void_p), ("x_elems", i64), ("dz_elems", i64),
                ("nseg", i32), ("seg", WgradSeg * ERD_MAX_SEG),
                ("Cin", i32), ("Cout", i32), ("ntaps", i32),
                ("dy", i32 * ERD_MAX_TAPS), ("dx", i32 * ERD_MAX_TAPS),
                ("in_stride", i32), ("out_stride", i32), ("oy", i32), ("ox", i32),
                ("part", C.c_void_p), ("nsplit", i32), ("bf16_multiplicands", i32),
                ("x_bf16", i32), ("dz_bf16", i32), ("limbs3", i32)]


class BnFoldItem(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean", C.c_void_p), ("var", C.c_void_p),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("n", i32), ("eps", f32)]


class WeightPrepItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("rowscale", C.c_void_p), ("dst", C.c_void_p),
                ("Cout", i32), ("ntaps", i32), ("Cin", i32), ("flip", i32), ("kind", i32), ("block0", i32)]


class Levels(C.Structure):
    _fields_ = [("nseg", i32), ("off", i64 * ERD_MAX_SEG), ("cnt", i64 * ERD_MAX_SEG)]


P = C.c_void_p
_SIGNATURES = {
    # name: argtypes (restype is always int unless noted)
    "erd_set_cu_reserve": [i32],
    "erd_usable_cus": [],
    "erd_conv_igemm": [C.POINTER(ConvDesc), P],
    "erd_conv_igemm_ws_bytes": [i32],
    "erd_to_bf16": [P, P, i64, P],
    "erd_split3": [P, P, i64, P],
    "erd_conv_thin_enable": [i32],
    "erd_conv_thin_ok": [C.POINTER(ConvDesc)],
    "erd_weight_transpose_x3": [P, P, P, i32, i32, i32, i32, P],
    "erd_wino_weights_elems": [i32, i32],
    "erd_wino_weights": [P, P, i32, i32, i32, P],
    "erd_wino_conv3x3": [P, i32, P, i32, i32, P, P, i32, P, i32, P, P],
    "erd_wino_weights_x3_elems": [i32, i32],
    "erd_wino_weights_x3": [P, P, i32, i32, i32, P],
    "erd_wino_conv3x3_x3": [P, i32, P, i32, i32, P, P, i32, P, i32, P, P],
    "erd_wino_x3_couts_per_item": [P, i32, i32],
    "erd_wino_conv3x3_x3_gn": [P, i32, P, i32, i32, P, P, i32, P, P, C.c_size_t, P],
    "erd_wino_gn_finalize": [P, i32, i32, P, P, f32, P],
    "erd_wino_x3_gn_ws_bytes": [P, i32, i32],
    "erd_conv_wgrad": [C.POINTER(WgradDesc), P],
    "erd_wgrad_row3_slices": [C.POINTER(WgradDesc)],
    "erd_wgrad_reduce": [P, i32, i32, i32, P, P, P, i32, P, P],
    "erd_weight_transpose": [P, P, P, i32, i32, i32, i32, P],
    "erd_weight_transpose_bf16": [P, P, P, i32, i32, i32, i32, P],
    "erd_stem_conv7x7_bn_relu": [P, P, P, P, P, i32, i32, i32, P],
    "erd_maxpool3x3s2": [P, P, i32, i32, i32, i32, i32, P],
    "erd_bn_fold": [P, P, P, P, f32, P, P, i64, P],
    "erd_bn_fold_batch": [P, i32, i32, P],
    "erd_weight_prep_blocks": [i32, i32, i32, i32],
    "erd_weight_prep_batch": [P, i32, i32, P],
    "erd_relu_bwd_colsum": [P, P, P, i64, i32, i64, i64, P, i32, i32, P],
    "erd_bn_dgamma": [P, P, i32, P, P, f32, P, P, i32, i32, P],
    "erd_gn_relu_fwd": [P, P, P, P, P, P, i32, i64, i32, i32, C.POINTER(Levels), f32, i32, P],
    "erd_gn_relu_apply": [P, P, P, P, P, i32, i64, i32, i32, C.POINTER(Levels), i32, P],
    "erd_gn_relu_bwd": [P, P, P, P, P, P, P, P, P, i32, i64, i32, i32, C.POINTER(Levels), i32, P],
    "erd_upsample2x_add": [P, P, i32, i32, i32, i32, i32, i32, i64, i64, i32, P],
    "erd_upsample2x_add_bwd": [P, P, i32, i32, i32, i32, i32, i32, i64, i64, i32, P],
    "erd_colsum": [P, i64, i32, P, i32, i32, P],
    "erd_level_scale": [P, P, P, i32, i64, i32, C.POINTER(Levels), P],
    "erd_level_scale_bwd": [P, P, P, P, P, i32, i64, i32, C.POINTER(Levels), P],
    "erd_sgd_momentum": [P, P, P, i64, f32, f32, f32, f32, i32, P],
    "erd_ers_select": [P, P, i32, i64, i32, i32, P, P, P, P, P, P, P, P],
    "erd_grid_anchors": [P, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i32, i32, P],
    "erd_atss_assign": [P, P, C.POINTER(i64), i32, i64, P, P, P, i32, i32, i32, i32, P, P, P, P, P, P],
    "erd_gfl_losses_fwd": [P, P, P, P, P, P, C.POINTER(i64), C.POINTER(i32), i32, i32, i64, i32, i32, P, P, P, P],
    "erd_gfl_losses_bwd": [P, P, P, P, P, P, C.POINTER(i64), C.POINTER(i32), i32, i32, i64, i32, i32, P, P, P, P,
                           P, P],
    "erd_l2_distill": [P, P, P, P, i32, i64, i32, i32, i32, P, P],
    "erd_l2_distill_bwd": [P, P, P, P, P, i32, i64, i32, i32, i32, P, P],
    "erd_distill_nms": [P, P, P, P, P, i32, i64, i32, f32, P, P, P, C.c_size_t, P],
    "erd_kd_kl": [P, P, P, P, i32, i64, i32, i32, f32, P, P],
    "erd_kd_kl_bwd": [P, P, P, P, P, i32, i64, i32, i32, f32, P, P],
    "erd_loss_avg": [P, i32, P, i32, P, P],
    "erd_loss_finalize": [P, P, P, P, P, i32, i32, i32, f32, f32, f32, f32, f32, P, P, P, P],
    "erd_preprocess_image": [P, i32, i32, i32, P, i32, i32, P, P, i32, f32, P],
    "erd_resize_normalize": [P, i32, i32, P, P, P, P, i32, i32, P, i32, i32, P, P, i32, i32, f32, P],
    "erd_predict_ws_bytes": [i32, i32, i32],
    "erd_predict_topk": [P, P, P, i32, i64, i32, P, P, P, f32, i32, P, P, P, P, P, C.c_size_t, P],
    "erd_predict_nms": [P, P, P, P, i32, i32, P, f32, f32, i32, P, P, P, P, C.c_size_t, P],
    "erd_qfl_rows": [P, P, P, i64, i32, P, P],
    "erd_qfl_bwd": [P, P, P, P, i64, i32, P, P],
    "erd_dfl": [P, P, P, i64, i32, P, P, P],
    "erd_kd_kl_rows": [P, P, P, i64, i32, f32, P, P, P],
    "erd_giou": [P, P, P, i64, f32, P, P, P],
    "erd_bbox_overlaps": [P, P, i64, i64, i32, i32, f32, P, P],
    "erd_integral": [P, P, i64, i32, P, P, P],
    "erd_distance2bbox": [P, P, P, i64, f32, f32, P, P, P],
    "erd_bbox2distance": [P, P, i64, f32, f32, P, P],
    "erd_weighted_sum": [P, P, i64, C.c_double, P, P],
    "erd_loss_coef": [P, P, i64, f32, P, P],
    "erd_rows_mul": [P, P, i64, f32, P, P],
    "erd_atss_result": [P, P, i64, P, P, P, P],
}

EXPORTS = ["erd_abi_version", "erd_probe_build", "erd_last_error", "erd_csrc_sha"] + sorted(_SIGNATURES)

_lib = None


class ErdHipError(RuntimeError):
    pass


def load():
    """dlopen liberd_hip.so (once).  Raises ErdHipError if it is not built -- no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ErdHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C erd_amd/csrc`).  erd_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.erd_abi_version.restype = C.c_int
    lib.erd_last_error.restype = C.c_char_p
    lib.erd_csrc_sha.restype = C.c_char_p
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_size_t if name.endswith(("_ws_bytes", "_elems")) else C.c_int
    if lib.erd_abi_version() != 6:
        raise ErdHipError("liberd_hip.so ABI version mismatch")
    lib.erd_probe_build.restype = C.c_int
    if lib.erd_probe_build() and not os.environ.get("ERD_HIP_LIB"):
        raise ErdHipError(f"{LIB_PATH} is a PROBE build (a kernel was compiled with a timing / accuracy / trace macro of "
                          "csrc/erd_probes.h): rebuild with `make -C erd_amd/csrc`, or select it explicitly with ERD_HIP_LIB")
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise ErdHipError(f"{name} failed (rc={rc}): {lib.erd_last_error().decode(errors='replace')}")

"""ctypes binding of ``libcellseg_hip.so`` (the C ABI declared in ``include/cellseg_hip.h``).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the shared object is
missing, or a kernel is asked to run on a non-GPU tensor, this module raises -- loudly.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# CELLSEG_LIB_FLAVOUR=ab: the A/B build (`make AB=1`: launch-rule knobs read from CELLSEG_* and cs_set_igemm_path compiled in).
# Only the forced-mode tests and the tools/ sweeps set it, in child processes; the product loads the production library.
FLAVOUR = os.environ.get("CELLSEG_LIB_FLAVOUR", "")
if FLAVOUR not in ("", "ab", "dbg"):
    raise RuntimeError(f"CELLSEG_LIB_FLAVOUR={FLAVOUR!r}: expected unset, 'ab' or 'dbg' (`make DEBUG=1`: A/B switches + s_memtime stamps)")
LIB_PATH = os.path.join(_HERE, f"libcellseg_hip_{FLAVOUR}.so" if FLAVOUR else "libcellseg_hip.so")

CS_F32, CS_BF16 = 0, 1
CS_ACT_NONE, CS_ACT_RELU, CS_ACT_SILU, CS_ACT_SIGMOID = 0, 1, 2, 3
CS_BN_BWD_OWN_RELU, CS_BN_BWD_FROZEN = 0x100, 0x200       # flags OR-ed into `act` of cs_bn_bwd_reduce / cs_bn_bwd_apply


class CsConvGeom(Structure):
    _fields_ = [(n, c_int32) for n in ("N", "H", "W", "C", "K", "R", "S", "stride", "pad", "P", "Q", "groups")]


class CsStageDesc(Structure):
    """One layer of cs_stage_conv_bn_multi (include/cellseg_hip.h)."""
    _fields_ = ([(n, c_void_p) for n in ("w", "gamma", "beta", "mean", "var", "conv_bias", "w_khwc", "w_chwk", "scale", "shift", "rstd")]
                + [("eps", c_float)] + [(n, c_int32) for n in ("K", "Cin", "R", "S", "Cp", "Kp", "block0", "fwd_packed", "bwd_packed")])


class CellsegLibraryMissing(RuntimeError):
    pass


_P = c_void_p
# entry points that exist in the A/B flavour only (include/cellseg_hip.h: #ifdef CS_AB_SWITCHES)
_AB_SIGNATURES = {
    "cs_set_igemm_path": (c_int, [c_int]),
}
_SIGNATURES = {
    # name: (restype, argtypes)
    "cs_abi_version": (c_int, []),
    "cs_last_error": (c_char_p, []),
    "cs_last_conv_variant": (c_char_p, []),
    "cs_nchw_to_nhwc": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cs_nhwc_to_nchw": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "cs_bn_fold": (c_int, [_P, _P, _P, _P, c_float, _P, _P, _P, _P, c_int, _P]),
    "cs_stage_conv_bn_blocks": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "cs_stage_conv_bn_multi": (c_int, [_P, c_int, c_int, c_int, _P]),
    "cs_stage_conv_bn": (c_int, [_P, _P, _P, _P, _P, c_float, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "cs_weight_prep": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "cs_conv2d_fwd": (c_int, [POINTER(CsConvGeom), c_int, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "cs_conv2d_stats_workspace": (c_size_t, [c_longlong, c_int]),
    "cs_igemm_tile": (c_int, [c_longlong, c_int]),
    "cs_conv2d_dgrad": (c_int, [POINTER(CsConvGeom), c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "cs_conv2d_fwd_bits": (c_int, [POINTER(CsConvGeom), c_int, _P, _P, _P, _P, _P, c_int, _P, _P, _P]),
    "cs_conv2d_dgrad_bits": (c_int, [POINTER(CsConvGeom), c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "cs_conv2d_packed_supported": (c_int, [POINTER(CsConvGeom), c_int]),
    "cs_conv2d_packed_weight_bytes": (c_size_t, [POINTER(CsConvGeom), c_int]),
    "cs_pack_conv_weights": (c_int, [POINTER(CsConvGeom), c_int, _P, _P, _P]),
    "cs_conv2d_packed_partial_rows": (c_int, [POINTER(CsConvGeom), c_int]),
    "cs_conv2d_fwd_packed": (c_int, [POINTER(CsConvGeom), _P, _P, _P, _P, c_int, _P, _P, _P]),
    "cs_conv2d_dgrad_packed": (c_int, [POINTER(CsConvGeom), _P, _P, _P, c_int, _P, _P, _P, _P]),
    "cs_conv2d_wgrad": (c_int, [POINTER(CsConvGeom), c_int, _P, _P, _P, c_int, _P]),
    "cs_wgrad_finalize": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P]),
    "cs_conv2d_wgrad_splits": (c_int, [POINTER(CsConvGeom), c_int]),
    "cs_weight_prep_grouped": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "cs_conv2d_wgrad_batched_splits": (c_int, [POINTER(CsConvGeom), c_int, c_int]),
    "cs_conv2d_wgrad2_supported": (c_int, [POINTER(CsConvGeom), c_int]),
    "cs_conv2d_wgrad_batched": (c_int, [POINTER(CsConvGeom), c_int, _P, _P, _P, c_int, c_int, _P]),
    "cs_wgrad_finalize_batched": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cs_conv2d_dgrad_partial_rows": (c_int, [POINTER(CsConvGeom)]),
    "cs_fold_partial_rows": (c_int, [_P, c_int, c_int, _P, _P]),
    "cs_wgrad_finalize_grouped": (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "cs_colsum": (c_int, [_P, c_int, c_longlong, c_int, _P, _P]),
    "cs_colsum_partial_rows": (c_int, [c_longlong]),
    "cs_colsum_partial": (c_int, [_P, c_int, c_longlong, c_int, _P, _P]),
    "cs_positive_bits": (c_int, [_P, c_int, c_longlong, c_int, _P, _P]),
    "cs_adam_chunk_elems": (c_int, []),
    "cs_adam_max_tensors": (c_int, []),
    "cs_adam_step": (c_int, [_P, _P, c_int, c_int, _P, c_int, c_double, c_double, c_double, c_double, c_double, c_double, _P]),
    "cs_adam_step_dev": (c_int, [_P, _P, c_int, c_int, _P, c_int, _P, _P, _P, c_double, c_double, c_double, c_double, c_double, _P]),
    "cs_sample_sum_workspace": (c_size_t, [c_int, c_int, c_int]),
    "cs_sample_sum": (c_int, [_P, _P, c_int, c_float, _P, _P, c_int, c_int, c_int, _P]),
    "cs_maxpool3x3s2_fwd": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cs_maxpool3x3s2_bwd": (c_int, [_P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cs_gap_avgmax_fwd": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "cs_gap_avgmax_bwd": (c_int, [_P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "cs_linear_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "cs_linear_bwd_workspace": (c_size_t, [c_int, c_int, c_int]),
    "cs_linear_bwd": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "cs_loss_words": (c_int, []),
    "cs_softmax_ce": (c_int, [_P, _P, c_float, _P, _P, c_int, c_int, _P]),
    "cs_softmax_prob1": (c_int, [_P, _P, c_int, c_int, _P]),
    "cs_softmax_argmax": (c_int, [_P, _P, c_int, c_int, _P]),
    "cs_mse": (c_int, [_P, _P, c_int, c_int, _P, _P, c_int, _P]),
    "cs_bn_accum_words": (c_size_t, [c_int]),
    "cs_bn_accum_read": (c_int, [_P, c_int, _P, _P]),
    "cs_bn_stats": (c_int, [_P, c_int, c_longlong, c_int, _P, _P, _P]),
    "cs_bn_partial_workspace": (c_size_t, [c_longlong, c_int]),
    "cs_bn_finalize": (c_int, [_P, c_longlong, c_float, c_float, _P, _P, _P, _P, c_int, _P]),
    "cs_bn_apply": (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int, _P, c_longlong, c_int, _P]),
    "cs_bn_apply_stats": (c_int, [_P, c_int, _P, c_float, c_float, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_longlong, c_int, _P]),
    "cs_bn_bwd_reduce": (c_int, [_P, _P, c_int, _P, _P, _P, _P, c_int, c_longlong, c_int, _P, _P, _P]),
    "cs_bn_bwd_apply": (c_int, [_P, _P, c_int, _P, _P, _P, _P, c_int, _P, c_longlong, c_int, _P, _P, _P, _P]),
    "cs_dwconv_fwd": (c_int, [POINTER(CsConvGeom), c_int, _P, _P, _P, _P, c_int, _P, _P]),
    "cs_dwconv_fwd_stats_workspace": (c_size_t, [POINTER(CsConvGeom)]),
    "cs_dwconv_fwd_stats": (c_int, [POINTER(CsConvGeom), c_int, _P, _P, _P, _P, POINTER(c_int), _P]),
    "cs_bn_partial_fold": (c_int, [_P, c_int, c_int, _P, _P]),
    "cs_dwconv_dgrad": (c_int, [POINTER(CsConvGeom), c_int, _P, _P, _P, _P]),
    "cs_dwconv_wgrad": (c_int, [POINTER(CsConvGeom), c_int, _P, _P, _P, _P, _P]),
    "cs_dwconv_wgrad_oihw": (c_int, [POINTER(CsConvGeom), c_int, _P, _P, _P, _P, _P]),
    "cs_dw_weights_hwc_multi": (c_int, [_P, c_int, ctypes.c_longlong, _P]),
    "cs_dwconv_wgrad_workspace": (ctypes.c_size_t, [POINTER(CsConvGeom)]),
    "cs_se_scale": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, _P]),
    "cs_se_scale_bwd": (c_int, [_P, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "cs_rowscale_add": (c_int, [_P, c_int, _P, _P, _P, c_int, c_longlong, _P]),
    "cs_bilinear_ac_fwd": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cs_bilinear_ac_bwd": (c_int, [_P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cs_concat_channels": (c_int, [_P, _P, c_int, _P, c_longlong, c_int, c_int, _P]),
    "cs_split_channels": (c_int, [_P, c_int, _P, _P, c_longlong, c_int, c_int, _P]),
    "cs_tile_gather": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_longlong, c_int, POINTER(c_float), POINTER(c_float), c_int, _P, _P]),
    "cs_dice_fwd": (c_int, [_P, _P, c_int, c_longlong, c_float, c_int, _P, _P, _P]),
    "cs_dice_bwd": (c_int, [_P, _P, _P, c_int, c_longlong, c_float, c_int, _P, _P]),
    "cs_softmax_channel_fwd": (c_int, [_P, _P, c_int, c_int, c_longlong, c_int, _P]),
    "cs_softmax_channel_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_longlong, c_int, _P]),
    "cs_segmented_topk_workspace": (c_size_t, [c_longlong]),
    "cs_segmented_topk": (c_int, [_P, _P, _P, _P, c_int, c_int, c_longlong, _P, _P, _P, c_size_t, _P]),
    "cs_stem_pair_input": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "cs_stem_pair_from_nchw": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "cs_stem_fwd_packed": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P, _P]),
    "cs_stem_pair_weights": (c_int, [_P, c_int, c_int, _P, _P]),
    "cs_stem_fwd": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "cs_stem_wgrad_splits": (c_int, [c_int, c_int, c_int, c_int]),
    "cs_stem_wgrad": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P]),
    "cs_stem_unpair_slabs": (c_int, [_P, c_int, c_int, _P, _P]),
    "cs_segmented_order": (c_int, [_P, _P, c_int, c_int, c_longlong, _P, _P]),
    "cs_threshold_select": (c_int, [_P, _P, c_longlong, c_float, _P, _P, _P, c_size_t, _P]),
    "cs_evaluate_tile_counts": (c_int, [_P, _P, _P, _P, c_float, c_longlong, _P, _P]),
    "cs_paint_tile_masks": (c_int, [_P, c_longlong, _P, _P, c_int, c_int, c_int, _P, _P]),
    "cs_prune_excess": (c_int, [_P, c_longlong, c_int, c_longlong, _P, _P, _P]),
}

_lib = None


def exported_symbols():
    """Names every build of the library must export (kept in sync with include/cellseg_hip.h)."""
    return sorted(_SIGNATURES)


def load():
    """Load the shared library once; raise CellsegLibraryMissing if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CellsegLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C cellsegmentation_amd/csrc`). There is no CPU fallback for the HIP hot path.")
    # torch first: its wheel bundles its own libamdhip64 / libhsa-runtime64, and both libraries must share ONE HIP runtime (streams,
    # device pointers).  Loaded the other way round, this library pulls /opt/rocm's runtime in first and torch's device queries then
    # fail with "no ROCm-capable device is detected" (seen when build() and smoke() ran in one process).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI mismatch, intentionally fatal
        fn.restype = restype
        fn.argtypes = argtypes
    if FLAVOUR in ("ab", "dbg"):
        for name, (restype, argtypes) in _AB_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().cs_last_error()
        raise RuntimeError(f"libcellseg_hip {what} failed (code {rc}): {msg.decode() if msg else ''}")

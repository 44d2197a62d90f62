"""ctypes binding of ``libtonal_hip.so`` (the C ABI declared in ``include/tonal_hip.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``csrc/Makefile``.  There is no CPU
fallback anywhere in this package: if the shared object is missing, or a kernel entry point
reports an error, a ``RuntimeError`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TONAL_HIP_LIB", os.path.join(_HERE, "libtonal_hip.so"))   # env override: A/B builds

# epilogue / loader codes of tl_gemm_nt_window (include/tonal_hip.h)
LOAD_DIRECT, LOAD_UNPOOL, LOAD_V = 0, 1, 2
EPI_STORE, EPI_LRELU, EPI_POOL, EPI_MASK, EPI_C1WGRAD, EPI_POOLV, EPI_MASKY, EPI_GY = 0, 1, 2, 3, 4, 5, 6, 7
LOAD_Y = 3


class NtParams(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("abits", C.c_void_p), ("Bw", C.c_void_p), ("bias", C.c_void_p), ("aux", C.c_void_p),
        ("out", C.c_void_p), ("obits", C.c_void_p),
        ("M", C.c_int64), ("A_rows", C.c_int64),
        ("N", C.c_int), ("K", C.c_int),
        ("lda", C.c_int), ("ldb", C.c_int), ("ldo", C.c_int), ("ldaux", C.c_int), ("ld_abits", C.c_int),
        ("ld_obits", C.c_int),
        ("J", C.c_int), ("row_shift", C.c_int),
        ("Tp", C.c_int), ("Tvalid", C.c_int), ("Tvalid_in", C.c_int),
        ("slope", C.c_float),
        ("loader", C.c_int), ("epilogue", C.c_int),
        ("splitk", C.c_int), ("slab_stride", C.c_int64),
        ("bm", C.c_int),
        ("osign", C.c_void_p), ("auxbits", C.c_void_p), ("ld_auxbits", C.c_int),
        ("c1x", C.c_void_p), ("c1bits", C.c_void_p), ("c1partial", C.c_void_p), ("c1T", C.c_int), ("c1kt", C.c_int),
        ("vout", C.c_void_p), ("vhalo", C.c_void_p), ("vout_quads", C.c_int64), ("ld_vout", C.c_int),
        ("out_tp", C.c_int), ("vout2", C.c_void_p),
    ]


class TnParams(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("bbits", C.c_void_p), ("slab", C.c_void_p),
        ("Krows", C.c_int64), ("A_rows", C.c_int64), ("B_rows", C.c_int64),
        ("Mdim", C.c_int), ("Ndim", C.c_int),
        ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int), ("ld_bbits", C.c_int),
        ("J", C.c_int), ("Tp", C.c_int), ("Tvalid", C.c_int), ("loader", C.c_int),
        ("splitk", C.c_int), ("slab_stride", C.c_int64),
        ("colsum", C.c_void_p),
        ("vd", C.c_void_p), ("ld_vd", C.c_int), ("part", C.c_int), ("bm", C.c_int), ("g_tp", C.c_int),
    ]


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float

#: every symbol include/tonal_hip.h declares -> (restype, argtypes)
SIGNATURES = {
    "tl_last_error": (C.c_char_p, []),
    "tl_version": (_I, []),
    "tl_device_count": (_I, []),
    "tl_gemm_nt_window": (_I, [C.POINTER(NtParams), _P]),
    "tl_gemm_tn_window": (_I, [C.POINTER(TnParams), _P]),
    "tl_wino43_weights": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "tl_wino63_xform2": (_I, [_P, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "tl_wino63_weights7": (_I, [_P, _P, _I, _I, _I, _P]),
    "tl_conv7_wino63v_nt": (_I, [C.POINTER(NtParams), _P]),
    "tl_wino63_unpool_rows6": (_I, [_P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "tl_wino63_weights1": (_I, [_P, _P, _I, _I, _I, _P]),
    "tl_conv1_wino63v_dgrad_nt": (_I, [C.POINTER(NtParams), _P]),
    "tl_comm_unique_id": (_I, [_P]),
    "tl_comm_init": (_I, [C.POINTER(C.c_void_p), _I, _I, _P]),
    "tl_comm_destroy": (_I, [_P]),
    "tl_allreduce": (_I, [_P, _P, _P, _L, _I, _P]),
    "tl_reduce_scatter": (_I, [_P, _P, _P, _L, _P]),
    "tl_all_gather": (_I, [_P, _P, _P, _L, _P]),
    "tl_wino43_wgrad_finalize": (_I, [_P, _P, _I, _I, _I, _P]),
    "tl_wino43_input_transform": (_I, [_P, _P, _L, _I, _I, _I, _I, _P]),
    "tl_conv3_wino43v_nt": (_I, [C.POINTER(NtParams), _P]),
    "tl_wino43_v_fixup": (_I, [_P, _P, _L, _L, _I, _I, _I, _P]),
    "tl_wino43_unpool_transform": (_I, [_P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _P]),
    "tl_conv3_wino43v_tn": (_I, [C.POINTER(TnParams), _P]),
    "tl_wino63_weights": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "tl_conv3_wino63v_nt": (_I, [C.POINTER(NtParams), _P]),
    "tl_wino63_nt_tile_rows": (_I, []),
    "tl_wino63_v_fixup": (_I, [_P, _P, _L, _L, _I, _I, _I, _P]),
    "tl_conv3_wino63v_tn": (_I, [C.POINTER(TnParams), _P]),
    "tl_wino63_wgrad_finalize": (_I, [_P, _P, _I, _I, _I, _P]),
    "tl_wino63_vd_fixup": (_I, [_P, _P, _L, _L, _I, _I, _I, _P]),
    "tl_wino63_unpool_yvd": (_I, [_P, _P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _P]),
    "tl_conv1_fwd_v6": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _F, _P]),
    "tl_sizeof_nt_params": (_I, []),
    "tl_sizeof_tn_params": (_I, []),
    "tl_conv1_fwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _F, _P]),
    "tl_conv1_fwd_v": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _F, _P]),
    "tl_dropout_scale": (_I, [_P, _L, _F, C.c_uint64, _P]),
    "tl_conv1_wgrad": (_I, [_P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _I, _P]),
    "tl_permute_reduce": (_I, [_P, _P, C.POINTER(_L), C.POINTER(_L), C.POINTER(_L), _I, _L, _P, _P]),
    "tl_colsum": (_I, [_P, _P, _I, _L, _I, _I, _I, _I, _P]),
    "tl_lstm_cell_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "tl_lstm_gw": (_I, [_P, _P, _P, _I, _L, _I, _L, _L, _I, _P]),
    "tl_lstm_infer_seq_fused": (_I, [_P, _L, _P, _P, _P, _P, _I, _I, _I, C.POINTER(C.c_int), _P]),
    "tl_lstm_cell_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "tl_lstm_ih_grad": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "tl_concat_pack": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, C.c_uint64, _L, _P]),
    "tl_concat_unpack_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F,
                                  C.c_uint64, _L, _P]),
    "tl_l1_mcd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "tl_nadam": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _F, _F, _P]),
    "tl_nadam_multi": (_I, [_P, _I, _L, _F, _F, _F, _F, _F, _F, _F, _F, _P]),
    "tl_nadam_multi_chunk": (_I, []),
    "tl_sum_slabs2": (_I, [_P, _P, _L, _P, _P, _L, _I, _P]),
    "tl_set_step_scalars": (_I, [_P, _P, _F, _F, _F, C.c_uint64, _P]),
    "tl_stage_step": (_I, [_P, _P, _F, _F, _F, C.c_uint64, _P, _P, _P, _I, _P]),
    "tl_nadam_multi_dev": (_I, [_P, _I, _L, _P, _F, _F, _F, _F, _F, _P]),
    "tl_nadam_lowrank": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _F, _F, _F, _F, _F, _P]),
    "tl_nadam_lowrank_dh": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _F, _F, _F, _F, _F, _P, _I, _I, _P]),
    "tl_tone_dynamics": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "tl_splitk_bias_lrelu": (_I, [_P, _P, _P, _I, _L, _I, _F, _P]),
    "tl_labels_from_scores": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "tl_lite_conv_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "tl_lite_bn_finalize": (_I, [_P, _P, _P, _P, _P, _I, _I, _L, _F, _F, _I, _P, _P]),
    "tl_lite_bn_act_pool_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "tl_lite_bn_act_pool_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P]),
    "tl_lite_conv_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "tl_lite_lstm_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "tl_lite_lstm_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "tl_lite_cat": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, C.c_uint64, _P]),
    "tl_lite_uncat": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, C.c_uint64, _P]),
    "tl_lite_cat_dev": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P]),
    "tl_lite_uncat_dev": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P]),
    "tl_row_zscore": (_I, [_P, _I, _P, _P, _I, _L, _L, _L, _I, _P]),
    "tl_car": (_I, [_P, _I, _P, _P, _I, _L, _I, _P]),
    "tl_rolling_zscore": (_I, [_P, _I, _P, _I, _L, _I, _I, _P]),
    "tl_fft_resample": (_I, [_P, _I, _P, _I, _L, _L, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P]),
    "tl_gauss_envelope": (_I, [_P, _I, _P, _P, _I, _L, _I, _I, _I, _I, _P]),
    "tl_gauss_envelope_sym": (_I, [_P, _I, _P, _P, _I, _L, _I, _I, _I, _P]),
    "tl_linear_rows": (_I, [_P, _P, _P, _P, _I, _I, _I, _L, _I, _P]),
    "tl_hilbert_ols": (_I, [_P, _I, _P, _P, _P, _I, _L, _I, _I, _I, _I, _P]),
    "tl_hilbert_ols_bl": (_I, [_P, _I, _P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _P]),
    "tl_hilbert_fft": (_I, [_P, _I, _P, _I, _L, _P, _I, _P, _P, _P, _I, _I, _P, _P]),
    "tl_filtfilt_f64": (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _L, _I, _P]),
    "tl_filtfilt_scan_f64": (_I, [_P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _I, _L, _I, _I, _P]),
    "tl_sosfilt_f64": (_I, [_P, _I, _P, _P, _I, _L, _I, _P]),
    "tl_fir_bank": (_I, [_P, _I, _P, _P, _I, _I, _L, _I, _I, _P]),
    "tl_fir_bank_ols": (_I, [_P, _I, _P, _P, _P, _I, _I, _L, _I, _I, _P]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared object (once) and type every entry point.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own HIP runtime (torch/lib/libamdhip64.so).  Loaded before it, this library would pull in
    # the system one (/opt/rocm) instead and its kernels would register with a runtime that torch's streams and allocations
    # do not belong to - every launch then fails with "no ROCm-capable device is detected" (seen with build() followed by
    # smoke() in one process).  With torch's runtime already in the process the dependency resolves to it.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C decode_tonal_langauge_amd/csrc`). This package has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.tl_sizeof_nt_params() != C.sizeof(NtParams) or lib.tl_sizeof_tn_params() != C.sizeof(TnParams):
        raise RuntimeError("libtonal_hip.so parameter structs do not match the ctypes binding; rebuild the library")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().tl_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_gpu(t, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: tensor is on '{t.device}'. The MI355X path runs HIP kernels only and has no CPU "
            "fallback; move the model and its inputs to a 'cuda' device.")

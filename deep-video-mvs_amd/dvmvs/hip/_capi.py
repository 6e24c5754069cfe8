"""ctypes binding of ``libdvmvs_hip.so`` (the C ABI declared in ``include/dvmvs_hip.h``).

The library is the only implementation of the hot path: there is no CPU or eager-PyTorch fallback.  If it has not
been built (``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C deep-video-mvs_amd/csrc``) every op
raises ``RuntimeError`` on first use.
"""
import ctypes
import os
import threading

# PyTorch-ROCm ships its own libamdhip64.so.  It must be in the process BEFORE libdvmvs_hip.so is dlopen'ed so that the
# library's DT_NEEDED libamdhip64.so.7 resolves to that already-loaded copy: two HIP runtimes in one process do not
# share streams or device memory (symptom: "no ROCm-capable device is detected" from the second one).  The same holds for
# libMIOpen.so.1 (dvmvs_conv_bias_act_fwd): the process has ONE MIOpen, the one torch's convolutions use.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVMVS_HIP_LIB", os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libdvmvs_hip.so")))

ABI_VERSION = 8
MAX_MEASUREMENTS = 8
MAX_DEPTH_LEVELS = 256
LAYOUT_NCHW, LAYOUT_NHWC = 0, 1

_c_fp = ctypes.c_void_p          # device pointer to float
_c_fpp = ctypes.POINTER(ctypes.c_void_p)  # host array of device pointers
_c_int = ctypes.c_int
_c_dbl = ctypes.c_double
_c_stream = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol the header declares (checked by tests/test_capi_symbols.py)
SIGNATURES = {
    "dvmvs_abi_version": (_c_int, []),
    "dvmvs_build_arch": (ctypes.c_char_p, []),
    "dvmvs_error_string": (ctypes.c_char_p, [_c_int]),
    "dvmvs_trace_marker": (_c_int, [_c_stream]),
    "dvmvs_host_pointer_device_visible": (_c_int, [ctypes.c_void_p]),
    "dvmvs_cost_volume_workspace_bytes": (ctypes.c_size_t, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "dvmvs_sweep_matrices": (_c_int, [_c_fp, _c_fpp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_stream]),
    "dvmvs_cost_volume_fwd": (_c_int, [_c_fp, _c_fpp, _c_fp, _c_fp, _c_fp,
                                       _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                       _c_dbl, _c_dbl, _c_int, _c_int, _c_int, _c_fp, ctypes.c_size_t, _c_stream]),
    "dvmvs_sweep_plan_stats": (_c_int, [_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_dbl, _c_dbl, _c_int,
                                        ctypes.POINTER(ctypes.c_longlong)]),
    "dvmvs_sweep_select_variant": (_c_int, [_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_dbl, _c_dbl]),
    "dvmvs_sweep_work_list_bytes": (ctypes.c_size_t, [_c_int, _c_int, _c_int, _c_int]),
    "dvmvs_sweep_work_list": (_c_int, [_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_dbl, _c_dbl, _c_int, _c_fp, ctypes.c_size_t]),
    "dvmvs_sweep_plan": (_c_int, [_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_dbl, _c_dbl, _c_int, _c_fp, ctypes.c_size_t]),
    "dvmvs_sweep_plan6": (_c_int, [_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_dbl, _c_dbl, _c_fp, ctypes.c_size_t]),
    "dvmvs_nchw_to_nhwc": (_c_int, [_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "dvmvs_copy_batch": (_c_int, [_c_fpp, _c_fpp, ctypes.POINTER(ctypes.c_longlong), _c_int, _c_stream]),
    "dvmvs_sweep_mfma_estimate": (_c_int, [_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_dbl, _c_dbl, ctypes.POINTER(ctypes.c_double)]),
    "dvmvs_cost_volume_planned_fwd": (_c_int, [_c_fp, _c_fpp, _c_fp, _c_fp, _c_fp,
                                               _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                               _c_dbl, _c_dbl, _c_int, _c_int, _c_int, _c_fp, ctypes.c_size_t, _c_fp, _c_stream]),
    "dvmvs_cost_volume_bwd": (_c_int, [_c_fp, _c_fp, _c_fpp, _c_fp, _c_fp, _c_fp, _c_fpp,
                                       _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                       _c_dbl, _c_dbl, _c_stream]),
    "dvmvs_hidden_warp_fwd": (_c_int, [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "dvmvs_hidden_warp_bwd": (_c_int, [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "dvmvs_relative_pose": (_c_int, [_c_fp, _c_fp, _c_fp, _c_int, _c_stream]),
    "dvmvs_lstm_gates_fwd": (_c_int, [_c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "dvmvs_lstm_gates_bwd": (_c_int, [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "dvmvs_lstm_gates_partials_fwd": (_c_int, [_c_fp, _c_int, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "dvmvs_bottleneck_conv_packed_bytes": (ctypes.c_size_t, [_c_int, _c_int]),
    "dvmvs_bottleneck_conv_pack": (_c_int, [_c_fp, _c_fp, _c_int, _c_int, _c_stream]),
    "dvmvs_bottleneck_conv_splits": (_c_int, [_c_int] * 6),
    "dvmvs_bottleneck_conv_fwd": (_c_int, [_c_fp, _c_fp, _c_fp] + [_c_int] * 6 + [_c_stream]),
    "dvmvs_bottleneck_conv_up2x_fwd": (_c_int, [_c_fp, _c_fp, _c_fp] + [_c_int] * 5 + [_c_stream]),
    "dvmvs_direct_conv_tile": (_c_int, [_c_int] * 7),
    "dvmvs_direct_conv_packed_bytes": (ctypes.c_size_t, [_c_int] * 4),
    "dvmvs_direct_conv_pack": (_c_int, [_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "dvmvs_direct_conv_fwd": (_c_int, [_c_fp, ctypes.c_longlong, _c_fp, _c_int, _c_fp, _c_fp, ctypes.c_longlong] + [_c_int] * 8 + [_c_stream]),
    "dvmvs_direct_conv_dual_fwd": (_c_int, [_c_fp, ctypes.c_longlong, _c_fp, _c_int, _c_fp, _c_fp, ctypes.c_longlong, _c_fp] + [_c_int] * 8 + [_c_stream]),
    "dvmvs_pointwise_conv_supported": (_c_int, [_c_int] * 7),
    "dvmvs_pointwise_conv_packed_bytes": (ctypes.c_size_t, [_c_int] * 2),
    "dvmvs_pointwise_conv_pack": (_c_int, [_c_fp, _c_fp, _c_int, _c_int, _c_stream]),
    "dvmvs_pointwise_conv_fwd": (_c_int, [_c_fp, ctypes.c_longlong, _c_fp, _c_fp, _c_fp, ctypes.c_longlong, _c_int, _c_fp, ctypes.c_longlong] + [_c_int] * 7 + [_c_stream]),
    "dvmvs_conv_head_fwd": (_c_int, [_c_fp, ctypes.c_longlong, _c_fp, _c_fp, _c_fp, ctypes.c_longlong] + [_c_int] * 5 + [ctypes.c_float, ctypes.c_float, _c_stream]),
    "dvmvs_partial_sums_bias_act_fwd": (_c_int, [_c_fp, _c_int, _c_fp, ctypes.c_longlong, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "dvmvs_depth_reproject_fwd": (_c_int, [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int,
                                           _c_int, _c_int, _c_int, _c_stream]),
    "dvmvs_bias_act_fwd": (_c_int, [_c_fp, _c_fp, ctypes.c_longlong, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                    ctypes.c_float, ctypes.c_float, _c_stream]),
    "dvmvs_bias_act_inplace": (_c_int, [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "dvmvs_conv_bias_act_fwd": (_c_int, [_c_fp, _c_fp, _c_fp, _c_fp, ctypes.c_longlong] + [_c_int] * 9 + [_c_stream]),
    "dvmvs_upsample2x_fwd": (_c_int, [_c_fp, _c_fp, ctypes.c_longlong, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "dvmvs_upsample2x_pair_fwd": (_c_int, [_c_fp, _c_fp, ctypes.c_longlong, _c_int, _c_fp, _c_fp, ctypes.c_longlong, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "dvmvs_upsample2x_bwd": (_c_int, [_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "dvmvs_depthwise_conv_bwd_workspace_bytes": (ctypes.c_size_t, [_c_int] * 6),
    "dvmvs_depthwise_conv_bwd": (_c_int, [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "dvmvs_depth_reproject_lowres_fwd": (_c_int, [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "dvmvs_depth_reproject_estimate_fwd": (_c_int, [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "dvmvs_tsdf_integrate": (_c_int, [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_float, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, ctypes.c_float, ctypes.c_float, _c_stream]),
    "dvmvs_depthwise_conv_fwd": (_c_int, [_c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                          _c_stream]),
}

_lib = None
_lock = threading.Lock()


def lib():
    """Loads the shared library once; raises RuntimeError (never falls back) when it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"dvmvs HIP library not found at {LIB_PATH}. Build it with `make -C deep-video-mvs_amd/csrc` "
                f"(or __graft_entry__.build()). The plane-sweep ops have no CPU / eager fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch: let it propagate loudly
            fn.restype = restype
            fn.argtypes = argtypes
        got = handle.dvmvs_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"libdvmvs_hip.so ABI {got} does not match the Python binding ({ABI_VERSION}); rebuild")
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().dvmvs_error_string(code).decode()
        raise RuntimeError(f"{what} failed with code {code}: {msg}")


def pointer_array(ptrs):
    """Host array of device pointers (``const float* const*`` in the header)."""
    arr = (ctypes.c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr

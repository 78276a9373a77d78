"""ctypes binding of the C ABI (include/reevr_amd/rvc.h). No fallback: if the HIP library
is missing this raises, and if there is no GPU every init() fails with RVC_ERR_NO_DEVICE."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# REEVR_AMD_LIB: development override (a scratch build of the library, tools/dev/)
LIB_PATH = os.environ.get("REEVR_AMD_LIB") or os.path.join(_HERE, "csrc", "libreevr_amd.so")

F32P = C.POINTER(C.c_float)
F32PP = C.POINTER(F32P)
SZP = C.POINTER(C.c_size_t)

RVC_OK, RVC_ERR_NO_DEVICE, RVC_ERR_HIP, RVC_ERR_BAD_ARG, RVC_ERR_UNSUPPORTED, RVC_ERR_NOT_INIT = range(6)
RVC_FLAG_BG_STREAM = 1
RVC_FLAG_TIMING = 2
RVC_FLAG_FFT_F64 = 4
RVC_FLAG_FIXED_PARTITIONS = 8
RVC_FLAG_NO_TIME_TILING = 16
RVC_FLAG_FORCE_TIME_TILING = 32
RVC_FLAG_PERSISTENT = 64
RVC_FLAG_FORCE_TWO_LEVEL = 128
RVC_FLAG_FFT_F32 = 256
RVC_FLAG_NO_SUBSETS = 512
RVC_FLAG_FFT_F64_LONG = 2048
RVC_FLAG_CHILD_SETS = 1024
RVC_MAX_BLOCK = 16384

# name -> (restype, argtypes); must list every symbol declared in include/reevr_amd/rvc.h
SIGNATURES = {
    "rvc_set_create": (C.c_void_p, [C.c_int, C.c_int, C.c_uint]),
    "rvc_set_destroy": (None, [C.c_void_p]),
    "rvc_set_init": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, F32PP, SZP, C.c_size_t]),
    "rvc_set_init_uniform": (C.c_int, [C.c_void_p, C.c_size_t, F32PP, SZP, C.c_size_t]),
    "rvc_set_process": (None, [C.c_void_p, F32PP, F32PP, C.c_size_t]),
    "rvc_set_process_begin": (None, [C.c_void_p, F32PP, C.c_size_t]),
    "rvc_set_process_end": (None, [C.c_void_p, F32PP]),
    "rvc_set_process_device": (None, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t]),
    "rvc_set_process_device_blocks": (None, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                                             C.c_size_t]),
    "rvc_set_host_buffers": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "rvc_set_process_host_blocks_timed": (None, [C.c_void_p, F32PP, F32PP, C.c_size_t, C.c_size_t, C.POINTER(C.c_double)]),
    "rvc_set_process_device_blocks_stamped": (C.c_long, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                                          C.POINTER(C.c_double)]),
    "rvc_set_clear": (None, [C.c_void_p]),
    "rvc_set_reset": (None, [C.c_void_p]),
    "rvc_set_is_finished": (C.c_int, [C.c_void_p]),
    "rvc_set_sync": (None, [C.c_void_p]),
    "rvc_set_channels": (C.c_int, [C.c_void_p]),
    "rvc_set_head_block": (C.c_size_t, [C.c_void_p]),
    "rvc_set_tail_block": (C.c_size_t, [C.c_void_p]),
    "rvc_set_max_len": (C.c_size_t, [C.c_void_p]),
    "rvc_set_partitions": (C.c_int, [C.c_void_p, C.c_int]),
    "rvc_set_stream": (C.c_void_p, [C.c_void_p, C.c_int]),
    "rvc_set_fork": (None, [C.c_void_p]),
    "rvc_set_join": (None, [C.c_void_p]),
    "rvc_set_subsets": (C.c_int, [C.c_void_p]),
    "rvc_set_tile_rows": (C.c_int, [C.c_void_p, C.c_int]),
    "rvc_last_error": (C.c_int, [C.c_void_p]),
    "rvc_last_error_string": (C.c_char_p, [C.c_void_p]),
    "rvc_set_kernel_time": (C.c_long, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "rvc_set_kernel_time_reset": (None, [C.c_void_p]),
    "rvc_set_kernel_intervals": (C.c_long, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_long]),
    "rvc_set_timing": (None, [C.c_void_p, C.c_int]),
    "rvc_create": (C.c_void_p, [C.c_int]),
    "rvc_init": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, F32P, C.c_size_t]),
    "rvc_process": (None, [C.c_void_p, F32P, F32P, C.c_size_t]),
    "rvc_clear": (None, [C.c_void_p]),
    "rvc_reset": (None, [C.c_void_p]),
    "rvc_is_finished": (C.c_int, [C.c_void_p]),
    "rvc_destroy": (None, [C.c_void_p]),
    "rvc_impulse_create": (C.c_void_p, [C.c_int]),
    "rvc_impulse_destroy": (None, [C.c_void_p]),
    "rvc_impulse_set_raw": (C.c_int, [C.c_void_p, C.c_int, F32PP, C.c_size_t]),
    "rvc_impulse_recalc": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rvc_impulse_stage_a": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rvc_impulse_stage_b": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rvc_impulse_decay_lut": (None, [F32P, C.c_double, C.c_float, C.POINTER(C.c_double)]),
    "rvc_impulse_channels": (C.c_int, [C.c_void_p]),
    "rvc_impulse_size": (C.c_size_t, [C.c_void_p]),
    "rvc_impulse_peak": (C.c_float, [C.c_void_p]),
    "rvc_impulse_trim_left_samples": (C.c_int, [C.c_void_p]),
    "rvc_impulse_trim_right_samples": (C.c_int, [C.c_void_p]),
    "rvc_impulse_read": (C.c_int, [C.c_void_p, C.c_int, F32P, C.c_size_t]),
    "rvc_impulse_write": (C.c_int, [C.c_void_p, C.c_int, F32P, C.c_size_t]),
    "rvc_impulse_device_ptr": (C.c_void_p, [C.c_void_p, C.c_int]),
    "rvc_impulse_last_error": (C.c_int, [C.c_void_p]),
    "rvc_impulse_last_error_string": (C.c_char_p, [C.c_void_p]),
    "rvc_set_init_impulse": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.POINTER(C.c_int), C.c_size_t]),
    "rvc_wet_mix_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "rvc_send_pre_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "rvc_debug_rfft": (C.c_int, [C.c_int, C.c_size_t, C.c_int, F32P, F32P, F32P]),
    "rvc_debug_irfft": (C.c_int, [C.c_int, C.c_size_t, C.c_int, F32P, F32P, F32P]),
    "rvc_debug_fdl": (C.c_int, [C.c_int] * 7 + [C.c_longlong, C.c_int, F32P, F32P, F32P, F32P, C.c_longlong, C.c_longlong]),
    "rvc_debug_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "rvc_debug_guard_check": (C.c_long, [C.c_void_p]),
    "rvc_debug_fence_probe": (C.c_int, [C.c_void_p]),
    "rvc_debug_plan": (C.c_int, [C.c_int, C.c_uint, C.c_size_t, C.c_size_t, C.c_size_t] + [C.POINTER(C.c_size_t)] * 3),
    "rvc_set_create_tuned": (C.c_void_p, [C.c_int, C.c_int, C.c_uint, C.c_char_p]),
    "rvc_debug_tuning_default": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "rvc_debug_tuning_keys": (C.c_char_p, []),
    "rvc_set_plan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rvc_device_count": (C.c_int, []),
    "rvc_version": (C.c_char_p, []),
    "rvc_abi_version": (C.c_int, []),
}
RVC_ABI_VERSION = 2

RVC_IMPULSE_FFT_SIZE = 4096
RVC_IMPULSE_LUT_SIZE = RVC_IMPULSE_FFT_SIZE // 2 + 1


class Plan(C.Structure):                # struct rvc_plan
    _fields_ = [("channels", C.c_int), ("subsets", C.c_int), ("initialised", C.c_int), ("live", C.c_int), ("two_stage", C.c_int),
                ("tail_on_second_stream", C.c_int), ("head_block", C.c_size_t), ("tail_block", C.c_size_t), ("max_len", C.c_size_t),
                ("zero_latency_samples", C.c_size_t), ("head_partitions", C.c_int), ("tail_partitions", C.c_int),
                ("wide_partitions", C.c_int), ("tail_delay", C.c_int), ("head_f64", C.c_int), ("tail_f64", C.c_int),
                ("head_tile_blocks", C.c_int), ("tail_tile_blocks", C.c_int), ("block_path", C.c_int),
                ("reference_structure", C.c_int), ("long_call_block", C.c_size_t), ("wide_block", C.c_size_t),
                ("head_patch_in_launch", C.c_int), ("tail_spread", C.c_int), ("tail_sweep_slices", C.c_int),
                ("tail_phase_groups", C.c_int), ("tail_third_level", C.c_int), ("head_third_level", C.c_int)]


class ImpulseParams(C.Structure):       # struct rvc_impulse_params
    _fields_ = [("reverse", C.c_int), ("trim_left", C.c_float), ("trim_right", C.c_float), ("gain", C.c_float),
                ("attack", C.c_float), ("decay", C.c_float), ("srate", C.c_double),
                ("decay_lut", C.POINTER(C.c_double))]


class WetParams(C.Structure):           # struct rvc_wet_params
    _fields_ = [("cur", C.c_void_p * 4), ("load", C.c_void_p * 2), ("xfade", C.c_longlong), ("xfadelen", C.c_longlong),
                ("yrev", C.c_void_p), ("width", C.c_float), ("drygain", C.c_float), ("wetgain", C.c_float),
                ("dry", C.c_void_p * 2), ("out", C.c_void_p * 2), ("n", C.c_size_t)]


class SendParams(C.Structure):          # struct rvc_send_params
    _fields_ = [("in_", C.c_void_p * 2), ("ysend", C.c_void_p), ("send", C.c_void_p * 2), ("delay_ring", C.c_void_p * 2),
                ("delay_size", C.c_longlong), ("delaypos", C.c_longlong), ("predelay", C.c_longlong),
                ("delayed", C.c_void_p * 2), ("warm_ring", C.c_void_p * 2), ("warm_size", C.c_longlong),
                ("warmwritepos", C.c_longlong), ("n", C.c_size_t)]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -m reevr_amd.build, or __graft_entry__.build()). There is no CPU fallback.")
        # One HIP runtime per process: PyTorch bundles its own libamdhip64.so (same SONAME,
        # libamdhip64.so.7). If it is already loaded the dynamic loader hands that copy to our
        # library too; loaded the other way round the process ends up with two runtimes and
        # torch then reports "No HIP GPUs are available". So when torch is installed, import it
        # first. (C/C++ clients that never load torch just get /opt/rocm's runtime.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib

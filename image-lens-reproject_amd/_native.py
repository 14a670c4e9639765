"""ctypes binding of liblrp_hip.so (the C ABI of include/lrp.h).

There is no Python/numpy/torch implementation of the hot path in this package:
if the shared library is missing, importing the binding raises.  PyTorch is used
by callers only as plumbing (device memory, streams, torch.distributed).
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblrp_hip.so")


class LrpLens(ctypes.Structure):
    """lrp_lens == reference LensInfo (src/config.hpp:15-37), 28 bytes."""

    _fields_ = [
        ("type", ctypes.c_int32),
        ("raw", ctypes.c_float * 4),
        ("sensor_width", ctypes.c_float),
        ("sensor_height", ctypes.c_float),
    ]


class LrpImage(ctypes.Structure):
    """lrp_image == reference Image (src/reproject.hpp:9-14), 56 bytes."""

    _fields_ = [
        ("lens", LrpLens),
        ("width", ctypes.c_int32),
        ("height", ctypes.c_int32),
        ("channels", ctypes.c_int32),
        ("data", ctypes.c_void_p),
        ("data_layout", ctypes.c_int32),
    ]


class LrpPost(ctypes.Structure):
    _fields_ = [("exposure", ctypes.c_float), ("reinhard", ctypes.c_float)]


class LrpGeometryCacheInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("bytes", "max_bytes", "entries", "fills", "hits", "bypasses", "evictions")]


assert ctypes.sizeof(LrpLens) == 28
assert ctypes.sizeof(LrpImage) == 56

# Every symbol include/lrp.h declares: (restype, argtypes)
_P = ctypes.POINTER
_FLOATP = ctypes.c_void_p  # float* passed as raw address (host or device)
SYMBOLS = {
    "lrp_abi_version": (ctypes.c_int, []),
    "lrp_device_count": (ctypes.c_int, []),
    "lrp_debug_kernel": (ctypes.c_int, [ctypes.c_int]),
    "lrp_debug_set": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "lrp_release_cached_tables": (None, []),
    "lrp_geometry_cache_configure": (ctypes.c_int, [ctypes.c_longlong, ctypes.c_int]),
    "lrp_geometry_cache_stats": (None, [ctypes.c_void_p]),
    "lrp_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "lrp_last_error": (ctypes.c_char_p, []),
    "lrp_reproject": (
        ctypes.c_int,
        [_P(LrpImage), _P(LrpImage), ctypes.c_int, ctypes.c_int, _FLOATP, _P(LrpPost), ctypes.c_int],
    ),
    "lrp_post_process": (ctypes.c_int, [_P(LrpImage), ctypes.c_float, ctypes.c_float, ctypes.c_int]),
    "lrp_reproject_device": (
        ctypes.c_int,
        [_P(LrpImage), _P(LrpImage), ctypes.c_int, ctypes.c_int, _FLOATP, _P(LrpPost), ctypes.c_int, ctypes.c_void_p],
    ),
    "lrp_post_process_device": (
        ctypes.c_int,
        [_P(LrpImage), ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_void_p],
    ),
    "lrp_reproject_multi_device": (
        ctypes.c_int,
        [
            _P(LrpImage),
            _P(LrpImage),
            ctypes.c_int,
            ctypes.c_int,
            ctypes.c_int,
            _FLOATP,
            _P(LrpPost),
            ctypes.c_int,
            ctypes.c_void_p,
        ],
    ),
    "lrp_reproject_rows_device": (
        ctypes.c_int,
        [_P(LrpImage), _P(LrpImage), ctypes.c_int, ctypes.c_int, _FLOATP, _P(LrpPost), ctypes.c_int, ctypes.c_int, ctypes.c_int,
         ctypes.c_void_p],
    ),
    "lrp_reproject_multi": (
        ctypes.c_int,
        [_P(LrpImage), _P(LrpImage), ctypes.c_int, ctypes.c_int, ctypes.c_int, _FLOATP, _P(LrpPost), _P(ctypes.c_int), ctypes.c_int],
    ),
    "lrp_reproject_batch_device": (
        ctypes.c_int,
        [
            _P(LrpImage),
            _P(LrpImage),
            ctypes.c_int,
            ctypes.c_int,
            ctypes.c_int,
            _FLOATP,
            _P(LrpPost),
            ctypes.c_int,
            ctypes.c_void_p,
        ],
    ),
    "lrp_context_create": (ctypes.c_int, [_P(ctypes.c_void_p), ctypes.c_int, ctypes.c_int]),
    "lrp_context_destroy": (None, [ctypes.c_void_p]),
    "lrp_context_submit": (
        ctypes.c_int,
        [ctypes.c_void_p, _P(LrpImage), _P(LrpImage), ctypes.c_int, ctypes.c_int, _FLOATP, _P(LrpPost)],
    ),
    "lrp_context_wait": (ctypes.c_int, [ctypes.c_void_p]),
    "lrp_context_submit_packed": (
        ctypes.c_int,
        [ctypes.c_void_p, _P(LrpImage), ctypes.c_int, ctypes.c_int, _P(LrpImage), ctypes.c_int, ctypes.c_int, ctypes.c_uint,
         ctypes.c_int, ctypes.c_int, _FLOATP, _P(LrpPost), _P(ctypes.c_int)],
    ),
    "lrp_context_wait_ticket": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "lrp_decode_pixels_device": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _FLOATP, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p],
    ),
    "lrp_encode_pixels_device": (
        ctypes.c_int,
        [_FLOATP, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_size_t, ctypes.c_int,
         ctypes.c_void_p],
    ),
    "lrp_pixel_tables": (None, [_P(ctypes.c_float), _P(ctypes.c_float)]),
    "lrp_host_alloc": (ctypes.c_int, [_P(ctypes.c_void_p), ctypes.c_size_t]),
    "lrp_host_free": (None, [ctypes.c_void_p]),
    "lrp_synth_fill_device": (
        ctypes.c_int,
        [_FLOATP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p],
    ),
    "lrp_checksum_device": (
        ctypes.c_int,
        [_FLOATP, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p],
    ),
    "lrp_math_eval_device": (
        ctypes.c_int,
        [ctypes.c_int, _FLOATP, _FLOATP, _FLOATP, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p],
    ),
    "lrp_rotation_matrix": (None, [ctypes.c_float, ctypes.c_float, ctypes.c_float, _P(ctypes.c_float)]),
    "lrp_lens_rectilinear": (
        None,
        [_P(LrpLens), ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float],
    ),
    "lrp_lens_equidistant": (None, [_P(LrpLens), ctypes.c_float]),
    "lrp_lens_equirectangular": (
        None,
        [_P(LrpLens), ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float],
    ),
    "lrp_lens_equirectangular_full": (None, [_P(LrpLens)]),
}

_lib = None


def load():
    """Load liblrp_hip.so and declare every entry point.  Raises if the library
    or any symbol is missing — there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with image-lens-reproject_amd/csrc/build.sh "
            "(or __graft_entry__.build()); the HIP library is the only implementation"
        )
    # One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so and load it by path; liblrp_hip.so
    # names libamdhip64.so.7 and, loaded FIRST, would bring in /opt/rocm's copy — two runtimes in one process, and the second
    # one to initialise finds no device ("No HIP GPUs are available" / LRP_ERR_NO_DEVICE).  Callers hand this package torch
    # tensors and streams, so where torch is installed its runtime is loaded first and the library binds to it.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.lrp_abi_version() != 3:
        raise ImportError("liblrp_hip.so ABI version mismatch")
    _lib = lib
    return lib

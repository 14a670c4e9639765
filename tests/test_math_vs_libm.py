"""CPU: the host build of lrp_math.h against the live libm of this machine, the
library the reference's std::sin / std::cos / std::atan / std::atan2 / std::asin
resolve to (reference src/reproject.cpp:182-263).  Exhaustive over all 2^32
binary32 inputs for the unary functions; structured + random pairs for atan2f.
Bit-exact (any NaN equals any NaN)."""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "native", "_build", "liblrp_math_check.so")
THREADS = min(32, os.cpu_count() or 1)


@pytest.fixture(scope="module")
def chk():
    L = ctypes.CDLL(SO)
    L.lrp_check_unary.restype = ctypes.c_uint64
    L.lrp_check_unary.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_uint32)]
    L.lrp_check_atan2.restype = ctypes.c_uint64
    L.lrp_check_atan2.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    L.lrp_eval_atan2.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_uint64]
    L.lrp_check_odd.restype = ctypes.c_uint64
    L.lrp_check_odd.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
    L.lrp_check_u8_quantiser.restype = ctypes.c_uint64
    L.lrp_check_u8_quantiser.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
    return L


@pytest.mark.parametrize("func,name", [(0, "sinf"), (1, "cosf"), (2, "sincosf.sin"), (3, "sincosf.cos"),
                                       (4, "atanf"), (5, "asinf")])
def test_unary_exhaustive(chk, func, name):
    first = ctypes.c_uint32(0)
    bad = chk.lrp_check_unary(func, 0, 1 << 32, 1, THREADS, ctypes.byref(first))
    assert bad == 0, f"{name}: {bad} of 2^32 inputs differ from libm, first bit pattern 0x{first.value:08x}"


@pytest.mark.parametrize("func,name", [(0, "sinf"), (4, "atanf"), (5, "asinf")])
def test_odd_functions_are_odd_bit_for_bit(chk, func, name):
    """f(-x) == -f(x) for all 2^31 magnitudes: the mirrored blocks of the window kernel share one
    evaluation between the four mirror images of a pixel (asinf for equirectangular sources)."""
    first = ctypes.c_uint32(0)
    bad = chk.lrp_check_odd(func, THREADS, ctypes.byref(first))
    assert bad == 0, f"{name}: {bad} magnitudes with f(-x) != -f(x), first 0x{first.value:08x}"


@pytest.mark.parametrize("mode,count", [(0, 1 << 27), (1, 1 << 28), (2, 1 << 26)])
def test_atan2_random_pairs(chk, mode, count):
    """atan2f_ == libm's atan2f, and both odd in their first argument (atan2f(-y, x) == -atan2f(y, x) bit for bit)."""
    by, bx = ctypes.c_uint32(0), ctypes.c_uint32(0)
    bad = chk.lrp_check_atan2(0xC0FFEE + mode, count, mode, THREADS, ctypes.byref(by), ctypes.byref(bx))
    assert bad == 0, f"atan2f mode {mode}: {bad} mismatches, first y=0x{by.value:08x} x=0x{bx.value:08x}"


def test_atan2_special_grid(chk):
    """Every pair from a grid of special values (signed zeros, infinities, NaN,
    denormals, the x == 1 shortcut, huge exponent gaps)."""
    vals = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38, 3e38, -3e38, 0.5, 2.0, 1e-20,
                     1e20, -1e-20, -1e20, 3.14159274, 0.4375, 2.4375], dtype=np.float32)
    y, x = [a.reshape(-1).copy() for a in np.meshgrid(vals, vals)]
    own, ref = np.empty_like(y), np.empty_like(y)
    chk.lrp_eval_atan2(y.ctypes.data, x.ctypes.data, own.ctypes.data, ref.ctypes.data, y.size)
    same = (own.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(own) & np.isnan(ref))
    assert same.all(), f"atan2f special grid: y={y[~same][:4]} x={x[~same][:4]}"
    # odd in y, signed zeros and infinities included (the columns-only mirror mode relies on it)
    own_n, ref_n = np.empty_like(y), np.empty_like(y)
    ny = (-y).copy()
    chk.lrp_eval_atan2(ny.ctypes.data, x.ctypes.data, own_n.ctypes.data, ref_n.ctypes.data, y.size)
    for a, b, what in ((own_n, -own, "atan2f_"), (ref_n, -ref, "libm atan2f")):
        odd = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert odd.all(), f"{what} is not odd in y at y={y[~odd][:4]} x={x[~odd][:4]}"


def test_u8_quantiser_equals_its_threshold_table(chk, lrp):
    """The device encode kernel of LRP_PIXEL_U8_GAMMA searches a 256-entry threshold table built with the
    host's powf instead of evaluating a pow on the device: for all 2^30 floats of [0, 1] the search must give
    uint8(255.9f * powf(s, 1 / 2.2f)), save_png's code (reference src/image_formats.cpp:155-158)."""
    dec = (ctypes.c_float * 256)()
    thr = (ctypes.c_float * 256)()
    lrp._native.load().lrp_pixel_tables(dec, thr)
    t = np.frombuffer(thr, dtype=np.float32)
    assert t[0] == 0.0 and np.all(np.diff(t) >= 0) and t[255] <= 1.0
    libm = ctypes.CDLL("libm.so.6")
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    d = np.frombuffer(dec, dtype=np.float32)
    for k in (0, 1, 2, 17, 128, 254, 255):  # read_png / read_jpeg: pow(float(p) / 255.0f, 2.2f)
        assert d[k] == np.float32(libm.powf(np.float32(k) / np.float32(255.0), np.float32(2.2)))
    first = ctypes.c_uint32(0)
    bad = chk.lrp_check_u8_quantiser(ctypes.addressof(thr), THREADS, ctypes.byref(first))
    assert bad == 0, f"{bad} floats of [0, 1] quantise differently, first bit pattern 0x{first.value:08x}"

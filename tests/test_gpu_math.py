"""-m gpu: the device build of lrp_math.h against the host libm (the library the
reference's std::sin/cos/atan/atan2/asin resolve to), bit for bit; plus IEEE
divide / sqrt / float->int conversion semantics on gfx950."""
import ctypes
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libm_eval():
    L = ctypes.CDLL(os.path.join(ROOT, "tests", "native", "_build", "liblrp_math_check.so"))
    L.lrp_eval_unary.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
    L.lrp_eval_atan2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
    return L


def _inputs(n, seed):
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    a = bits.view(np.float32).copy()
    lens_like = ((rng.random(n) * 2 - 1) * 4).astype(np.float32)      # angles
    unit = ((rng.random(n) * 2 - 1) * 1.001).astype(np.float32)        # asin domain
    big = ((rng.random(n) * 2 - 1) * 1e5).astype(np.float32)           # reduce_large
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38, 3.4e38,
                        0.4375, 0.6875, 1.1875, 2.4375, 0.975, 120.0, 119.99, 0.78539816, 3.14159265, 1.5707964],
                       dtype=np.float32)
    return np.concatenate([a, lens_like, unit, big, special])


@pytest.mark.parametrize("func,name", [(0, "sinf"), (1, "cosf"), (2, "sincosf.sin"), (3, "sincosf.cos"),
                                       (4, "atanf"), (5, "asinf")])
def test_unary_device_math_equals_host_libm(lrp, torch_cuda, func, name):
    torch = torch_cuda
    x = _inputs(1 << 20, seed=func)
    own = np.empty_like(x)
    ref = np.empty_like(x)
    _libm_eval().lrp_eval_unary(func, x.ctypes.data, own.ctypes.data, ref.ctypes.data, x.size)
    dev = lrp.math_eval(func, torch.from_numpy(x).cuda()).cpu().numpy()
    cases.assert_same_bits(dev, ref, f"device {name} vs host libm")
    cases.assert_same_bits(own, ref, f"host build of lrp_math {name} vs host libm")


def test_atan2_device_equals_host_libm(lrp, torch_cuda):
    torch = torch_cuda
    y = _inputs(1 << 20, seed=11)
    x = _inputs(1 << 20, seed=12)[::-1].copy()
    own = np.empty_like(x)
    ref = np.empty_like(x)
    _libm_eval().lrp_eval_atan2(y.ctypes.data, x.ctypes.data, own.ctypes.data, ref.ctypes.data, x.size)
    dev = lrp.math_eval(6, torch.from_numpy(y).cuda(), torch.from_numpy(x).cuda()).cpu().numpy()
    cases.assert_same_bits(dev, ref, "device atan2f vs host libm")


def test_ieee_divide_sqrt_and_x86_truncation(lrp, torch_cuda):
    torch = torch_cuda
    a = _inputs(1 << 20, seed=21)
    b = _inputs(1 << 20, seed=22)[::-1].copy()
    with np.errstate(all="ignore"):
        want_div = (a / b).astype(np.float32)
        want_sqrt = np.sqrt(a).astype(np.float32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    cases.assert_same_bits(lrp.math_eval(7, ta, tb).cpu().numpy(), want_div, "IEEE divide (denormals kept)")
    cases.assert_same_bits(lrp.math_eval(8, ta).cpu().numpy(), want_sqrt, "IEEE sqrt")
    # int(float) with the cvttss2si convention: out of range / NaN -> INT_MIN
    with np.errstate(all="ignore"):
        inr = np.abs(a) < np.float32(2147483648.0)
        want_i = np.where(inr, np.trunc(np.where(inr, a, 0)).astype(np.int64), -(2**31)).astype(np.float32)
    cases.assert_same_bits(lrp.math_eval(9, ta).cpu().numpy(), want_i, "x86 float->int")

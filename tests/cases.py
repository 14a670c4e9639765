"""Shared case builders for the parity tests."""
import math

import numpy as np


def hash_noise(h, w, c, seed, planted=True):
    """Half-representable noise in [0,1) plus a few planted special texels
    (-0.0, denormal, inf, NaN, large HDR), like SURVEY.md §8c asks."""
    rng = np.random.default_rng(seed)
    a = (rng.integers(0, 2048, size=(h, w, c)).astype(np.float32) / np.float32(2048.0)).astype(np.float32)
    if planted and h * w >= 16:
        flat = a.reshape(-1)
        idx = rng.choice(flat.size, size=min(6, flat.size), replace=False)
        specials = np.array([-0.0, 1e-41, 65504.0, 3.0e38, -7.25, 1.0], dtype=np.float32)
        flat[idx[: len(specials)]] = specials[: len(idx)]
    return a


def lenses(lrp, w, h):
    """The lens set of Appendix B: name -> LensInfo for an image of w x h."""
    L = lrp.LensInfo
    return {
        "rect": L.rectilinear(18.0, 36.0, w, h),
        "rect_tele": L.rectilinear(50.0, 36.0, w, h),
        "eqd180": L.equidistant(math.pi),
        "eqd120": L.equidistant(2.0943951),
        "eqr_full": L.equirectangular(),
        "eqr_part": L.equirectangular(-1.0, 1.5, -0.6, 0.7),
    }


ROTATIONS_DEG = [None, (0.0, 0.0, 0.0), (30.0, -15.0, 5.0), (180.0, 0.0, 0.0), (0.0, 90.0, 0.0)]


def rotation(lrp, deg):
    if deg is None:
        return None
    pan, pitch, roll = [np.float32(d) * np.float32(math.pi) / np.float32(180.0) for d in deg]
    return lrp.rotation_matrix(float(pan), float(pitch), float(roll))


def same_bits(a, b):
    """Bit-exact comparison where any NaN matches any NaN."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    ua, ub = a.view(np.uint32), b.view(np.uint32)
    ok = (ua == ub) | (np.isnan(a) & np.isnan(b))
    return ok


def assert_same_bits(a, b, what=""):
    ok = same_bits(a, b)
    if not ok.all():
        bad = np.argwhere(~ok)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {ok.size} values differ; first at {i}: "
                             f"{a[i]!r} (0x{a.view(np.uint32)[i]:08x}) vs {b[i]!r} (0x{b.view(np.uint32)[i]:08x})")

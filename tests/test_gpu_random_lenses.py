"""-m gpu: randomized lens pairs, rotations, sizes, channel counts and sub-sample
counts (fixed seeds), HIP vs the oracle, bit for bit, for all three kernel
families.  Exercises magnification and strong minification (LDS window too large
-> direct taps), partial equirectangular ranges, tele / wide lenses, views that
fall outside the source (clamped taps, the degenerate-tap shortcut), seams and
poles."""
import math

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def random_lens(lrp, rng, w, h):
    kind = rng.integers(0, 4)
    L = lrp.LensInfo
    if kind == 0:
        return L.rectilinear(float(rng.uniform(6.0, 80.0)), float(rng.choice([24.0, 36.0, 50.0])), w, h)
    if kind == 1:
        return L.equidistant(float(rng.uniform(0.6, 2.0 * math.pi)))
    if kind == 2:
        return L.equirectangular()
    lon0 = float(rng.uniform(-3.0, 0.5))
    lat0 = float(rng.uniform(-1.5, 0.2))
    return L.equirectangular(lon0, lon0 + float(rng.uniform(0.5, 3.0)), lat0, lat0 + float(rng.uniform(0.3, 1.4)))


def random_rotation(lrp, rng):
    r = rng.integers(0, 6)
    if r == 0:
        return None
    if r == 1:
        return cases.rotation(lrp, (0.0, 0.0, 0.0))
    if r == 2:
        return cases.rotation(lrp, (float(rng.uniform(-180.0, 180.0)), 0.0, 0.0))
    if r == 3:
        return cases.rotation(lrp, (0.0, float(rng.uniform(-180.0, 180.0)), 0.0))
    return cases.rotation(lrp, tuple(float(v) for v in rng.uniform(-180.0, 180.0, size=3)))


@pytest.mark.parametrize("seed", range(24))
def test_random_configuration(lrp, oracle, torch_cuda, seed):
    torch = torch_cuda
    rng = np.random.default_rng(1000 + seed)
    in_w, in_h = int(rng.integers(40, 400)), int(rng.integers(30, 300))
    out_w, out_h = int(rng.integers(17, 330)), int(rng.integers(9, 250))
    c = int(rng.choice([3, 4, 4, 4, 5, 1]))
    ns = int(rng.choice([1, 1, 1, 2, 3]))
    lin, lout = random_lens(lrp, rng, in_w, in_h), random_lens(lrp, rng, out_w, out_h)
    rot = random_rotation(lrp, rng)
    src = cases.hash_noise(in_h, in_w, c, seed=seed, planted=bool(rng.integers(0, 2)))
    d_in = torch.from_numpy(src).cuda()
    for interp in (0, 1, 2):
        want = oracle.reproject(lin, src, lout, out_w, out_h, ns, interp, rot)
        for family in (2, 3, 1, 0):
            prev = lrp.debug_kernel(family)
            try:
                d_out = torch.full((out_h, out_w, c), -777.0, dtype=torch.float32, device="cuda")
                lrp.reproject(lrp.Image(lin, in_w, in_h, c, d_in), lrp.Image(lout, out_w, out_h, c, d_out), ns, interp, rot)
                torch.cuda.synchronize()
            finally:
                lrp.debug_kernel(prev)
            cases.assert_same_bits(d_out.cpu().numpy(), want,
                                   f"seed {seed}: {in_w}x{in_h}x{c} -> {out_w}x{out_h}, lens {lin.type}->{lout.type}, "
                                   f"ns={ns}, interp={interp}, family={family}")


@pytest.mark.parametrize("scale", [0.25, 0.5, 2.0, 3.0])
def test_scaled_output_window_paths(lrp, oracle, torch_cuda, scale):
    """--scale use case: 1024x768 source, output scaled down (minification: the bicubic
    window outgrows LDS and the kernel takes direct taps) and up (magnification)."""
    torch = torch_cuda
    in_w, in_h = 1024, 768
    out_w, out_h = int(in_w * scale), int(in_h * scale)
    src = cases.hash_noise(in_h, in_w, 4, seed=int(scale * 100))
    d_in = torch.from_numpy(src).cuda()
    lin = lrp.LensInfo.equidistant(math.pi)
    lout = lrp.LensInfo.equidistant(math.pi)
    rot = cases.rotation(lrp, (10.0, 5.0, 20.0))
    want = oracle.reproject(lin, src, lout, out_w, out_h, 1, 2, rot, threads=8)
    for family in (2, 3, 1, 0):
        prev = lrp.debug_kernel(family)
        try:
            d_out = torch.empty((out_h, out_w, 4), dtype=torch.float32, device="cuda")
            lrp.reproject(lrp.Image(lin, in_w, in_h, 4, d_in), lrp.Image(lout, out_w, out_h, 4, d_out), 1, 2, rot)
            torch.cuda.synchronize()
        finally:
            lrp.debug_kernel(prev)
        cases.assert_same_bits(d_out.cpu().numpy(), want, f"scale {scale} family {family}")


@pytest.mark.parametrize("pair", [("eqd180", "rect"), ("eqr_full", "rect"), ("rect", "rect"), ("eqd180", "eqd120"),
                                  ("eqr_part", "rect_tele")])
def test_magnified_bicubic_shared_tap_coefficients(lrp, oracle, torch_cuda, pair):
    """Magnified RGBA bicubic: the window kernel evaluates the weight-independent part of
    the vertical Catmull-Rom cubics once per tap column of the staged window and shares
    it between pixels.  Small sources blown up 3-8x keep every block in that tier; 6 % of
    the texels are special values (infinities, NaN, signed zeros, denormals, values whose
    2x / 4x / 5x multiples overflow) so that any change of operation order would show."""
    torch = torch_cuda
    in_name, out_name = pair
    in_w, in_h, out_w, out_h = 131, 97, 640, 512
    rng = np.random.default_rng(len(in_name) * 31 + len(out_name))
    src = cases.hash_noise(in_h, in_w, 4, seed=3, planted=False)
    flat = src.reshape(-1)
    specials = np.array([np.inf, -np.inf, np.nan, -0.0, 0.0, 1e-41, -1e-45, 65504.0, 3.0e38, -3.0e38, 1.7e38, 8.6e37,
                         -6.9e37, 1.2e-38], dtype=np.float32)
    idx = rng.choice(flat.size, size=flat.size // 16, replace=False)
    flat[idx] = specials[rng.integers(0, len(specials), size=idx.size)]
    d_in = torch.from_numpy(src).cuda()
    lin, lout = cases.lenses(lrp, in_w, in_h)[in_name], cases.lenses(lrp, out_w, out_h)[out_name]
    for deg in (None, (12.0, -7.0, 3.0)):
        rot = cases.rotation(lrp, deg)
        with np.errstate(all="ignore"):
            want = oracle.reproject(lin, src, lout, out_w, out_h, 1, 2, rot, threads=8)
        for family in (2, 3):
            prev = lrp.debug_kernel(family)
            try:
                d_out = torch.full((out_h, out_w, 4), -777.0, dtype=torch.float32, device="cuda")
                lrp.reproject(lrp.Image(lin, in_w, in_h, 4, d_in), lrp.Image(lout, out_w, out_h, 4, d_out), 1, 2, rot)
                torch.cuda.synchronize()
            finally:
                lrp.debug_kernel(prev)
            cases.assert_same_bits(d_out.cpu().numpy(), want, f"{in_name}->{out_name} rot={deg} family={family}")


@pytest.mark.parametrize("chunk", range(int(__import__("os").environ.get("LRP_STRESS_CHUNKS", "8"))))
def test_kernel_families_agree_on_many_random_configurations(lrp, torch_cuda, chunk):
    """40 random configurations per chunk (sizes up to 1500, all lens pairs, unrotated / identity /
    pan-only / pitch-only / general rotations, 3-5 channels, 1-3 sub-samples), every sampler: the default
    family (all work sharing on) must produce the bytes of the one-pixel-per-lane kernel, which
    shares nothing.  No oracle involved, so the configurations can be large and many."""
    torch = torch_cuda
    rng = np.random.default_rng(777 + chunk)
    for case in range(40):
        in_w, in_h = int(rng.integers(8, 1500)), int(rng.integers(8, 1100))
        out_w, out_h = int(rng.integers(1, 1500)), int(rng.integers(1, 1100))
        c = int(rng.choice([3, 4, 4, 5]))
        ns = int(rng.choice([1, 1, 1, 1, 2, 3]))
        lin, lout = random_lens(lrp, rng, in_w, in_h), random_lens(lrp, rng, out_w, out_h)
        kind = int(rng.integers(0, 6))
        if kind == 0:
            rot = None
        elif kind == 1:
            rot = cases.rotation(lrp, (0.0, 0.0, 0.0))
        elif kind == 2:  # pan only: top / bottom mirror blocks
            rot = cases.rotation(lrp, (float(rng.choice([90.0, 180.0, 270.0, 33.0, -71.5])), 0.0, 0.0))
        elif kind == 3:  # pitch only: left / right mirror blocks (rectilinear targets)
            rot = cases.rotation(lrp, (0.0, float(rng.choice([90.0, -90.0, 180.0, 33.0, -71.5])), 0.0))
        else:
            rot = cases.rotation(lrp, tuple(float(v) for v in rng.uniform(-180.0, 180.0, size=3)))
        d_in = torch.empty((in_h, in_w, c), dtype=torch.float32, device="cuda")
        lrp.synth_fill(d_in, in_w, in_h, c, 0x1234 + case + 100 * chunk)
        for interp in (0, 1, 2):
            outs = []
            for family in (2, 0):
                prev = lrp.debug_kernel(family)
                try:
                    d_out = torch.full((out_h, out_w, c), -777.0, dtype=torch.float32, device="cuda")
                    lrp.reproject(lrp.Image(lin, in_w, in_h, c, d_in), lrp.Image(lout, out_w, out_h, c, d_out), ns, interp, rot)
                    torch.cuda.synchronize()
                finally:
                    lrp.debug_kernel(prev)
                outs.append(d_out)
            a, b = outs[0].view(torch.int32), outs[1].view(torch.int32)
            same = (a == b) | (torch.isnan(outs[0]) & torch.isnan(outs[1]))
            assert bool(same.all()), (f"chunk {chunk} case {case}: {in_w}x{in_h}x{c} -> {out_w}x{out_h}, lens {lin.type}->{lout.type}, "
                                      f"rot kind {kind}, ns={ns}, interp={interp}: {int((~same).sum())} values differ")

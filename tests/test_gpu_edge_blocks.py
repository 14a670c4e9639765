"""-m gpu: blocks of the window kernel that lie wholly beyond one SIDE of a rectilinear source (lrp_kernel_v2.h,
WinBlockT::edge): a narrow view rendered into a panorama or a fisheye frame.  There sample_bicubic
(src/reproject.cpp:109-148) clamps the four tap rows (or columns) to the first / last source row (column) and the
weight of that axis to 0 / 1; the kernel stages that one row / column instead of gathering 16 taps per pixel.
Against the oracle, bit for bit: every channel count, single launches, batches whose wavefronts walk several frames,
row bands, rotations that move the view off centre, non-finite texels on the border of the source (cubic(t, t, t, t, f)
is evaluated, not replaced by t), and sources so small that a block sees the whole border row."""
import math

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

BICUBIC = 2


def render(lrp, torch, lin, src, lout, out_w, out_h, rot, channels, what, want):
    d_in = torch.from_numpy(src).cuda()
    for family in (2, 3):  # everything on; the window kernel without its sharing paths
        prev = lrp.debug_kernel(family)
        try:
            d_out = torch.full((out_h, out_w, channels), -777.0, dtype=torch.float32, device="cuda")
            lrp.reproject(lrp.Image(lin, src.shape[1], src.shape[0], channels, d_in),
                          lrp.Image(lout, out_w, out_h, channels, d_out), 1, BICUBIC, rot)
            torch.cuda.synchronize()
        finally:
            lrp.debug_kernel(prev)
        cases.assert_same_bits(d_out.cpu().numpy(), want, f"{what}, family {family}")


@pytest.mark.parametrize("channels", [3, 4, 5])
@pytest.mark.parametrize("deg", [None, (0.0, 0.0, 0.0), (90.0, 0.0, 0.0), (0.0, 40.0, 0.0), (30.0, -15.0, 5.0), (0.0, 0.0, 90.0)])
@pytest.mark.parametrize("out_name,out_w,out_h", [("eqr_full", 512, 256), ("eqr_full", 333, 190), ("eqd180", 320, 320),
                                                  ("eqr_part", 400, 208)])
def test_narrow_view_in_a_wide_target(lrp, oracle, torch_cuda, channels, deg, out_name, out_w, out_h):
    in_w, in_h = 300, 200
    src = cases.hash_noise(in_h, in_w, channels, seed=17 * channels + out_w)
    lin, lout = cases.lenses(lrp, in_w, in_h)["rect"], cases.lenses(lrp, out_w, out_h)[out_name]
    rot = cases.rotation(lrp, deg)
    want = oracle.reproject(lin, src, lout, out_w, out_h, 1, BICUBIC, rot, threads=8)
    render(lrp, torch_cuda, lin, src, lout, out_w, out_h, rot, channels, f"rect->{out_name} {out_w}x{out_h} C={channels} rot={deg}", want)


@pytest.mark.parametrize("channels", [3, 4, 5])
def test_non_finite_texels_on_the_border(lrp, oracle, torch_cuda, channels):
    """inf, -inf, NaN, the largest finite value and denormals in the first / last rows and columns of the source: the
    clamped cubics of those texels are not the texels themselves (inf - inf, 5 * FLT_MAX ...)."""
    in_w, in_h, out_w, out_h = 240, 160, 512, 256
    src = cases.hash_noise(in_h, in_w, channels, seed=99 + channels)
    specials = np.array([np.inf, -np.inf, np.nan, 3.4028235e38, -3.4028235e38, 1e-45, -0.0, 0.0], dtype=np.float32)
    rng = np.random.default_rng(5 + channels)
    for border in (src[0], src[-1], src[:, 0], src[:, -1], src[1], src[:, -2]):
        mask = rng.random(border.shape) < 0.2
        border[mask] = specials[rng.integers(0, len(specials), size=int(mask.sum()))]
    lin = cases.lenses(lrp, in_w, in_h)["rect"]
    for out_name, deg in (("eqr_full", None), ("eqr_full", (45.0, 20.0, 0.0)), ("eqd180", (0.0, 0.0, 0.0))):
        lout = cases.lenses(lrp, out_w, out_h)[out_name]
        rot = cases.rotation(lrp, deg)
        want = oracle.reproject(lin, src, lout, out_w, out_h, 1, BICUBIC, rot, threads=8)
        render(lrp, torch_cuda, lin, src, lout, out_w, out_h, rot, channels, f"non-finite border rect->{out_name} C={channels} rot={deg}", want)


@pytest.mark.parametrize("in_w,in_h", [(4, 4), (5, 3), (17, 9), (64, 2), (2, 64), (700, 40), (40, 700)])
def test_small_and_long_sources(lrp, oracle, torch_cuda, in_w, in_h):
    """Sources a block sees whole, and sources whose border row / column is longer than one 64-texel fetch."""
    out_w, out_h = 384, 192
    lout = cases.lenses(lrp, out_w, out_h)["eqr_full"]
    for channels in (4, 5):
        src = cases.hash_noise(in_h, in_w, channels, seed=in_w * 3 + in_h)
        for focal in (18.0, 6.0, 80.0):
            lin = lrp.LensInfo.rectilinear(focal, 36.0, in_w, in_h)
            for deg in (None, (10.0, 5.0, 0.0)):
                rot = cases.rotation(lrp, deg)
                want = oracle.reproject(lin, src, lout, out_w, out_h, 1, BICUBIC, rot, threads=8)
                render(lrp, torch_cuda, lin, src, lout, out_w, out_h, rot, channels,
                       f"{in_w}x{in_h} f={focal} C={channels} rot={deg}", want)


@pytest.mark.parametrize("channels", [3, 4, 5])
def test_batches_and_row_bands(lrp, oracle, torch_cuda, channels):
    """Frames of a batch share the window plan of an edge block (1, 2, 3 and 5 frames per wavefront); a row band renders
    the same pixels as the whole frame."""
    import os

    torch = torch_cuda
    in_w, in_h, out_w, out_h, n = 200, 140, 512, 256, 5
    lin, lout = cases.lenses(lrp, in_w, in_h)["rect"], cases.lenses(lrp, out_w, out_h)["eqr_full"]
    rot = cases.rotation(lrp, (0.0, 0.0, 0.0))
    srcs = [cases.hash_noise(in_h, in_w, channels, seed=300 + 7 * k + channels) for k in range(n)]
    wants = [oracle.reproject(lin, s, lout, out_w, out_h, 1, BICUBIC, rot, threads=8) for s in srcs]
    d_in = [torch.from_numpy(s).cuda() for s in srcs]
    prev = lrp.debug_set("batch_frames", -1)
    try:
        for frames in ("1", "2", "3", "5"):
            lrp.debug_set("batch_frames", int(frames))
            d_out = [torch.full((out_h, out_w, channels), -777.0, dtype=torch.float32, device="cuda") for _ in range(n)]
            lrp.reproject_batch([lrp.Image(lin, in_w, in_h, channels, t) for t in d_in],
                                [lrp.Image(lout, out_w, out_h, channels, t) for t in d_out], 1, BICUBIC, rot)
            torch.cuda.synchronize()
            for k in range(n):
                cases.assert_same_bits(d_out[k].cpu().numpy(), wants[k], f"batch frame {k}, {frames} frames per wavefront, C={channels}")
    finally:
        lrp.debug_set("batch_frames", prev)
    d_out = torch.full((out_h, out_w, channels), -777.0, dtype=torch.float32, device="cuda")
    im_in, im_out = lrp.Image(lin, in_w, in_h, channels, d_in[0]), lrp.Image(lout, out_w, out_h, channels, d_out)
    for first, count in ((0, 40), (40, 100), (140, 116)):
        lrp.reproject_rows(im_in, im_out, 1, BICUBIC, first, count, rot)
    torch.cuda.synchronize()
    cases.assert_same_bits(d_out.cpu().numpy(), wants[0], f"row bands, C={channels}")

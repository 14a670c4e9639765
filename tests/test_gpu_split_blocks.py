"""-m gpu: blocks of the window kernel whose source window is a little larger than the LDS buffer (lrp_kernel_v2.h,
WinBlockT::split): a panorama rendered into views a few times smaller — the 2048^2 faces of an 8192^2 panorama, BASELINE
configs[4] — minifies by 1.0-1.6, the window of a 16 x 16 block no longer fits, the windows of its two 16 x 8 halves do and
are staged one after the other.  Same 4 : 1 geometry at sizes the oracle renders in seconds, every channel count, the six
face rotations and general ones, full and partial panoramas, ratios on both sides of the limit; bit for bit."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

BICUBIC = 2
FACES = [(0.0, 0.0, 0.0), (90.0, 0.0, 0.0), (180.0, 0.0, 0.0), (270.0, 0.0, 0.0), (0.0, 90.0, 0.0), (0.0, -90.0, 0.0)]


def render(lrp, torch, lin, src, lout, out_w, out_h, rot, channels, what, want):
    d_in = torch.from_numpy(src).cuda()
    for family in (2, 3):  # everything on; the window kernel without split blocks and sharing
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
@pytest.mark.parametrize("deg", FACES + [(30.0, -15.0, 5.0), None])
def test_cubemap_geometry_at_a_quarter_of_the_size(lrp, oracle, torch_cuda, channels, deg):
    in_w, in_h, face = 2048, 1024, 256  # the 8192 x 4096 -> 2048 ratio of panorama pixels per face pixel
    src = cases.hash_noise(in_h, in_w, channels, seed=11 * channels + 1)
    lin = cases.lenses(lrp, in_w, in_h)["eqr_full"]
    lout = lrp.LensInfo.rectilinear(18.0, 36.0, face, face)
    rot = cases.rotation(lrp, deg)
    want = oracle.reproject(lin, src, lout, face, face, 1, BICUBIC, rot, threads=8)
    render(lrp, torch_cuda, lin, src, lout, face, face, rot, channels, f"eqr {in_w}x{in_h} -> rect {face} C={channels} rot={deg}", want)


@pytest.mark.parametrize("in_w,in_h,out_w,out_h", [(1536, 768, 256, 256), (2048, 1024, 300, 200), (2560, 1280, 256, 256),
                                                   (3072, 1536, 256, 192), (1024, 1024, 256, 256), (2048, 512, 192, 256)])
def test_ratios_around_the_limit(lrp, oracle, torch_cuda, in_w, in_h, out_w, out_h):
    """From windows that fit whole, over windows whose halves fit, to windows whose halves do not fit either."""
    for in_name, channels, deg in (("eqr_full", 4, (15.0, 10.0, -5.0)), ("eqr_part", 3, None), ("eqr_part", 5, (0.0, 0.0, 30.0))):
        src = cases.hash_noise(in_h, in_w, channels, seed=in_w + channels)
        lin = cases.lenses(lrp, in_w, in_h)[in_name]
        lout = lrp.LensInfo.rectilinear(18.0, 36.0, out_w, out_h)
        rot = cases.rotation(lrp, deg)
        want = oracle.reproject(lin, src, lout, out_w, out_h, 1, BICUBIC, rot, threads=8)
        render(lrp, torch_cuda, lin, src, lout, out_w, out_h, rot, channels, f"{in_name} {in_w}x{in_h} -> {out_w}x{out_h} C={channels} rot={deg}", want)


@pytest.mark.parametrize("channels", [3, 4, 5])
@pytest.mark.parametrize("deg", [(0.0, 90.0, 0.0), (0.0, -90.0, 0.0), (0.0, 60.0, 0.0), (20.0, 75.0, 10.0)])
def test_pole_views_of_a_square_panorama(lrp, oracle, torch_cuda, channels, deg):
    """The bench's configs[4] geometry (a SQUARE 8192^2 panorama -> 2048^2 faces) at an eighth of the size: towards the pole
    neither the block's window nor its halves fit; the passes whose own 16 x 4 window fits stage that (pass windows), the
    rest gathers per pixel."""
    n, face = 1024, 256
    src = cases.hash_noise(n, n, channels, seed=5 * channels + 3)
    lin = cases.lenses(lrp, n, n)["eqr_full"]
    lout = lrp.LensInfo.rectilinear(18.0, 36.0, face, face)
    rot = cases.rotation(lrp, deg)
    want = oracle.reproject(lin, src, lout, face, face, 1, BICUBIC, rot, threads=8)
    render(lrp, torch_cuda, lin, src, lout, face, face, rot, channels, f"square eqr {n} -> rect {face} C={channels} rot={deg}", want)


def test_faces_through_the_multi_output_entry_point_and_row_bands(lrp, oracle, torch_cuda):
    torch = torch_cuda
    in_w, in_h, face = 2048, 1024, 256
    src = cases.hash_noise(in_h, in_w, 3, seed=77)
    lin = cases.lenses(lrp, in_w, in_h)["eqr_full"]
    lout = lrp.LensInfo.rectilinear(18.0, 36.0, face, face)
    rots = np.stack([cases.rotation(lrp, d) for d in FACES])
    wants = [oracle.reproject(lin, src, lout, face, face, 1, BICUBIC, r, threads=8) for r in rots]
    d_in = torch.from_numpy(src).cuda()
    d_outs = [torch.full((face, face, 3), -777.0, dtype=torch.float32, device="cuda") for _ in FACES]
    lrp.reproject_multi(lrp.Image(lin, in_w, in_h, 3, d_in), [lrp.Image(lout, face, face, 3, t) for t in d_outs], 1, BICUBIC, rots)
    torch.cuda.synchronize()
    for d, t, w in zip(FACES, d_outs, wants):
        cases.assert_same_bits(t.cpu().numpy(), w, f"multi face {d}")
    d_out = torch.full((face, face, 3), -777.0, dtype=torch.float32, device="cuda")
    im_in, im_out = lrp.Image(lin, in_w, in_h, 3, d_in), lrp.Image(lout, face, face, 3, d_out)
    for first, count in ((0, 24), (24, 100), (124, 132)):
        lrp.reproject_rows(im_in, im_out, 1, BICUBIC, first, count, rots[4])
    torch.cuda.synchronize()
    cases.assert_same_bits(d_out.cpu().numpy(), wants[4], "row bands of the pole face")

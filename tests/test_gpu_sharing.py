"""-m gpu: the three work-sharing paths of the tile / window kernels against the oracle, bit
for bit, at the edges of their applicability.

* column-separable source x (lrp_tables.hip build_xsep_kernel): no rotation, identity, pan-only
  rotations (cubemap side faces), and rotations for which it must NOT be used;
* dropped identity rotation (ray tables without -0.0f / non-finite values);
* mirrored blocks of the window kernel: odd and even sizes (the centre column / row is its own
  mirror image), non-square and tiny images, every source lens, equidistant target;
* shared tap-column coefficients: magnified windows around the image centre and at the seam.
Each case renders with the default family (everything on), with the window kernel's sharing
switched off (family 3) and with the one-pixel-per-lane kernel (family 0)."""
import math

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def render_all(lrp, torch, lin, src, lout, out_w, out_h, interp, rot, what, want, channels=4, families=(2, 3, 0)):
    d_in = torch.from_numpy(src).cuda()
    for family in families:
        prev = lrp.debug_kernel(family)
        try:
            d_out = torch.full((out_h, out_w, channels), -777.0, dtype=torch.float32, device="cuda")
            lrp.reproject(lrp.Image(lin, src.shape[1], src.shape[0], channels, d_in),
                          lrp.Image(lout, out_w, out_h, channels, d_out), 1, interp, rot)
            torch.cuda.synchronize()
        finally:
            lrp.debug_kernel(prev)
        cases.assert_same_bits(d_out.cpu().numpy(), want, f"{what}, family {family}")


SIZES = [(256, 192), (255, 193), (257, 191), (64, 48), (33, 17), (16, 16), (5, 3), (130, 1), (1, 77)]


@pytest.mark.parametrize("out_w,out_h", SIZES)
@pytest.mark.parametrize("in_name,out_name", [("eqd180", "rect"), ("eqr_full", "rect"), ("rect_tele", "rect"),
                                              ("eqd180", "eqd120"), ("rect", "eqd120"), ("eqr_part", "rect")])
def test_mirrored_blocks_all_sizes(lrp, oracle, torch_cuda, out_w, out_h, in_name, out_name):
    """No rotation and the identity matrix: the window kernel renders a block and its three
    mirror images from one evaluation of the coordinate math (where the lens pair allows)."""
    in_w, in_h = 150, 110
    src = cases.hash_noise(in_h, in_w, 4, seed=out_w * 7 + out_h)
    lin, lout = cases.lenses(lrp, in_w, in_h)[in_name], cases.lenses(lrp, out_w, out_h)[out_name]
    for deg in (None, (0.0, 0.0, 0.0)):
        rot = cases.rotation(lrp, deg)
        want = oracle.reproject(lin, src, lout, out_w, out_h, 1, 2, rot, threads=8)
        render_all(lrp, torch_cuda, lin, src, lout, out_w, out_h, 2, rot, f"{in_name}->{out_name} {out_w}x{out_h} rot={deg}", want)


@pytest.mark.parametrize("deg", [(90.0, 0.0, 0.0), (180.0, 0.0, 0.0), (270.0, 0.0, 0.0), (37.5, 0.0, 0.0), (-120.0, 0.0, 0.0),
                                 (0.0, 90.0, 0.0), (0.0, -90.0, 0.0), (0.0, 0.0, 45.0), (20.0, 1e-3, 0.0), (20.0, 0.0, 1e-3)])
@pytest.mark.parametrize("interp", [0, 1, 2])
def test_column_separable_source_x_and_its_limits(lrp, oracle, torch_cuda, deg, interp):
    """Pan-only rotations keep the ray's x and z independent of the row (column table used);
    pitch / roll do not (table must not be used).  Equirectangular and rectilinear sources,
    rectilinear and equirectangular targets, RGB and RGBA."""
    in_w, in_h, out_w, out_h = 384, 192, 200, 136
    rot = cases.rotation(lrp, deg)
    for in_name, out_name, c in (("eqr_full", "rect", 4), ("eqr_part", "rect", 3), ("rect", "eqr_full", 4),
                                 ("eqr_full", "eqr_part", 4), ("rect_tele", "rect", 5)):
        src = cases.hash_noise(in_h, in_w, c, seed=interp + 10 * c)
        lin, lout = cases.lenses(lrp, in_w, in_h)[in_name], cases.lenses(lrp, out_w, out_h)[out_name]
        want = oracle.reproject(lin, src, lout, out_w, out_h, 1, interp, rot, threads=8)
        fam = (2, 3, 0) if c == 4 else (2, 0)
        render_all(lrp, torch_cuda, lin, src, lout, out_w, out_h, interp, rot, f"{in_name}->{out_name} C={c} rot={deg} interp={interp}",
                   want, channels=c, families=fam)


def test_identity_is_not_dropped_when_the_tables_hold_negative_zero(lrp, oracle, torch_cuda):
    """A negative focal length turns the centre column's +0 into -0: R v then differs from v in
    the sign of a zero, and the identity matrix must be applied as given."""
    in_w, in_h, out_w, out_h = 96, 64, 63, 41  # odd output size: centre column and row are exactly 0
    L = lrp.LensInfo
    lout = L.rectilinear(-18.0, 36.0, out_w, out_h)
    src = cases.hash_noise(in_h, in_w, 4, seed=5)
    for lin in (L.equirectangular(), L.equidistant(math.pi), L.rectilinear(24.0, 36.0, in_w, in_h)):
        for deg in (None, (0.0, 0.0, 0.0), (90.0, 0.0, 0.0)):
            rot = cases.rotation(lrp, deg)
            for interp in (0, 2):
                with np.errstate(all="ignore"):
                    want = oracle.reproject(lin, src, lout, out_w, out_h, 1, interp, rot)
                render_all(lrp, torch_cuda, lin, src, lout, out_w, out_h, interp, rot, f"negative focal, lens {lin.type}, rot={deg}", want)


def test_mirrored_4k_frame_with_odd_size_matches_pixel_kernel(lrp, torch_cuda):
    """4095 x 4097 output (odd both ways) from a 4K fisheye frame: mirrored window kernel against
    the one-pixel-per-lane kernel over the whole frame."""
    torch = torch_cuda
    n = 4096
    d_in = torch.empty((n, n, 4), dtype=torch.float32, device="cuda")
    lrp.synth_fill(d_in, n, n, 4, 0x5EED0042)
    out_w, out_h = 4095, 4097
    lin, lout = lrp.LensInfo.equidistant(math.pi), lrp.LensInfo.rectilinear(18.0, 36.0, out_w, out_h)
    outs = []
    for family in (2, 0):
        prev = lrp.debug_kernel(family)
        try:
            d_out = torch.full((out_h, out_w, 4), -1.0, dtype=torch.float32, device="cuda")
            lrp.reproject(lrp.Image(lin, n, n, 4, d_in), lrp.Image(lout, out_w, out_h, 4, d_out), 1, 2, None)
            torch.cuda.synchronize()
        finally:
            lrp.debug_kernel(prev)
        outs.append(d_out)
    assert bool(torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)))


@pytest.mark.parametrize("out_w,out_h", [(200, 136), (201, 137), (65, 33), (7, 5)])
@pytest.mark.parametrize("deg", [None, (30.0, -15.0, 5.0), (0.0, 90.0, 0.0), (180.0, 0.0, 0.0)])
def test_mirrored_rays_equidistant_target(lrp, oracle, torch_cuda, out_w, out_h, deg):
    """Equidistant target: the four mirror pixels share the ray through the output lens under any
    rotation (tile kernels); the centre column / row of an odd-sized image is its own mirror
    image and keeps its +0 ray component."""
    in_w, in_h = 300, 160
    rot = cases.rotation(lrp, deg)
    lout = cases.lenses(lrp, out_w, out_h)["eqd180"]
    for in_name in ("eqr_full", "eqr_part", "rect", "eqd120"):
        lin = cases.lenses(lrp, in_w, in_h)[in_name]
        for c, interp in ((4, 0), (4, 1), (3, 2), (5, 1)):
            src = cases.hash_noise(in_h, in_w, c, seed=c + interp)
            with np.errstate(all="ignore"):
                want = oracle.reproject(lin, src, lout, out_w, out_h, 1, interp, rot, threads=8)
            render_all(lrp, torch_cuda, lin, src, lout, out_w, out_h, interp, rot,
                       f"{in_name}->eqd180 {out_w}x{out_h} C={c} interp={interp} rot={deg}", want, channels=c, families=(2, 0))


@pytest.mark.parametrize("span", [None, (-1.25, 1.25, -0.5, 0.5), (-0.5, 0.5, -1.0, 1.0), (-math.pi / 2, math.pi / 2, -math.pi, math.pi),
                                  (-1.0, 1.0, -2.0, 2.0)])
def test_alias_pairs_plain_and_mirrored_strips(lrp, oracle, torch_cuda, span):
    """Rectilinear view into a panorama: the window kernel deals the strip that renders the view and the
    strip that renders its copy behind the camera to neighbouring workgroups (plain strips: tile (t, r)
    with (t + tiles_x/2, tiles_y-1-r), walked bottom-up; mirrored strips, which some symmetric
    panoramas take: columns from both ends inwards, mirror images in reverse).  A permutation of the
    work only: every channel count, even and odd tile counts, against the oracle."""
    L = lrp.LensInfo
    in_w, in_h = 300, 200
    lin = L.rectilinear(18.0, 36.0, in_w, in_h)
    for out_w, out_h in ((512, 256), (496, 250), (528, 272), (272, 80)):  # 32 / 31 / 33 / 17 tile columns
        lout = L.equirectangular() if span is None else L.equirectangular(*span)
        for c in (4, 5, 3):
            src = cases.hash_noise(in_h, in_w, c, seed=out_w + c)
            with np.errstate(all="ignore"):
                want = oracle.reproject(lin, src, lout, out_w, out_h, 1, 2, None, threads=8)
            render_all(lrp, torch_cuda, lin, src, lout, out_w, out_h, 2, None, f"rect->eqr{span} {out_w}x{out_h} C={c}", want,
                       channels=c, families=(2,))


# ---- one-axis mirror modes of the window kernel (lrp_kernel_v2.h QMode 2 / 3) -----------------------------------------
PAN_DEG = [(90.0, 0.0, 0.0), (180.0, 0.0, 0.0), (270.0, 0.0, 0.0), (37.5, 0.0, 0.0), (-120.0, 0.0, 0.0), (1e-4, 0.0, 0.0)]
PITCH_DEG = [(0.0, 90.0, 0.0), (0.0, -90.0, 0.0), (0.0, 33.0, 0.0), (0.0, -71.5, 0.0), (0.0, 180.0, 0.0), (0.0, 1e-4, 0.0)]


@pytest.mark.parametrize("out_w,out_h", [(256, 192), (255, 193), (64, 48), (33, 17), (16, 16), (5, 3), (130, 1), (1, 77)])
@pytest.mark.parametrize("deg", PAN_DEG + PITCH_DEG)
def test_one_axis_mirror_modes_all_cells_and_sizes(lrp, oracle, torch_cuda, out_w, out_h, deg):
    """Pan-only rotations render a block and its top / bottom mirror image from one evaluation of stage 1
    (rectilinear / equirectangular lenses on both sides), pitch-only rotations a block and its left / right mirror
    image (rectilinear target, every source).  Odd sizes: the centre row / column is its own mirror image.  Each case
    against the oracle with the mode on (family 2), with every sharing path off (3) and per pixel (0)."""
    in_w, in_h = 150, 110
    rot = cases.rotation(lrp, deg)
    pan = deg[1] == 0.0
    cells = ([("eqr_full", "rect"), ("eqr_part", "rect"), ("rect_tele", "rect"), ("rect", "eqr_full"), ("eqr_full", "eqr_part")] if pan
             else [("eqr_full", "rect"), ("eqr_part", "rect"), ("rect_tele", "rect"), ("eqd180", "rect")])
    for in_name, out_name in cells:
        src = cases.hash_noise(in_h, in_w, 4, seed=out_w * 7 + out_h)
        lin, lout = cases.lenses(lrp, in_w, in_h)[in_name], cases.lenses(lrp, out_w, out_h)[out_name]
        with np.errstate(all="ignore"):
            want = oracle.reproject(lin, src, lout, out_w, out_h, 1, 2, rot, threads=8)
        render_all(lrp, torch_cuda, lin, src, lout, out_w, out_h, 2, rot, f"{in_name}->{out_name} {out_w}x{out_h} rot={deg}", want)


@pytest.mark.parametrize("channels", [3, 5])
@pytest.mark.parametrize("deg", [(90.0, 0.0, 0.0), (-37.5, 0.0, 0.0), (0.0, 90.0, 0.0), (0.0, -40.0, 0.0)])
def test_one_axis_mirror_modes_rgb_and_rgbaz(lrp, oracle, torch_cuda, channels, deg):
    in_w, in_h, out_w, out_h = 384, 192, 201, 137
    rot = cases.rotation(lrp, deg)
    pan = deg[1] == 0.0
    for in_name, out_name in ([("eqr_full", "rect"), ("rect", "eqr_full"), ("rect_tele", "rect")] if pan
                              else [("eqr_full", "rect"), ("eqd180", "rect"), ("rect_tele", "rect")]):
        src = cases.hash_noise(in_h, in_w, channels, seed=channels)
        lin, lout = cases.lenses(lrp, in_w, in_h)[in_name], cases.lenses(lrp, out_w, out_h)[out_name]
        want = oracle.reproject(lin, src, lout, out_w, out_h, 1, 2, rot, threads=8)
        render_all(lrp, torch_cuda, lin, src, lout, out_w, out_h, 2, rot, f"{in_name}->{out_name} C={channels} rot={deg}", want,
                   channels=channels, families=(2, 0))


@pytest.mark.parametrize("deg", [(20.0, 1e-3, 0.0), (20.0, 0.0, 1e-3), (1e-3, 40.0, 0.0), (0.0, 40.0, 1e-3), (0.0, 0.0, 45.0)])
def test_one_axis_mirror_modes_are_not_used_beyond_their_rotations(lrp, oracle, torch_cuda, deg):
    """A trace of pitch or roll next to a pan (or of pan / roll next to a pitch) breaks the symmetry: plain blocks."""
    in_w, in_h, out_w, out_h = 300, 160, 201, 137
    rot = cases.rotation(lrp, deg)
    for in_name, out_name in (("eqr_full", "rect"), ("rect_tele", "rect"), ("eqd180", "rect"), ("rect", "eqr_full")):
        src = cases.hash_noise(in_h, in_w, 4, seed=3)
        lin, lout = cases.lenses(lrp, in_w, in_h)[in_name], cases.lenses(lrp, out_w, out_h)[out_name]
        want = oracle.reproject(lin, src, lout, out_w, out_h, 1, 2, rot, threads=8)
        render_all(lrp, torch_cuda, lin, src, lout, out_w, out_h, 2, rot, f"{in_name}->{out_name} rot={deg}", want, families=(2, 0))


def test_one_axis_mirror_modes_negative_zero_in_the_tables(lrp, oracle, torch_cuda):
    """A negative focal length puts -0.0f into the centre column / row of an odd-sized target: the tables are then not
    'plain', no column table is built and the rows-only mode must not be chosen; the columns-only mode sees a centre
    column that is its own mirror image."""
    in_w, in_h, out_w, out_h = 96, 64, 63, 41
    L = lrp.LensInfo
    lout = L.rectilinear(-18.0, 36.0, out_w, out_h)
    src = cases.hash_noise(in_h, in_w, 4, seed=5)
    for lin in (L.equirectangular(), L.equidistant(math.pi), L.rectilinear(24.0, 36.0, in_w, in_h)):
        for deg in ((90.0, 0.0, 0.0), (0.0, 90.0, 0.0), (0.0, -30.0, 0.0)):
            rot = cases.rotation(lrp, deg)
            with np.errstate(all="ignore"):
                want = oracle.reproject(lin, src, lout, out_w, out_h, 1, 2, rot)
            render_all(lrp, torch_cuda, lin, src, lout, out_w, out_h, 2, rot, f"negative focal, lens {lin.type}, rot={deg}", want)


def test_hand_made_pan_and_pitch_matrices(lrp, oracle, torch_cuda):
    """Matrices a caller may hand over directly (signed zeros, a pan by exactly 90 degrees with an exact 0 / 1 pattern,
    mirror-like matrices with a negative diagonal entry) — whatever mode the host picks, the bits are the oracle's."""
    in_w, in_h, out_w, out_h = 300, 160, 137, 95
    mats = [
        [0.0, 0.0, 1.0, 0.0, 1.0, 0.0, -1.0, 0.0, 0.0],        # exact quarter turn about y
        [0.0, -0.0, 1.0, -0.0, 1.0, 0.0, -1.0, 0.0, -0.0],     # the same with signed zeros
        [0.6, 0.0, 0.8, 0.0, -1.0, 0.0, -0.8, 0.0, 0.6],       # pan with a flipped vertical axis
        [1.0, 0.0, 0.0, 0.0, 0.6, -0.8, 0.0, 0.8, 0.6],        # pitch
        [-1.0, -0.0, 0.0, 0.0, 0.6, -0.8, -0.0, 0.8, 0.6],     # pitch with a flipped horizontal axis
        [1.0, 0.0, 0.0, 0.0, 0.0, -1.0, 0.0, 1.0, 0.0],        # exact quarter turn about x (R4 == R8 == 0)
        [1.0, 0.0, 0.0, 0.0, 1e-30, 0.0, 0.0, 0.0, 1.0],       # degenerate: tiny R4
    ]
    for m in mats:
        rot = np.array(m, dtype=np.float32)
        for in_name, out_name in (("eqr_full", "rect"), ("rect_tele", "rect"), ("eqd180", "rect"), ("rect", "eqr_part")):
            src = cases.hash_noise(in_h, in_w, 4, seed=9)
            lin, lout = cases.lenses(lrp, in_w, in_h)[in_name], cases.lenses(lrp, out_w, out_h)[out_name]
            with np.errstate(all="ignore"):
                want = oracle.reproject(lin, src, lout, out_w, out_h, 1, 2, rot, threads=8)
            render_all(lrp, torch_cuda, lin, src, lout, out_w, out_h, 2, rot, f"{in_name}->{out_name} matrix {m}", want, families=(2, 0))


@pytest.mark.parametrize("out_w,out_h", [(200, 136), (201, 137), (65, 33), (16, 16), (7, 5), (130, 1), (1, 77)])
@pytest.mark.parametrize("deg", [None, (0.0, 0.0, 0.0), (30.0, -15.0, 5.0), (0.0, 90.0, 0.0), (180.0, 0.0, 0.0)])
def test_shared_rays_equidistant_target_bicubic(lrp, oracle, torch_cuda, out_w, out_h, deg):
    """Equidistant target, bicubic (window kernel, QMode 4): the four mirror pixels share the ray through the output lens
    under any rotation, each image runs its own rotation and source lens; the centre column / row of an odd-sized image
    is its own mirror image.  Every source lens, RGBA / RGB / RGBAZ."""
    in_w, in_h = 300, 160
    rot = cases.rotation(lrp, deg)
    lout = cases.lenses(lrp, out_w, out_h)["eqd180"]
    for in_name, c in (("eqr_full", 4), ("eqr_part", 3), ("rect", 4), ("eqd120", 5), ("eqd180", 4)):
        src = cases.hash_noise(in_h, in_w, c, seed=out_w + 3 * c)
        lin = cases.lenses(lrp, in_w, in_h)[in_name]
        with np.errstate(all="ignore"):
            want = oracle.reproject(lin, src, lout, out_w, out_h, 1, 2, rot, threads=8)
        render_all(lrp, torch_cuda, lin, src, lout, out_w, out_h, 2, rot, f"{in_name}->eqd180 C={c} {out_w}x{out_h} rot={deg}", want,
                   channels=c, families=(2, 3, 0) if c == 4 else (2, 0))

"""-m gpu: tap DMA of the big-window variant (csrc/lrp_win_kernel.h request_taps / the rolled loop over the passes of a block with
nothing staged).

A rectilinear view rendered into a panorama of far fewer pixels: 16 output pixels span more than 128 source texels, so no
block, half block or 16 x 4 pass has a window that fits the LDS buffer — every pass that lies in the source whole fetches its
taps quad by quad (knob win_tapdma 1) or gathers them per lane (0), the passes that touch the border of the view gather.  Both
settings must reproduce the oracle (src/reproject.cpp:92-148: the same 16 taps, the same operations) bit for bit: RGB, RGBA and
RGBAZ (whose taps fill the buffer to the last byte, the exchange buffer of its stores included), with the fused tonemap, single
launches and a batch, enumerated and listed launches."""
import json
import os

import numpy as np
import pytest

import cases
import fullframe_cases as ffc
import golden_cases

pytestmark = pytest.mark.gpu
USES_GEO_CACHE = True

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "fullframe_golden.json")) as _f:
    FULL = json.load(_f)


@pytest.fixture(autouse=True)
def _fresh_cache(lrp):
    lrp.debug_set("geo_cache", 1)
    prev_lists = lrp.debug_set("geo_lists", 1)
    prev_tap = lrp.debug_set("win_tapdma", 1)
    prev_big = lrp.debug_set("geo_big", 1)
    lrp.geometry_cache_configure(1 << 30, 1)
    lrp.release_cached_tables()
    yield
    lrp.debug_set("geo_lists", prev_lists)
    lrp.debug_set("win_tapdma", prev_tap)
    lrp.debug_set("geo_big", prev_big)
    lrp.geometry_cache_configure(1 << 30, 1)
    lrp.release_cached_tables()


CASES = [
    # (source w, h, output w, h, source lens, rotation, post)
    (1500, 1100, 160, 90, "rect", "none", None),        # ~17 x 8 source texels per output pixel
    (2000, 1500, 300, 140, "rect", "r30", (1.5, 3.0)),  # a general rotation: the windows are skewed as well as wide
    (1900, 900, 333, 201, "rect", "pitch90", None),      # the view around a pole: passes of every shape
    (1280, 1280, 210, 130, "rect_tele", "none", (0.75, 2.0)),  # a narrow view: mostly corner blocks, the rest minified 20 x
]


@pytest.mark.parametrize("channels", [3, 4, 5])
@pytest.mark.parametrize("lists", [0, 2])
def test_minified_views_against_the_live_oracle(lrp, oracle, torch_cuda, channels, lists):
    torch = torch_cuda
    lrp.debug_set("geo_lists", lists)
    for iw, ih, ow, oh, in_name, rot_name, post in CASES:
        src = cases.hash_noise(ih, iw, channels, seed=0x7A9 + ow + channels, planted=True)
        lin = cases.lenses(lrp, iw, ih)[in_name]
        lout = cases.lenses(lrp, ow, oh)["eqr_full"]
        rot = cases.rotation(lrp, golden_cases.ROTS[rot_name])
        want = oracle.reproject(lin, src, lout, ow, oh, 1, 2, rot)
        if post:
            want = oracle.post_process(want, *post)
        d_in = torch.from_numpy(src).cuda()
        img_in = lrp.Image(lin, iw, ih, channels, d_in)

        def render(batch=0):
            outs = [torch.full((oh, ow, channels), -12345.0, dtype=torch.float32, device="cuda") for _ in range(max(batch, 1))]
            if batch:
                lrp.reproject_batch([img_in] * batch, [lrp.Image(lout, ow, oh, channels, o) for o in outs], 1, 2, rot, post=post)
            else:
                lrp.reproject(img_in, lrp.Image(lout, ow, oh, channels, outs[0]), 1, 2, rot, post=post)
            torch.cuda.synchronize()
            return [o.cpu().numpy() for o in outs]

        what = f"{in_name} {iw}x{ih} -> eqr_full {ow}x{oh} C={channels} {rot_name} post={post} lists={lists}"
        cases.assert_same_bits(render()[0], want, "the launch that fills the entry, " + what)
        for tap in (1, 0, 1):
            lrp.debug_set("win_tapdma", tap)
            cases.assert_same_bits(render()[0], want, f"a launch that reads the entry, win_tapdma={tap}, " + what)
        for i, got in enumerate(render(batch=3)):
            cases.assert_same_bits(got, want, f"frame {i} of a batch of three, " + what)


FOCAL_MM = {"rect": 18.0, "rect_tele": 50.0}  # cases.lenses: 36 mm sensors


def test_the_source_is_minified_beyond_every_window():
    """What the cases above rely on: 16 output pixels of the view's centre span more than 128 source texels (the widest window
    the big-window variant stages), so the in-view passes cannot come from a window."""
    for iw, ih, ow, oh, in_name, rot_name, post in CASES:
        f_px = FOCAL_MM[in_name] / 36.0 * iw  # texels per unit of tan(angle)
        texels_per_pixel = f_px * np.pi / 180.0 * 360.0 / ow  # a panorama column is 360 / ow degrees
        assert 16 * texels_per_pixel > 128, (in_name, iw, ow, texels_per_pixel)


PANORAMA_CASES = [
    # (source w, h, source lens, output w, h, output lens, rotation, post)
    (1024, 512, "eqr_full", 200, 200, "rect", "pitch90", None),        # a pole in view: the rows of the panorama converge
    (2048, 1024, "eqr_full", 128, 96, "rect", "r30", (1.5, 3.0)),      # minified four times, the seam of the panorama in view
    (1536, 768, "eqr_full", 160, 160, "eqd180", "pitch90", None),      # a fisheye frame around the pole
    (1200, 700, "eqr_part", 150, 130, "rect_tele", "none", (0.75, 2.0)),  # a partial panorama (no wrap-around), minified
    (1400, 1000, "rect", 180, 140, "eqd180", "none", None),            # a rectilinear view into a fisheye frame: corner blocks + minified in-view blocks
]


@pytest.mark.parametrize("channels", [3, 4, 5])
def test_big_window_variant_of_every_source_against_the_live_oracle(lrp, oracle, torch_cuda, channels):
    """geo_big 2 sends every launch that reads a geometry-cache entry through the big-window variant of its source (rectilinear,
    panorama, wrapping panorama): windows of single passes, tap DMA, per-pixel gathers at the seam — against the
    four-wavefront instantiations (geo_big 0), the automatic choice (1) and the oracle."""
    torch = torch_cuda
    for iw, ih, in_name, ow, oh, out_name, rot_name, post in PANORAMA_CASES:
        src = cases.hash_noise(ih, iw, channels, seed=0xB16 + ow + channels, planted=True)
        lin = cases.lenses(lrp, iw, ih)[in_name]
        lout = cases.lenses(lrp, ow, oh)[out_name]
        rot = cases.rotation(lrp, golden_cases.ROTS[rot_name])
        want = oracle.reproject(lin, src, lout, ow, oh, 1, 2, rot)
        if post:
            want = oracle.post_process(want, *post)
        d_in = torch.from_numpy(src).cuda()
        img_in = lrp.Image(lin, iw, ih, channels, d_in)

        def render(batch=0):
            outs = [torch.full((oh, ow, channels), -12345.0, dtype=torch.float32, device="cuda") for _ in range(max(batch, 1))]
            if batch:
                lrp.reproject_batch([img_in] * batch, [lrp.Image(lout, ow, oh, channels, o) for o in outs], 1, 2, rot, post=post)
            else:
                lrp.reproject(img_in, lrp.Image(lout, ow, oh, channels, outs[0]), 1, 2, rot, post=post)
            torch.cuda.synchronize()
            return [o.cpu().numpy() for o in outs]

        what = f"{in_name} {iw}x{ih} -> {out_name} {ow}x{oh} C={channels} {rot_name} post={post}"
        cases.assert_same_bits(render()[0], want, "the launch that fills the entry, " + what)
        for big in (2, 0, 1):
            lrp.debug_set("geo_big", big)
            b0 = lrp.debug_set("big_launches", -1)
            cases.assert_same_bits(render()[0], want, f"a launch that reads the entry, geo_big={big}, " + what)
            if big != 1:
                assert lrp.debug_set("big_launches", -1) == b0 + (1 if big == 2 else 0), what
            for i, got in enumerate(render(batch=3)):
                cases.assert_same_bits(got, want, f"frame {i} of a batch of three, geo_big={big}, " + what)


@pytest.mark.parametrize("face,big", [(4, True), (1, False)])
def test_the_census_picks_the_variant_per_cubemap_face(lrp, torch_cuda, face, big):
    """BASELINE configs[4]: the pole faces of the 8192^2 -> 2048^2 cubemap have windows no 10 KiB buffer stages (the census of
    their entry says so: the big-window variant renders them), the side faces have none (the four-wavefront instantiation
    stays); either way the face reproduces the committed oracle digest."""
    torch = torch_cuda
    name = f"config4_8k_rgb_face{face}"
    case, want = ffc.frame_cases()[name], FULL["frames"][name]
    n, m, c = case["size"], case["out_size"], case["c"]
    lin, lout = cases.lenses(lrp, n, n)[case["inp"]], cases.lenses(lrp, m, m)[case["out"]]

    def frame(seed):
        d_in = torch.empty((n, n, c), dtype=torch.float32, device="cuda")
        lrp.synth_fill(d_in, n, n, c, seed, case.get("depth", -1))
        d_out = torch.full((m, m, c), -12345.0, dtype=torch.float32, device="cuda")
        lrp.reproject(lrp.Image(lin, n, n, c, d_in), lrp.Image(lout, m, m, c, d_out), 1, case["interp"], cases.rotation(lrp, case["deg"]))
        torch.cuda.synchronize()
        return d_out

    frame(case["seed"] + 99)  # fills the entry; its census follows the records to the host
    b0 = lrp.debug_set("big_launches", -1)
    d_out = frame(case["seed"])
    assert lrp.debug_set("big_launches", -1) == b0 + (1 if big else 0)
    sha, bands, n_nan = ffc.frame_digests(d_out.cpu().numpy())
    bad = [b for b in range(ffc.BANDS) if bands[b] != want["bands"][b]]
    assert not bad, f"{name}: row bands {bad} of {ffc.BANDS} differ from the committed oracle digest"
    assert sha == want["sha256"] and n_nan == want["nan"]

"""-m gpu: rendering by block class (csrc/lrp_geo_lists.hip, lrp_params.h "Block lists").

Once the block lists of a geometry-cache entry are known, a bicubic launch of that geometry is two kernels: the fill kernel
writes the corner runs (every pixel the one clamped corner texel, src/reproject.cpp:114-131) and the window kernel walks the
work list, which holds every other block.  Same bits as the reference: every comparison is against the live oracle or the
committed whole-frame digests; `listed_launches` proves that the listed path is the one that ran."""
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
    prev = lrp.debug_set("geo_lists", 2)
    lrp.geometry_cache_configure(1 << 30, 1)
    lrp.release_cached_tables()
    yield
    lrp.debug_set("geo_lists", prev)
    lrp.debug_set("geo_fill_stream", 0)
    lrp.geometry_cache_configure(1 << 30, 1)
    lrp.release_cached_tables()


def _setup(lrp, torch, case):
    src = cases.hash_noise(case["ih"], case["iw"], case["c"], seed=case["seed"], planted=True)
    lin = cases.lenses(lrp, case["iw"], case["ih"])[case["inp"]]
    lout = cases.lenses(lrp, case["ow"], case["oh"])[case["out"]]
    rot = cases.rotation(lrp, golden_cases.ROTS[case["rot"]])
    return src, lin, lout, rot


def _listed(lrp):
    return lrp.debug_set("listed_launches", -1)


@pytest.mark.parametrize("channels", [3, 4, 5])
@pytest.mark.parametrize("fill_stream", [0, 1])
def test_listed_launches_against_the_live_oracle(lrp, oracle, torch_cuda, channels, fill_stream):
    """Views with many, few and no corner blocks; sizes that are not multiples of the block, of a run (16 blocks) or of four
    pixels (the partial last vector of an RGB / RGBAZ row segment); every source that has corners (rectilinear, fisheye,
    partial panorama) and one that has none (full panorama: the work list is the whole frame); fused tonemap; single
    launches and a batch of five; the fill kernel in front of the window kernel and beside it on a side stream."""
    torch = torch_cuda
    lrp.debug_set("geo_fill_stream", fill_stream)
    todo = [((96, 64, 80, 200), "eqr_full", "rect_tele", "none", None), ((64, 64, 147, 331), "eqr_full", "rect_tele", "r30", (1.5, 3.0)),
            ((200, 120, 64, 130), "eqr_full", "rect", "pitch90", (0.75, 2.0)), ((90, 70, 33, 97), "eqr_full", "rect", "pan180", None),
            ((120, 90, 531, 77), "eqr_full", "rect", "none", None), ((80, 60, 290, 150), "eqd180", "rect_tele", "r30", (2.0, 4.0)),
            ((100, 100, 301, 203), "rect", "rect_tele", "none", None), ((64, 48, 270, 131), "eqr_full", "eqd180", "r30", None),
            ((70, 50, 259, 120), "eqr_full", "eqr_part", "none", (1.25, 3.0)), ((128, 64, 140, 90), "rect", "eqr_full", "r30", None)]
    for (iw, ih, ow, oh), out_name, in_name, rot_name, post in todo:
        case = dict(iw=iw, ih=ih, ow=ow, oh=oh, out=out_name, inp=in_name, interp=2, c=channels, ns=1, rot=rot_name, seed=0xB10C + ow + channels)
        src, lin, lout, rot = _setup(lrp, torch, case)
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

        what = f"{in_name} {iw}x{ih} -> {out_name} {ow}x{oh} C={channels} {rot_name} post={post} fill_stream={fill_stream}"
        n0 = _listed(lrp)
        lists = 1 if in_name.startswith("rect") else 0  # (the lists are built for the rectilinear source's kernels)
        cases.assert_same_bits(render()[0], want, "the launch that fills the entry and builds its lists, " + what)
        assert _listed(lrp) == n0, "the filling launch enumerates the frame"
        cases.assert_same_bits(render()[0], want, "listed launch, " + what)
        assert _listed(lrp) == n0 + lists, "the second launch of a geometry is rendered by block class: " + what
        for got in render(batch=5):
            cases.assert_same_bits(got, want, "listed batch of five, " + what)
        assert _listed(lrp) == n0 + 2 * lists
        for knob_name in ("geo_fill_fused", "geo_lists"):  # the fill kernel instead of a share per wavefront; no lists at all
            prev = lrp.debug_set(knob_name, 0)
            cases.assert_same_bits(render()[0], want, f"{knob_name} 0, " + what)
            lrp.debug_set(knob_name, prev)
        assert _listed(lrp) == n0 + 3 * lists


def test_all_corner_one_block_and_no_corner_frames(lrp, oracle, torch_cuda):
    """The lists at their extremes: a frame of nothing but corner blocks (empty work list: no window launch at all), a frame of
    ONE block, a frame without any corner block (no runs: no fill launch), a frame whose only non-corner blocks are edge blocks."""
    torch = torch_cuda
    tele = lrp.LensInfo.rectilinear(4000.0, 36.0, 64, 64)  # a view 0.5 degrees wide
    for (ow, oh), lin, lout, rot_name, c in (((256, 128), tele, lrp.LensInfo.equirectangular(0.5, 2.5, 0.3, 0.9), "none", 4),  # looks away: all corners
                                             ((256, 128), tele, lrp.LensInfo.equirectangular(0.5, 2.5, 0.3, 0.9), "none", 5),
                                             ((13, 9), lrp.LensInfo.rectilinear(18.0, 36.0, 64, 64), lrp.LensInfo.equirectangular(), "none", 3),
                                             ((16, 16), tele, lrp.LensInfo.equirectangular(), "r30", 5),
                                             ((96, 96), lrp.LensInfo.rectilinear(18.0, 36.0, 64, 64), lrp.LensInfo.rectilinear(60.0, 36.0, 96, 96), "none", 4),  # inside the view: no corner
                                             ((200, 40), tele, lrp.LensInfo.equirectangular(-0.001, 0.001, -1.2, 1.2), "none", 4)):  # a sliver: rows above / below the view
        src = cases.hash_noise(64, 64, c, seed=0xA11C + ow + c, planted=True)
        rot = cases.rotation(lrp, golden_cases.ROTS[rot_name])
        want = oracle.reproject(lin, src, lout, ow, oh, 1, 2, rot)
        d_in = torch.from_numpy(src).cuda()
        n0 = _listed(lrp)
        for k in range(3):
            d_out = torch.full((oh, ow, c), -12345.0, dtype=torch.float32, device="cuda")
            lrp.reproject(lrp.Image(lin, 64, 64, c, d_in), lrp.Image(lout, ow, oh, c, d_out), 1, 2, rot)
            torch.cuda.synchronize()
            cases.assert_same_bits(d_out.cpu().numpy(), want, f"launch {k}: {ow}x{oh} C={c} lens {lin.type}->{lout.type}")
        assert n0 + 2 <= _listed(lrp) <= n0 + 3  # (the RGBA and the RGBAZ case share one geometry: the entry is there)


@pytest.mark.parametrize("name", ["config3_4k_rgbaz_rect_eqr_bc_post", "config3_4k_rgbz_rect_eqr_bc_post", "4k_rect_eqr_bc"])
def test_whole_frames_by_block_class(lrp, torch_cuda, name):
    """BASELINE configs[3] at full size: the entry and its lists are made by a launch on another frame, the case's frame is
    then rendered by block class (forced: the automatic rule wants 45 % corner blocks, this frame has 37 %) and must reproduce the committed
    oracle digest, band by band — the fill kernel in front of the window kernel, and beside it."""
    torch = torch_cuda
    case, want = ffc.frame_cases()[name], FULL["frames"][name]
    n, m, c = case["size"], case["out_size"], case["c"]
    lin, lout = cases.lenses(lrp, n, n)[case["inp"]], cases.lenses(lrp, m, m)[case["out"]]

    def frame(seed):
        d_in = torch.empty((n, n, c), dtype=torch.float32, device="cuda")
        lrp.synth_fill(d_in, n, n, c, seed, case.get("depth", -1))
        d_out = torch.full((m, m, c), -12345.0, dtype=torch.float32, device="cuda")
        lrp.reproject(lrp.Image(lin, n, n, c, d_in), lrp.Image(lout, m, m, c, d_out), 1, case["interp"],
                      cases.rotation(lrp, case["deg"]), post=tuple(case["post"]) if case.get("post") else None)
        torch.cuda.synchronize()
        return d_out

    frame(case["seed"] + 99)
    for fill_stream in (0, 1):
        lrp.debug_set("geo_fill_stream", fill_stream)
        n0 = _listed(lrp)
        d_out = frame(case["seed"])
        assert _listed(lrp) == n0 + 1
        sha, bands, n_nan = ffc.frame_digests(d_out.cpu().numpy())
        bad = [b for b in range(ffc.BANDS) if bands[b] != want["bands"][b]]
        assert not bad, f"{name}: row bands {bad} of {ffc.BANDS} differ from the committed oracle digest (fill_stream {fill_stream})"
        assert sha == want["sha256"] and n_nan == want["nan"]


@pytest.mark.parametrize("channels", [3, 4, 5])
def test_outputs_of_one_source(lrp, oracle, torch_cuda, channels):
    """lrp_reproject_multi_device (BASELINE configs[4]: six faces of one panorama): a launch per face, dealt over the caller's stream
    and a side stream; the first call fills six geometry-cache entries, later calls read them.  Six and nine outputs, odd sizes, a
    rectilinear source as well, fused tonemap — every face of three consecutive calls against the live oracle."""
    torch = torch_cuda
    faces = [(0.0, 0.0, 0.0), (90.0, 0.0, 0.0), (180.0, 0.0, 0.0), (270.0, 0.0, 0.0), (0.0, 90.0, 0.0), (0.0, -90.0, 0.0),
             (30.0, -15.0, 5.0), (45.0, 45.0, 0.0), (10.0, 0.0, 80.0)]
    for (iw, ih, ow, oh), in_name, n_faces, post in (((256, 128, 72, 72), "eqr_full", 6, None), ((200, 100, 53, 41), "eqr_full", 9, (1.5, 3.0)),
                                                     ((96, 80, 64, 48), "rect", 6, None), ((128, 128, 40, 56), "eqd180", 3, (0.75, 2.0))):
        src = cases.hash_noise(ih, iw, channels, seed=0xFACE + iw + channels, planted=True)
        lin = cases.lenses(lrp, iw, ih)[in_name]
        lout = lrp.LensInfo.rectilinear(18.0, 36.0, ow, oh)
        rots = np.stack([cases.rotation(lrp, f) for f in faces[:n_faces]])
        wants = []
        for f in range(n_faces):
            w = oracle.reproject(lin, src, lout, ow, oh, 1, 2, rots[f])
            wants.append(oracle.post_process(w, *post) if post else w)
        d_in = torch.from_numpy(src).cuda()

        def render():
            outs = [torch.full((oh, ow, channels), -12345.0, dtype=torch.float32, device="cuda") for _ in range(n_faces)]
            lrp.reproject_multi(lrp.Image(lin, iw, ih, channels, d_in), [lrp.Image(lout, ow, oh, channels, o) for o in outs], 1, 2, rots, post=post)
            torch.cuda.synchronize()
            return [o.cpu().numpy() for o in outs]

        what = f"{in_name} {iw}x{ih} -> {n_faces} x {ow}x{oh} C={channels} post={post}"
        hits0 = lrp.geometry_cache_stats()["hits"]
        for call in ("first call (fills the entries)", "second call (reads them)", "third call"):
            for f, got in enumerate(render()):
                cases.assert_same_bits(got, wants[f], f"{call}, face {f}, " + what)
        assert lrp.geometry_cache_stats()["hits"] >= hits0 + 2 * n_faces, what

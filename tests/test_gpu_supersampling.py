"""-m gpu: bicubic with num_samples 2, 3 and 4 through the window kernel's supersampling instantiations (csrc/lrp_win_kernel.h
SS) — the settings the reference's own help text prescribes: --scale 0.5 --samples 2, --scale 0.33334 --samples 3, --scale 0.25
--samples 4 (src/main.cpp:192-196).

The reference sums the ns x ns sub-samples of a pixel in ssx-outer, ssy-inner order into a zero-initialised accumulator and
multiplies by 1 / (ns * ns) (src/reproject.cpp:290-298, 334-341).  In the SS instantiations a block is 16 x 4 output pixels, a
lane owns one pixel and takes its sub-samples in that order in rounds of four (one round for 2 x 2, three for 3 x 3 — the last
holds one sub-sample —, four for 4 x 4), summed in registers; num_samples 5 and more stay with the tile kernel.  Every lens pair, channel count,
rotation, odd sizes, strong down-scaling (what --samples is for), corner / edge blocks, fused tonemap, row bands and batches,
against the live oracle at 0 ULP; `win_ss` 0 (the tile kernel) must give the same bits."""
import numpy as np
import pytest

import cases
import golden_cases

pytestmark = pytest.mark.gpu
USES_GEO_CACHE = True  # (the product's default: the first launch of a geometry fills an entry of sub-samples, later launches read it)

BICUBIC = 2


@pytest.fixture(autouse=True)
def _fresh_cache(lrp):
    lrp.debug_set("geo_cache", 1)
    lrp.geometry_cache_configure(1 << 30, 1)
    lrp.release_cached_tables()
    yield
    lrp.geometry_cache_configure(1 << 30, 1)
    lrp.release_cached_tables()


def _render(lrp, torch, lin, d_in, lout, ow, oh, rot, post=None, ns=2):
    h, w, c = d_in.shape
    d_out = torch.full((oh, ow, c), -12345.0, dtype=torch.float32, device="cuda")
    lrp.reproject(lrp.Image(lin, w, h, c, d_in), lrp.Image(lout, ow, oh, c, d_out), ns, BICUBIC, rot, post=post)
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


@pytest.mark.parametrize("ns", [2, 3, 4])
@pytest.mark.parametrize("channels", [3, 4, 5])
def test_every_lens_pair_against_the_live_oracle(lrp, oracle, torch_cuda, channels, ns):
    torch = torch_cuda
    k = ns
    for out_name in ("rect", "eqd180", "eqr_full", "eqr_part"):
        for in_name in ("rect", "rect_tele", "eqd180", "eqr_full", "eqr_part"):
            k += 1
            rot_name = list(golden_cases.ROTS)[k % 5]
            iw, ih, ow, oh = [(61, 47, 53, 41), (160, 120, 72, 67), (256, 128, 35, 19), (96, 80, 131, 90)][k % 4]
            post = (1.5, 3.0) if k % 3 == 0 else None
            src = cases.hash_noise(ih, iw, channels, seed=0x552 + 8 * k + channels, planted=(k % 2 == 0))
            lin, lout = cases.lenses(lrp, iw, ih)[in_name], cases.lenses(lrp, ow, oh)[out_name]
            rot = cases.rotation(lrp, golden_cases.ROTS[rot_name])
            want = oracle.reproject(lin, src, lout, ow, oh, ns, BICUBIC, rot)
            if post:
                want = oracle.post_process(want, *post)
            d_in = torch.from_numpy(src).cuda()
            what = f"{in_name} {iw}x{ih} -> {out_name} {ow}x{oh} C={channels} ns={ns} {rot_name} post={post}"
            fills0, hits0 = (lrp.geometry_cache_stats()[key] for key in ("fills", "hits"))
            cases.assert_same_bits(_render(lrp, torch, lin, d_in, lout, ow, oh, rot, post, ns=ns), want, "window kernel (SS), the launch that fills the entry of sub-samples, " + what)
            cases.assert_same_bits(_render(lrp, torch, lin, d_in, lout, ow, oh, rot, post, ns=ns), want, "window kernel (SS), a launch that reads it, " + what)
            stats = lrp.geometry_cache_stats()
            cached = 0 if (in_name.startswith("rect") and out_name.startswith("eqr")) else 1  # (a rectilinear view into a panorama computes: lrp_plan.cpp)
            assert stats["fills"] == fills0 + cached and stats["hits"] == hits0 + cached, what
            prev_geo = lrp.debug_set("geo_cache", 0)
            cases.assert_same_bits(_render(lrp, torch, lin, d_in, lout, ow, oh, rot, post, ns=ns), want, "window kernel (SS), cache off, " + what)
            lrp.debug_set("geo_cache", prev_geo)
            prev = lrp.debug_set("win_ss", 0)
            cases.assert_same_bits(_render(lrp, torch, lin, d_in, lout, ow, oh, rot, post, ns=ns), want, "tile kernel, " + what)
            lrp.debug_set("win_ss", prev)


@pytest.mark.parametrize("interp", [0, 1])
@pytest.mark.parametrize("ns", [2, 3, 4])
def test_nearest_and_bilinear_share_the_entry_of_sub_samples(lrp, oracle, torch_cuda, ns, interp):
    """Nearest / bilinear with num_samples 2-4: the launch that fills an entry of sub-samples (tile kernel, computing), a launch that
    reads it (lrp_ss_gather_kernel.h: a lane per sub-sample), a batch of three that reads it, the cache off — and bicubic reading the
    entry a nearest / bilinear launch wrote (one entry serves the three samplers) — against the live oracle, RGB / RGBA / RGBAZ, every
    target lens over every source lens, odd sizes (partial rows of pixels, the idle lane of num_samples 3)."""
    torch = torch_cuda
    k = 3 * ns + interp
    for out_name in ("rect", "eqd180", "eqr_full", "eqr_part"):
        for in_name in ("rect_tele", "eqd180", "eqr_full", "eqr_part"):
            k += 1
            channels = 3 + k % 3
            rot_name = list(golden_cases.ROTS)[k % 5]
            iw, ih, ow, oh = [(61, 47, 53, 41), (160, 120, 72, 67), (256, 128, 35, 19), (96, 80, 131, 90)][k % 4]
            post = (1.5, 3.0) if k % 3 == 0 else None
            src = cases.hash_noise(ih, iw, channels, seed=0x7155 + 8 * k + channels, planted=(k % 2 == 0))
            lin, lout = cases.lenses(lrp, iw, ih)[in_name], cases.lenses(lrp, ow, oh)[out_name]
            rot = cases.rotation(lrp, golden_cases.ROTS[rot_name])
            d_in = torch.from_numpy(src).cuda()

            def render(interpolation, batch=0):
                outs = [torch.full((oh, ow, channels), -12345.0, dtype=torch.float32, device="cuda") for _ in range(max(batch, 1))]
                im_in = lrp.Image(lin, iw, ih, channels, d_in)
                if batch:
                    lrp.reproject_batch([im_in] * batch, [lrp.Image(lout, ow, oh, channels, o) for o in outs], ns, interpolation, rot, post=post)
                else:
                    lrp.reproject(im_in, lrp.Image(lout, ow, oh, channels, outs[0]), ns, interpolation, rot, post=post)
                torch.cuda.synchronize()
                return [o.cpu().numpy() for o in outs]

            def want_of(interpolation):
                w = oracle.reproject(lin, src, lout, ow, oh, ns, interpolation, rot)
                return oracle.post_process(w, *post) if post else w

            want = want_of(interp)
            what = f"{in_name} {iw}x{ih} -> {out_name} {ow}x{oh} C={channels} ns={ns} interp={interp} {rot_name} post={post}"
            fills0, hits0 = (lrp.geometry_cache_stats()[key] for key in ("fills", "hits"))
            cases.assert_same_bits(render(interp)[0], want, "the launch that fills the entry, " + what)
            cases.assert_same_bits(render(interp)[0], want, "a launch that reads it, " + what)
            for i, got in enumerate(render(interp, batch=3)):
                cases.assert_same_bits(got, want, f"a batch of three, frame {i}, " + what)
            stats = lrp.geometry_cache_stats()
            # cheap coordinates are computed (a rectilinear source under a rectilinear / panorama target; a source x from the column
            # table: lrp_plan.cpp); a fisheye target never has them
            cached = stats["fills"] - fills0
            assert cached in (0, 1) and stats["hits"] >= hits0 + 2 * cached, what
            if out_name.startswith("eqd"):
                assert cached == 1, what
            if cached:  # ... and the bicubic SS instantiations read the entry this sampler wrote
                hits1 = lrp.geometry_cache_stats()["hits"]
                cases.assert_same_bits(render(BICUBIC)[0], want_of(BICUBIC), "bicubic on the entry a nearest / bilinear launch wrote, " + what)
                if not (in_name.startswith("rect") and out_name.startswith("eqr")):
                    assert lrp.geometry_cache_stats()["hits"] == hits1 + 1, what
            prev_geo = lrp.debug_set("geo_cache", 0)
            cases.assert_same_bits(render(interp)[0], want, "cache off, " + what)
            lrp.debug_set("geo_cache", prev_geo)


def test_bands_batches_and_the_other_sample_counts(lrp, oracle, torch_cuda):
    """Row bands (lrp_reproject_rows_device: bands that start and end inside a 4-row block) and a batch of five frames for every
    sample count of the SS instantiations, and num_samples 1 / 5 around them (their own kernels) on one geometry."""
    torch = torch_cuda
    iw, ih, ow, oh, c = 300, 200, 147, 101, 4
    lin, lout = lrp.LensInfo.equirectangular(), lrp.LensInfo.rectilinear(18.0, 36.0, ow, oh)
    rot = lrp.rotation_matrix(0.4, -0.2, 0.05)
    srcs = [cases.hash_noise(ih, iw, c, seed=0xBA2D + i, planted=True) for i in range(5)]
    d_ins = [torch.from_numpy(s).cuda() for s in srcs]
    for ns in (1, 2, 3, 4, 5):
        want = oracle.reproject(lin, srcs[0], lout, ow, oh, ns, BICUBIC, rot)
        cases.assert_same_bits(_render(lrp, torch, lin, d_ins[0], lout, ow, oh, rot, ns=ns), want, f"num_samples {ns}")
    for ns in (2, 3, 4):
        wants = [oracle.reproject(lin, s, lout, ow, oh, ns, BICUBIC, rot) for s in srcs]
        outs = [torch.full((oh, ow, c), -1.0, dtype=torch.float32, device="cuda") for _ in srcs]
        lrp.reproject_batch([lrp.Image(lin, iw, ih, c, d) for d in d_ins], [lrp.Image(lout, ow, oh, c, o) for o in outs], ns, BICUBIC, rot)
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            cases.assert_same_bits(o.cpu().numpy(), wants[i], f"num_samples {ns}: batch of five, frame {i}")
        banded = torch.full((oh, ow, c), -7.0, dtype=torch.float32, device="cuda")
        cuts = [0, 3, 10, 11, 50, 97, oh]
        for a, b in zip(cuts[:-1], cuts[1:]):
            lrp.reproject_rows(lrp.Image(lin, iw, ih, c, d_ins[0]), lrp.Image(lout, ow, oh, c, banded), ns, BICUBIC, a, b - a, rot)
        torch.cuda.synchronize()
        cases.assert_same_bits(banded.cpu().numpy(), wants[0], f"num_samples {ns}: row bands")


@pytest.mark.parametrize("ns,m", [(2, 2048), (3, 1365), (4, 1024)])
def test_downscale_4k_whole_frame_rows(lrp, oracle, torch_cuda, ns, m):
    """The cases --samples exists for (src/main.cpp:192-196: --scale 0.5 --samples 2, 0.33334 / 3, 0.25 / 4): 4096^2 -> 2048^2,
    1365^2 (int(4096 * 0.33334), src/main.cpp:581-587) and 1024^2, bicubic RGBA — sampled rows against the oracle, the whole frame
    against the tile kernel and the one-pixel-per-lane kernel."""
    torch = torch_cuda
    n, c = 4096, 4
    d_in = torch.empty((n, n, c), dtype=torch.float32, device="cuda")
    lrp.synth_fill(d_in, n, n, c, 0x5EED0002)
    torch.cuda.synchronize()
    src = d_in.cpu().numpy()
    for in_name, out_name, deg in (("eqd180", "rect", None), ("eqr_full", "rect", (30.0, -15.0, 5.0)), ("rect", "eqr_full", (0.0, 0.0, 0.0))):
        lin, lout = cases.lenses(lrp, n, n)[in_name], cases.lenses(lrp, m, m)[out_name]
        rot = cases.rotation(lrp, deg)
        first = _render(lrp, torch, lin, d_in, lout, m, m, rot, ns=ns)  # fills the entry of sub-samples
        got = _render(lrp, torch, lin, d_in, lout, m, m, rot, ns=ns)    # reads it
        assert np.array_equal(got.view(np.uint32), first.view(np.uint32)), f"{in_name} -> {out_name}: the filling and the reading launch differ"
        rows = sorted({0, 1, 2, 3, 4, m // 4 - 1, m // 2 - 1, m // 2, (3 * m) // 4 + 1, m - 4, m - 3, m - 2, m - 1})
        want = oracle.reproject_rows(lin, src, lout, m, m, ns, BICUBIC, rot, rows)
        for y in rows:
            cases.assert_same_bits(got[y], want[y], f"{in_name} -> {out_name} {deg}: row {y}")
        prev = lrp.debug_set("win_ss", 0)
        tile = _render(lrp, torch, lin, d_in, lout, m, m, rot, ns=ns)
        lrp.debug_set("win_ss", prev)
        assert np.array_equal(got.view(np.uint32), tile.view(np.uint32)), f"{in_name} -> {out_name}: window (SS) and tile kernels differ"
        prev = lrp.debug_kernel(0)
        pixel = _render(lrp, torch, lin, d_in, lout, m, m, rot, ns=ns)
        lrp.debug_kernel(prev)
        assert np.array_equal(got.view(np.uint32), pixel.view(np.uint32)), f"{in_name} -> {out_name}: window (SS) and pixel kernels differ"


@pytest.mark.parametrize("name", ["4k_eqd_rect_bc_half_ns2", "4k_eqd_rect_bc_third_ns3", "4k_eqd_rect_bc_quarter_ns4", "4k_rgbaz_eqr_rect_bc_rot_third_ns3_post",
                                  "4k_eqr_eqd_bl_rot_third_ns3", "4k_rgb_eqr_rect_nn_rot_quarter_ns4"])
def test_whole_frames_equal_the_committed_oracle_digests(lrp, torch_cuda, name):
    """The reference's --scale / --samples pairs at full size against digests the oracle wrote in the build container
    (tests/golden/fullframe_golden.json; no oracle call on the box): the launch that fills the entry of sub-samples, a launch
    that reads it (bicubic: the GeoRead + SS window instantiations; nearest / bilinear: the gather kernel), a launch with the
    cache off — all three must carry the committed bits."""
    import json
    import os

    import fullframe_cases as ffc

    torch = torch_cuda
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullframe_golden.json")) as f:
        want = json.load(f)["frames"][name]
    case = ffc.frame_cases()[name]
    n, m, c = case["size"], case["out_size"], case["c"]
    d_in = torch.empty((n, n, c), dtype=torch.float32, device="cuda")
    lrp.synth_fill(d_in, n, n, c, case["seed"], case.get("depth", -1))
    lin, lout = cases.lenses(lrp, n, n)[case["inp"]], cases.lenses(lrp, m, m)[case["out"]]
    rot = cases.rotation(lrp, case["deg"])

    def render():
        d_out = torch.full((m, m, c), -12345.0, dtype=torch.float32, device="cuda")
        lrp.reproject(lrp.Image(lin, n, n, c, d_in), lrp.Image(lout, m, m, c, d_out), case["ns"], case["interp"], rot,
                      post=tuple(case["post"]) if case.get("post") else None)
        torch.cuda.synchronize()
        return d_out

    stats0 = lrp.geometry_cache_stats()
    for what in ("the launch that fills the entry", "a launch that reads it", "cache off"):
        prev = lrp.debug_set("geo_cache", 0) if what == "cache off" else None
        d_out = render()
        if prev is not None:
            lrp.debug_set("geo_cache", prev)
        sha, bands, n_nan = ffc.frame_digests(d_out.cpu().numpy())
        bad = [b for b in range(ffc.BANDS) if bands[b] != want["bands"][b]]
        assert not bad, f"{name}, {what}: row bands {bad} of {ffc.BANDS} differ from the committed oracle digest"
        assert sha == want["sha256"] and n_nan == want["nan"], f"{name}, {what}"
    stats = lrp.geometry_cache_stats()
    assert stats["fills"] == stats0["fills"] + 1 and stats["hits"] == stats0["hits"] + 1, name

"""-m gpu: the geometry cache (csrc/lrp_geocache.cpp, the GeoRead window kernels).

The first single bicubic launch of a geometry writes the source coordinates of every output pixel and the window
extremes of every block as a side output; later launches of the same geometry load them.  Both must give the
reference's bits: every comparison here is against COMMITTED oracle digests (tests/golden/), once for the launch that
fills an entry and once for a launch that reads it — with different pixels in between, the geometry is what is cached."""
import importlib
import json
import os
import threading

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
with open(os.path.join(HERE, "golden", "oracle_golden.json")) as _f:
    SMALL = json.load(_f)


@pytest.fixture(autouse=True)
def _fresh_cache(lrp):
    lrp.debug_set("geo_cache", 1)
    lrp.geometry_cache_configure(1 << 30, 1)
    lrp.release_cached_tables()
    yield
    lrp.geometry_cache_configure(1 << 30, 1)
    lrp.release_cached_tables()


class _DeviceSynth:
    def __init__(self, lrp, torch):
        self.lrp, self.torch = lrp, torch

    def synth_frame(self, width, height, channels, seed, depth_channel=-1):
        t = self.torch.empty((height, width, channels), dtype=self.torch.float32, device="cuda")
        self.lrp.synth_fill(t, width, height, channels, seed, depth_channel)
        self.torch.cuda.synchronize()
        return t.cpu().numpy()


def _bicubic_cases(lrp):
    return [(n, c) for n, c in golden_cases.all_cases(lrp) if c["interp"] == 2 and c["ns"] == 1]


def _single_sample_cases(lrp):
    """every sampler: the window kernel (bicubic) and the tile kernels (nearest, bilinear) read the same coordinate map"""
    return [(n, c) for n, c in golden_cases.all_cases(lrp) if c["ns"] == 1]


def _small_setup(lrp, torch, case):
    src = golden_cases.planted_input(_DeviceSynth(lrp, torch), case["iw"], case["ih"], case["c"], case["seed"])
    lin = cases.lenses(lrp, case["iw"], case["ih"])[case["inp"]]
    lout = cases.lenses(lrp, case["ow"], case["oh"])[case["out"]]
    rot = cases.rotation(lrp, golden_cases.ROTS[case["rot"]])
    return src, lin, lout, rot


def _small_render(lrp, torch, case, d_in, lin, lout, rot, stream=None, interp=None):
    # (the fill of the output runs on the stream the launch goes to: torch's streams do not wait for its default stream)
    with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream()):
        d_out = torch.full((case["oh"], case["ow"], case["c"]), -12345.0, dtype=torch.float32, device="cuda")
    lrp.reproject(lrp.Image(lin, case["iw"], case["ih"], case["c"], d_in),
                  lrp.Image(lout, case["ow"], case["oh"], case["c"], d_out), 1, case["interp"] if interp is None else interp, rot,
                  stream=stream)
    return d_out


def test_small_matrix_fill_then_read(lrp, torch_cuda):
    """Every single-sample case of the 216-case matrix (nearest, bilinear, bicubic): the launch that fills the entry, a
    launch on OTHER pixels that reads it, and a third launch on the case's pixels that reads it again."""
    torch = torch_cuda
    todo = _single_sample_cases(lrp)
    assert len(todo) >= 60 and {c["interp"] for _n, c in todo} == {0, 1, 2}
    for name, case in todo:
        src, lin, lout, rot = _small_setup(lrp, torch, case)
        d_in = torch.from_numpy(src).cuda()
        before = lrp.geometry_cache_stats()
        first = _small_render(lrp, torch, case, d_in, lin, lout, rot)
        other = torch.rand_like(d_in)
        _small_render(lrp, torch, case, other, lin, lout, rot)
        again = _small_render(lrp, torch, case, d_in, lin, lout, rot)
        torch.cuda.synchronize()
        after = lrp.geometry_cache_stats()
        delta = tuple(after[k] - before[k] for k in ("fills", "hits", "bypasses"))
        # (tile-kernel launches whose coordinates are cheap — nearest without a rotation, a rectilinear source under a
        # rectilinear / panorama target — run without the cache: all three launches compute)
        assert delta == (1, 2, 0) or (case["interp"] != 2 and delta == (0, 0, 0)), (name, before, after)
        assert golden_cases.digest(first.cpu().numpy()) == SMALL["reproject"][name], f"{name}: the filling launch"
        assert golden_cases.digest(again.cpu().numpy()) == SMALL["reproject"][name], f"{name}: the reading launch"


@pytest.mark.parametrize("channels", [3, 4, 5])
def test_lens_pairs_channels_rotations_against_the_live_oracle(lrp, oracle, torch_cuda, channels):
    """The committed matrix only holds RGB cases for single-sample bicubic: every lens pair x rotation for RGB, RGBA and
    RGBAZ against the oracle run here — filling launch, reading launch, and every strip length of the reading kernel."""
    torch = torch_cuda
    k = 0
    for out_name in ("rect", "eqd180", "eqr_full"):
        for in_name in ("rect", "eqd180", "eqr_full", "eqr_part"):
            k += 1
            rot_name = list(golden_cases.ROTS)[k % 5]
            iw, ih, ow, oh = (61, 47, 53, 41) if k % 2 else (96, 80, 72, 67)
            case = dict(iw=iw, ih=ih, ow=ow, oh=oh, out=out_name, inp=in_name, interp=2, c=channels, ns=1, rot=rot_name,
                        seed=0x6E0 + 16 * k + channels)
            want = golden_cases.run_oracle(oracle, lrp, case)
            src, lin, lout, rot = _small_setup(lrp, torch, case)
            d_in = torch.from_numpy(src).cuda()
            first = _small_render(lrp, torch, case, d_in, lin, lout, rot)
            torch.cuda.synchronize()
            cases.assert_same_bits(first.cpu().numpy(), want, f"fill {out_name} <- {in_name} C={channels} {rot_name}")
            for strip in (0, 1, 2, 4):
                prev = lrp.debug_set("geo_strip", strip)
                again = _small_render(lrp, torch, case, d_in, lin, lout, rot)
                torch.cuda.synchronize()
                lrp.debug_set("geo_strip", prev)
                cases.assert_same_bits(again.cpu().numpy(), want, f"read (strip {strip}) {out_name} <- {in_name} C={channels} {rot_name}")


@pytest.mark.parametrize("channels", [3, 4, 5])
def test_big_window_variant_against_the_live_oracle(lrp, oracle, torch_cuda, channels):
    """A rectilinear view rendered into a panorama: the reading launch is the big-window variant (20 KiB of LDS per
    wavefront, pass windows up to 128 texels wide).  Minifying and magnifying views, with and without a pan."""
    torch = torch_cuda
    for (iw, ih, ow, oh), in_name, rot_name in (((256, 192, 64, 48), "rect", "none"), ((320, 320, 128, 64), "rect", "pan180"),
                                                ((64, 64, 128, 96), "rect_tele", "ident"), ((300, 200, 96, 64), "rect", "none"),
                                                ((700, 500, 90, 61), "rect", "r30")):
        case = dict(iw=iw, ih=ih, ow=ow, oh=oh, out="eqr_full", inp=in_name, interp=2, c=channels, ns=1, rot=rot_name,
                    seed=0x9A1 + iw + channels)
        want = golden_cases.run_oracle(oracle, lrp, case)
        src, lin, lout, rot = _small_setup(lrp, torch, case)
        d_in = torch.from_numpy(src).cuda()
        first = _small_render(lrp, torch, case, d_in, lin, lout, rot)
        again = _small_render(lrp, torch, case, d_in, lin, lout, rot)
        torch.cuda.synchronize()
        cases.assert_same_bits(first.cpu().numpy(), want, f"fill {in_name} {iw}x{ih} -> {ow}x{oh} C={channels} {rot_name}")
        cases.assert_same_bits(again.cpu().numpy(), want, f"read {in_name} {iw}x{ih} -> {ow}x{oh} C={channels} {rot_name}")


@pytest.mark.parametrize("channels", [3, 4, 5])
def test_corner_classes_under_every_strip_length(lrp, oracle, torch_cuda, channels):
    """The class bytes of a geometry-cache entry (lrp_params.h geo_class_offset): a wavefront of the big-window variant
    reads the classes of its strip when it starts and renders corner blocks without their record and coordinates.
    Outputs of many block rows that are mostly out of view (tele lens: almost all corner blocks), sizes that are not
    multiples of the block and of the strip, every strip length (8 and 16: the kernel plans every block from its
    record), a pitch that moves the view to the top edge, the fused tonemap — the filling launch and the reading
    launches against the live oracle."""
    torch = torch_cuda
    for (iw, ih, ow, oh), in_name, rot_name, post in (((96, 64, 80, 200), "rect_tele", "none", None), ((64, 64, 147, 331), "rect_tele", "r30", (1.5, 3.0)),
                                                      ((200, 120, 64, 130), "rect", "pitch90", (0.75, 2.0)), ((90, 70, 33, 97), "rect", "pan180", None)):
        case = dict(iw=iw, ih=ih, ow=ow, oh=oh, out="eqr_full", inp=in_name, interp=2, c=channels, ns=1, rot=rot_name, seed=0xC1A5 + ow + channels)
        want = golden_cases.run_oracle(oracle, lrp, case)
        if post:
            want = oracle.post_process(want, *post)
        src, lin, lout, rot = _small_setup(lrp, torch, case)
        d_in = torch.from_numpy(src).cuda()

        def render():
            d_out = torch.full((oh, ow, channels), -12345.0, dtype=torch.float32, device="cuda")
            lrp.reproject(lrp.Image(lin, iw, ih, channels, d_in), lrp.Image(lout, ow, oh, channels, d_out), 1, 2, rot, post=post)
            torch.cuda.synchronize()
            return d_out.cpu().numpy()

        what = f"{in_name} {iw}x{ih} -> {ow}x{oh} C={channels} {rot_name} post={post}"
        cases.assert_same_bits(render(), want, "fill " + what)
        for strip in (0, 1, 2, 4, 8, 16):
            prev = lrp.debug_set("geo_strip", strip)
            got = render()
            lrp.debug_set("geo_strip", prev)
            cases.assert_same_bits(got, want, f"read (strip {strip}) " + what)
        prev = lrp.debug_set("geo_big", 0)  # the four-wavefront instantiation reads the same entry (and ignores the classes)
        got = render()
        lrp.debug_set("geo_big", prev)
        cases.assert_same_bits(got, want, "read (geo_big 0) " + what)
        stats = lrp.geometry_cache_stats()
        assert stats["fills"] >= 1 and stats["hits"] >= 7


@pytest.mark.parametrize("chunk", range(6))
def test_random_configurations_through_the_cache(lrp, oracle, torch_cuda, chunk):
    """Randomised lens pairs, rotations, sizes and channel counts (the generators of test_gpu_random_lenses.py, one sample per
    pixel): the launch that fills the entry, a single launch that reads it, a batch of five that reads it and — for a
    rectilinear view into a panorama — the four-wavefront instantiation, all three samplers, against the live oracle."""
    import test_gpu_random_lenses as rl

    torch = torch_cuda
    for k in range(8):
        seed = 7000 + 8 * chunk + k
        rng = np.random.default_rng(seed)
        in_w, in_h = int(rng.integers(40, 400)), int(rng.integers(30, 300))
        out_w, out_h = int(rng.integers(17, 330)), int(rng.integers(9, 250))
        c = int(rng.choice([3, 4, 4, 5]))
        lin, lout = rl.random_lens(lrp, rng, in_w, in_h), rl.random_lens(lrp, rng, out_w, out_h)
        if k % 4 == 0:  # every fourth: the mapping of the big-window variant and the class bytes
            lin, lout = lrp.LensInfo.rectilinear(float(rng.uniform(10.0, 90.0)), 36.0, in_w, in_h), lrp.LensInfo.equirectangular()
        rot = rl.random_rotation(lrp, rng)
        post = (1.25, 3.0) if k % 3 == 0 else None
        src = cases.hash_noise(in_h, in_w, c, seed=seed, planted=bool(rng.integers(0, 2)))
        d_in = torch.from_numpy(src).cuda()
        img_in = lrp.Image(lin, in_w, in_h, c, d_in)

        def render(interp, batch=0):
            outs = [torch.full((out_h, out_w, c), -777.0, dtype=torch.float32, device="cuda") for _ in range(max(batch, 1))]
            if batch:
                lrp.reproject_batch([img_in] * batch, [lrp.Image(lout, out_w, out_h, c, o) for o in outs], 1, interp, rot, post=post)
            else:
                lrp.reproject(img_in, lrp.Image(lout, out_w, out_h, c, outs[0]), 1, interp, rot, post=post)
            torch.cuda.synchronize()
            return [o.cpu().numpy() for o in outs]

        for interp in (2, 1, 0):
            want = oracle.reproject(lin, src, lout, out_w, out_h, 1, interp, rot)
            if post:
                want = oracle.post_process(want, *post)
            what = f"seed {seed}: {in_w}x{in_h}x{c} -> {out_w}x{out_h}, lens {lin.type}->{lout.type}, interp={interp}, post={post}"
            cases.assert_same_bits(render(interp)[0], want, "first launch, " + what)
            cases.assert_same_bits(render(interp)[0], want, "second launch, " + what)
            for got in render(interp, batch=5):
                cases.assert_same_bits(got, want, "batch of five, " + what)
            if interp == 2 and k % 4 == 0:
                prev = lrp.debug_set("geo_big", 0)
                got = render(interp)[0]
                lrp.debug_set("geo_big", prev)
                cases.assert_same_bits(got, want, "geo_big 0, " + what)


FRAMES = ["config0_512_eqr_rect_nn", "config2_4k_eqr_eqd_bl_rot", "4k_eqr_rect_bl", "4k_eqr_rect_nn", "config1_4k_eqd_rect_bc", "northstar_4k_eqr_rect_bc", "scaling_4k_eqr_eqd_bc_rot", "config3_4k_rgbaz_rect_eqr_bc_post",
          "config3_4k_rgbz_rect_eqr_bc_post", "config4_8k_rgb_face0", "config4_8k_rgb_face1", "config4_8k_rgb_face4",
          "4k_eqr_rect_bc_rot", "4k_eqr_rect_bc_pan90", "4k_eqr_rect_bc_pitch90", "4k_rect_rect_bc_rot", "4k_eqd_eqd_bc_rot",
          "4k_eqr_eqr_bc_rot", "4k_rect_eqr_bc", "4k_rgb_eqd_rect_bc", "4k_rgb_eqr_rect_bc_rot", "4k_rgbaz_eqd_rect_bc",
          "4k_rgbaz_eqr_rect_bc_rot"]


def _frame(lrp, torch, case, seed):
    n, m, c = case["size"], case["out_size"], case["c"]
    d_in = torch.empty((n, n, c), dtype=torch.float32, device="cuda")
    lrp.synth_fill(d_in, n, n, c, seed, case.get("depth", -1))
    lin, lout = cases.lenses(lrp, n, n)[case["inp"]], cases.lenses(lrp, m, m)[case["out"]]
    d_out = torch.full((m, m, c), -12345.0, dtype=torch.float32, device="cuda")
    lrp.reproject(lrp.Image(lin, n, n, c, d_in), lrp.Image(lout, m, m, c, d_out), 1, case["interp"],
                  cases.rotation(lrp, case["deg"]), post=tuple(case["post"]) if case.get("post") else None)
    torch.cuda.synchronize()
    return d_out


@pytest.mark.parametrize("name", FRAMES)
def test_whole_frames_filled_and_read(lrp, torch_cuda, name):
    """Whole frames at BASELINE.json's sizes: the entry is filled by a launch on another frame (seed + 99), then the
    case's frame is rendered from the cached coordinates and must reproduce the committed oracle digest, band by band."""
    torch = torch_cuda
    case, want = ffc.frame_cases()[name], FULL["frames"][name]
    _frame(lrp, torch, case, case["seed"] + 99)
    st = lrp.geometry_cache_stats()
    unrotated = case["deg"] is None or not any(case["deg"])
    cached = not (case["interp"] == 0 and unrotated)  # (nearest without a rotation computes for itself: mirrored pixels)
    assert (st["fills"] >= 1 and st["entries"] == 1) if cached else st["entries"] == 0, st
    d_out = _frame(lrp, torch, case, case["seed"])
    assert lrp.geometry_cache_stats()["hits"] == st["hits"] + (1 if cached else 0)
    sha, bands, n_nan = ffc.frame_digests(d_out.cpu().numpy())
    bad = [b for b in range(ffc.BANDS) if bands[b] != want["bands"][b]]
    assert not bad, f"{name}: row bands {bad} of {ffc.BANDS} differ from the committed oracle digest (cached coordinates)"
    assert sha == want["sha256"] and n_nan == want["nan"]


def test_filling_launch_equals_committed_digest(lrp, torch_cuda):
    """... and the launch that FILLS the entry (plain blocks + side output) on the case's own pixels."""
    torch = torch_cuda
    for name in ("config1_4k_eqd_rect_bc", "config3_4k_rgbaz_rect_eqr_bc_post", "config4_8k_rgb_face4"):
        case, want = ffc.frame_cases()[name], FULL["frames"][name]
        fills = lrp.geometry_cache_stats()["fills"]
        d_out = _frame(lrp, torch, case, case["seed"])
        assert lrp.geometry_cache_stats()["fills"] == fills + 1
        assert ffc.frame_digests(d_out.cpu().numpy())[0] == want["sha256"], name


def test_one_entry_serves_all_three_samplers(lrp, oracle, torch_cuda):
    """The coordinates do not depend on the sampler: a nearest launch fills the map, a bilinear launch reads it, the first
    bicubic launch adds the window extremes (a second side output), later bicubic launches read both — one entry."""
    torch = torch_cuda
    for k, (out_name, in_name, rot_name) in enumerate((("rect", "eqr_full", "r30"), ("eqd180", "rect", "none"), ("eqr_full", "eqd180", "pitch90"))):
        base = dict(iw=96, ih=80, ow=72, oh=67, out=out_name, inp=in_name, c=4, ns=1, rot=rot_name, seed=0x77A0 + k)
        wants = {i: golden_cases.run_oracle(oracle, lrp, dict(base, interp=i)) for i in (0, 1, 2)}
        src, lin, lout, rot = _small_setup(lrp, torch, dict(base, interp=0))
        d_in = torch.from_numpy(src).cuda()
        s0 = lrp.geometry_cache_stats()
        for interp in (0, 1, 2, 2, 1, 0, 2):
            got = _small_render(lrp, torch, base, d_in, lin, lout, rot, interp=interp)
            torch.cuda.synchronize()
            cases.assert_same_bits(got.cpu().numpy(), wants[interp], f"{out_name} <- {in_name} {rot_name} interp {interp}")
        s1 = lrp.geometry_cache_stats()
        assert s1["entries"] == s0["entries"] + 1
        # the map by the first launch that uses the cache, the extremes by the first bicubic one; a nearest launch without a
        # rotation computes for itself (mirrored pixels are as fast as a load per pixel)
        assert s1["fills"] - s0["fills"] == 2 and 3 <= s1["hits"] - s0["hits"] <= 5, (s0, s1)


def test_eviction_under_a_small_cap(lrp, torch_cuda):
    """Three geometries alternate under a cap that holds two: entries are evicted and re-filled, the bits stay."""
    torch = torch_cuda
    todo = _bicubic_cases(lrp)[:3]
    assert len(todo) == 3
    setups = []
    for name, case in todo:
        src, lin, lout, rot = _small_setup(lrp, torch, case)
        setups.append((name, case, torch.from_numpy(src).cuda(), lin, lout, rot))
    one = 53 * 41 * 8 + 4 * 16 * 32  # bytes of one entry of the odd-sized cases (map + boxes, rows of blocks padded to 16)
    lrp.geometry_cache_configure(int(2.5 * one), 1)
    before = lrp.geometry_cache_stats()
    for rnd in range(4):
        for name, case, d_in, lin, lout, rot in setups:
            out = _small_render(lrp, torch, case, d_in, lin, lout, rot)
            out2 = _small_render(lrp, torch, case, d_in, lin, lout, rot)
            torch.cuda.synchronize()
            assert golden_cases.digest(out.cpu().numpy()) == SMALL["reproject"][name], (rnd, name)
            assert golden_cases.digest(out2.cpu().numpy()) == SMALL["reproject"][name], (rnd, name)
    after = lrp.geometry_cache_stats()
    assert after["evictions"] > before["evictions"] and after["entries"] <= 2 and after["bytes"] <= after["max_bytes"], after
    assert after["hits"] >= before["hits"] + 12


def test_min_sightings_and_switch(lrp, torch_cuda):
    torch = torch_cuda
    name, case = _bicubic_cases(lrp)[0]
    src, lin, lout, rot = _small_setup(lrp, torch, case)
    d_in = torch.from_numpy(src).cuda()
    lrp.geometry_cache_configure(-1, 2)
    s0 = lrp.geometry_cache_stats()
    outs = [_small_render(lrp, torch, case, d_in, lin, lout, rot) for _ in range(3)]
    torch.cuda.synchronize()
    s1 = lrp.geometry_cache_stats()
    assert (s1["bypasses"] - s0["bypasses"], s1["fills"] - s0["fills"], s1["hits"] - s0["hits"]) == (1, 1, 1)
    prev = lrp.debug_set("geo_cache", 0)
    outs.append(_small_render(lrp, torch, case, d_in, lin, lout, rot))
    torch.cuda.synchronize()
    lrp.debug_set("geo_cache", prev)
    assert lrp.geometry_cache_stats()["hits"] == s1["hits"]
    lrp.geometry_cache_configure(0, -1)  # off: frees everything, every launch computes
    assert lrp.geometry_cache_stats()["entries"] == 0
    outs.append(_small_render(lrp, torch, case, d_in, lin, lout, rot))
    torch.cuda.synchronize()
    for o in outs:
        assert golden_cases.digest(o.cpu().numpy()) == SMALL["reproject"][name]


def test_concurrent_callers_share_entries(lrp, torch_cuda):
    """Four host threads, each with its own stream, render the same three geometries at the same time (the reference's
    -j N pool threads, src/main.cpp:538-544): one of them fills each entry, the others wait for it on the device or
    compute for themselves, everybody gets the oracle's bits."""
    torch = torch_cuda
    todo = _bicubic_cases(lrp)[:3]
    setups = []
    for name, case in todo:
        src, lin, lout, rot = _small_setup(lrp, torch, case)
        setups.append((name, case, torch.from_numpy(src).cuda(), lin, lout, rot))
    errors = []
    barrier = threading.Barrier(4)

    def worker(k):
        try:
            stream = torch.cuda.Stream()
            barrier.wait()
            for rnd in range(6):
                for name, case, d_in, lin, lout, rot in setups[k % 3:] + setups[:k % 3]:
                    out = _small_render(lrp, torch, case, d_in, lin, lout, rot, stream=stream)
                    stream.synchronize()
                    if golden_cases.digest(out.cpu().numpy()) != SMALL["reproject"][name]:
                        errors.append((k, rnd, name))
        except Exception as exc:  # noqa: BLE001
            errors.append((k, repr(exc)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    st = lrp.geometry_cache_stats()
    assert st["entries"] == 3 and st["hits"] > 0


def test_concurrent_whole_frames_on_two_streams(lrp, torch_cuda):
    """A reader on another stream than the writer's must wait for the writer on the DEVICE: the fill of a 4K frame
    takes ~200 us, the second stream's launch is enqueued microseconds later."""
    torch = torch_cuda
    name = "4k_eqr_rect_bc_rot"
    case, want = ffc.frame_cases()[name], FULL["frames"][name]
    n, c = case["size"], case["c"]
    lin, lout = cases.lenses(lrp, n, n)[case["inp"]], cases.lenses(lrp, n, n)[case["out"]]
    rot = cases.rotation(lrp, case["deg"])
    d_in = torch.empty((n, n, c), dtype=torch.float32, device="cuda")
    lrp.synth_fill(d_in, n, n, c, case["seed"])
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = [torch.full((n, n, c), -1.0, dtype=torch.float32, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        lrp.reproject(lrp.Image(lin, n, n, c, d_in), lrp.Image(lout, n, n, c, o), 1, 2, rot, stream=(s1 if i % 2 == 0 else s2))
    torch.cuda.synchronize()
    st = lrp.geometry_cache_stats()
    assert st["fills"] >= 1 and st["hits"] >= 2, st
    for o in outs:
        assert ffc.frame_digests(o.cpu().numpy())[0] == want["sha256"]


def test_host_buffer_entry_points_use_the_cache(lrp, torch_cuda):
    """lrp_reproject (what the C++ drop-in calls) and the batch context on host buffers."""
    name, case = _bicubic_cases(lrp)[3]
    src, lin, lout, rot = _small_setup(lrp, torch_cuda, case)
    s0 = lrp.geometry_cache_stats()
    for _ in range(3):
        out = np.full((case["oh"], case["ow"], case["c"]), -1.0, dtype=np.float32)
        lrp.reproject(lrp.Image(lin, case["iw"], case["ih"], case["c"], src), lrp.Image(lout, case["ow"], case["oh"], case["c"], out),
                      1, 2, rot)
        assert golden_cases.digest(out) == SMALL["reproject"][name]
    s1 = lrp.geometry_cache_stats()
    assert s1["fills"] == s0["fills"] + 1 and s1["hits"] == s0["hits"] + 2


def test_eight_threads_one_evicting(lrp, oracle, torch_cuda):
    """Seven host threads render their own geometry over and over (hits) while an eighth walks through new geometries under a
    cap that holds about eight entries, so that every one of its calls evicts somebody's entry: retired buffers are taken over
    behind their events or freed outside the lock — nothing on the launch path synchronises the device
    (csrc/lrp_geocache.cpp).  Every image against the live oracle; the launch calls of the non-evicting threads stay short."""
    import time

    torch = torch_cuda
    iw, ih, ow, oh, c = 96, 80, 72, 67, 4
    src = cases.hash_noise(ih, iw, c, seed=0xE71C, planted=True)
    d_in = torch.from_numpy(src).cuda()
    lin, lout = lrp.LensInfo.equirectangular(), lrp.LensInfo.rectilinear(18.0, 36.0, ow, oh)
    rots = [lrp.rotation_matrix(0.05 * k, -0.03 * k, 0.01 * k) for k in range(7 + 24)]
    wants = [oracle.reproject(lin, src, lout, ow, oh, 1, 2, r) for r in rots]
    one = lrp.geometry_cache_stats()
    entry_bytes = ((ow * oh * 8 + 255) // 256) * 256 + 8192  # (map + records + lists, roughly)
    lrp.geometry_cache_configure(8 * entry_bytes, 1)
    errors, latencies = [], [[] for _ in range(7)]
    barrier = threading.Barrier(8)

    def render(k, stream):
        with torch.cuda.stream(stream):  # (the fill on the stream of the launch: torch's streams do not wait for its default stream)
            out = torch.full((oh, ow, c), -12345.0, dtype=torch.float32, device="cuda")
        t0 = time.perf_counter()
        lrp.reproject(lrp.Image(lin, iw, ih, c, d_in), lrp.Image(lout, ow, oh, c, out), 1, 2, rots[k], stream=stream)
        dt = time.perf_counter() - t0
        stream.synchronize()
        if not np.array_equal(out.cpu().numpy().view(np.uint32), wants[k].view(np.uint32)):
            errors.append(("bits", k))
        return dt

    def steady(t):
        try:
            stream = torch.cuda.Stream()
            barrier.wait()
            for _ in range(40):
                latencies[t].append(render(t, stream))
        except Exception as exc:  # noqa: BLE001
            errors.append((t, repr(exc)))

    def evictor():
        try:
            stream = torch.cuda.Stream()
            barrier.wait()
            for rnd in range(3):
                for k in range(7, 7 + 24):
                    render(k, stream)
        except Exception as exc:  # noqa: BLE001
            errors.append(("evictor", repr(exc)))

    threads = [threading.Thread(target=steady, args=(t,)) for t in range(7)] + [threading.Thread(target=evictor)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
    st = lrp.geometry_cache_stats()
    assert st["evictions"] - one["evictions"] >= 40 and st["hits"] - one["hits"] >= 100, st
    assert st["bytes"] <= st["max_bytes"]
    med = sorted(x for lat in latencies for x in lat[2:])
    assert med[len(med) // 2] < 2e-3, f"median launch call of the steady threads {med[len(med) // 2] * 1e3:.2f} ms (max {med[-1] * 1e3:.2f} ms)"


def test_two_8k_output_geometries_fit_the_default_cap(lrp, torch_cuda):
    """Two 8192^2 output geometries (537 MB of coordinates each) alternate: under the default cap — min(4 GiB, 2 % of the
    device's memory), at least two entries of the largest geometry seen — both stay resident (hits, no eviction), and the cached
    launches render what the computing launches render (device checksums)."""
    torch = torch_cuda
    lrp.geometry_cache_configure(-2, 1)  # the default cap
    n_in, n_out, c = 1024, 8192, 3
    d_in = torch.empty((n_in, n_in, c), dtype=torch.float32, device="cuda")
    lrp.synth_fill(d_in, n_in, n_in, c, 0x8192)
    lin, lout = lrp.LensInfo.equirectangular(), lrp.LensInfo.rectilinear(18.0, 36.0, n_out, n_out)
    rots = [lrp.rotation_matrix(0.0, 0.0, 0.0), lrp.rotation_matrix(1.0, -0.4, 0.1)]
    d_out = torch.empty((n_out, n_out, c), dtype=torch.float32, device="cuda")

    def render(k):
        d_out.fill_(-1.0)
        lrp.reproject(lrp.Image(lin, n_in, n_in, c, d_in), lrp.Image(lout, n_out, n_out, c, d_out), 1, 2, rots[k])
        torch.cuda.synchronize()
        return lrp.checksums([d_out])[0]

    prev = lrp.debug_set("geo_cache", 0)
    want = [render(0), render(1)]
    lrp.debug_set("geo_cache", prev)
    s0 = lrp.geometry_cache_stats()
    for rnd in range(3):
        for k in (0, 1):
            assert render(k) == want[k], (rnd, k)
    s1 = lrp.geometry_cache_stats()
    assert s1["fills"] - s0["fills"] == 2 and s1["hits"] - s0["hits"] == 4 and s1["evictions"] == s0["evictions"], (s0, s1)
    assert s1["entries"] == 2 and s1["bytes"] > 2 * 8192 * 8192 * 8 and s1["max_bytes"] >= s1["bytes"]


def test_key_ignores_the_union_members_a_lens_does_not_have(lrp, torch_cuda):
    """A C caller sets focal_length of a rectilinear lens and leaves the other three floats of the union indeterminate: the
    geometry is the same, the cache key must be too (every such call missed, filled and evicted)."""
    torch = torch_cuda
    name, case = _bicubic_cases(lrp)[0]
    src, lin, lout, rot = _small_setup(lrp, torch, case)
    d_in = torch.from_numpy(src).cuda()
    s0 = lrp.geometry_cache_stats()
    for junk in (0.0, 1.5, -7.0e30, float("nan")):
        def garbage(lens):
            if int(lens.type) == int(lrp.LensType.EQUIRECTANGULAR):
                return lens
            p = list(lens.params)
            keep = 1
            return lrp.LensInfo(lens.type, tuple(p[:keep] + [junk] * (4 - keep)), lens.sensor_width, lens.sensor_height)

        d_out = torch.full((case["oh"], case["ow"], case["c"]), -1.0, dtype=torch.float32, device="cuda")
        lrp.reproject(lrp.Image(garbage(lin), case["iw"], case["ih"], case["c"], d_in), lrp.Image(garbage(lout), case["ow"], case["oh"], case["c"], d_out),
                      1, 2, rot)
        torch.cuda.synchronize()
        assert golden_cases.digest(d_out.cpu().numpy()) == SMALL["reproject"][name]
    s1 = lrp.geometry_cache_stats()
    if int(lin.type) != int(lrp.LensType.EQUIRECTANGULAR) or int(lout.type) != int(lrp.LensType.EQUIRECTANGULAR):
        assert s1["fills"] - s0["fills"] == 1 and s1["hits"] - s0["hits"] == 3, (s0, s1)


def test_two_streams_race_on_one_geometry_with_different_pixels(lrp, oracle, torch_cuda):
    """Single launches of ONE geometry dealt alternately to two streams — what lrp_context does with consecutive images so that
    the tail of one launch overlaps the head of the next —: different pixels in every launch, the entry filled by the first
    launch while the second is already enqueued on the other stream; and the same through a BatchContext with three slots
    (two compute streams inside).  Every image against the live oracle."""
    torch = torch_cuda
    iw, ih, ow, oh, c = 200, 120, 147, 131, 4
    lin, lout = lrp.LensInfo.equidistant(3.14159265), lrp.LensInfo.rectilinear(18.0, 36.0, ow, oh)
    rot = lrp.rotation_matrix(0.3, -0.2, 0.1)
    srcs = [cases.hash_noise(ih, iw, c, seed=0x2570 + k, planted=(k % 2 == 0)) for k in range(8)]
    wants = [oracle.reproject(lin, s, lout, ow, oh, 1, 2, rot) for s in srcs]
    d_ins = [torch.from_numpy(s).cuda() for s in srcs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for rnd in range(3):
        outs = [torch.full((oh, ow, c), -12345.0, dtype=torch.float32, device="cuda") for _ in srcs]
        torch.cuda.synchronize()
        for k in range(8):
            lrp.reproject(lrp.Image(lin, iw, ih, c, d_ins[k]), lrp.Image(lout, ow, oh, c, outs[k]), 1, 2, rot, stream=streams[k & 1])
        torch.cuda.synchronize()
        for k in range(8):
            cases.assert_same_bits(outs[k].cpu().numpy(), wants[k], f"round {rnd}, image {k} on stream {k & 1}")
    st = lrp.geometry_cache_stats()
    assert st["fills"] >= 1 and st["hits"] >= 16, st
    for streams_knob in (1, 0):
        prev = lrp.debug_set("context_streams", streams_knob)
        host_outs = [np.full((oh, ow, c), -1.0, dtype=np.float32) for _ in srcs]
        with lrp.BatchContext(0, 3) as ctx:
            for k in range(8):
                ctx.submit(lrp.Image(lin, iw, ih, c, srcs[k]), lrp.Image(lout, ow, oh, c, host_outs[k]), 1, 2, rot)
            ctx.wait()
        lrp.debug_set("context_streams", prev)
        for k in range(8):
            cases.assert_same_bits(host_outs[k], wants[k], f"BatchContext (context_streams {streams_knob}), image {k}")

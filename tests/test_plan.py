"""CPU: the launch planner (csrc/lrp_plan.cpp) — every decision of lrp_capi.cpp's enqueue_reproject as pure functions of the
request, the switches and what the host learns on the way (output-lens tables, geometry cache) — asked without a GPU through
tests/native/plan_driver.cpp.  Table-driven: the five BASELINE configs, a cubemap's pole face against a side
face, batches of 16 / 256, row bands, supersampling, the fallbacks to the pixel kernel, the A/B switches.  A changed threshold or
rule (kListedCornerPercent, kBigWidePercent, kMinWavesForFusedFill, who goes to the geometry cache) fails a row here."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "image-lens-reproject_amd", "csrc")
BUILD = os.path.join(ROOT, "tests", "native", "_build")
RECT, EQD, EQR = 0, 1, 4                     # lens types (include/lrp.h numbering)
IN_RECT, IN_EQD, IN_EQR, IN_LOOP = 0, 1, 2, 3  # input modes (lrp_params.h)
NN, BL, BC = 0, 1, 2
TWO_PI = float(np.float32(math.pi) - np.float32(-math.pi))


def rotation(pan, pitch, roll):
    """R = R_y(pan) R_x(pitch) R_z(roll) in binary32 (src/main.cpp:110-142), degrees."""
    p, t, r = (np.float32(math.radians(v)) for v in (pan, pitch, roll))
    c, s = np.cos, np.sin
    ry = np.array([[c(p), 0, s(p)], [0, 1, 0], [-s(p), 0, c(p)]], dtype=np.float32)
    rx = np.array([[1, 0, 0], [0, c(t), -s(t)], [0, s(t), c(t)]], dtype=np.float32)
    rz = np.array([[c(r), -s(r), 0], [s(r), c(r), 0], [0, 0, 1]], dtype=np.float32)
    return ",".join(repr(float(v)) for v in (ry @ rx @ rz).reshape(-1))


IDENTITY = "1,0,0,0,1,0,0,0,1"
GEN = rotation(30, -15, 5)
HEAD = dict(out_type=RECT, in_type=EQD, in_mode=IN_EQD)                                         # configs[1]: fisheye -> rect bicubic
C0 = dict(out_type=RECT, in_type=EQR, in_mode=IN_LOOP, out_w=512, out_h=512, in_w=512, in_h=512, interp=NN, rot=IDENTITY)
C2 = dict(out_type=EQD, in_type=EQR, in_mode=IN_LOOP, interp=BL, rot=GEN)                       # configs[2]
C3 = dict(out_type=EQR, in_type=RECT, in_mode=IN_RECT, channels=5, rot=IDENTITY, out_lon_span=TWO_PI)  # configs[3]
FACE = dict(out_type=RECT, in_type=EQR, in_mode=IN_LOOP, in_w=8192, in_h=8192, out_w=2048, out_h=2048, channels=3)  # configs[4]
POLE, SIDE = dict(FACE, rot=rotation(0, 90, 0)), dict(FACE, rot=rotation(90, 0, 0))
C3_LISTS = {"g.mode": 2, "g.lists": 1, "g.n_blocks": 65536, "g.n_corner_blocks": 24248, "g.n_work": 41288, "g.n_runs": 3000, "g.n_inview": 15000, "g.n_wide": 9000}

CASES = [
    # BASELINE configs[0]: nearest without a rotation (the CLI's identity is dropped) — mirrored pixels, no geometry cache
    ("configs[0]", C0, dict(family="tile", has_rot=0, wants_xsep=1, quad=1, wants_geo=0, geo_mode=0)),
    ("configs[0], the tables hold a -0.0f: the identity stays", dict(C0, **{"t.plain": 0}), dict(has_rot=1, wants_xsep=0, quad=0, wants_geo=1)),
    # configs[1], the headline: first call fills the entry, later calls read it; without the cache the mirrored blocks run
    ("configs[1] first call", dict(HEAD, **{"g.mode": 1}), dict(family="window", wants_tables=1, wants_geo=1, geo_want_boxes=1, geo_mode=1, win_mode=0, quad=0, win_coef=1)),
    ("configs[1] later calls", dict(HEAD, **{"g.mode": 2}), dict(family="window", geo_mode=2, big_windows=0, listed=0, win_mode=0)),
    ("configs[1], cache off", dict(HEAD, **{"s.geo_cache": 0, "g.mode": 2}), dict(family="window", wants_geo=0, geo_mode=0, win_mode=1, quad=1)),
    ("configs[1], a batch of 16", dict(HEAD, n_batch=16, **{"g.mode": 2}), dict(family="window", geo_mode=2, frames_per_wave=0, big_windows=0)),
    ("configs[1], a row band", dict(HEAD, band=1, **{"g.mode": 2}), dict(family="window", wants_geo=0, win_mode=0, quad=0)),
    # north_star's case: the CLI's identity matrix is dropped, the panorama's x is column-separable
    ("equirect -> rect bicubic, identity", dict(out_type=RECT, in_type=EQR, in_mode=IN_LOOP, rot=IDENTITY, **{"s.geo_cache": 0}), dict(has_rot=0, wants_xsep=1, win_mode=1)),
    ("equirect -> rect bicubic, general rotation", dict(out_type=RECT, in_type=EQR, in_mode=IN_LOOP, rot=GEN, **{"s.geo_cache": 0}), dict(has_rot=1, wants_xsep=0, win_mode=0, quad=0)),
    # configs[2]: bilinear into a fisheye frame, rotated — shared rays without the cache, the coordinate map with it; batches read it too
    ("configs[2] single", dict(C2, **{"g.mode": 2}), dict(family="tile", wants_tables=0, wants_geo=1, geo_want_boxes=0, geo_mode=2, quad=0)),
    ("configs[2] single, cache off", dict(C2, **{"s.geo_cache": 0}), dict(family="tile", quad=2, wants_geo=0)),
    ("configs[2] batch of 256", dict(C2, n_batch=256, **{"g.mode": 2}), dict(family="tile", wants_geo=1, geo_mode=2, quad=0)),
    ("configs[2] shape, nearest batch: the frame loop, no cache", dict(C2, interp=NN, n_batch=256), dict(family="tile", wants_geo=0, quad=0)),
    ("configs[2] shape, bicubic, single: shared rays before the entry exists", dict(C2, interp=BC, **{"s.geo_cache": 0}), dict(family="window", win_mode=4)),
    ("configs[2] shape, bicubic, batch: plain blocks", dict(C2, interp=BC, n_batch=16, **{"s.geo_cache": 0}), dict(family="window", win_mode=0)),
    # configs[3]: RGBAZ rect -> full panorama; alias pairs, the big-window variant, rendering by block class from 30 % corner blocks on
    ("configs[3] first call", dict(C3, **{"g.mode": 1}), dict(family="window", has_rot=0, alias_pairs=1, geo_mode=1, rgbaz_runs=1, big_windows=1, listed=0)),
    ("configs[3] reading, lists known (37 % corner blocks)", dict(C3, **C3_LISTS),
     dict(geo_mode=2, big_windows=1, listed=1, list_recs=1, fill_stride=13, fill_per_wave=16, win_tapdma=1, win_split=1, win_edge=1)),
    ("configs[3] reading, lists not yet known", dict(C3, **dict(C3_LISTS, **{"g.lists": 0})), dict(geo_mode=2, big_windows=1, listed=0, fill_stride=0)),
    ("a wider view: 20 % corner blocks stay enumerated", dict(C3, **dict(C3_LISTS, **{"g.n_corner_blocks": 13107})), dict(listed=0)),
    ("... 30 % exactly are listed", dict(C3, **dict(C3_LISTS, **{"g.n_corner_blocks": 19661})), dict(listed=1)),
    ("... and any share with geo_lists 2", dict(C3, **dict(C3_LISTS, **{"g.n_corner_blocks": 100, "s.geo_lists": 2})), dict(listed=1)),
    ("a frame of (almost) nothing but corner blocks: the fill kernel at its own occupancy", dict(C3, **dict(C3_LISTS, **{"g.n_corner_blocks": 64000, "g.n_work": 1536})),
     dict(listed=1, fill_stride=0, fill_per_wave=0)),
    ("configs[3], partial panorama: no second copy behind the camera", dict(C3, out_lon_span=3.0, **{"g.mode": 1}), dict(alias_pairs=0)),
    ("configs[3], pitched: no alias pairs", dict(C3, rot=rotation(0, 20, 0), **{"g.mode": 1}), dict(alias_pairs=0, has_rot=1)),
    ("configs[3] RGBA, geo_big 0", dict(C3, channels=4, **dict(C3_LISTS, **{"s.geo_big": 0})), dict(big_windows=0, listed=1)),
    ("rect -> rect: lists but few corners, no wide block", dict(out_type=RECT, in_type=RECT, in_mode=IN_RECT, **dict(C3_LISTS, **{"g.n_corner_blocks": 0, "g.n_wide": 0})),
     dict(rgbaz_runs=0, big_windows=0, listed=0)),
    ("rect -> fisheye: 49 % corners, wide blocks: listed, big windows", dict(out_type=EQD, in_type=RECT, in_mode=IN_RECT, **dict(C3_LISTS, **{"g.n_corner_blocks": 32000, "g.n_wide": 6000, "g.n_inview": 15000})),
     dict(rgbaz_runs=0, big_windows=1, listed=1)),
    # configs[4]: the census decides per face; before the entry exists a side face mirrors its rows, a pole face its columns
    ("cubemap pole face, census: 73 % of the in-view blocks too wide", dict(POLE, **{"g.mode": 2, "g.lists": 1, "g.n_inview": 16384, "g.n_wide": 12000}), dict(family="window", big_windows=1, listed=0)),
    ("cubemap side face, census: none", dict(SIDE, **{"g.mode": 2, "g.lists": 1, "g.n_inview": 16384, "g.n_wide": 0}), dict(big_windows=0, listed=0)),
    ("a face at 29 % stays with the four-wavefront kernel", dict(POLE, **{"g.mode": 2, "g.lists": 1, "g.n_inview": 10000, "g.n_wide": 2999}), dict(big_windows=0)),
    ("... 30 % takes the variant", dict(POLE, **{"g.mode": 2, "g.lists": 1, "g.n_inview": 10000, "g.n_wide": 3000}), dict(big_windows=1)),
    ("cubemap side face without the cache: rows mirrored (pan)", dict(SIDE, **{"s.geo_cache": 0}), dict(wants_xsep=1, win_mode=2)),
    ("cubemap pole face without the cache: columns mirrored (pitch into rect)", dict(POLE, **{"s.geo_cache": 0}), dict(wants_xsep=0, win_mode=3)),
    ("... mirror modes off", dict(POLE, **{"s.geo_cache": 0, "s.mirror_modes": 0}), dict(win_mode=0)),
    ("an equidistant source never takes the variant by census", dict(HEAD, **{"g.mode": 2, "g.lists": 1, "g.n_inview": 100, "g.n_wide": 100}), dict(big_windows=0)),
    # supersampling (the reference's --samples): the window kernel's SS instantiations, never the cache
    ("--samples 2 bicubic, first call: fills an entry of sub-samples", dict(HEAD, ns=2, out_w=2048, out_h=2048, **{"g.mode": 1}),
     dict(family="window", wants_geo=1, geo_want_boxes=0, geo_mode=1, quad=0, win_mode=0, big_windows=0, listed=0)),
    ("--samples 2 bicubic, later calls read it", dict(HEAD, ns=2, out_w=2048, out_h=2048, **{"g.mode": 2, "g.lists": 1, "g.n_inview": 9, "g.n_wide": 9}),
     dict(family="window", geo_mode=2, big_windows=0, listed=0, blocks_per_wave=0)),
    ("--samples 2 bicubic, rect -> equirect: cheap coordinates are computed, not loaded", dict(C3, channels=4, ns=2, out_w=2048, out_h=2048, **{"g.mode": 2}), dict(family="window", wants_geo=0, geo_mode=0)),
    ("--samples 2 bicubic, rect -> rect reads its entry like everybody", dict(out_type=RECT, in_type=RECT, in_mode=IN_RECT, ns=2, out_w=2048, out_h=2048, **{"g.mode": 2}), dict(family="window", wants_geo=1, geo_mode=2)),
    ("--samples 2 bicubic, cache off", dict(HEAD, ns=2, out_w=2048, out_h=2048, **{"s.geo_cache": 0, "g.mode": 2}), dict(family="window", wants_geo=0, geo_mode=0)),
    ("--samples 2 bicubic, a row band", dict(HEAD, ns=2, out_w=2048, out_h=2048, band=1), dict(family="window", wants_geo=0)),
    ("--samples 4 of a huge output: 2^31 sub-samples and more are not cached", dict(HEAD, ns=4, out_w=16384, out_h=8192), dict(family="window", wants_geo=0)),
    ("--samples 2 bicubic, win_ss 0", dict(HEAD, ns=2, **{"s.win_ss": 0}), dict(family="tile", wants_geo=0)),
    ("--samples 3 bicubic (--scale 0.33334)", dict(HEAD, ns=3, out_w=1365, out_h=1365, **{"g.mode": 2}), dict(family="window", wants_geo=1, geo_mode=2, quad=0, win_mode=0)),
    ("--samples 4 bicubic (--scale 0.25)", dict(HEAD, ns=4, out_w=1024, out_h=1024, **{"g.mode": 1}), dict(family="window", wants_geo=1, geo_mode=1)),
    ("--samples 5 bicubic: the tile kernel", dict(HEAD, ns=5, out_w=819, out_h=819), dict(family="tile", wants_geo=0)),
    ("--samples 2 bilinear: the tile kernel on the same entry of sub-samples", dict(HEAD, ns=2, interp=BL, **{"g.mode": 2}), dict(family="tile", wants_geo=1, geo_want_boxes=0, geo_mode=2, quad=0)),
    ("--samples 2 nearest into a fisheye frame, first call", dict(C2, ns=2, interp=NN, **{"g.mode": 1}), dict(family="tile", wants_geo=1, geo_mode=1, quad=0)),
    ("--samples 2 bilinear, a batch: the first frame writes, the rest read", dict(C2, ns=2, n_batch=16, **{"g.mode": 2}), dict(family="tile", wants_geo=1, geo_mode=2)),
    ("--samples 3 bilinear reads the entry (a lane per sub-sample: lrp_ss_gather_kernel.h)", dict(C2, ns=3, **{"g.mode": 2}), dict(family="tile", wants_geo=1, geo_mode=2)),
    ("--samples 4 nearest, first call", dict(C2, ns=4, interp=NN, **{"g.mode": 1}), dict(family="tile", wants_geo=1, geo_mode=1)),
    ("--samples 2 bilinear, rect -> equirect: cheap coordinates are computed", dict(C3, channels=4, ns=2, interp=BL, **{"g.mode": 2}), dict(family="tile", wants_geo=0, geo_mode=0)),
    ("--samples 5 bilinear: computed", dict(C2, ns=5, **{"g.mode": 2}), dict(family="tile", wants_geo=0)),
    ("--samples 3 bilinear, equirect -> rect without a rotation: the source x comes from the column table, computed", dict(C0, interp=BL, ns=3, **{"g.mode": 2}), dict(family="tile", wants_xsep=1, wants_geo=0)),
    ("... rotated: loaded", dict(C0, interp=BL, ns=3, rot=GEN, **{"g.mode": 2}), dict(family="tile", wants_xsep=0, wants_geo=1, geo_mode=2)),
    # where the map does not pay: a rectilinear source under a rectilinear / panorama target, nearest / bilinear
    ("rect -> equirect bilinear: four divides beat 8 B per pixel", dict(C3, channels=4, interp=BL), dict(family="tile", wants_geo=0)),
    ("rect -> fisheye bilinear reads the map", dict(out_type=EQD, in_type=RECT, in_mode=IN_RECT, interp=BL, **{"g.mode": 2}), dict(wants_geo=1, geo_mode=2)),
    # fallbacks to the one-pixel-per-lane kernel
    ("seven channels", dict(HEAD, channels=7), dict(family="pixel", wants_tables=0, wants_geo=0)),
    ("an image of 4 GiB and more", dict(HEAD, fits=0), dict(family="pixel")),
    ("no memory for the output-lens tables", dict(HEAD, **{"t.built": 0}), dict(family="pixel", wants_geo=0)),
    ("a source wider than 65535 texels", dict(HEAD, in_w=70000, in_h=4), dict(family="pixel")),
    ("kernel family 0", dict(HEAD, **{"s.kernel": 0}), dict(family="pixel")),
    ("kernel family 1: tile kernels only, no cache", dict(HEAD, **{"s.kernel": 1}), dict(family="tile", wants_geo=0, quad=1)),
    ("kernel family 3: no work sharing, raw taps only", dict(HEAD, **{"s.kernel": 3}), dict(family="window", quad=0, win_mode=0, win_coef=0, win_edge=0, win_split=0, wants_geo=0)),
    ("switches reach the plan", dict(C3, **dict(C3_LISTS, **{"s.geo_strip": 4, "s.batch_frames": 8, "s.geo_list_recs": 0, "s.geo_fill_fused": 0, "s.win_tapdma": 0})),
     dict(blocks_per_wave=4, frames_per_wave=8, list_recs=0, fill_stride=0, win_tapdma=0, listed=1)),
]


@pytest.fixture(scope="module")
def planner():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "plan_driver")
    # plain g++: the planner includes no HIP header
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + CSRC, os.path.join(ROOT, "tests", "native", "plan_driver.cpp"),
                    os.path.join(CSRC, "lrp_plan.cpp"), "-o", exe], check=True, cwd=ROOT)

    def ask(request):
        line = " ".join(f"{k}={v}" for k, v in request.items())
        r = subprocess.run([exe], input=line + "\n", capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        return json.loads(r.stdout)

    return ask


@pytest.mark.parametrize("what,request_,expect", CASES, ids=[c[0] for c in CASES])
def test_plan(planner, what, request_, expect):
    plan = planner(request_)
    got = {k: plan[k] for k in expect}
    assert got == expect, f"{what}: {plan}"


def test_fused_fill_shares_cover_every_run(planner):
    """The shares of the fused corner fill: every stride-th wavefront (odd stride: all XCDs) writes fill_per_wave row segments, and
    together they cover all 16 x n_runs of them — for every size of the two lists."""
    for n_work in (2048, 2049, 5000, 41288, 65536, 200000):
        for n_runs in (1, 7, 500, 3000, 65536):
            p = planner(dict(C3, **dict(C3_LISTS, **{"g.n_work": n_work, "g.n_runs": n_runs, "g.n_corner_blocks": 30000})))
            assert p["listed"] == 1 and p["fill_stride"] % 2 == 1 and p["fill_per_wave"] >= 1, (n_work, n_runs, p)
            fillers = -(-n_work // p["fill_stride"])
            assert fillers * p["fill_per_wave"] >= 16 * n_runs, (n_work, n_runs, p)
            assert fillers >= min(n_work, 1024) // 2, (n_work, n_runs, p)  # spread over at least ~a thousand wavefronts
    p = planner(dict(C3, **dict(C3_LISTS, **{"g.n_work": 2047})))
    assert p["listed"] == 1 and p["fill_per_wave"] == 0  # below kMinWavesForFusedFill: the stand-alone fill kernel


def test_the_planner_has_no_hip_dependency():
    for name in ("lrp_plan.h", "lrp_plan.cpp"):
        text = open(os.path.join(CSRC, name)).read()
        assert "hip/" not in text and "hipStream" not in text and "getenv" not in text, name
    capi = open(os.path.join(CSRC, "lrp_capi.cpp")).read()
    assert "getenv" not in capi  # (round 6: the library reads no environment variable)

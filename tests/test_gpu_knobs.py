"""-m gpu: every A/B switch of the library (lrp_debug_set; DESIGN.md section 2 "knob matrix") under the driver's eyes.

Each setting switches one sharing / staging path of the tile / window kernels off (or forces a batch shape): the bits
must not change.  For every setting: the 216-case matrix of tests/golden/oracle_golden.json (single launches, and as
batches of five frames where the setting is about batches) and four whole frames — BASELINE configs[1], configs[3], the
north_star mapping under a general rotation and the pole face of the configs[4] cubemap — against the COMMITTED oracle
digests.  The cubemap itself (six faces through lrp_reproject_multi_device) runs under the multi_fork settings."""
import json
import os

import numpy as np
import pytest

import cases
import fullframe_cases as ffc
import golden_cases

pytestmark = pytest.mark.gpu
USES_GEO_CACHE = True  # (every setting names the geometry-cache switch itself)

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "fullframe_golden.json")) as _f:
    FULL = json.load(_f)
with open(os.path.join(HERE, "golden", "oracle_golden.json")) as _f:
    SMALL = json.load(_f)

# name -> switches (everything not named keeps its default; geo_cache 0 so that the compute kernels — mirror modes, edge
# and split blocks, the frame loop — are what runs, except in the settings that are about the cache)
SETTINGS = {
    "defaults": {"geo_cache": 0},
    "geo_cache": {"geo_cache": 1},
    "geo_cache_strip1": {"geo_cache": 1, "geo_strip": 1},
    "geo_cache_strip4": {"geo_cache": 1, "geo_strip": 4},
    "geo_cache_big0": {"geo_cache": 1, "geo_big": 0},  # (a rectilinear view into a panorama through the four-wavefront instantiation)
    "xsep0": {"geo_cache": 0, "xsep": 0},
    "quad0": {"geo_cache": 0, "quad": 0},
    "mirror_modes0": {"geo_cache": 0, "mirror_modes": 0},
    "win_edge0": {"geo_cache": 0, "win_edge": 0},
    "win_edge0_geo": {"geo_cache": 1, "win_edge": 0},
    "win_split0": {"geo_cache": 0, "win_split": 0},
    "win_split0_geo": {"geo_cache": 1, "win_split": 0},
    "kernel_tile": {"geo_cache": 0, "kernel": 1},
    "kernel_window_raw": {"geo_cache": 0, "kernel": 3},
    "kernel_pixel": {"geo_cache": 0, "kernel": 0},
    "batch_frames1": {"geo_cache": 0, "batch_frames": 1},
    "batch_frames5": {"geo_cache": 0, "batch_frames": 5},
    "batch_frames16_geo": {"geo_cache": 1, "batch_frames": 16},
    "batch_frames3_geo": {"geo_cache": 1, "batch_frames": 3},
    "batch_frames0_geo": {"geo_cache": 1},  # (batched launches read the geometry cache; the first frame of the first batch writes it)
    "multi_fork0": {"geo_cache": 0, "multi_fork": 0},
    "multi_fork3": {"geo_cache": 0, "multi_fork": 3},
    "multi_fork3_geo": {"geo_cache": 1, "multi_fork": 3},
    # rendering by block class (round 5): never / whenever the lists of an entry are known, the corner runs by the fill kernel
    # instead of a share per wavefront, that kernel on a side stream
    "geo_lists0": {"geo_cache": 1, "geo_lists": 0},
    "geo_lists2": {"geo_cache": 1, "geo_lists": 2},
    "geo_lists2_fill_kernel": {"geo_cache": 1, "geo_lists": 2, "geo_fill_fused": 0},
    "geo_lists2_fill_stream": {"geo_cache": 1, "geo_lists": 2, "geo_fill_fused": 0, "geo_fill_stream": 1},
    "win_tapdma0": {"geo_cache": 1, "win_tapdma": 0},  # (the big-window variant: passes whose window fits no buffer gather per lane)
    "geo_lists2_tapdma0": {"geo_cache": 1, "geo_lists": 2, "win_tapdma": 0},
    "geo_lists2_recs0": {"geo_cache": 1, "geo_lists": 2, "geo_list_recs": 0},  # (a listed wavefront loads its block's record from the box array)
    "win_ss0": {"geo_cache": 0, "win_ss": 0},  # (bicubic with num_samples == 2 through the tile kernel)
}
FRAMES = ["config1_4k_eqd_rect_bc", "config3_4k_rgbaz_rect_eqr_bc_post", "4k_eqr_rect_bc_rot", "config4_8k_rgb_face4"]


class _Knobs:
    def __init__(self, lrp, values):
        self.lrp, self.values, self.prev = lrp, values, {}

    def __enter__(self):
        self.lrp.release_cached_tables()
        for k, v in self.values.items():
            self.prev[k] = self.lrp.debug_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            self.lrp.debug_set(k, v)
        self.lrp.release_cached_tables()


class _DeviceSynth:
    def __init__(self, lrp, torch):
        self.lrp, self.torch = lrp, torch

    def synth_frame(self, width, height, channels, seed, depth_channel=-1):
        t = self.torch.empty((height, width, channels), dtype=self.torch.float32, device="cuda")
        self.lrp.synth_fill(t, width, height, channels, seed, depth_channel)
        self.torch.cuda.synchronize()
        return t.cpu().numpy()


def test_switch_names_and_ranges(lrp):
    for name in ("kernel", "xsep", "quad", "mirror_modes", "win_edge", "win_split", "batch_frames", "multi_fork", "geo_cache", "geo_strip", "geo_big",
                 "geo_lists", "geo_fill_fused", "geo_fill_stream", "context_streams", "win_ss", "win_tapdma", "geo_list_recs", "geo_census"):
        now = lrp.debug_set(name, -1)
        assert lrp.debug_set(name, now) == now  # setting the current value returns it
        assert lrp.debug_set(name, 10 ** 6) == now and lrp.debug_set(name, -1) == now  # out of range: a query
    with pytest.raises(ValueError):
        lrp.debug_set("no_such_switch", 1)


@pytest.mark.parametrize("setting", sorted(SETTINGS))
def test_small_matrix_under_setting(lrp, torch_cuda, setting):
    torch = torch_cuda
    batched = setting.startswith("batch_frames")
    synth = _DeviceSynth(lrp, torch)
    with _Knobs(lrp, SETTINGS[setting]):
        for name, case in golden_cases.all_cases(lrp):
            src = golden_cases.planted_input(synth, case["iw"], case["ih"], case["c"], case["seed"])
            lin = cases.lenses(lrp, case["iw"], case["ih"])[case["inp"]]
            lout = cases.lenses(lrp, case["ow"], case["oh"])[case["out"]]
            rot = cases.rotation(lrp, golden_cases.ROTS[case["rot"]])
            n = 5 if batched else 2  # (twice: with the geometry cache on, the launch that fills an entry and one that reads it)
            d_ins = [torch.from_numpy(src).cuda() for _ in range(n if batched else 1)]
            d_outs = [torch.full((case["oh"], case["ow"], case["c"]), -12345.0, dtype=torch.float32, device="cuda") for _ in range(n)]
            if batched:
                lrp.reproject_batch([lrp.Image(lin, case["iw"], case["ih"], case["c"], d) for d in d_ins],
                                    [lrp.Image(lout, case["ow"], case["oh"], case["c"], d) for d in d_outs], case["ns"], case["interp"], rot)
            else:
                for d in d_outs:
                    lrp.reproject(lrp.Image(lin, case["iw"], case["ih"], case["c"], d_ins[0]),
                                  lrp.Image(lout, case["ow"], case["oh"], case["c"], d), case["ns"], case["interp"], rot)
                    torch.cuda.synchronize()  # (the block lists of an entry are known once the launch that filled it has completed)
            torch.cuda.synchronize()
            for i, d in enumerate(d_outs):
                assert golden_cases.digest(d.cpu().numpy()) == SMALL["reproject"][name], f"{setting}: {name} (output {i})"


def _frame(lrp, torch, case, batched):
    n, m, c = case["size"], case["out_size"], case["c"]
    d_in = torch.empty((n, n, c), dtype=torch.float32, device="cuda")
    lrp.synth_fill(d_in, n, n, c, case["seed"], case.get("depth", -1))
    lin, lout = cases.lenses(lrp, n, n)[case["inp"]], cases.lenses(lrp, m, m)[case["out"]]
    rot = cases.rotation(lrp, case["deg"])
    post = tuple(case["post"]) if case.get("post") else None
    count = 3 if batched else 2
    outs = [torch.full((m, m, c), -12345.0, dtype=torch.float32, device="cuda") for _ in range(count)]
    if batched:
        lrp.reproject_batch([lrp.Image(lin, n, n, c, d_in)] * count, [lrp.Image(lout, m, m, c, o) for o in outs], 1, case["interp"], rot,
                            post=post)
    else:
        for o in outs:
            lrp.reproject(lrp.Image(lin, n, n, c, d_in), lrp.Image(lout, m, m, c, o), 1, case["interp"], rot, post=post)
            torch.cuda.synchronize()  # (the block lists of an entry are known once the launch that filled it has completed)
    torch.cuda.synchronize()
    return outs


@pytest.mark.parametrize("setting", sorted(SETTINGS))
def test_whole_frames_under_setting(lrp, torch_cuda, setting):
    torch = torch_cuda
    with _Knobs(lrp, SETTINGS[setting]):
        for name in FRAMES:
            case, want = ffc.frame_cases()[name], FULL["frames"][name]
            outs = _frame(lrp, torch, case, setting.startswith("batch_frames"))
            for i in (0, len(outs) - 1):  # the first and the last launch / frame (geometry cache: the filling and a reading launch)
                sha, _bands, n_nan = ffc.frame_digests(outs[i].cpu().numpy())
                assert sha == want["sha256"] and n_nan == want["nan"], f"{setting}: {name} (output {i})"
            del outs
            torch.cuda.empty_cache()


@pytest.mark.parametrize("setting", ["multi_fork0", "multi_fork3", "multi_fork3_geo", "defaults", "geo_cache"])
def test_cubemap_through_multi_under_setting(lrp, torch_cuda, setting):
    """BASELINE configs[4]: the six faces of an 8192^2 RGB panorama in ONE lrp_reproject_multi_device call, twice."""
    torch = torch_cuda
    names = [f"config4_8k_rgb_face{i}" for i in range(6)]
    case0 = ffc.frame_cases()[names[0]]
    n, m, c = case0["size"], case0["out_size"], case0["c"]
    d_in = torch.empty((n, n, c), dtype=torch.float32, device="cuda")
    lrp.synth_fill(d_in, n, n, c, case0["seed"])
    lin, lout = cases.lenses(lrp, n, n)[case0["inp"]], cases.lenses(lrp, m, m)[case0["out"]]
    rots = np.stack([cases.rotation(lrp, ffc.frame_cases()[nm]["deg"]) for nm in names])
    with _Knobs(lrp, SETTINGS[setting]):
        for rnd in range(2):
            outs = [torch.full((m, m, c), -1.0, dtype=torch.float32, device="cuda") for _ in names]
            lrp.reproject_multi(lrp.Image(lin, n, n, c, d_in), [lrp.Image(lout, m, m, c, o) for o in outs], 1, case0["interp"], rots)
            torch.cuda.synchronize()
            for nm, o in zip(names, outs):
                assert ffc.frame_digests(o.cpu().numpy())[0] == FULL["frames"][nm]["sha256"], f"{setting}: {nm} (round {rnd})"

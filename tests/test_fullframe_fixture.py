"""CPU: tests/golden/fullframe_golden.json is complete, matches the case definitions, and the oracle
of THIS checkout still reproduces it (a sample of whole frames: the 512^2 plumbing config, one 4K
headline frame, a cubemap face, the super-sampled case, two images of the bench batch) — so an edit
of the oracle, its flags or the host libm cannot drift away from the committed digests unnoticed."""
import hashlib
import importlib
import json
import os

import pytest

import cases
import fullframe_cases as ffc

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "fullframe_golden.json")) as _f:
    FULL = json.load(_f)

THREADS = max(1, len(os.sched_getaffinity(0)))


def _render(lrp, oracle, case, seed=None):
    n, m, c = case["size"], case["out_size"], case["c"]
    src = oracle.synth_frame(n, n, c, case["seed"] if seed is None else seed, depth_channel=case.get("depth", -1))
    lin, lout = cases.lenses(lrp, n, n)[case["inp"]], cases.lenses(lrp, m, m)[case["out"]]
    out = oracle.reproject(lin, src, lout, m, m, case.get("ns", 1), case["interp"], cases.rotation(lrp, case["deg"]), threads=THREADS)
    if case.get("post"):
        oracle.post_process(out, *case["post"])
    return out


def test_fixture_covers_every_case_and_every_baseline_config():
    names = set(ffc.frame_cases())
    assert names == set(FULL["frames"]), "re-run tests/golden/make_fullframe_golden.py"
    for prefix in ("config0_", "config1_", "config2_", "config3_", "config4_"):
        assert any(n.startswith(prefix) for n in names), prefix
    for name, case in ffc.frame_cases().items():
        stored = FULL["frames"][name]
        assert {k: (list(v) if isinstance(v, tuple) else v) for k, v in case.items() if k != "name"} == stored["case"], name
        assert len(stored["bands"]) == ffc.BANDS and len(stored["sha256"]) == 64
    sums = FULL["bench_batch"]["fisheye_to_rect_bicubic"]["checksums"]
    assert len(sums) == ffc.BENCH_BATCH == 256 and len(set(sums)) == 256


@pytest.mark.parametrize("name", ["config0_512_eqr_rect_nn", "config1_4k_eqd_rect_bc", "config4_8k_rgb_face4",
                                  "config3_4k_rgbaz_rect_eqr_bc_post", "2k_eqd_rect_bc_ns2"])
def test_oracle_reproduces_committed_frame(lrp, oracle, name):
    out = _render(lrp, oracle, ffc.frame_cases()[name])
    sha, bands, n_nan = ffc.frame_digests(out)
    want = FULL["frames"][name]
    assert bands == want["bands"] and sha == want["sha256"] and n_nan == want["nan"]
    if want["checksum"] is not None:
        assert f"{oracle.checksum(out):016x}" == want["checksum"]


def test_oracle_reproduces_bench_batch_checksums_and_digest(lrp, oracle):
    wl = FULL["bench_batch"]["fisheye_to_rect_bicubic"]
    case = dict(wl["case"], out_size=wl["case"]["size"], seed=0)
    for i in (0, 255):
        assert f"{oracle.checksum(_render(lrp, oracle, case, seed=0x5EED0000 + i)):016x}" == wl["checksums"][i], i
    # bench.py's digest of a batch is the sha256 of the comma-joined checksums in image order
    bench = importlib.import_module("bench")
    assert bench.golden_batch_digest("fisheye_to_rect_bicubic", 256) == hashlib.sha256(",".join(wl["checksums"]).encode()).hexdigest()
    assert bench.golden_batch_digest("fisheye_to_rect_bicubic", 257) is None


def test_host_checksum_twins_agree(lrp, oracle):
    a = oracle.synth_frame(123, 45, 5, 99, depth_channel=4)
    assert oracle.checksum(a) == lrp.checksum_host(a)

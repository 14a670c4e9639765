"""CPU: the C-ABI library loads, exports every symbol include/lrp.h declares,
keeps the reference's struct layouts, validates in the reference's dispatch
order, and fails loudly (never falls back) when there is no GPU."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lrp.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lrp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(lrp):
    lib = lrp._native.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/lrp.h but not exported by liblrp_hip.so"
    assert set(names) == set(lrp._native.SYMBOLS), "python binding and header disagree"


def test_no_oracle_or_cpu_fallback_linked(lrp):
    """The product library must not pull in the oracle."""
    out = subprocess.run(["ldd", lrp._native.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    syms = subprocess.run(["nm", "-D", lrp._native.LIB_PATH], capture_output=True, text=True).stdout
    assert "lrpo_" not in syms


def test_struct_layouts_match_reference(lrp):
    # reference: sizeof(LensInfo)==28 (union @4, sensor @20/24); sizeof(Image)==56
    # (width @28, height @32, channels @36, data @40, data_layout @48) — SURVEY.md §8a.
    from importlib import import_module

    nat = lrp._native
    assert ctypes.sizeof(nat.LrpLens) == 28
    assert nat.LrpLens.raw.offset == 4 and nat.LrpLens.sensor_width.offset == 20 and nat.LrpLens.sensor_height.offset == 24
    assert ctypes.sizeof(nat.LrpImage) == 56
    assert (nat.LrpImage.width.offset, nat.LrpImage.height.offset, nat.LrpImage.channels.offset,
            nat.LrpImage.data.offset, nat.LrpImage.data_layout.offset) == (28, 32, 36, 40, 48)


def test_enums_numbered_like_the_reference(lrp):
    assert [int(v) for v in lrp.LensType] == [0, 1, 2, 3, 4]
    assert [int(v) for v in lrp.Interpolation] == [0, 1, 2]
    assert [int(v) for v in lrp.DataLayout] == [0, 1, 2, 3]


def test_error_strings_are_the_reference_messages(lrp):
    lib = lrp._native.load()
    assert lib.lrp_strerror(1) == b"Output lens type not supported."
    assert lib.lrp_strerror(2) == b"Input lens type not supported."
    assert lib.lrp_strerror(3) == b"Interpolation method not supported."
    assert lib.lrp_strerror(0) == b"ok"


def test_rotation_matrix_and_lens_helpers(lrp, oracle):
    for ang in [(0.0, 0.0, 0.0), (0.5235988, -0.2617994, 0.0872665), (3.1415927, 0.0, 0.0), (0.0, 1.5707964, 0.0)]:
        a = lrp.rotation_matrix(*ang)
        b = oracle.rotation_matrix(*ang)
        assert (a.view(np.uint32) == b.view(np.uint32)).all()
    ident = lrp.rotation_matrix(0.0, 0.0, 0.0)
    assert (ident == np.eye(3, dtype=np.float32).reshape(9)).all()
    r = lrp.LensInfo.rectilinear(18.0, 36.0, 4096, 2048)
    assert r.type == 0 and r.params[0] == 18.0 and r.sensor_width == 36.0 and r.sensor_height == 18.0
    e = lrp.LensInfo.equidistant(3.14159265)
    assert e.type == 1 and e.sensor_width == 36.0 and e.sensor_height == 36.0
    q = lrp.LensInfo.equirectangular()
    assert q.type == 4 and q.sensor_width == 0.0
    # params = latitude_min, latitude_max, longitude_min, longitude_max (union order)
    assert q.params[2] == float(np.float32(-np.pi)) and q.params[3] == float(np.float32(np.pi))
    assert q.params[0] == float(np.float32(-np.pi * np.float32(0.5)))


def test_validation_order_and_no_gpu_behaviour(lrp):
    """Dispatch errors are reported before any device is touched, in the
    reference's order; with no GPU the compute entry points fail with
    NO_DEVICE instead of computing anything on the CPU."""
    import torch

    a = np.zeros((4, 4, 4), dtype=np.float32)
    good = lrp.LensInfo.rectilinear(18.0, 36.0, 4, 4)
    bad = lrp.LensInfo(lrp.LensType.FISHEYE_EQUISOLID, (10.0, 3.0), 36.0, 36.0)
    with pytest.raises(lrp.LrpError) as e:
        lrp.reproject(lrp.Image(bad, 4, 4, 4, a), lrp.Image(bad, 4, 4, 4, a.copy()), 1, 9)
    assert e.value.status == lrp.Status.OUTPUT_LENS
    with pytest.raises(lrp.LrpError) as e:
        lrp.reproject(lrp.Image(bad, 4, 4, 4, a), lrp.Image(good, 4, 4, 4, a.copy()), 1, 9)
    assert e.value.status == lrp.Status.INPUT_LENS
    with pytest.raises(lrp.LrpError) as e:
        lrp.reproject(lrp.Image(good, 4, 4, 4, a), lrp.Image(good, 4, 4, 4, a.copy()), 1, 9)
    assert e.value.status == lrp.Status.INTERPOLATION
    with pytest.raises(lrp.LrpError) as e:
        lrp.reproject(lrp.Image(good, 4, 4, 4, a), lrp.Image(good, 4, 4, 3, np.zeros((4, 4, 3), np.float32)), 1, 0)
    assert e.value.status == lrp.Status.CHANNELS
    if not torch.cuda.is_available():
        assert lrp.device_count() == 0
        out = np.full((4, 4, 4), 7.0, dtype=np.float32)
        with pytest.raises(lrp.LrpError) as e:
            lrp.reproject(lrp.Image(good, 4, 4, 4, a), lrp.Image(good, 4, 4, 4, out), 1, 0)
        assert e.value.status == lrp.Status.NO_DEVICE
        assert (out == 7.0).all(), "output was written without a GPU"
        with pytest.raises(lrp.LrpError):
            lrp.post_process(lrp.Image(good, 4, 4, 4, out), 2.0, 4.0)
        with pytest.raises(lrp.LrpError):
            lrp.BatchContext(0, 2)


def test_missing_library_fails_loudly(lrp, tmp_path, monkeypatch):
    nat = lrp._native
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setattr(nat, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        nat.load()

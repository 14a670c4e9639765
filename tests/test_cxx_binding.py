"""The C++ boundary: integration/reproject_hip.cpp (the binding a maintainer of the
reference adds) compiled against the declarations of reference src/reproject.hpp
(tests/native/reproject.hpp) and driven like the reference's worker
(src/main.cpp:576-603).  CPU part: dispatch errors print the reference's messages
and exit(1); without a GPU a valid call throws (no CPU fallback).  GPU part: the
result equals the oracle's."""
import os
import subprocess

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "native", "_build", "binding_driver")


def build_driver():
    os.makedirs(os.path.dirname(DRIVER), exist_ok=True)
    lib_dir = os.path.join(ROOT, "image-lens-reproject_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "tests", "native"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "integration", "reproject_hip.cpp"), os.path.join(ROOT, "tests", "native", "binding_driver.cpp"),
           "-L" + lib_dir, "-llrp_hip", "-Wl,-rpath," + lib_dir, "-o", DRIVER]
    subprocess.run(cmd, check=True, cwd=ROOT)


@pytest.fixture(scope="module")
def driver(lrp):
    """Always rebuilt from the sources under review (file times mean nothing after a checkout)."""
    lrp._native.load()  # the library must exist
    build_driver()
    return DRIVER


@pytest.mark.parametrize("mode,message", [("out_lens", "Output lens type not supported."),
                                          ("in_lens", "Input lens type not supported."),
                                          ("interp", "Interpolation method not supported.")])
def test_unsupported_dispatch_prints_reference_message_and_exits_1(driver, mode, message):
    r = subprocess.run([driver, mode], capture_output=True, text=True)
    assert r.returncode == 1
    assert r.stdout.strip() == message  # src/reproject.cpp:365,396,416


def test_header_only_wrapper_compiles(tmp_path):
    """include/lens_reproject.hpp alone (inline definitions) is a usable C++ API."""
    src = tmp_path / "t.cpp"
    src.write_text('#include "lens_reproject.hpp"\nint main(){ reproject::Image a{}, b{}; (void)a; (void)b; '
                   "static_assert(sizeof(reproject::Image)==56, \"\"); reproject::test_conversion_math(); return 0; }\n")
    lib_dir = os.path.join(ROOT, "image-lens-reproject_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), str(src), "-L" + lib_dir, "-llrp_hip",
                    "-Wl,-rpath," + lib_dir, "-o", str(tmp_path / "t")], check=True)
    subprocess.run([str(tmp_path / "t")], check=True)


def test_valid_call_without_gpu_throws_instead_of_falling_back(driver, lrp):
    if lrp.device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([driver, "nodevice"], capture_output=True, text=True)
    assert r.returncode == 3 and r.stdout.startswith("Error: no usable HIP device")


@pytest.mark.gpu
def test_binding_result_equals_oracle(driver, lrp, oracle, torch_cuda, tmp_path):
    """Every output float of reproject() + post_process() through the C++ binding, bit for bit."""
    dump = tmp_path / "out.f32"
    r = subprocess.run([driver, "run", str(dump)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    # the driver's inputs, restated
    w, h, c = 64, 32, 4
    i = np.arange(w * h * c, dtype=np.uint64)
    src = ((((i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) >> np.uint64(21)).astype(np.float32)
           / np.float32(2048.0)).reshape(h, w, c)
    lin = lrp.LensInfo.equirectangular()
    lout = lrp.LensInfo.rectilinear(18.0, 36.0, w, h)
    rot = lrp.rotation_matrix(0.5, -0.25, 0.1)
    want = oracle.reproject(lin, src, lout, w, h, 1, 2, rot)
    oracle.post_process(want, 2.0, 4.0)
    got = np.fromfile(dump, dtype=np.float32).reshape(h, w, c)
    cases.assert_same_bits(got, want, "C++ binding")

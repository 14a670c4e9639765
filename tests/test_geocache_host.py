"""CPU: the host logic of the geometry cache (csrc/lrp_geocache.cpp) on a fake HIP runtime (tests/native/geocache_driver.cpp
defines the dozen runtime calls the cache makes and decides when "device work" completes): fill / read / lists, a map-only entry
that gets its records later, eviction and take-over of buffers behind events — no device synchronisation, no free of a buffer a
launch may touch —, independent devices, the default cap, failed launches, the canonical key; then eight threads on two devices
under ThreadSanitizer and AddressSanitizer + UBSan (the one process-wide mutex and the
hipDeviceSynchronize under it)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "native", "_build")
SCENARIOS = ["fill_read_lists", "map_then_boxes", "layout", "sub_sample_entries", "eviction", "devices", "default_cap", "failed_launch_and_key", "release_one_device",
             "marks_are_pruned", "failed_writer_without_mark", "threads"]
HIP_INCLUDE = os.environ.get("HIP_INCLUDE", "/opt/rocm/include")


def build(name, extra):
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, name)
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-D__HIP_PLATFORM_AMD__", *extra, "-I" + HIP_INCLUDE,
                    "-I" + os.path.join(ROOT, "image-lens-reproject_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "native", "geocache_driver.cpp"), os.path.join(ROOT, "image-lens-reproject_amd", "csrc", "lrp_geocache.cpp"),
                    "-pthread", "-o", out], check=True, cwd=ROOT)
    return out


@pytest.fixture(scope="module", params=["plain", "tsan", "asan"])
def driver(request):
    if not os.path.isdir(os.path.join(HIP_INCLUDE, "hip")):
        pytest.skip("no HIP headers (the cache is compiled against the real declarations)")
    extra = {"plain": [], "tsan": ["-fsanitize=thread"], "asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]}[request.param]
    return build("geocache_driver_" + request.param, extra)


@pytest.mark.parametrize("scenario", SCENARIOS)
def test_geometry_cache_host_logic(driver, scenario):
    r = subprocess.run([driver, scenario], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and f"ok {scenario}" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]

"""CPU: lrp_reproject_multi (one source, several outputs, several GPUs — BASELINE configs[4] over a node) on a fake HIP runtime
with 2, 3 and 8 DISTINCT devices.  The library's real host code (lrp_capi.cpp, lrp_plan.cpp,
lrp_geocache.cpp) is linked against tests/native/multi_driver.cpp, which defines the runtime calls and the kernel launchers and
records them: the binary-tree fan-out of the source (participant k copies from k - 2^floor(log2 k), on its own stream, behind the
event of the copy that filled its parent; one upload in all), repeated GPUs reading their copy in place, every runtime call with
the right device current, every output row rendered and downloaded exactly once by the participant the split names, every exit path
— a failing peer copy, launch, download, upload, allocation — draining every participant's stream, one global lock order (opposite
device lists at once).  The GPU tests run the same entry point with GPU 0 named eight times (tests/test_gpu_multi_gpu.py); what a
node of eight physical GPUs adds is exactly what is checked here.  Plain, under ThreadSanitizer and under ASan + UBSan."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "native", "_build")
CSRC = os.path.join(ROOT, "image-lens-reproject_amd", "csrc")
SCENARIOS = ["tree2", "tree3", "tree8", "repeats", "failures", "lock_order", "arguments"]
HIP_INCLUDE = os.environ.get("HIP_INCLUDE", "/opt/rocm/include")


@pytest.fixture(scope="module", params=["plain", "tsan", "asan"])
def driver(request):
    if not os.path.isdir(os.path.join(HIP_INCLUDE, "hip")):
        pytest.skip("no HIP headers (the host code is compiled against the real declarations)")
    extra = {"plain": [], "tsan": ["-fsanitize=thread"], "asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]}[request.param]
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, "multi_driver_" + request.param)
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-D__HIP_PLATFORM_AMD__", *extra, "-I" + HIP_INCLUDE, "-I" + CSRC, "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "native", "multi_driver.cpp"), *[os.path.join(CSRC, f) for f in ("lrp_capi.cpp", "lrp_plan.cpp", "lrp_geocache.cpp", "lrp_host_util.cpp")],
                    "-pthread", "-o", out], check=True, cwd=ROOT)
    return out


@pytest.mark.parametrize("scenario", SCENARIOS)
def test_one_source_over_several_gpus_host_logic(driver, scenario):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")  # (the library keeps its participants, streams and the geometry cache for the process's life)
    r = subprocess.run([driver, scenario], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and f"ok {scenario}" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]

"""The N > 1 path of bench.py itself ("multi-GPU evidence is hollow").

bench.py shards ONE sorted list of images in static contiguous blocks over the ranks
(image-lens-reproject_amd/sharding.py — the reference's one-file-per-pool-thread split,
src/main.cpp:538-544,624-655) with no collective on the data path and reports per-image
checksums.  -m gpu: the same 24-image batch rendered by 1 rank and by 2 ranks (two fresh
child processes started by torch.distributed.run, gloo for the barrier / timing reduce so
that both may share one GPU) must give identical per-image checksums, and they must be the
oracle's.  CPU: launcher argument checks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
SIZE, BATCH = 512, 24


def check_driver_line(stdout, n):
    """What the driver does with bench.py's output: the LAST stdout line is one compact JSON object with the contract keys
    (round 5's had grown to 24.7 KB and was recorded as `parsed: null`)."""
    sys.path.insert(0, ROOT)
    import bench

    last = stdout.rstrip("\n").splitlines()[-1]
    assert last.startswith("{") and len(last) < bench.COMPACT_LIMIT, len(last)
    rec = json.loads(last)
    assert set(bench.TOP_KEYS) - {"cpu_baseline", "detail_file"} <= set(rec) <= set(bench.TOP_KEYS), sorted(rec)
    assert rec["n_gpus"] == n and rec["unit"] == "Mpix/s" and rec["value"] > 0 and rec["ms_per_step"] > 0
    assert rec["roofline"]["bound"] == "hbm" and rec["roofline"]["peak"] == 8000.0 and 0 < rec["roofline"]["frac"]
    assert ("single_launch_us" in rec["roofline"]) == (n == 1)  # N > 1: the timed region only, the other ranks are released at once
    return rec


def run_bench(n, tmp_path, extra=(), backend="gloo"):
    """-> (the detail record bench.py wrote, the per-image checksums); the compact line is checked on the way."""
    out = tmp_path / f"sums_{n}.json"
    detail = tmp_path / f"detail_{n}.json"
    args = ["--gpus", str(n), "--steps", "2", "--warmup", "1", "--batch", str(BATCH), "--size", str(SIZE),
            "--no-cpu-baseline", "--secondary", "", "--settle-seconds", "0", "--dist-backend", backend, "--checksums-file", str(out),
            "--detail-file", str(detail), *extra]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # N > 1: bench.py starts torch.distributed.run itself (as a child, before touching the GPU)
    r = subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = check_driver_line(r.stdout, n)
    rec = json.load(open(detail))
    assert line["value"] == pytest.approx(rec["value"], rel=1e-5) and line["detail_file"] == str(detail)
    return rec, json.load(open(out))


@pytest.mark.gpu
def test_one_rank_and_two_ranks_render_identical_images(lrp, oracle, torch_cuda, tmp_path):
    one, sums1 = run_bench(1, tmp_path)
    two, sums2 = run_bench(2, tmp_path)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["scaling"] == two["scaling"] == "strong"
    assert one["config"]["images_per_gpu_per_step"] == BATCH and two["config"]["images_per_gpu_per_step"] == BATCH // 2
    assert len(sums1["checksums"]) == len(sums2["checksums"]) == BATCH
    assert sums1["checksums"] == sums2["checksums"]
    assert one["outputs_digest"] == two["outputs_digest"] is not None
    # ... and they are the oracle's images (first, one from the second rank's block, last)
    import bench

    wl = bench.WORKLOADS["fisheye_to_rect_bicubic"]
    lin, lout = bench.make_lens(lrp, wl["in_lens"], SIZE, SIZE), bench.make_lens(lrp, wl["out_lens"], SIZE, SIZE)
    for i in (0, BATCH // 2 + 1, BATCH - 1):
        src = oracle.synth_frame(SIZE, SIZE, 4, 0x5EED0000 + i)
        want = oracle.reproject(lin, src, lout, SIZE, SIZE, 1, wl["interp"], None)
        assert f"{lrp.checksum_host(want):016x}" == sums2["checksums"][i], f"image {i}"
    assert one["roofline"]["single_launch_us"] > 0
    assert one["roofline"]["single_launch_us_uncached"] > 0 and one["roofline"]["single_launch_frac_uncached"] > 0
    assert "single_launch_us" not in two["roofline"] and two["secondary"] == {}
    for rec in (one, two):
        assert 0 < rec["roofline"]["frac_read_only"] < rec["roofline"]["frac"]
        # per-rank times (what explains a bad scaling line): one entry per rank, the whole-job time is the slowest rank's
        n = rec["n_gpus"]
        assert len(rec["per_rank_elapsed_s"]) == n and len(rec["per_rank_device_busy_s"]) == n
        assert abs(max(rec["per_rank_elapsed_s"]) - rec["timed_region_s"]) < 1e-9
        assert rec["rank_skew_s"] == pytest.approx(max(rec["per_rank_elapsed_s"]) - min(rec["per_rank_elapsed_s"]))
        assert all(0 < b <= e * 1.05 for e, b in zip(rec["per_rank_elapsed_s"], rec["per_rank_device_busy_s"]))
    assert one["staged"] is None or one["staged"]["f32_pinned"] > 0  # (N = 1 only; None when tools/staged_bench is not built)
    assert "staged" not in two


@pytest.mark.gpu
def test_weak_scaling_mode_and_uneven_shards(lrp, torch_cuda, tmp_path):
    """--scaling weak: every rank renders its own --batch images; strong with a batch that does not
    divide: ceil blocks (13 + 12)."""
    rec, sums = run_bench(2, tmp_path, extra=("--scaling", "weak"))
    assert rec["scaling"] == "weak" and rec["config"]["images_per_step"] == 2 * BATCH
    assert len(sums["checksums"]) == 2 * BATCH
    odd, sums_odd = run_bench(2, tmp_path, extra=("--batch", "25"))
    assert len(sums_odd["checksums"]) == 25 and sums_odd["checksums"][:BATCH] == sums["checksums"][:BATCH]


@pytest.mark.gpu
def test_four_ranks_share_one_gpu_with_uneven_shards(lrp, torch_cuda, tmp_path):
    """More ranks than the two of the tests above, uneven blocks (42 images over 4 ranks: 11 + 11 + 11 + 9): the same per-image
    checksums as one rank, one entry per rank in every per-rank list.  Four, not eight: a GPU box admits at most six
    processes on its card (this test's parent is one of them); the world-size-8 sharding runs over gloo on the CPU
    (tests/test_sharding_gloo.py)."""
    one, sums1 = run_bench(1, tmp_path, extra=("--batch", "42"))
    four, sums4 = run_bench(4, tmp_path, extra=("--batch", "42"))
    assert four["n_gpus"] == 4 and four["config"]["images_per_gpu_per_step"] == 11
    assert len(sums4["checksums"]) == 42 and sums4["checksums"] == sums1["checksums"]
    assert one["outputs_digest"] == four["outputs_digest"]
    assert len(four["per_rank_elapsed_s"]) == 4 and len(four["per_rank_device_busy_s"]) == 4
    assert abs(max(four["per_rank_elapsed_s"]) - four["timed_region_s"]) < 1e-9


def test_gpus_flag_must_match_the_launcher():
    """`--gpus 2` inside a 1-rank launch is an error with the torchrun command line, not a silent 1-GPU run."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode != 0
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node 2" in r.stderr


def test_checksum_host_matches_its_definition(lrp):
    a = np.array([0.0, -0.0, 1.5, np.nan], dtype=np.float32)
    bits = a.view(np.uint32)

    def mix32(seed, index):
        h = (index * 0x9E3779B9 + seed) & 0xFFFFFFFF
        h ^= h >> 16
        h = (h * 0x7FEB352D) & 0xFFFFFFFF
        h ^= h >> 15
        h = (h * 0x846CA68B) & 0xFFFFFFFF
        h ^= h >> 16
        return h

    want = 0
    for i, b in enumerate(int(v) for v in bits):
        want += (mix32(b ^ 0xA5A5A5A5, (2 * i + 0x7F4A7C15) & 0xFFFFFFFF) << 32) | mix32(b, i)
    assert lrp.checksum_host(a) == want & 0xFFFFFFFFFFFFFFFF


@pytest.mark.gpu
def test_two_ranks_over_rccl_on_two_gpus(torch_cuda, tmp_path):
    """The launch the driver uses for N > 1 — `--dist-backend nccl` (RCCL), one rank per GPU — on the first two GPUs of
    the box.  Needs two GPUs (a one-GPU box cannot host two RCCL ranks): skipped there, the gloo tests above cover the
    sharding and the reporting, and RCCL only carries the barrier and the one-float MAX of the elapsed time."""
    if torch_cuda.cuda.device_count() < 2:
        pytest.skip("one GPU visible: two RCCL ranks need two")
    one, sums1 = run_bench(1, tmp_path)
    two, sums2 = run_bench(2, tmp_path, backend="nccl")
    assert two["n_gpus"] == 2 and sums2["checksums"] == sums1["checksums"]


@pytest.mark.gpu
def test_one_rank_through_the_rccl_path(lrp, torch_cuda, tmp_path):
    """The N > 1 code path on real RCCL with the one world size a one-GPU box admits: `--force-dist` starts
    ONE rank under torch.distributed.run (as a child, before anything touches the GPU), `init_process_group("nccl",
    device_id=...)`, every barrier, both `all_gather_object`s and the device-side float64 `all_reduce(MAX)` of
    sharding.max_over_ranks execute on a world-size-1 RCCL communicator; the images are the ones a plain run renders."""
    plain, sums_plain = run_bench(1, tmp_path)
    forced, sums_forced = run_bench(1, tmp_path, extra=("--force-dist",), backend="nccl")
    assert plain["dist"] is None and forced["dist"] == "nccl process group, world size 1 (--force-dist)"
    assert sums_forced["checksums"] == sums_plain["checksums"] and forced["outputs_digest"] == plain["outputs_digest"]
    assert len(forced["per_rank_elapsed_s"]) == 1 and forced["per_rank_elapsed_s"][0] == pytest.approx(forced["timed_region_s"], abs=1e-6)
    assert forced["roofline"]["single_launch_us"] > 0  # (N = 1: rank 0 still measures the single-launch legs)


@pytest.mark.gpu
def test_device_visibility_is_honoured(torch_cuda, tmp_path):
    """HIP_VISIBLE_DEVICES subsets: the library sees exactly the listed GPUs — the last one alone renders the
    same image as device 0 of the full list, and with none visible every compute entry point fails loudly
    (LRP_ERR_NO_DEVICE): there is no CPU fallback to fall into."""
    code = r'''
import importlib, sys, numpy as np
sys.path.insert(0, %r)
lrp = importlib.import_module("image-lens-reproject_amd")
n = lrp.device_count()
print("count", n)
src = (np.arange(64 * 32 * 4, dtype=np.float32).reshape(32, 64, 4) %% 17) / 16
out = np.full((24, 40, 4), -1.0, dtype=np.float32)
try:
    lrp.reproject(lrp.Image(lrp.LensInfo.equirectangular(), 64, 32, 4, src), lrp.Image(lrp.LensInfo.rectilinear(18.0, 36.0, 40, 24), 40, 24, 4, out), 1, 2, None)
    print("sum", float(out.sum()))
except lrp.LrpError as e:
    print("error", int(e.status))
''' % ROOT
    def run(visible):
        env = dict(os.environ)
        if visible is None:
            env.pop("HIP_VISIBLE_DEVICES", None)
        else:
            env["HIP_VISIBLE_DEVICES"] = visible
        env.pop("ROCR_VISIBLE_DEVICES", None)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        return dict(ln.split(" ", 1) for ln in r.stdout.splitlines() if " " in ln)

    full = run(None)
    n = int(full["count"])
    assert n >= 1 and "sum" in full
    last = run(str(n - 1))
    assert int(last["count"]) == 1 and last["sum"] == full["sum"]
    none = run("")
    if int(none["count"]) == 0:  # (an empty list hides every GPU on this runtime; if it does not, there is nothing to check)
        lrp = __import__("importlib").import_module("image-lens-reproject_amd")
        assert int(none["error"]) == int(lrp.Status.NO_DEVICE)

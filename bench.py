#!/usr/bin/env python3
"""bench.py — Mpix/s of the lens-reprojection hot path on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`;
for N > 1 it is launched by torch.distributed.run, one rank per GPU.  Prints ONE
JSON line on rank 0.

A step = one pass of the hot path over one batch of `--batch` synthetic 4096x4096
RGBA float frames that are already resident in HBM (generated on the device by
the counter-based generator; seeds 0x5EED0000 + i).  The batch is sharded
one-shard-per-GPU with no collective on the data path (images are independent,
reference src/main.cpp:540-622), i.e. weak scaling: every rank renders `--batch`
frames per step.  The frames of a step share one geometry (a directory of frames
from one camera, the reference's --input-dir path), so a step is ONE kernel launch
(lrp_reproject_batch_device, 16 frames per launch: the chip neither drains nor
refills between frames); `--per-frame-launches` issues one launch per frame instead,
round-robin over `--streams` HIP streams.  `roofline.achieved` = algorithmic bytes
of the frames one launch renders / that launch's duration (HIP events on its stream).
"""
import argparse
import importlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # BASELINE.json configs[1] (the config the metric is quoted on).  The reference
    # rejects FISHEYE_EQUISOLID (src/reproject.cpp:395-397), so the fisheye is the
    # equidistant one with fov = pi, as SURVEY.md §8d prescribes for parity.
    "fisheye_to_rect_bicubic": dict(in_lens="eqd", out_lens="rect", interp=2, rot=None, channels=4, size=4096),
    # north_star's roofline target case.
    "equirect_to_rect_bicubic": dict(in_lens="eqr", out_lens="rect", interp=2, rot=(0.0, 0.0, 0.0), channels=4,
                                     size=4096),
    # BASELINE.json configs[2] shape.
    "equirect_to_fisheye_bilinear": dict(in_lens="eqr", out_lens="eqd", interp=1, rot=(30.0, -15.0, 5.0), channels=4,
                                         size=4096),
    # BASELINE.json configs[0] shape at 4K (plumbing case, nearest).
    "equirect_to_rect_nearest": dict(in_lens="eqr", out_lens="rect", interp=0, rot=(0.0, 0.0, 0.0), channels=4,
                                     size=4096),
}

KERNEL_NAMES = {0: "reproject_tile_kernel (nearest)", 1: "reproject_tile_kernel (bilinear)",
                2: "reproject_bicubic_win_kernel (LDS window)"}


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC
    summary (profiles/traffic_r01.json; made by tools/collect_traffic.sh on an MI355X)."""
    path = os.path.join(ROOT, "profiles", "traffic_r01.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload)
    except (OSError, ValueError):
        return None



def make_lens(pkg, kind, w, h):
    if kind == "rect":
        return pkg.LensInfo.rectilinear(18.0, 36.0, w, h)
    if kind == "eqd":
        return pkg.LensInfo.equidistant(3.14159265)
    return pkg.LensInfo.equirectangular()


def make_rot(pkg, deg):
    if deg is None:
        return None
    r = [d * math.pi / 180.0 for d in deg]
    return pkg.rotation_matrix(*r)


def usable_cpus():
    """CPUs this process may actually use: affinity mask, capped by the cgroup v2 CPU quota
    (a GPU box hands a 1-GPU job a 16-CPU share of a 256-thread host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(pkg, wl, seconds_target):
    """The oracle (kind "port": our C restatement of the reference loop, built with
    the reference's flags) timed on this host's cores.  The reference schedules
    one image per pool thread (src/main.cpp:538-544) and its loop is
    single-threaded per image; here every host thread renders its own row band
    of each frame (rows are independent, src/reproject.cpp:284), which is the
    same work per thread without needing one 256 MiB output per thread."""
    import ctypes
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np
    import oracle_binding as oracle

    cores = usable_cpus()
    size = wl["size"]
    c = wl["channels"]
    src = oracle.synth_frame(size, size, c, 0x5EED0000)
    lin, lout = make_lens(pkg, wl["in_lens"], size, size), make_lens(pkg, wl["out_lens"], size, size)
    rot = make_rot(pkg, wl["rot"])
    out = np.empty((size, size, c), dtype=np.float32)
    L = oracle.lib()
    cin = oracle._image(lin, size, size, c, src)
    cout = oracle._image(lout, size, size, c, out)
    keep, rp = oracle._rot(rot)
    bands = [(size * i // cores, size * (i + 1) // cores) for i in range(cores)]

    def run(frames):
        def work(b):
            for _ in range(frames):
                L.lrpo_reproject_rows(ctypes.byref(cin), ctypes.byref(cout), 1, wl["interp"], rp, b[0], b[1])

        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(work, bands))
        return time.perf_counter() - t0

    dt = run(1)  # calibration (also warms the pages)
    frames = int(max(1, min(4096, seconds_target / max(dt, 1e-3))))
    dt = run(frames)
    return {
        "value": frames * size * size / dt / 1e6,
        "unit": "Mpix/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{frames} frames of {size}x{size}x{c}, each split into {cores} row bands (one per host thread = "
                  f"one per usable CPU: affinity {len(os.sched_getaffinity(0))}, cgroup quota honoured), {dt:.1f} s wall",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="frames per GPU per step")
    ap.add_argument("--distinct", type=int, default=16, help="distinct resident source / destination frames per GPU")
    ap.add_argument("--per-frame-launches", action="store_true",
                    help="one kernel launch per frame (lrp_reproject_device) instead of one per step "
                         "(lrp_reproject_batch_device: all frames of a step share one geometry)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the frames of a step round-robin over (one frame fills the chip; >1 only "
                         "overlaps kernel tails and makes per-kernel durations in a profile overlap)")
    ap.add_argument("--workload", default="fisheye_to_rect_bicubic", choices=sorted(WORKLOADS))
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the barrier / max-time reduction for --gpus > 1 (nccl = RCCL; "
                         "gloo lets two ranks share one GPU in a smoke test of the multi-rank path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch

    pkg = importlib.import_module("image-lens-reproject_amd")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend="gloo")
    dev = torch.device("cuda", dev_index)

    wl = WORKLOADS[args.workload]
    size, c = wl["size"], wl["channels"]
    lin, lout = make_lens(pkg, wl["in_lens"], size, size), make_lens(pkg, wl["out_lens"], size, size)
    rot = make_rot(pkg, wl["rot"])

    # resident frames: `distinct` sources and as many destinations, cycled, so the
    # working set (distinct x 512 MiB) is far beyond the 256 MiB Infinity Cache.
    n_res = max(1, min(args.distinct, args.batch))
    srcs, dsts = [], []
    for i in range(n_res):
        s = torch.empty((size, size, c), dtype=torch.float32, device=dev)
        pkg.synth_fill(s, size, size, c, 0x5EED0000 + rank * args.batch + i)
        srcs.append(s)
        dsts.append(torch.empty((size, size, c), dtype=torch.float32, device=dev))
    im_in = [pkg.Image(lin, size, size, c, s) for s in srcs]
    im_out = [pkg.Image(lout, size, size, c, d) for d in dsts]
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]
    torch.cuda.synchronize()

    batch_in = [im_in[i % n_res] for i in range(args.batch)]
    batch_out = [im_out[i % n_res] for i in range(args.batch)]
    launches_per_step = args.batch if args.per_frame_launches else (args.batch + 15) // 16
    frames_per_launch = args.batch / launches_per_step

    def step():
        if args.per_frame_launches:
            for i in range(args.batch):
                st = streams[i % len(streams)]
                pkg.reproject(im_in[i % n_res], im_out[i % n_res], 1, wl["interp"], rot, stream=st)
        else:  # the frames of a step share one geometry: one launch per 16 frames
            pkg.reproject_batch(batch_in, batch_out, 1, wl["interp"], rot, stream=streams[0])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel timed live with HIP events on the stream it is launched on
    # (torch events recorded on that same stream), one frame per launch.
    kst = streams[0]
    reps = 40
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    if not args.per_frame_launches:
        reps = 10
        ev = ev[:reps]
    for i in range(reps):
        ev[i][0].record(kst)
        if args.per_frame_launches:
            pkg.reproject(im_in[i % n_res], im_out[i % n_res], 1, wl["interp"], rot, stream=kst)
        else:
            pkg.reproject_batch(batch_in[:16], batch_out[:16], 1, wl["interp"], rot, stream=kst)
        ev[i][1].record(kst)
    torch.cuda.synchronize()
    k_ms = sorted(a.elapsed_time(b) for a, b in ev)
    k_avg_ms = sum(k_ms) / len(k_ms)

    if rank == 0:
        pix_per_step = world * args.batch * size * size
        value = pix_per_step * args.steps / elapsed / 1e6
        frames_in_timed_launch = 1 if args.per_frame_launches else min(16, args.batch)
        # (inW*inH + outW*outH)*C*4 per frame x the frames one launch renders
        algo_bytes = 2 * size * size * c * 4 * frames_in_timed_launch
        traffic = measured_traffic(args.workload)
        achieved = algo_bytes / (k_avg_ms * 1e-3) / 1e9
        out = {
            "metric": "Mpix/s reprojected (4K RGBA float, bicubic)",
            "value": value,
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {size}x{size}x{c} f32 -> {size}x{size}x{c}, "
                            f"{wl['in_lens']}->{wl['out_lens']}, interp={wl['interp']}, num_samples=1",
                "frames_per_gpu_per_step": args.batch,
                "resident_distinct_frames": n_res,
                "streams": len(streams),
                "launches_per_step": launches_per_step,
                "frames_per_launch": frames_per_launch,
                "parallelism": f"image-sharded x{world}, no collective",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": ((traffic or {}).get("hbm_bytes_per_launch") or 0) * frames_in_timed_launch or None,
                "traffic_detail": traffic,
                "kernel": KERNEL_NAMES[wl["interp"]],
                "kernel_ms_avg": k_avg_ms,
                "kernel_ms_min": k_ms[0],
                "algorithmic_bytes_per_launch": algo_bytes,
                "frames_per_launch": frames_in_timed_launch,
                "traffic_note": "PMC bytes measured per single-frame launch (traffic_detail) x frames_per_launch",
                "note": "the un-fused IEEE arithmetic of the reference makes this kernel VALU-bound, not HBM-bound "
                        "(DESIGN.md section 5); frac is algorithmic bytes / kernel time / 8 TB/s as the contract asks",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, wl, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

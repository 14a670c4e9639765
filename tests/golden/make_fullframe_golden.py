#!/usr/bin/env python3
"""Regenerates tests/golden/fullframe_golden.json: whole-frame digests of the ORACLE's output
(oracle/liblrp_oracle.so, host libm = glibc 2.35 of the build container) at BASELINE.json's
full sizes, plus the per-image checksums of bench.py's 256-image batch.

Like oracle_golden.json these are outputs of our C restatement, not of the reference binary
(which cannot be built in this image, see oracle/lrp_oracle.h): they make the GPU parity
evidence independent of the GPU box (tests/test_gpu_golden.py compares the HIP output with
these committed values and never calls the oracle) and cover whole frames, but they do not
pin the oracle to the reference.

Run from the repo root (takes ~10 minutes on 8 cores):
    python tests/golden/make_fullframe_golden.py [--only NAME_SUBSTRING] [--skip-bench]
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
import fullframe_cases as ffc  # noqa: E402
import oracle_binding as oracle  # noqa: E402

lrp = importlib.import_module("image-lens-reproject_amd")
OUT = os.path.join(HERE, "fullframe_golden.json")
THREADS = max(1, len(os.sched_getaffinity(0)))


def render(case, seed=None):
    n, m, c = case["size"], case["out_size"], case["c"]
    src = oracle.synth_frame(n, n, c, case["seed"] if seed is None else seed, depth_channel=case.get("depth", -1))
    lin, lout = cases.lenses(lrp, n, n)[case["inp"]], cases.lenses(lrp, m, m)[case["out"]]
    rot = cases.rotation(lrp, case["deg"])
    out = oracle.reproject(lin, src, lout, m, m, case.get("ns", 1), case["interp"], rot, threads=THREADS)
    if case.get("post"):
        oracle.post_process(out, *case["post"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--skip-bench", action="store_true")
    args = ap.parse_args()
    try:
        with open(OUT) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        doc = {}
    doc["generator"] = "oracle/liblrp_oracle.so (gcc -O3 -ffp-contract=off, glibc 2.35), tests/golden/make_fullframe_golden.py"
    frames = doc.setdefault("frames", {})
    for name, case in ffc.frame_cases().items():
        if args.only and args.only not in name:
            continue
        t0 = time.time()
        out = render(case)
        sha, bands, n_nan = ffc.frame_digests(out)
        frames[name] = {"sha256": sha, "bands": bands, "nan": n_nan,
                        "checksum": f"{oracle.checksum(out):016x}" if n_nan == 0 else None,
                        "case": {k: v for k, v in case.items() if k != "name"}}
        print(f"{name}: {sha[:16]} nan={n_nan} ({time.time() - t0:.1f} s)", flush=True)
    if not args.skip_bench and not args.only:
        bench = doc.setdefault("bench_batch", {})
        for wname, wl in ffc.BENCH_WORKLOADS.items():
            sums = []
            t0 = time.time()
            case = dict(wl, out_size=wl["size"], seed=0)
            for i in range(ffc.BENCH_BATCH):
                sums.append(f"{oracle.checksum(render(case, seed=0x5EED0000 + i)):016x}")
                if i % 16 == 15:
                    print(f"{wname}: {i + 1} / {ffc.BENCH_BATCH} images ({time.time() - t0:.0f} s)", flush=True)
            bench[wname] = {"checksums": sums, "case": wl}
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print(f"wrote {OUT}: {len(frames)} frames")


if __name__ == "__main__":
    main()

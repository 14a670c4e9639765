#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_golden.json + oracle_golden_arrays.npz.

What these fixtures are: outputs of oracle/liblrp_oracle.so (our C restatement,
host libm = glibc 2.35 of the build container) on seeded inputs.  They are NOT
outputs of the reference binary — the reference translation unit cannot be built
in this image (see oracle/lrp_oracle.h) — so they pin the oracle against drift
(compiler, libm, edits), not against the reference.  Inputs are regenerated from
the counter-based generator (lrpo_synth_fill), only digests / small arrays are
stored.

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_cases  # noqa: E402
import oracle_binding as oracle  # noqa: E402

lrp = importlib.import_module("image-lens-reproject_amd")


def main():
    digests = {}
    arrays = {}
    for name, case in golden_cases.all_cases(lrp):
        out = golden_cases.run_oracle(oracle, lrp, case)
        digests[name] = golden_cases.digest(out)
        if case.get("store"):
            arrays[name] = out
    post = {}
    for name, arr, exposure, reinhard in golden_cases.post_cases(oracle):
        a = arr.copy()
        oracle.post_process(a, exposure, reinhard)
        post[name] = golden_cases.digest(a)
    with open(os.path.join(HERE, "oracle_golden.json"), "w") as f:
        json.dump({"reproject": digests, "post_process": post,
                   "generator": "oracle/liblrp_oracle.so (gcc -O3 -ffp-contract=off, glibc 2.35)"}, f, indent=1,
                  sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "oracle_golden_arrays.npz"), **arrays)
    print(f"{len(digests)} reproject digests, {len(arrays)} arrays, {len(post)} post_process digests")


if __name__ == "__main__":
    main()

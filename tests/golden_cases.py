"""The golden-case matrix: every lens pair x interpolation x
channel count x num_samples x rotation at tiny odd sizes, plus a power-of-two
wrap case.  Shared by tests/golden/make_golden.py and tests/test_oracle.py."""
import hashlib
import math

import numpy as np

import cases

INTERPS = {"nn": 0, "bl": 1, "bc": 2}
ROTS = {"none": None, "ident": (0.0, 0.0, 0.0), "r30": (30.0, -15.0, 5.0), "pan180": (180.0, 0.0, 0.0),
        "pitch90": (0.0, 90.0, 0.0)}


def planted_input(oracle, w, h, c, seed):
    a = oracle.synth_frame(w, h, c, seed, depth_channel=(c - 1 if c == 5 else -1))
    flat = a.reshape(-1)
    if flat.size >= 64:
        specials = np.array([-0.0, 1e-41, 65504.0, 3.0e38, -7.25, np.inf], dtype=np.float32)
        for k, v in enumerate(specials):
            flat[(k * 37 + 11) % flat.size] = v
    return a


def all_cases(lrp):
    out = []
    sizes = {"odd": (61, 47, 53, 41), "pow2": (64, 32, 32, 32)}
    for sz_name, (iw, ih, ow, oh) in sizes.items():
        for out_name in ("rect", "eqd180", "eqr_full"):
            for in_name in ("rect", "eqd180", "eqr_full", "eqr_part"):
                for iname, interp in INTERPS.items():
                    for c in (3, 4, 5):
                        # a rotating subset keeps the matrix small but touches every value of every axis
                        key = (len(out)) % 15
                        ns = 1 + key % 3
                        rot = list(ROTS)[key % 5]
                        name = f"{sz_name}-{out_name}-{in_name}-{iname}-c{c}-n{ns}-{rot}"
                        out.append((name, dict(iw=iw, ih=ih, ow=ow, oh=oh, out=out_name, inp=in_name, interp=interp,
                                               c=c, ns=ns, rot=rot, seed=0x5EED0000 + len(out),
                                               store=(len(out) % 29 == 0))))
    return out


def run_oracle(oracle, lrp, case):
    src = planted_input(oracle, case["iw"], case["ih"], case["c"], case["seed"])
    lin = cases.lenses(lrp, case["iw"], case["ih"])[case["inp"]]
    lout = cases.lenses(lrp, case["ow"], case["oh"])[case["out"]]
    rot = cases.rotation(lrp, ROTS[case["rot"]])
    return oracle.reproject(lin, src, lout, case["ow"], case["oh"], case["ns"], case["interp"], rot)


def post_cases(oracle):
    out = []
    for c in (3, 4, 5):
        arr = oracle.synth_frame(31, 17, c, 0xABCD0000 + c) * np.float32(6.0)
        for k, (e, r) in enumerate(((2.0, 4.0), (0.5, 1.0), (1.0, 2.5))):
            out.append((f"post-c{c}-{k}", arr, e, r))
    return out


def digest(a):
    """SHA-256 of the float32 bits with every NaN canonicalised."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).copy()
    u[np.isnan(a)] = 0x7FC00000
    return hashlib.sha256(u.tobytes()).hexdigest()

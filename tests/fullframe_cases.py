"""Whole-frame fixtures at BASELINE.json's full sizes ("SHA-256 of the five
BASELINE-shaped full-size outputs").  Shared by tests/golden/make_fullframe_golden.py (runs the
oracle once, in the build container, and commits digests) and tests/test_gpu_golden.py (compares
the HIP output with the committed digests: no oracle call on the GPU box, so the evidence does
not depend on that box's libm).

A case renders ONE frame: source = the counter-based synthetic frame (seed, depth channel), lenses
by name (cases.lenses), rotation in degrees or None, optional post_process (exposure, reinhard).
Digest of a frame: SHA-256 over the float32 bit patterns with every NaN canonicalised, for the
whole frame and for 16 row bands (a failing test names the band), plus the 64-bit order-independent
checksum the device computes (lrp_checksum_device; raw bits, only stored for NaN-free frames)."""
import hashlib

import numpy as np

NEAREST, BILINEAR, BICUBIC = 0, 1, 2
BANDS = 16

CUBE_FACES_DEG = [(0, 0, 0), (90, 0, 0), (180, 0, 0), (270, 0, 0), (0, 90, 0), (0, -90, 0)]


def _case(name, size, out_size, c, inp, out, interp, deg, seed, depth=-1, post=None, ns=1):
    return dict(name=name, size=size, out_size=out_size, c=c, inp=inp, out=out, interp=interp, deg=deg, seed=seed,
                depth=depth, post=post, ns=ns)


def frame_cases():
    """name -> case.  The first block is BASELINE.json configs[0..4]; the second the mappings whose kernels are
    being optimised (every window-kernel tier, rotated and not), so that a kernel change is checked on whole frames."""
    cs = [
        # --- BASELINE.json configs ---------------------------------------------------------------
        _case("config0_512_eqr_rect_nn", 512, 512, 4, "eqr_full", "rect", NEAREST, (0.0, 0.0, 0.0), 0x5EED0000),
        _case("config1_4k_eqd_rect_bc", 4096, 4096, 4, "eqd180", "rect", BICUBIC, None, 0x5EED0000),
        _case("northstar_4k_eqr_rect_bc", 4096, 4096, 4, "eqr_full", "rect", BICUBIC, (0.0, 0.0, 0.0), 0x5EED0000),
        _case("config2_4k_eqr_eqd_bl_rot", 4096, 4096, 4, "eqr_full", "eqd180", BILINEAR, (30.0, -15.0, 5.0), 0x5EED0000),
        _case("scaling_4k_eqr_eqd_bc_rot", 4096, 4096, 4, "eqr_full", "eqd180", BICUBIC, (30.0, -15.0, 5.0), 0x5EED0000),
        _case("config3_4k_rgbaz_rect_eqr_bc_post", 4096, 4096, 5, "rect", "eqr_full", BICUBIC, (0.0, 0.0, 0.0), 0x5EED0007,
              depth=4, post=(2.0, 4.0)),
        _case("config3_4k_rgbz_rect_eqr_bc_post", 4096, 4096, 4, "rect", "eqr_full", BICUBIC, (0.0, 0.0, 0.0), 0x5EED0007,
              depth=3, post=(2.0, 4.0)),
    ]
    for i, deg in enumerate(CUBE_FACES_DEG):
        cs.append(_case(f"config4_8k_rgb_face{i}", 8192, 2048, 3, "eqr_full", "rect", BICUBIC,
                        tuple(float(d) for d in deg), 0x5EED0005))
    cs += [
        # --- every other tier of the bicubic kernels, whole 4K frames ----------------------------
        _case("4k_eqr_rect_bc_rot", 4096, 4096, 4, "eqr_full", "rect", BICUBIC, (30.0, -15.0, 5.0), 0x5EED0001),
        _case("4k_eqr_rect_bc_pan90", 4096, 4096, 4, "eqr_full", "rect", BICUBIC, (90.0, 0.0, 0.0), 0x5EED0001),
        _case("4k_eqr_rect_bc_pitch90", 4096, 4096, 4, "eqr_full", "rect", BICUBIC, (0.0, 90.0, 0.0), 0x5EED0001),
        _case("4k_rect_rect_bc_rot", 4096, 4096, 4, "rect", "rect", BICUBIC, (30.0, -15.0, 5.0), 0x5EED0002),
        _case("4k_eqd_eqd_bc_rot", 4096, 4096, 4, "eqd180", "eqd180", BICUBIC, (30.0, -15.0, 5.0), 0x5EED0002),
        _case("4k_eqr_eqr_bc_rot", 4096, 4096, 4, "eqr_full", "eqr_full", BICUBIC, (30.0, -15.0, 5.0), 0x5EED0002),
        _case("4k_rect_eqr_bc", 4096, 4096, 4, "rect", "eqr_full", BICUBIC, (0.0, 0.0, 0.0), 0x5EED0003),
        _case("4k_rgb_eqd_rect_bc", 4096, 4096, 3, "eqd180", "rect", BICUBIC, None, 0x5EED0004),
        _case("4k_rgb_eqr_rect_bc_rot", 4096, 4096, 3, "eqr_full", "rect", BICUBIC, (30.0, -15.0, 5.0), 0x5EED0004),
        _case("4k_rgbaz_eqd_rect_bc", 4096, 4096, 5, "eqd180", "rect", BICUBIC, None, 0x5EED0006, depth=4),
        _case("4k_rgbaz_eqr_rect_bc_rot", 4096, 4096, 5, "eqr_full", "rect", BICUBIC, (30.0, -15.0, 5.0), 0x5EED0006, depth=4),
        _case("4k_eqr_rect_bl", 4096, 4096, 4, "eqr_full", "rect", BILINEAR, (0.0, 0.0, 0.0), 0x5EED0008),
        _case("4k_eqr_rect_nn", 4096, 4096, 4, "eqr_full", "rect", NEAREST, (0.0, 0.0, 0.0), 0x5EED0008),
        _case("2k_eqd_rect_bc_ns2", 2048, 1024, 4, "eqd180", "rect", BICUBIC, (10.0, 20.0, 30.0), 0x5EED0009, ns=2),
        # --- the --scale / --samples pairs the reference's help text prescribes (src/main.cpp:192-196; int(4096 * 0.33334) = 1365),
        # bench.py's supersampling secondaries, and the other two samplers on the same entries of sub-samples
        _case("4k_eqd_rect_bc_half_ns2", 4096, 2048, 4, "eqd180", "rect", BICUBIC, None, 0x5EED000A, ns=2),
        _case("4k_eqd_rect_bc_third_ns3", 4096, 1365, 4, "eqd180", "rect", BICUBIC, None, 0x5EED000A, ns=3),
        _case("4k_eqd_rect_bc_quarter_ns4", 4096, 1024, 4, "eqd180", "rect", BICUBIC, None, 0x5EED000A, ns=4),
        _case("4k_rgbaz_eqr_rect_bc_rot_third_ns3_post", 4096, 1365, 5, "eqr_full", "rect", BICUBIC, (30.0, -15.0, 5.0), 0x5EED000B, depth=4,
              post=(2.0, 4.0), ns=3),
        _case("4k_eqr_eqd_bl_rot_third_ns3", 4096, 1365, 4, "eqr_full", "eqd180", BILINEAR, (30.0, -15.0, 5.0), 0x5EED000C, ns=3),
        _case("4k_rgb_eqr_rect_nn_rot_quarter_ns4", 4096, 1024, 3, "eqr_full", "rect", NEAREST, (30.0, -15.0, 5.0), 0x5EED000C, ns=4),
    ]
    return {c["name"]: c for c in cs}


# bench.py's batch: image i of the 256-image list has seed 0x5EED0000 + i; per image the 64-bit checksum
BENCH_BATCH = 256
BENCH_WORKLOADS = {
    "fisheye_to_rect_bicubic": dict(size=4096, c=4, inp="eqd180", out="rect", interp=BICUBIC, deg=None),
}


def frame_digests(a):
    """(whole-frame sha256, [band sha256] * BANDS, nan count) of an (H, W, C) float32 array."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).copy()
    nan = np.isnan(a)
    n_nan = int(nan.sum())
    if n_nan:
        u[nan] = 0x7FC00000
    h = a.shape[0]
    bands = []
    for b in range(BANDS):
        y0, y1 = h * b // BANDS, h * (b + 1) // BANDS
        bands.append(hashlib.sha256(u[y0:y1].tobytes()).hexdigest())
    return hashlib.sha256(u.tobytes()).hexdigest(), bands, n_nan

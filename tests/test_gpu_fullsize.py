"""-m gpu: parity at BASELINE.json's full sizes.

The oracle is single-threaded C (~5-13 Mpix/s), so at 4096^2 / 8192^2 it checks
a bounded sample — whole output rows spread over the frame, computed by
lrpo_reproject_rows (rows are independent in the reference loop,
src/reproject.cpp:284) — bit for bit, and size-independent properties cover the
rest of the frame: the three HIP kernel families (one-pixel-per-lane, tile,
LDS-window) must produce identical bytes for the whole frame, repeated runs are
identical, and a frame rendered alone equals the same frame rendered inside a
multi-stream batch.  Tolerance: 0 ULP (north_star allows 1 ULP for
bilinear/bicubic; any NaN equals any NaN)."""
import math

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
# The product's default: the first launch of a geometry computes and fills its geometry-cache entry, later launches read it
# (tests/conftest.py switches the cache off for modules that do not say this).  Every test here renders its geometry more than
# once — families, `again`, fused vs stand-alone tonemap, 16-frame launches — so both kinds of launch meet the oracle rows.
USES_GEO_CACHE = True

NEAREST, BILINEAR, BICUBIC = 0, 1, 2


def sample_rows(h, n=12, seed=0):
    rng = np.random.default_rng(seed)
    rows = {0, 1, h // 2 - 1, h // 2, h - 2, h - 1}
    rows.update(int(v) for v in rng.integers(0, h, size=n))
    return sorted(r for r in rows if 0 <= r < h)


def gpu_frame(lrp, torch, w, h, c, seed, depth=-1):
    t = torch.empty((h, w, c), dtype=torch.float32, device="cuda")
    lrp.synth_fill(t, w, h, c, seed, depth)
    torch.cuda.synchronize()
    return t


def render(lrp, torch, lin, d_in, lout, out_w, out_h, ns, interp, rot, post=None, kernel=None):
    h, w, c = d_in.shape
    prev = lrp.debug_kernel(kernel) if kernel is not None else None
    try:
        d_out = torch.full((out_h, out_w, c), -12345.0, dtype=torch.float32, device="cuda")
        lrp.reproject(lrp.Image(lin, w, h, c, d_in), lrp.Image(lout, out_w, out_h, c, d_out), ns, interp, rot, post=post)
        torch.cuda.synchronize()
    finally:
        if prev is not None:
            lrp.debug_kernel(prev)
    return d_out


def check_rows(oracle, lin, src_host, lout, out_w, out_h, ns, interp, rot, d_out, what, post=None, n=12):
    rows = sample_rows(out_h, n)
    want = oracle.reproject_rows(lin, src_host, lout, out_w, out_h, ns, interp, rot, rows)
    got = d_out.cpu().numpy()
    for y in rows:
        w_row = want[y]
        if post is not None:
            w_row = w_row.reshape(1, out_w, -1).copy()
            oracle.post_process(w_row, post[0], post[1])
            w_row = w_row[0]
        cases.assert_same_bits(got[y], w_row, f"{what} row {y}")


def same_bytes(torch, a, b):
    return bool(torch.equal(a.view(torch.int32), b.view(torch.int32)))


def test_config1_512_equirect_to_rect_nearest_full_frame(lrp, oracle, torch_cuda):
    """BASELINE configs[0]: 512x512x4 equirect(full) -> rectilinear(18, 36), nearest, identity R."""
    torch = torch_cuda
    src = oracle.synth_frame(512, 512, 4, 0x5EED0000)
    lin, lout = lrp.LensInfo.equirectangular(), lrp.LensInfo.rectilinear(18.0, 36.0, 512, 512)
    rot = cases.rotation(lrp, (0.0, 0.0, 0.0))
    want = oracle.reproject(lin, src, lout, 512, 512, 1, NEAREST, rot, threads=8)
    for kern in (0, 1, 2):
        got = render(lrp, torch, lin, torch.from_numpy(src).cuda(), lout, 512, 512, 1, NEAREST, rot, kernel=kern)
        cases.assert_same_bits(got.cpu().numpy(), want, f"config 1, kernel family {kern}")


@pytest.mark.parametrize("name,in_kind,out_kind,interp,deg", [
    ("config2 fisheye->rect bicubic", "eqd180", "rect", BICUBIC, None),
    ("north-star equirect->rect bicubic", "eqr_full", "rect", BICUBIC, (0.0, 0.0, 0.0)),
    ("config3 equirect->fisheye bilinear rotated", "eqr_full", "eqd180", BILINEAR, (30.0, -15.0, 5.0)),
    ("config5 face equirect->rect bicubic pan 90", "eqr_full", "rect", BICUBIC, (90.0, 0.0, 0.0)),
    ("pole face equirect->rect bicubic pitch 90", "eqr_full", "rect", BICUBIC, (0.0, 90.0, 0.0)),
    ("seam equirect->equirect nearest pan 180", "eqr_full", "eqr_full", NEAREST, (180.0, 0.0, 0.0)),
    ("rect->equirect nearest (alias-paired tiles)", "rect", "eqr_full", NEAREST, None),
    ("rect->equirect bilinear yaw 40 (alias-paired tiles)", "rect", "eqr_full", BILINEAR, (40.0, 0.0, 0.0)),
    ("equirect->rect bilinear rotated (plain tile path)", "eqr_full", "rect", BILINEAR, (30.0, -15.0, 5.0)),
])
def test_4k_rgba_rows_against_oracle_and_kernel_families_agree(lrp, oracle, torch_cuda, name, in_kind, out_kind, interp,
                                                               deg):
    torch = torch_cuda
    n = 4096
    d_in = gpu_frame(lrp, torch, n, n, 4, 0x5EED0000)
    src_host = d_in.cpu().numpy()
    lin, lout = cases.lenses(lrp, n, n)[in_kind], cases.lenses(lrp, n, n)[out_kind]
    rot = cases.rotation(lrp, deg)
    outs = [render(lrp, torch, lin, d_in, lout, n, n, 1, interp, rot, kernel=k) for k in (2, 1, 0, 3)]
    check_rows(oracle, lin, src_host, lout, n, n, 1, interp, rot, outs[0], name)
    assert same_bytes(torch, outs[0], outs[1]), f"{name}: window/tile kernels differ"
    assert same_bytes(torch, outs[0], outs[2]), f"{name}: tile/pixel kernels differ"
    assert same_bytes(torch, outs[0], outs[3]), f"{name}: window kernel with / without shared tap coefficients differ"
    again = render(lrp, torch, lin, d_in, lout, n, n, 1, interp, rot)
    assert same_bytes(torch, outs[0], again), f"{name}: not deterministic"


@pytest.mark.parametrize("channels,depth", [(5, 4), (4, 3)])
def test_config4_rect_to_equirect_bicubic_tonemapped(lrp, oracle, torch_cuda, channels, depth):
    """BASELINE configs[3] shape: RGBAZ / RGBZ, rectilinear -> equirect(full), bicubic, exposure 2, Reinhard 4."""
    torch = torch_cuda
    n = 4096
    d_in = gpu_frame(lrp, torch, n, n, channels, 0x5EED0007, depth)
    src_host = d_in.cpu().numpy()
    lin, lout = cases.lenses(lrp, n, n)["rect"], cases.lenses(lrp, n, n)["eqr_full"]
    rot = cases.rotation(lrp, (0.0, 0.0, 0.0))
    post = (2.0, 4.0)
    fused = render(lrp, torch, lin, d_in, lout, n, n, 1, BICUBIC, rot, post=post)
    check_rows(oracle, lin, src_host, lout, n, n, 1, BICUBIC, rot, fused, f"config 4 C={channels}", post=post, n=8)
    plain = render(lrp, torch, lin, d_in, lout, n, n, 1, BICUBIC, rot)
    lrp.post_process(lrp.Image(lout, n, n, channels, plain), *post)
    torch.cuda.synchronize()
    assert same_bytes(torch, fused, plain), "fused and stand-alone post_process differ"
    # the tile kernel (RGBAZ / RGBZ variants) against the one-pixel-per-lane kernel, whole frame
    pixel = render(lrp, torch, lin, d_in, lout, n, n, 1, BICUBIC, rot, post=post, kernel=0)
    assert same_bytes(torch, fused, pixel), "tile and pixel kernels differ"


def test_config5_8k_rgb_to_six_cubemap_faces(lrp, oracle, torch_cuda):
    """BASELINE configs[4] shape: one resident 8192x8192x3 source (PNG path: C = 3),
    six rectilinear 90-degree faces of 2048x2048, bicubic."""
    torch = torch_cuda
    n, face = 8192, 2048
    d_in = gpu_frame(lrp, torch, n, n, 3, 0x5EED0005)
    src_host = d_in.cpu().numpy()
    lin = lrp.LensInfo.equirectangular()
    lout = lrp.LensInfo.rectilinear(18.0, 36.0, face, face)
    degs = [(0, 0, 0), (90, 0, 0), (180, 0, 0), (270, 0, 0), (0, 90, 0), (0, -90, 0)]
    rots = np.stack([cases.rotation(lrp, d) for d in degs])
    outs = [torch.empty((face, face, 3), dtype=torch.float32, device="cuda") for _ in degs]
    lrp.reproject_multi(lrp.Image(lin, n, n, 3, d_in), [lrp.Image(lout, face, face, 3, t) for t in outs], 1, BICUBIC, rots)
    torch.cuda.synchronize()
    for d, t, r in zip(degs, outs, rots):
        check_rows(oracle, lin, src_host, lout, face, face, 1, BICUBIC, r, t, f"face {d}", n=4)
    # RGB tile kernel against the one-pixel-per-lane kernel, whole faces
    for interp in (NEAREST, BILINEAR, BICUBIC):
        a = render(lrp, torch, lin, d_in, lout, face, face, 1, interp, rots[4], kernel=2)
        b = render(lrp, torch, lin, d_in, lout, face, face, 1, interp, rots[4], kernel=0)
        assert same_bytes(torch, a, b), f"RGB interp={interp}: tile and pixel kernels differ"


def test_supersampled_downscale_4k_to_1k(lrp, oracle, torch_cuda):
    """--samples use case: 4096^2 -> 1024^2 with 3x3 sub-samples."""
    torch = torch_cuda
    d_in = gpu_frame(lrp, torch, 4096, 4096, 4, 0x5EED0003)
    src_host = d_in.cpu().numpy()
    lin, lout = cases.lenses(lrp, 4096, 4096)["eqr_full"], cases.lenses(lrp, 1024, 1024)["eqd180"]
    rot = cases.rotation(lrp, (30.0, -15.0, 5.0))
    for interp in (BILINEAR, BICUBIC):
        outs = [render(lrp, torch, lin, d_in, lout, 1024, 1024, 3, interp, rot, kernel=k) for k in (2, 0)]
        check_rows(oracle, lin, src_host, lout, 1024, 1024, 3, interp, rot, outs[0], f"ns=3 interp={interp}", n=6)
        assert same_bytes(torch, outs[0], outs[1])


def test_batch_context_full_size_equals_single_calls(lrp, torch_cuda):
    """The --input-dir path: a frame rendered inside a multi-stream batch of host
    buffers equals the same frame rendered alone (device-resident)."""
    torch = torch_cuda
    n = 2048
    lin, lout = cases.lenses(lrp, n, n)["eqr_full"], cases.lenses(lrp, n, n)["eqd180"]
    rot = cases.rotation(lrp, (30.0, -15.0, 5.0))
    frames = [gpu_frame(lrp, torch, n, n, 4, 0x5EED0000 + i) for i in range(5)]
    singles = [render(lrp, torch, lin, f, lout, n, n, 1, BILINEAR, rot).cpu().numpy() for f in frames]
    hosts = [f.cpu().numpy() for f in frames]
    outs = [np.empty((n, n, 4), dtype=np.float32) for _ in frames]
    with lrp.BatchContext(device=0, n_streams=3) as ctx:
        for s, o in zip(hosts, outs):
            ctx.submit(lrp.Image(lin, n, n, 4, s), lrp.Image(lout, n, n, 4, o), 1, BILINEAR, rot)
        ctx.wait()
    for i, (o, s) in enumerate(zip(outs, singles)):
        cases.assert_same_bits(o, s, f"batch frame {i}")


@pytest.mark.parametrize("in_kind,deg", [("eqd180", None), ("eqr_full", (0.0, 0.0, 0.0))])
def test_bench_launch_shape_16_frames_of_4k_against_oracle(lrp, oracle, torch_cuda, in_kind, deg):
    """Exactly the launch bench.py times: lrp_reproject_batch_device with 16 distinct 4096^2 RGBA frames
    (blockIdx.y = frame), bicubic, fisheye -> rect (configs[1]) and equirect -> rect (north_star).
    Three frames of the batch (first, middle, last) are checked on sampled rows against the oracle;
    every frame must equal the same frame rendered by a single-frame launch."""
    torch = torch_cuda
    n, nb = 4096, 16
    lin, lout = cases.lenses(lrp, n, n)[in_kind], cases.lenses(lrp, n, n)["rect"]
    rot = cases.rotation(lrp, deg)
    srcs = [gpu_frame(lrp, torch, n, n, 4, 0x5EED0000 + i) for i in range(nb)]
    dsts = [torch.full((n, n, 4), -7.0, dtype=torch.float32, device="cuda") for _ in range(nb)]
    lrp.reproject_batch([lrp.Image(lin, n, n, 4, s) for s in srcs], [lrp.Image(lout, n, n, 4, d) for d in dsts], 1, BICUBIC,
                        rot)
    torch.cuda.synchronize()
    for i in (0, 7, 15):
        check_rows(oracle, lin, srcs[i].cpu().numpy(), lout, n, n, 1, BICUBIC, rot, dsts[i], f"batched frame {i}", n=4)
    for i in range(nb):
        single = render(lrp, torch, lin, srcs[i], lout, n, n, 1, BICUBIC, rot)
        assert same_bytes(torch, single, dsts[i]), f"frame {i}: batched launch differs from the single-frame launch"
        del single


def test_images_of_4_gib_and_the_int_index_limit(lrp, oracle, torch_cuda):
    """The reference addresses texels with `int` (src/reproject.cpp:49-51): up to 2^31 floats.  A 16384 x 16384 RGBA source
    is 2^30 floats = 4 GiB — beyond the 32-bit byte offsets of the tile / window kernels, which hand such an image to the
    one-pixel-per-lane kernel (32-bit ELEMENT offsets, 64-bit pointers).  Rendered for all three samplers into a 1024^2
    view and compared with oracle rows; an image of more than 2^31 floats is refused with a clear error, like nothing the
    reference could have addressed either."""
    torch = torch_cuda
    n, m, c = 16384, 1024, 4
    free_b, _total = torch.cuda.mem_get_info()
    if free_b < 6 * (1 << 30):
        pytest.skip("less than 6 GiB of device memory free")
    d_in = gpu_frame(lrp, torch, n, n, c, 0x5EED4A11)
    src = oracle.synth_frame(n, n, c, 0x5EED4A11)
    assert src.nbytes == 1 << 32
    # the far end of the buffer is the far end of the image: the last texel, bit for bit
    assert np.array_equal(d_in[n - 1, n - 64:].cpu().numpy().view(np.uint32), src[n - 1, n - 64:].view(np.uint32))
    lin, lout = lrp.LensInfo.equirectangular(), lrp.LensInfo.rectilinear(18.0, 36.0, m, m)
    for interp, deg in ((BICUBIC, (30.0, -15.0, 5.0)), (BILINEAR, (200.0, 60.0, 0.0)), (NEAREST, (0.0, -89.0, 10.0))):
        rot = cases.rotation(lrp, deg)
        d_out = render(lrp, torch, lin, d_in, lout, m, m, 1, interp, rot)
        check_rows(oracle, lin, src, lout, m, m, 1, interp, rot, d_out, f"4 GiB source, interp {interp}", n=10)
    del d_in, src
    torch.cuda.empty_cache()
    # 2^31 floats + one row: not addressable (straight through the C ABI: the sizes are checked before any data is touched)
    import ctypes

    nat, lib = lrp._native, lrp._native.load()
    tiny = torch.zeros((4, 4, 4), dtype=torch.float32, device="cuda")

    def c_image(lens, w, h):
        im = nat.LrpImage()
        im.lens, im.width, im.height, im.channels, im.data = lens.to_c(), w, h, 4, tiny.data_ptr()
        return im

    small = c_image(lout, 2, 2)
    st = lib.lrp_reproject_device(ctypes.byref(c_image(lin, 32768, 16385)), ctypes.byref(small), 1, BICUBIC, None, None, 0, None)
    assert st == int(lrp.Status.BAD_DIMS) and b"2^31" in lib.lrp_last_error()
    # exactly 2^31 floats passes the size check (dispatch errors come first, as in the reference: the unsupported lens is what is reported)
    st = lib.lrp_reproject_device(ctypes.byref(c_image(lrp.LensInfo(lrp.LensType.FISHEYE_EQUISOLID, (10.0, 3.0)), 32768, 16384)),
                                  ctypes.byref(small), 1, BICUBIC, None, None, 0, None)
    assert st == int(lrp.Status.INPUT_LENS)

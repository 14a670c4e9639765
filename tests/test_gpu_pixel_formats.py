"""-m gpu: the device-side pixel-format kernels of the file path against the host
codecs' conversions, bit for bit, and the packed pipeline (upload in the file format, convert on the
device, reproject, convert back, download) against the oracle pipeline.

References: read_exr widens HALF (src/image_formats.cpp:266-295), save_exr narrows to HALF (:318-333),
read_png / read_jpeg v = pow(p / 255, 2.2) (:196-198, :64-66), save_png uint8(255.9 * pow(clamp(v), 1 / 2.2))
(:155-158)."""
import ctypes

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def host_powf():
    libm = ctypes.CDLL("libm.so.6")
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    return libm.powf


def save_png_code(powf, v):
    """The reference's quantiser on one float (std::max / std::min argument order: NaN -> 1, -0 -> +0)."""
    s = np.float32(v)
    m = s if s < np.float32(1.0) else np.float32(1.0)
    m = m if np.float32(0.0) < m else np.float32(0.0)
    return int(np.float32(255.9) * np.float32(powf(m, np.float32(1.0) / np.float32(2.2)))) & 0xFF  # 1.0f / 2.2f is a float division


def test_half_decode_all_65536_values(lrp, torch_cuda):
    torch = torch_cuda
    bits = np.arange(65536, dtype=np.uint16).reshape(256, 256, 1)
    src = torch.from_numpy(bits.view(np.int16)).cuda()
    dst = torch.empty((256, 256, 1), dtype=torch.float32, device="cuda")
    lrp.decode_pixels(src, lrp.PixelFormat.F16, dst)
    torch.cuda.synchronize()
    want = bits.view(np.float16).astype(np.float32)  # widening is exact
    cases.assert_same_bits(dst.cpu().numpy(), want, "half -> float")
    # NaN payloads too: the widened NaNs keep their top mantissa bits
    got_bits = dst.cpu().numpy().view(np.uint32).reshape(-1)
    nan = np.isnan(want.reshape(-1))
    assert np.array_equal(got_bits[nan] >> 13 & 0x3FF, bits.reshape(-1)[nan] & 0x3FF)


def test_half_encode_round_to_nearest_even(lrp, torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(5)
    vals = rng.integers(0, 1 << 32, size=1 << 20, dtype=np.uint64).astype(np.uint32)
    # plus every float that sits exactly on or next to a half rounding boundary
    halves = np.arange(0x7C00, dtype=np.uint16).view(np.float16).astype(np.float32).view(np.uint32)
    mids = ((halves[:-1].astype(np.uint64) + halves[1:].astype(np.uint64)) // 2).astype(np.uint32)
    edge = np.concatenate([mids - 1, mids, mids + 1, halves, halves | np.uint32(0x80000000)])
    vals = np.concatenate([vals, edge, np.array([0x33000000, 0x33000001, 0x477FEFFF, 0x477FF000, 0x7F800000, 0xFF800000,
                                                 0x7FC00000, 0x7F800001, 0xFFFFFFFF], dtype=np.uint32)])
    vals = np.resize(vals, (vals.size // 4 * 4,)).reshape(-1, 4)
    src = torch.from_numpy(vals.view(np.float32).copy()).cuda()
    dst = torch.zeros((vals.shape[0], 4), dtype=torch.int16, device="cuda")
    lrp.encode_pixels(src, dst, lrp.PixelFormat.F16)
    torch.cuda.synchronize()
    got = dst.cpu().numpy().view(np.uint16)
    with np.errstate(over="ignore", invalid="ignore"):
        want = vals.view(np.float32).astype(np.float16).view(np.uint16)  # IEEE round to nearest even
    finite = ~np.isnan(vals.view(np.float32))
    assert np.array_equal(got[finite], want[finite])
    assert np.all((got[~finite] & 0x7C00) == 0x7C00) and np.all((got[~finite] & 0x3FF) != 0)  # NaN stays NaN


def test_u8_gamma_decode_and_encode_match_the_host_powf(lrp, torch_cuda):
    torch = torch_cuda
    powf = host_powf()
    # decode: RGBA8 as libpng delivers it -> RGB float (alpha dropped)
    rgba = np.arange(256 * 4, dtype=np.uint32).reshape(256, 4).astype(np.uint8)
    rgba[:, 0] = np.arange(256)
    dst = torch.empty((256, 3), dtype=torch.float32, device="cuda")
    lrp.decode_pixels(torch.from_numpy(rgba).cuda(), lrp.PixelFormat.U8_GAMMA, dst)
    torch.cuda.synchronize()
    want = np.array([[powf(np.float32(v) / np.float32(255.0), np.float32(2.2)) for v in px[:3]] for px in rgba], dtype=np.float32)
    cases.assert_same_bits(dst.cpu().numpy(), want, "u8 -> float")
    # encode: floats around every threshold, out-of-range values, specials; RGB float -> RGBA8 with alpha 255
    _, thr = lrp.pixel_tables()
    tb = thr.view(np.uint32).astype(np.int64)
    near = np.concatenate([tb - 2, tb - 1, tb, tb + 1, tb + 2]).clip(0, 0x3F800000).astype(np.uint32).view(np.float32)
    rng = np.random.default_rng(9)
    extra = np.array([-1.0, -0.0, 0.0, 1.0, 1.0000001, 2.5, np.inf, -np.inf, np.nan, 1e-30, 0.5, 0.21404114], dtype=np.float32)
    vals = np.concatenate([near, rng.random(30000, dtype=np.float32), rng.normal(0.5, 1.0, 3000).astype(np.float32), extra])
    vals = np.resize(vals, (vals.size // 3 * 3,)).reshape(-1, 3)
    out = torch.zeros((vals.shape[0], 4), dtype=torch.uint8, device="cuda")
    lrp.encode_pixels(torch.from_numpy(vals).cuda(), out, lrp.PixelFormat.U8_GAMMA, fill=255)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    want = np.array([[save_png_code(powf, v) for v in px] for px in vals], dtype=np.uint8)
    assert np.array_equal(got[:, :3], want)
    assert np.all(got[:, 3] == 255)


@pytest.mark.parametrize("fmt,channels", [("half", 4), ("half", 5), ("half", 3), ("u8", 3)])
def test_packed_pipeline_equals_oracle_pipeline(lrp, oracle, torch_cuda, fmt, channels):
    """Frames cross the boundary in their file format: HALF in / HALF out (EXR), RGBA8 in / RGBA8 out (PNG)."""
    powf = host_powf()
    in_w, in_h, out_w, out_h = 160, 96, 128, 80
    lin = cases.lenses(lrp, in_w, in_h)["eqr_full"]
    lout = cases.lenses(lrp, out_w, out_h)["rect"]
    rot = cases.rotation(lrp, (30.0, -15.0, 5.0))
    post = (2.0, 4.0)
    rng = np.random.default_rng(21)
    n_img = 5
    ins, outs, wants = [], [], []
    for i in range(n_img):
        if fmt == "half":
            packed_in = (rng.random((in_h, in_w, channels), dtype=np.float32) * 4).astype(np.float16)
            src = packed_in.astype(np.float32)
            packed_out = np.zeros((out_h, out_w, channels), dtype=np.float16)
        else:
            packed_in = rng.integers(0, 256, size=(in_h, in_w, 4), dtype=np.uint8)
            dec, _ = lrp.pixel_tables()
            src = dec[packed_in[..., :3]]
            packed_out = np.zeros((out_h, out_w, 4), dtype=np.uint8)
        want = oracle.reproject(lin, src, lout, out_w, out_h, 1, 2, rot)
        oracle.post_process(want, *post)
        if fmt == "half":
            with np.errstate(over="ignore"):
                want = want.astype(np.float16)
        else:
            q = np.array([save_png_code(powf, v) for v in want.reshape(-1)], dtype=np.uint8).reshape(out_h, out_w, 3)
            want = np.concatenate([q, np.full((out_h, out_w, 1), 255, np.uint8)], axis=-1)
        ins.append(packed_in)
        outs.append(packed_out)
        wants.append(want)
    pf = lrp.PixelFormat.F16 if fmt == "half" else lrp.PixelFormat.U8_GAMMA
    with lrp.BatchContext(device=0, n_streams=3) as ctx:
        tickets = [ctx.submit_packed(lrp.Image(lin, in_w, in_h, channels, None), pf, a, lrp.Image(lout, out_w, out_h, channels, None),
                                     pf, b, 255, 1, 2, rot, post=post) for a, b in zip(ins, outs)]
        for t in reversed(tickets):  # out of order on purpose
            ctx.wait_ticket(t)
        ctx.wait()
    for i, (got, want) in enumerate(zip(outs, wants)):
        if fmt == "half":
            assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), f"image {i}"
        else:
            assert np.array_equal(got, want), f"image {i}"

"""-m gpu: RGBAZ (colour + depth, 20-byte pixels) output written as whole 16-byte chunks.

The tile kernels and the window kernel exchange a wavefront's 64 RGBAZ pixels through LDS and store
them as runs (lrp_kernel_v2.h store_rgbaz_run) when the whole row of 64 / pass of 16 x 4 pixels
lies in the image, per lane otherwise.  Every geometry of that choice against the oracle, bit for
bit: whole and partial tiles in one image, odd sizes (the centre column is its own mirror image:
two runs overlap there), mirrored pixels / rays / blocks (runs written right to left and bottom
up), rotated mappings (plain order), super-sampling (normalised values), the fused tonemap, corner
blocks of a narrow view inside a panorama, a row band, a batch."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

C = 5
SIZES = [(256, 192), (255, 193), (320, 130), (129, 67), (64, 16), (63, 5)]


def render(lrp, torch, lin, src, lout, out_w, out_h, ns, interp, rot, post=None):
    d_in = torch.from_numpy(np.ascontiguousarray(src)).cuda()
    d_out = torch.full((out_h, out_w, C), -4321.0, dtype=torch.float32, device="cuda")
    lrp.reproject(lrp.Image(lin, src.shape[1], src.shape[0], C, d_in), lrp.Image(lout, out_w, out_h, C, d_out), ns, interp,
                  rot, post=post)
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


@pytest.mark.parametrize("out_w,out_h", SIZES)
@pytest.mark.parametrize("interp", [0, 1, 2])
def test_runs_every_kernel_family(lrp, oracle, torch_cuda, out_w, out_h, interp):
    in_w, in_h = 170, 120
    src = cases.hash_noise(in_h, in_w, C, seed=out_w + 31 * out_h + interp)
    # (source, target, rotation): mirrored pixels / blocks, mirrored rays, plain order, narrow view in a panorama
    for in_name, out_name, deg in (("eqr_full", "rect", None), ("eqd180", "rect", (0.0, 0.0, 0.0)), ("eqr_full", "eqd180", (30.0, -15.0, 5.0)),
                                   ("eqr_full", "rect", (20.0, 10.0, -5.0)), ("rect", "eqr_full", None), ("rect_tele", "eqr_full", (90.0, 0.0, 0.0))):
        lin, lout = cases.lenses(lrp, in_w, in_h)[in_name], cases.lenses(lrp, out_w, out_h)[out_name]
        rot = cases.rotation(lrp, deg)
        with np.errstate(all="ignore"):
            want = oracle.reproject(lin, src, lout, out_w, out_h, 1, interp, rot, threads=8)
        got = render(lrp, torch_cuda, lin, src, lout, out_w, out_h, 1, interp, rot)
        cases.assert_same_bits(got, want, f"{in_name}->{out_name} {out_w}x{out_h} interp={interp} rot={deg}")


@pytest.mark.parametrize("interp", [0, 1, 2])
def test_runs_supersampled_and_tonemapped(lrp, oracle, torch_cuda, interp):
    in_w, in_h, out_w, out_h = 150, 110, 192, 72
    with np.errstate(all="ignore"):
        src = cases.hash_noise(in_h, in_w, C, seed=77 + interp) * np.float32(3.0)
    lin, lout = cases.lenses(lrp, in_w, in_h)["rect"], cases.lenses(lrp, out_w, out_h)["eqr_full"]
    for ns, post in ((2, None), (1, (2.0, 4.0)), (3, (0.5, 1.5))):
        with np.errstate(all="ignore"):
            want = oracle.reproject(lin, src, lout, out_w, out_h, ns, interp, None, threads=8)
        if post:
            oracle.post_process(want, *post)
        got = render(lrp, torch_cuda, lin, src, lout, out_w, out_h, ns, interp, None, post=post)
        cases.assert_same_bits(got, want, f"ns={ns} post={post} interp={interp}")


def test_runs_row_band_and_batch(lrp, oracle, torch_cuda):
    torch = torch_cuda
    in_w, in_h, out_w, out_h = 150, 110, 256, 160
    lin, lout = cases.lenses(lrp, in_w, in_h)["rect"], cases.lenses(lrp, out_w, out_h)["eqr_full"]
    srcs = [cases.hash_noise(in_h, in_w, C, seed=900 + i) for i in range(3)]
    wants = [oracle.reproject(lin, s, lout, out_w, out_h, 1, 2, None, threads=8) for s in srcs]
    # rows [37, 37 + 70) of the first image: everything else keeps the poison value
    d_in = torch.from_numpy(srcs[0]).cuda()
    d_out = torch.full((out_h, out_w, C), -4321.0, dtype=torch.float32, device="cuda")
    lrp.reproject_rows(lrp.Image(lin, in_w, in_h, C, d_in), lrp.Image(lout, out_w, out_h, C, d_out), 1, 2, 37, 70)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    cases.assert_same_bits(got[37:107], wants[0][37:107], "row band")
    assert (got[:37] == np.float32(-4321.0)).all() and (got[107:] == np.float32(-4321.0)).all()
    # one launch for the three frames
    ins = [lrp.Image(lin, in_w, in_h, C, torch.from_numpy(s).cuda()) for s in srcs]
    outs_t = [torch.full((out_h, out_w, C), -4321.0, dtype=torch.float32, device="cuda") for _ in srcs]
    outs = [lrp.Image(lout, out_w, out_h, C, t) for t in outs_t]
    lrp.reproject_batch(ins, outs, 1, 2, None)
    torch.cuda.synchronize()
    for i, t in enumerate(outs_t):
        cases.assert_same_bits(t.cpu().numpy(), wants[i], f"batch frame {i}")

"""-m gpu: bicubic with num_samples == 2 through the window kernel's supersampling instantiations (csrc/lrp_win_kernel.h SS).

The reference sums the ns x ns sub-samples of a pixel in ssx-outer, ssy-inner order into a zero-initialised accumulator and
multiplies by 1 / (ns * ns) (src/reproject.cpp:290-298, 334-341).  In the SS instantiations a block is 16 x 4 output pixels and
its four passes are the four sub-samples of every pixel, summed in registers in that order.  Every lens pair, channel count,
rotation, odd sizes, strong down-scaling (what --samples is for), corner / edge blocks, fused tonemap, row bands and batches,
against the live oracle at 0 ULP; `win_ss` 0 (the tile kernel) must give the same bits."""
import numpy as np
import pytest

import cases
import golden_cases

pytestmark = pytest.mark.gpu

BICUBIC = 2


def _render(lrp, torch, lin, d_in, lout, ow, oh, rot, post=None, ns=2):
    h, w, c = d_in.shape
    d_out = torch.full((oh, ow, c), -12345.0, dtype=torch.float32, device="cuda")
    lrp.reproject(lrp.Image(lin, w, h, c, d_in), lrp.Image(lout, ow, oh, c, d_out), ns, BICUBIC, rot, post=post)
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


@pytest.mark.parametrize("channels", [3, 4, 5])
def test_every_lens_pair_against_the_live_oracle(lrp, oracle, torch_cuda, channels):
    torch = torch_cuda
    k = 0
    for out_name in ("rect", "eqd180", "eqr_full", "eqr_part"):
        for in_name in ("rect", "rect_tele", "eqd180", "eqr_full", "eqr_part"):
            k += 1
            rot_name = list(golden_cases.ROTS)[k % 5]
            iw, ih, ow, oh = [(61, 47, 53, 41), (160, 120, 72, 67), (256, 128, 35, 19), (96, 80, 131, 90)][k % 4]
            post = (1.5, 3.0) if k % 3 == 0 else None
            src = cases.hash_noise(ih, iw, channels, seed=0x552 + 8 * k + channels, planted=(k % 2 == 0))
            lin, lout = cases.lenses(lrp, iw, ih)[in_name], cases.lenses(lrp, ow, oh)[out_name]
            rot = cases.rotation(lrp, golden_cases.ROTS[rot_name])
            want = oracle.reproject(lin, src, lout, ow, oh, 2, BICUBIC, rot)
            if post:
                want = oracle.post_process(want, *post)
            d_in = torch.from_numpy(src).cuda()
            what = f"{in_name} {iw}x{ih} -> {out_name} {ow}x{oh} C={channels} {rot_name} post={post}"
            cases.assert_same_bits(_render(lrp, torch, lin, d_in, lout, ow, oh, rot, post), want, "window kernel (SS), " + what)
            prev = lrp.debug_set("win_ss", 0)
            cases.assert_same_bits(_render(lrp, torch, lin, d_in, lout, ow, oh, rot, post), want, "tile kernel, " + what)
            lrp.debug_set("win_ss", prev)


def test_bands_batches_and_the_other_sample_counts(lrp, oracle, torch_cuda):
    """Row bands (lrp_reproject_rows_device: bands that start and end inside a 4-row block), a batch of five frames, and
    num_samples 1 / 3 around it (their own kernels) on one geometry."""
    torch = torch_cuda
    iw, ih, ow, oh, c = 300, 200, 147, 101, 4
    lin, lout = lrp.LensInfo.equirectangular(), lrp.LensInfo.rectilinear(18.0, 36.0, ow, oh)
    rot = lrp.rotation_matrix(0.4, -0.2, 0.05)
    srcs = [cases.hash_noise(ih, iw, c, seed=0xBA2D + i, planted=True) for i in range(5)]
    d_ins = [torch.from_numpy(s).cuda() for s in srcs]
    for ns in (1, 2, 3):
        want = oracle.reproject(lin, srcs[0], lout, ow, oh, ns, BICUBIC, rot)
        cases.assert_same_bits(_render(lrp, torch, lin, d_ins[0], lout, ow, oh, rot, ns=ns), want, f"num_samples {ns}")
    wants = [oracle.reproject(lin, s, lout, ow, oh, 2, BICUBIC, rot) for s in srcs]
    outs = [torch.full((oh, ow, c), -1.0, dtype=torch.float32, device="cuda") for _ in srcs]
    lrp.reproject_batch([lrp.Image(lin, iw, ih, c, d) for d in d_ins], [lrp.Image(lout, ow, oh, c, o) for o in outs], 2, BICUBIC, rot)
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        cases.assert_same_bits(o.cpu().numpy(), wants[i], f"batch of five, frame {i}")
    banded = torch.full((oh, ow, c), -7.0, dtype=torch.float32, device="cuda")
    cuts = [0, 3, 10, 11, 50, 97, oh]
    for a, b in zip(cuts[:-1], cuts[1:]):
        lrp.reproject_rows(lrp.Image(lin, iw, ih, c, d_ins[0]), lrp.Image(lout, ow, oh, c, banded), 2, BICUBIC, a, b - a, rot)
    torch.cuda.synchronize()
    cases.assert_same_bits(banded.cpu().numpy(), wants[0], "row bands")


def test_downscale_4k_to_2k_whole_frame_rows(lrp, oracle, torch_cuda):
    """The case --samples exists for (README: raise it when scaling down): 4096^2 -> 2048^2, num_samples 2, bicubic RGBA —
    sampled rows against the oracle, the whole frame against the tile kernel and the one-pixel-per-lane kernel."""
    torch = torch_cuda
    n, m, c = 4096, 2048, 4
    d_in = torch.empty((n, n, c), dtype=torch.float32, device="cuda")
    lrp.synth_fill(d_in, n, n, c, 0x5EED0002)
    torch.cuda.synchronize()
    src = d_in.cpu().numpy()
    for in_name, out_name, deg in (("eqd180", "rect", None), ("eqr_full", "rect", (30.0, -15.0, 5.0)), ("rect", "eqr_full", (0.0, 0.0, 0.0))):
        lin, lout = cases.lenses(lrp, n, n)[in_name], cases.lenses(lrp, m, m)[out_name]
        rot = cases.rotation(lrp, deg)
        got = _render(lrp, torch, lin, d_in, lout, m, m, rot)
        rows = [0, 1, 2, 3, 4, 511, 1023, 1024, 1500, 2044, 2045, 2046, 2047]
        want = oracle.reproject_rows(lin, src, lout, m, m, 2, BICUBIC, rot, rows)
        for y in rows:
            cases.assert_same_bits(got[y], want[y], f"{in_name} -> {out_name} {deg}: row {y}")
        prev = lrp.debug_set("win_ss", 0)
        tile = _render(lrp, torch, lin, d_in, lout, m, m, rot)
        lrp.debug_set("win_ss", prev)
        assert np.array_equal(got.view(np.uint32), tile.view(np.uint32)), f"{in_name} -> {out_name}: window (SS) and tile kernels differ"
        prev = lrp.debug_kernel(0)
        pixel = _render(lrp, torch, lin, d_in, lout, m, m, rot)
        lrp.debug_kernel(prev)
        assert np.array_equal(got.view(np.uint32), pixel.view(np.uint32)), f"{in_name} -> {out_name}: window (SS) and pixel kernels differ"

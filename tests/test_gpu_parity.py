"""-m gpu: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

north_star tolerance: bit-exact for nearest, within 1 ULP for bilinear/bicubic.
Because the device math reproduces the host libm bit for bit and every float
operation keeps the reference's order and rounding, the tests demand the
stronger bar — identical bits — for all three interpolation modes
(TOL_ULP = 0; any NaN equals any NaN)."""
import itertools
import math

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

TOL_ULP = 0
NEAREST, BILINEAR, BICUBIC = 0, 1, 2


def run_gpu(lrp, torch, in_lens, src, out_lens, out_w, out_h, ns, interp, rot=None, post=None, poison=True):
    h, w, c = src.shape
    d_in = torch.from_numpy(np.ascontiguousarray(src)).cuda()
    d_out = torch.full((out_h, out_w, c), -12345.0, dtype=torch.float32, device="cuda")
    im_in = lrp.Image(in_lens, w, h, c, d_in)
    im_out = lrp.Image(out_lens, out_w, out_h, c, d_out)
    lrp.reproject(im_in, im_out, ns, interp, rot, post=post)
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


LENS_PAIRS = [(o, i) for o in ("rect", "eqd180", "eqr_full") for i in ("rect", "eqd180", "eqr_full", "eqr_part")]


@pytest.mark.parametrize("interp", [NEAREST, BILINEAR, BICUBIC])
@pytest.mark.parametrize("out_name,in_name", LENS_PAIRS)
def test_all_lens_pairs_small(lrp, oracle, torch_cuda, out_name, in_name, interp):
    """Every reachable cell of SURVEY Appendix B at odd sizes (centre pixel has
    cx == cy == 0 -> the NaN paths), RGBA, identity-rotation and rotated."""
    torch = torch_cuda
    in_w, in_h, out_w, out_h = 61, 47, 53, 41
    src = cases.hash_noise(in_h, in_w, 4, seed=sum(map(ord, out_name + in_name)))
    lin = cases.lenses(lrp, in_w, in_h)[in_name]
    lout = cases.lenses(lrp, out_w, out_h)[out_name]
    for deg in (None, (0.0, 0.0, 0.0), (30.0, -15.0, 5.0)):
        rot = cases.rotation(lrp, deg)
        want = oracle.reproject(lin, src, lout, out_w, out_h, 1, interp, rot)
        got = run_gpu(lrp, torch, lin, src, lout, out_w, out_h, 1, interp, rot)
        cases.assert_same_bits(got, want, f"{out_name}<-{in_name} interp={interp} rot={deg}")


@pytest.mark.parametrize("channels", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("interp", [NEAREST, BILINEAR, BICUBIC])
def test_channel_counts(lrp, oracle, torch_cuda, channels, interp):
    torch = torch_cuda
    in_w, in_h, out_w, out_h = 64, 32, 32, 32  # power-of-two wrap width
    src = cases.hash_noise(in_h, in_w, channels, seed=channels * 7 + interp)
    lin = cases.lenses(lrp, in_w, in_h)["eqr_full"]
    lout = cases.lenses(lrp, out_w, out_h)["rect"]
    rot = cases.rotation(lrp, (180.0, 0.0, 0.0))  # look at the +-pi seam
    want = oracle.reproject(lin, src, lout, out_w, out_h, 1, interp, rot)
    got = run_gpu(lrp, torch, lin, src, lout, out_w, out_h, 1, interp, rot)
    cases.assert_same_bits(got, want, f"C={channels} interp={interp}")


@pytest.mark.parametrize("ns", [1, 2, 3])
@pytest.mark.parametrize("interp", [NEAREST, BILINEAR, BICUBIC])
def test_supersampling(lrp, oracle, torch_cuda, ns, interp):
    torch = torch_cuda
    in_w, in_h, out_w, out_h = 61, 47, 29, 23  # down-scaling, the --samples use case
    for in_name, out_name in (("eqr_full", "eqd180"), ("eqd180", "rect"), ("rect", "eqr_full")):
        src = cases.hash_noise(in_h, in_w, 3, seed=ns * 31 + interp)
        lin = cases.lenses(lrp, in_w, in_h)[in_name]
        lout = cases.lenses(lrp, out_w, out_h)[out_name]
        rot = cases.rotation(lrp, (30.0, -15.0, 5.0))
        want = oracle.reproject(lin, src, lout, out_w, out_h, ns, interp, rot)
        got = run_gpu(lrp, torch, lin, src, lout, out_w, out_h, ns, interp, rot)
        cases.assert_same_bits(got, want, f"ns={ns} interp={interp} {out_name}<-{in_name}")


@pytest.mark.parametrize("deg", cases.ROTATIONS_DEG)
def test_rotations_seam_and_poles(lrp, oracle, torch_cuda, deg):
    """Wrapping source with a non-power-of-two width, every rotation of §8c
    (seam crossing at pan=180, pole at pitch=90)."""
    torch = torch_cuda
    in_w, in_h, out_w, out_h = 100, 50, 48, 36
    src = cases.hash_noise(in_h, in_w, 4, seed=99)
    lin = cases.lenses(lrp, in_w, in_h)["eqr_full"]
    rot = cases.rotation(lrp, deg)
    for out_name in ("rect", "eqd180", "eqr_full"):
        lout = cases.lenses(lrp, out_w, out_h)[out_name]
        for interp in (NEAREST, BILINEAR, BICUBIC):
            want = oracle.reproject(lin, src, lout, out_w, out_h, 1, interp, rot)
            got = run_gpu(lrp, torch, lin, src, lout, out_w, out_h, 1, interp, rot)
            cases.assert_same_bits(got, want, f"rot={deg} out={out_name} interp={interp}")


def test_tiny_and_ragged_sizes(lrp, oracle, torch_cuda):
    """1x1 images, single rows/columns, sizes that are not multiples of the tile."""
    torch = torch_cuda
    for (in_w, in_h, out_w, out_h) in [(1, 1, 1, 1), (1, 1, 7, 5), (5, 1, 3, 9), (2, 3, 33, 9), (33, 9, 65, 17),
                                       (257, 3, 31, 1)]:
        src = cases.hash_noise(in_h, in_w, 4, seed=in_w * 100 + out_w, planted=False)
        for in_name, out_name in (("eqr_full", "rect"), ("rect", "eqd180"), ("eqd180", "eqr_full")):
            lin = cases.lenses(lrp, in_w, in_h)[in_name]
            lout = cases.lenses(lrp, out_w, out_h)[out_name]
            for interp in (NEAREST, BILINEAR, BICUBIC):
                want = oracle.reproject(lin, src, lout, out_w, out_h, 1, interp, cases.rotation(lrp, (0, 0, 0)))
                got = run_gpu(lrp, torch, lin, src, lout, out_w, out_h, 1, interp, cases.rotation(lrp, (0, 0, 0)))
                cases.assert_same_bits(got, want, f"{in_w}x{in_h}->{out_w}x{out_h} {out_name}<-{in_name} i={interp}")


def test_special_texels_and_negative_zero(lrp, oracle, torch_cuda):
    """-0.0 texels come out as +0.0 for n=1 (0.0f + s) * 1.0f; NaN/inf texels
    propagate exactly as on the CPU; denormals are not flushed."""
    torch = torch_cuda
    in_w, in_h = 16, 8
    src = np.zeros((in_h, in_w, 4), dtype=np.float32)
    src[..., 0] = -0.0
    src[..., 1] = np.float32(1e-40)
    src[2, 3, 2] = np.inf
    src[5, 9, 3] = np.nan
    lin = cases.lenses(lrp, in_w, in_h)["eqr_full"]
    lout = cases.lenses(lrp, 16, 8)["eqr_full"]
    for interp in (NEAREST, BILINEAR, BICUBIC):
        want = oracle.reproject(lin, src, lout, 16, 8, 1, interp, None)
        got = run_gpu(lrp, torch, lin, src, lout, 16, 8, 1, interp, None)
        cases.assert_same_bits(got, want, f"special texels interp={interp}")
    got = run_gpu(lrp, torch, lin, src, lout, 16, 8, 1, NEAREST, None)
    assert not np.signbit(got[..., 0]).any()
    assert (got[..., 1] == np.float32(1e-40)).all()


def test_post_process_fused_and_standalone(lrp, oracle, torch_cuda):
    torch = torch_cuda
    in_w, in_h, out_w, out_h = 61, 47, 53, 41
    for c in (3, 4, 5):
        src = cases.hash_noise(in_h, in_w, c, seed=c) * np.float32(4.0)
        lin = cases.lenses(lrp, in_w, in_h)["rect"]
        lout = cases.lenses(lrp, out_w, out_h)["eqr_full"]
        for (exposure, reinhard) in ((2.0, 4.0), (0.5, 1.0), (1.0, 2.5)):
            want = oracle.reproject(lin, src, lout, out_w, out_h, 1, BICUBIC, None)
            oracle.post_process(want, exposure, reinhard)
            fused = run_gpu(lrp, torch, lin, src, lout, out_w, out_h, 1, BICUBIC, None, post=(exposure, reinhard))
            cases.assert_same_bits(fused, want, f"fused post C={c}")
            plain = run_gpu(lrp, torch, lin, src, lout, out_w, out_h, 1, BICUBIC, None)
            t = torch.from_numpy(plain).cuda()
            lrp.post_process(lrp.Image(lout, out_w, out_h, c, t), exposure, reinhard)
            torch.cuda.synchronize()
            cases.assert_same_bits(t.cpu().numpy(), want, f"standalone post C={c}")


def test_num_samples_zero_leaves_output_untouched(lrp, torch_cuda):
    torch = torch_cuda
    src = cases.hash_noise(8, 8, 4, seed=1)
    l = cases.lenses(lrp, 8, 8)["rect"]
    got = run_gpu(lrp, torch, l, src, l, 8, 8, 0, BILINEAR)
    assert (got == np.float32(-12345.0)).all()


def test_host_buffer_path_and_batch_context(lrp, oracle, torch_cuda):
    """The reference's own calling convention: host pointers in, host pointers out."""
    in_w, in_h, out_w, out_h = 96, 48, 64, 40
    lin = cases.lenses(lrp, in_w, in_h)["eqr_full"]
    lout = cases.lenses(lrp, out_w, out_h)["eqd180"]
    rot = cases.rotation(lrp, (30.0, -15.0, 5.0))
    srcs = [cases.hash_noise(in_h, in_w, 4, seed=s) for s in range(7)]
    wants = [oracle.reproject(lin, s, lout, out_w, out_h, 1, BILINEAR, rot) for s in srcs]
    out = np.empty((out_h, out_w, 4), dtype=np.float32)
    lrp.reproject(lrp.Image(lin, in_w, in_h, 4, srcs[0]), lrp.Image(lout, out_w, out_h, 4, out), 1, BILINEAR, rot)
    cases.assert_same_bits(out, wants[0], "host path")
    outs = [np.empty((out_h, out_w, 4), dtype=np.float32) for _ in srcs]
    with lrp.BatchContext(device=0, n_streams=3) as ctx:
        for s, o in zip(srcs, outs):
            ctx.submit(lrp.Image(lin, in_w, in_h, 4, s), lrp.Image(lout, out_w, out_h, 4, o), 1, BILINEAR, rot)
        ctx.wait()
    for i, (o, w) in enumerate(zip(outs, wants)):
        cases.assert_same_bits(o, w, f"batch image {i}")
    # host post_process
    img = wants[0].copy() * np.float32(3.0)
    ref = img.copy()
    oracle.post_process(ref, 2.0, 4.0)
    lrp.post_process(lrp.Image(lout, out_w, out_h, 4, img), 2.0, 4.0)
    cases.assert_same_bits(img, ref, "host post_process")


def test_multi_output_cubemap_faces(lrp, oracle, torch_cuda):
    """BASELINE config 5 shape at small size: one resident source, six faces."""
    torch = torch_cuda
    in_w, in_h, face = 128, 64, 24
    src = cases.hash_noise(in_h, in_w, 3, seed=5)
    lin = cases.lenses(lrp, in_w, in_h)["eqr_full"]
    lout = lrp.LensInfo.rectilinear(18.0, 36.0, face, face)
    degs = [(0, 0, 0), (90, 0, 0), (180, 0, 0), (270, 0, 0), (0, 90, 0), (0, -90, 0)]
    rots = np.stack([cases.rotation(lrp, d) for d in degs])
    d_in = torch.from_numpy(src).cuda()
    d_outs = [torch.empty((face, face, 3), dtype=torch.float32, device="cuda") for _ in degs]
    lrp.reproject_multi(lrp.Image(lin, in_w, in_h, 3, d_in), [lrp.Image(lout, face, face, 3, t) for t in d_outs], 1,
                        BICUBIC, rots)
    torch.cuda.synchronize()
    for d, t, r in zip(degs, d_outs, rots):
        want = oracle.reproject(lin, src, lout, face, face, 1, BICUBIC, r)
        cases.assert_same_bits(t.cpu().numpy(), want, f"face {d}")


def test_multi_output_stays_ordered_on_the_callers_stream(lrp, oracle, torch_cuda):
    """lrp_reproject_multi_device deals its launches over an internal side stream (fork / join by events): work queued on
    the caller's stream before the call (the copy that fills the source) must be seen by every face, work queued after it
    (the copy that takes the faces away, the next source) must see every face; repeated so that a missing edge shows.
    Captured into a hipGraph the call keeps to the one stream and replays to the same bits."""
    torch = torch_cuda
    in_w, in_h, face = 512, 256, 160
    lin = cases.lenses(lrp, in_w, in_h)["eqr_full"]
    lout = lrp.LensInfo.rectilinear(18.0, 36.0, face, face)
    degs = [(0, 0, 0), (90, 0, 0), (180, 0, 0), (270, 0, 0), (0, 90, 0), (0, -90, 0)]
    rots = np.stack([cases.rotation(lrp, d) for d in degs])
    srcs = [cases.hash_noise(in_h, in_w, 3, seed=40 + k) for k in range(4)]
    pinned = [torch.from_numpy(s).pin_memory() for s in srcs]
    wants = [[oracle.reproject(lin, s, lout, face, face, 1, BICUBIC, r) for r in rots] for s in srcs]
    d_in = torch.zeros((in_h, in_w, 3), dtype=torch.float32, device="cuda")
    d_outs = [torch.zeros((face, face, 3), dtype=torch.float32, device="cuda") for _ in degs]
    im_in, im_outs = lrp.Image(lin, in_w, in_h, 3, d_in), [lrp.Image(lout, face, face, 3, t) for t in d_outs]
    lrp.reproject_multi(im_in, im_outs, 1, BICUBIC, rots)  # tables
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    taken = [[torch.empty((face, face, 3), dtype=torch.float32).pin_memory() for _ in degs] for _ in range(12)]
    with torch.cuda.stream(stream):
        for it in range(12):
            d_in.copy_(pinned[it % 4], non_blocking=True)
            lrp.reproject_multi(im_in, im_outs, 1, BICUBIC, rots)
            for t, h in zip(d_outs, taken[it]):
                h.copy_(t, non_blocking=True)
    stream.synchronize()
    for it in range(12):
        for k, d in enumerate(degs):
            cases.assert_same_bits(taken[it][k].numpy(), wants[it % 4][k], f"round {it} face {d}")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        lrp.reproject_multi(im_in, im_outs, 1, BICUBIC, rots)
    for k in (2, 1):
        d_in.copy_(pinned[k])
        for t in d_outs:
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        for j, d in enumerate(degs):
            cases.assert_same_bits(d_outs[j].cpu().numpy(), wants[k][j], f"graph replay source {k} face {d}")


def test_unsupported_dispatch_matches_reference_messages(lrp, torch_cuda):
    torch = torch_cuda
    t = torch.zeros((4, 4, 4), dtype=torch.float32, device="cuda")
    good = cases.lenses(lrp, 4, 4)["rect"]
    equisolid = lrp.LensInfo(lrp.LensType.FISHEYE_EQUISOLID, (10.5, 3.14), 36.0, 36.0)
    stereo = lrp.LensInfo(lrp.LensType.FISHEYE_STEREOGRAPHIC, (), 36.0, 36.0)
    with pytest.raises(lrp.LrpError, match="Output lens type not supported."):
        lrp.reproject(lrp.Image(good, 4, 4, 4, t), lrp.Image(equisolid, 4, 4, 4, t.clone()), 1, NEAREST)
    with pytest.raises(lrp.LrpError, match="Input lens type not supported."):
        lrp.reproject(lrp.Image(stereo, 4, 4, 4, t), lrp.Image(good, 4, 4, 4, t.clone()), 1, NEAREST)
    with pytest.raises(lrp.LrpError, match="Interpolation method not supported."):
        lrp.reproject(lrp.Image(good, 4, 4, 4, t), lrp.Image(good, 4, 4, 4, t.clone()), 1, 7)
    # output-lens failure wins over input-lens failure, like the reference's dispatch order
    with pytest.raises(lrp.LrpError, match="Output lens type not supported."):
        lrp.reproject(lrp.Image(stereo, 4, 4, 4, t), lrp.Image(equisolid, 4, 4, 4, t.clone()), 1, 7)


def test_synth_frames_match_host_generator(lrp, oracle, torch_cuda):
    torch = torch_cuda
    for (w, h, c, depth) in ((64, 32, 4, -1), (33, 17, 5, 4), (40, 8, 4, 3), (16, 16, 3, -1)):
        t = torch.empty((h, w, c), dtype=torch.float32, device="cuda")
        lrp.synth_fill(t, w, h, c, 0x5EED0000 + c, depth)
        torch.cuda.synchronize()
        cases.assert_same_bits(t.cpu().numpy(), oracle.synth_frame(w, h, c, 0x5EED0000 + c, depth), "synth")

"""-m gpu: behaviour of the drop-in boundary beyond numerics — concurrent callers
(the reference calls reproject() from -j N pool threads, src/main.cpp:538-544),
hipGraph capture of the device-resident entry point, the table cache, the kernel
family knob."""
import threading

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def test_concurrent_host_callers_like_the_reference_thread_pool(lrp, oracle, torch_cuda):
    """8 host threads, each rendering its own images through the synchronous host-buffer
    entry point (the reference's calling convention), different lenses per thread."""
    in_w, in_h, out_w, out_h = 160, 96, 128, 80
    names = ["rect", "eqd180", "eqr_full", "eqr_part"]
    jobs = []
    for t in range(8):
        lin = cases.lenses(lrp, in_w, in_h)[names[t % 4]]
        lout = cases.lenses(lrp, out_w, out_h)[names[(t + 1) % 3]]
        rot = cases.rotation(lrp, (10.0 * t, -5.0 * t, 3.0 * t))
        src = cases.hash_noise(in_h, in_w, 4, seed=100 + t)
        want = oracle.reproject(lin, src, lout, out_w, out_h, 1, t % 3, rot)
        jobs.append((lin, lout, rot, src, want, t % 3))
    errors = []

    def worker(job):
        lin, lout, rot, src, want, interp = job
        try:
            for _ in range(5):
                out = np.full((out_h, out_w, 4), np.float32(-1.0), dtype=np.float32)
                lrp.reproject(lrp.Image(lin, in_w, in_h, 4, src), lrp.Image(lout, out_w, out_h, 4, out), 1, interp, rot)
                cases.assert_same_bits(out, want, "threaded host call")
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[0]


def test_device_entry_point_is_capturable_into_a_graph(lrp, oracle, torch_cuda):
    """After one warm-up call (which builds the output-lens tables) lrp_reproject_device
    neither allocates nor synchronises: it can be captured into a hipGraph and replayed."""
    torch = torch_cuda
    in_w, in_h, out_w, out_h = 256, 128, 192, 112
    lin = cases.lenses(lrp, in_w, in_h)["eqr_full"]
    lout = cases.lenses(lrp, out_w, out_h)["rect"]
    rot = cases.rotation(lrp, (30.0, -15.0, 5.0))
    src = cases.hash_noise(in_h, in_w, 4, seed=7)
    d_in = torch.from_numpy(src).cuda()
    d_out = torch.zeros((out_h, out_w, 4), dtype=torch.float32, device="cuda")
    im_in, im_out = lrp.Image(lin, in_w, in_h, 4, d_in), lrp.Image(lout, out_w, out_h, 4, d_out)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for interp in (0, 1, 2):
            lrp.reproject(im_in, im_out, 1, interp, rot)  # warm-up: tables cached
    side.synchronize()
    for interp in (0, 1, 2):
        want = oracle.reproject(lin, src, lout, out_w, out_h, 1, interp, rot)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            lrp.reproject(im_in, im_out, 1, interp, rot)
        d_out.zero_()
        g.replay()
        torch.cuda.synchronize()
        cases.assert_same_bits(d_out.cpu().numpy(), want, f"graph replay interp={interp}")
        # new source contents, same graph
        src2 = cases.hash_noise(in_h, in_w, 4, seed=8 + interp)
        d_in.copy_(torch.from_numpy(src2))
        g.replay()
        torch.cuda.synchronize()
        cases.assert_same_bits(d_out.cpu().numpy(), oracle.reproject(lin, src2, lout, out_w, out_h, 1, interp, rot),
                               f"graph replay 2 interp={interp}")
        d_in.copy_(torch.from_numpy(src))


def test_table_cache_release_and_rebuild(lrp, oracle, torch_cuda):
    torch = torch_cuda
    in_w, in_h, out_w, out_h = 90, 60, 70, 50
    lin = cases.lenses(lrp, in_w, in_h)["eqd180"]
    src = cases.hash_noise(in_h, in_w, 4, seed=3)
    d_in = torch.from_numpy(src).cuda()
    for round_ in range(2):
        for out_name in ("rect", "rect_tele", "eqr_full", "eqr_part"):
            lout = cases.lenses(lrp, out_w, out_h)[out_name]
            for ns in (1, 2):
                d_out = torch.empty((out_h, out_w, 4), dtype=torch.float32, device="cuda")
                lrp.reproject(lrp.Image(lin, in_w, in_h, 4, d_in), lrp.Image(lout, out_w, out_h, 4, d_out), ns, 1, None)
                torch.cuda.synchronize()
                cases.assert_same_bits(d_out.cpu().numpy(), oracle.reproject(lin, src, lout, out_w, out_h, ns, 1, None),
                                       f"{out_name} ns={ns} round {round_}")
        lrp._native.load().lrp_release_cached_tables()


def test_kernel_family_knob(lrp, torch_cuda):
    prev = lrp.debug_kernel(-1)
    assert prev in (0, 1, 2)
    assert lrp.debug_kernel(0) == prev
    assert lrp.debug_kernel(-1) == 0
    assert lrp.debug_kernel(7) == 0  # out of range: query only
    lrp.debug_kernel(prev)
    assert lrp.debug_kernel(-1) == prev


@pytest.mark.parametrize("channels,interp,deg", [(4, 2, None), (4, 2, (30.0, -15.0, 5.0)), (3, 2, (90.0, 0.0, 0.0)), (4, 1, None),
                                                 (5, 2, None), (4, 0, (0.0, 0.0, 0.0)), (2, 2, None)])
def test_batched_launch_equals_single_calls(lrp, torch_cuda, channels, interp, deg):
    """lrp_reproject_batch_device: 37 frames of one geometry (three launches: 16 + 16 + 5; the
    2-channel case goes through the per-pixel kernel frame by frame) against 37 single calls."""
    torch = torch_cuda
    in_w, in_h, out_w, out_h, n = 320, 200, 290, 210, 37
    lin = cases.lenses(lrp, in_w, in_h)["eqd180"]
    lout = cases.lenses(lrp, out_w, out_h)["rect"]
    rot = cases.rotation(lrp, deg)
    srcs = [torch.empty((in_h, in_w, channels), dtype=torch.float32, device="cuda") for _ in range(n)]
    for i, s in enumerate(srcs):
        lrp.synth_fill(s, in_w, in_h, channels, 0xBA7C0000 + i)
    single = [torch.full((out_h, out_w, channels), -1.0, dtype=torch.float32, device="cuda") for _ in range(n)]
    batched = [torch.full((out_h, out_w, channels), -2.0, dtype=torch.float32, device="cuda") for _ in range(n)]
    for s, d in zip(srcs, single):
        lrp.reproject(lrp.Image(lin, in_w, in_h, channels, s), lrp.Image(lout, out_w, out_h, channels, d), 1, interp, rot)
    lrp.reproject_batch([lrp.Image(lin, in_w, in_h, channels, s) for s in srcs],
                        [lrp.Image(lout, out_w, out_h, channels, d) for d in batched], 1, interp, rot)
    torch.cuda.synchronize()
    for i in range(n):
        assert bool(torch.equal(single[i].view(torch.int32), batched[i].view(torch.int32))), f"frame {i} differs"


@pytest.mark.parametrize("frames", ["2", "3", "16"])
@pytest.mark.parametrize("channels,in_name,out_name,deg", [
    (4, "eqd180", "rect", None),                  # mirrored in both axes
    (4, "eqr_full", "rect", (90.0, 0.0, 0.0)),    # rows only
    (3, "eqr_full", "rect", (0.0, -90.0, 0.0)),   # columns only
    (4, "eqr_full", "eqd180", (30.0, -15.0, 5.0)),  # shared rays
    (4, "rect_tele", "rect", (30.0, -15.0, 5.0)),   # plain blocks, raw taps
    (5, "eqr_part", "rect", (12.0, 7.0, 0.0)),      # RGBAZ, plain blocks
    (4, "rect", "eqr_full", None),                  # corner blocks, direct taps, alias pairs
])
def test_batched_launch_with_several_frames_per_wavefront(lrp, torch_cuda, frames, channels, in_name, out_name, deg):
    """Bicubic batches: a wavefront of the window kernel renders its strip for several consecutive frames and shares the
    coordinate math and the window plan between them (lrp_debug_set "batch_frames" forces the count; by default small images keep
    one frame per wavefront).  21 frames (so that the last group is short: 21 = 16 + 5 launches, groups of 2 / 3 / 16)
    against 21 single calls, bit for bit, in every mirror mode."""
    import os

    torch = torch_cuda
    in_w, in_h, out_w, out_h, n = 300, 180, 201, 137, 21
    lin, lout = cases.lenses(lrp, in_w, in_h)[in_name], cases.lenses(lrp, out_w, out_h)[out_name]
    rot = cases.rotation(lrp, deg)
    srcs = [torch.empty((in_h, in_w, channels), dtype=torch.float32, device="cuda") for _ in range(n)]
    for i, s in enumerate(srcs):
        lrp.synth_fill(s, in_w, in_h, channels, 0xF4A30000 + i, 4 if channels == 5 else -1)
    single = [torch.full((out_h, out_w, channels), -1.0, dtype=torch.float32, device="cuda") for _ in range(n)]
    batched = [torch.full((out_h, out_w, channels), -2.0, dtype=torch.float32, device="cuda") for _ in range(n)]
    for s, d in zip(srcs, single):
        lrp.reproject(lrp.Image(lin, in_w, in_h, channels, s), lrp.Image(lout, out_w, out_h, channels, d), 1, 2, rot, post=(2.0, 4.0))
    prev = lrp.debug_set("batch_frames", int(frames))
    try:
        lrp.reproject_batch([lrp.Image(lin, in_w, in_h, channels, s) for s in srcs],
                            [lrp.Image(lout, out_w, out_h, channels, d) for d in batched], 1, 2, rot, post=(2.0, 4.0))
        torch.cuda.synchronize()
    finally:
        lrp.debug_set("batch_frames", prev)
    for i in range(n):
        assert bool(torch.equal(single[i].view(torch.int32), batched[i].view(torch.int32))), f"frame {i} differs"


@pytest.mark.parametrize("frames", ["2", "5", "16"])
@pytest.mark.parametrize("interp", [0, 1])
@pytest.mark.parametrize("channels,in_name,out_name,deg", [(4, "eqr_full", "rect", (30.0, -15.0, 5.0)), (3, "eqd180", "rect", (0.0, 40.0, 0.0)),
                                                          (5, "eqr_part", "eqr_full", (10.0, 5.0, 20.0)), (4, "rect", "eqd120", (30.0, -15.0, 5.0))])
def test_batched_nearest_and_bilinear_with_several_frames_per_wavefront(lrp, torch_cuda, frames, interp, channels, in_name, out_name, deg):
    """Tile kernels, rotated mappings: the source coordinates of a wavefront's pixels stay in registers while it walks
    the frames of its group.  21 frames against 21 single calls."""
    import os

    torch = torch_cuda
    in_w, in_h, out_w, out_h, n = 300, 180, 201, 137, 21
    lin, lout = cases.lenses(lrp, in_w, in_h)[in_name], cases.lenses(lrp, out_w, out_h)[out_name]
    rot = cases.rotation(lrp, deg)
    srcs = [torch.empty((in_h, in_w, channels), dtype=torch.float32, device="cuda") for _ in range(n)]
    for i, s in enumerate(srcs):
        lrp.synth_fill(s, in_w, in_h, channels, 0xF4A40000 + i, 4 if channels == 5 else -1)
    single = [torch.full((out_h, out_w, channels), -1.0, dtype=torch.float32, device="cuda") for _ in range(n)]
    batched = [torch.full((out_h, out_w, channels), -2.0, dtype=torch.float32, device="cuda") for _ in range(n)]
    for s, d in zip(srcs, single):
        lrp.reproject(lrp.Image(lin, in_w, in_h, channels, s), lrp.Image(lout, out_w, out_h, channels, d), 1, interp, rot)
    prev = lrp.debug_set("batch_frames", int(frames))
    try:
        lrp.reproject_batch([lrp.Image(lin, in_w, in_h, channels, s) for s in srcs],
                            [lrp.Image(lout, out_w, out_h, channels, d) for d in batched], 1, interp, rot)
        torch.cuda.synchronize()
    finally:
        lrp.debug_set("batch_frames", prev)
    for i in range(n):
        assert bool(torch.equal(single[i].view(torch.int32), batched[i].view(torch.int32))), f"frame {i} differs"


def test_batched_launch_rejects_mixed_geometries(lrp, torch_cuda):
    torch = torch_cuda
    a = torch.zeros((64, 64, 4), dtype=torch.float32, device="cuda")
    b = torch.zeros((64, 48, 4), dtype=torch.float32, device="cuda")
    o1 = torch.zeros((32, 32, 4), dtype=torch.float32, device="cuda")
    o2 = torch.zeros((32, 32, 4), dtype=torch.float32, device="cuda")
    L = lrp.LensInfo
    with pytest.raises(Exception):
        lrp.reproject_batch([lrp.Image(L.equirectangular(), 64, 64, 4, a), lrp.Image(L.equirectangular(), 48, 64, 4, b)],
                            [lrp.Image(L.rectilinear(18.0, 36.0, 32, 32), 32, 32, 4, o1),
                             lrp.Image(L.rectilinear(18.0, 36.0, 32, 32), 32, 32, 4, o2)], 1, 2, None)


@pytest.mark.parametrize("channels,interp,ns", [(9, 2, 1), (12, 1, 2), (17, 0, 1), (16, 2, 1)])
def test_wide_texels_are_rendered_in_channel_groups(lrp, oracle, torch_cuda, channels, interp, ns):
    """The reference loop is generic in the channel count (src/reproject.cpp:50,76,134); texels wider
    than 8 floats go through the per-pixel kernel 8 channels per launch, with the fused post_process
    on channels 0-2 only."""
    torch = torch_cuda
    in_w, in_h, out_w, out_h = 97, 61, 83, 59
    lin = cases.lenses(lrp, in_w, in_h)["eqr_full"]
    lout = cases.lenses(lrp, out_w, out_h)["rect"]
    rot = cases.rotation(lrp, (30.0, -15.0, 5.0))
    src = cases.hash_noise(in_h, in_w, channels, seed=40 + channels)
    want = oracle.reproject(lin, src, lout, out_w, out_h, ns, interp, rot)
    oracle.post_process(want, 2.0, 4.0)
    d_in = torch.from_numpy(src).cuda()
    d_out = torch.full((out_h, out_w, channels), -1.0, dtype=torch.float32, device="cuda")
    lrp.reproject(lrp.Image(lin, in_w, in_h, channels, d_in), lrp.Image(lout, out_w, out_h, channels, d_out), ns, interp,
                  rot, post=(2.0, 4.0))
    torch.cuda.synchronize()
    cases.assert_same_bits(d_out.cpu().numpy(), want, f"C={channels}")


def test_table_cache_eviction_while_other_threads_launch(lrp, oracle, torch_cuda):
    """More output geometries than the 256-entry table cache holds, from 4 threads at once: an entry
    handed to a caller is pinned until its kernel is enqueued and eviction synchronises the device
    before it frees unpinned entries, so no launch ever reads a freed (or re-used) table."""
    torch = torch_cuda
    in_w, in_h = 64, 48
    lin = cases.lenses(lrp, in_w, in_h)["eqr_full"]
    src = cases.hash_noise(in_h, in_w, 4, seed=11)
    d_in = torch.from_numpy(src).cuda()
    errors = []

    def worker(t):
        try:
            stream = torch.cuda.Stream()
            for i in range(110):
                out_w, out_h = 24 + (i % 7), 16 + (i % 5)
                # a distinct output lens per (thread, i): 440 table builds, > 256 cache slots
                lout = lrp.LensInfo.rectilinear(18.0 + t + 0.01 * i, 36.0, out_w, out_h)
                rot = cases.rotation(lrp, (90.0, 0.0, 0.0)) if i % 2 else None  # pan only: a column table too
                with torch.cuda.stream(stream):
                    d_out = torch.full((out_h, out_w, 4), -1.0, dtype=torch.float32, device="cuda")
                    lrp.reproject(lrp.Image(lin, in_w, in_h, 4, d_in), lrp.Image(lout, out_w, out_h, 4, d_out), 1, 2, rot)
                stream.synchronize()
                if i % 10 == 0:
                    cases.assert_same_bits(d_out.cpu().numpy(), oracle.reproject(lin, src, lout, out_w, out_h, 1, 2, rot),
                                           f"thread {t} geometry {i}")
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    lrp._native.load().lrp_release_cached_tables()
    assert not errors, errors[0]

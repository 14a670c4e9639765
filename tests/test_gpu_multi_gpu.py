"""-m gpu: the pieces of a job that splits ONE image, or one source with several outputs, over GPUs
(SURVEY §8e "intra-image split", BASELINE configs[4]: an 8192^2 panorama -> six cubemap faces on 8 GPUs).
Rows of the reference loop are independent (src/reproject.cpp:284): a row band rendered alone must carry
the bytes of the same rows of a whole-image call, and lrp_reproject_multi — source uploaded once, copied
device to device, whole outputs dealt round-robin or band d of every output on GPU d — must give the bytes of n_out single calls.  On a
one-GPU box the device list names GPU 0 several times, which runs the same code (peer copy = device copy)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("channels,interp,out_kind,deg", [(4, 2, "rect", None), (4, 2, "rect", (30.0, -15.0, 5.0)), (3, 2, "eqd180", None),
                                                          (5, 2, "rect", (90.0, 0.0, 0.0)), (4, 1, "eqd180", (30.0, -15.0, 5.0)),
                                                          (4, 0, "eqr_full", None), (2, 2, "rect", None)])
def test_row_bands_carry_the_bytes_of_the_whole_image(lrp, oracle, torch_cuda, channels, interp, out_kind, deg):
    torch = torch_cuda
    in_w, in_h, out_w, out_h = 256, 128, 200, 150
    lin = cases.lenses(lrp, in_w, in_h)["eqr_full"]
    lout = cases.lenses(lrp, out_w, out_h)[out_kind]
    rot = cases.rotation(lrp, deg)
    src = cases.hash_noise(in_h, in_w, channels, seed=77)
    d_in = torch.from_numpy(src).cuda()
    im_in = lrp.Image(lin, in_w, in_h, channels, d_in)
    whole = torch.full((out_h, out_w, channels), -1.0, dtype=torch.float32, device="cuda")
    lrp.reproject(im_in, lrp.Image(lout, out_w, out_h, channels, whole), 1, interp, rot, post=(2.0, 4.0))
    want = oracle.reproject(lin, src, lout, out_w, out_h, 1, interp, rot)
    oracle.post_process(want, 2.0, 4.0)
    torch.cuda.synchronize()
    cases.assert_same_bits(whole.cpu().numpy(), want, "whole image")
    banded = torch.full((out_h, out_w, channels), -1.0, dtype=torch.float32, device="cuda")
    im_out = lrp.Image(lout, out_w, out_h, channels, banded)
    lrp.reproject_rows(im_in, im_out, 1, interp, 37, 50, rot, post=(2.0, 4.0))
    torch.cuda.synchronize()
    got = banded.cpu().numpy()
    assert (got[:37] == -1.0).all() and (got[87:] == -1.0).all(), "a band wrote outside its rows"
    cases.assert_same_bits(got[37:87], want[37:87], "band 37..87")
    for first, count in ((0, 37), (87, 1), (88, 62), (150, 0)):
        lrp.reproject_rows(im_in, im_out, 1, interp, first, count, rot, post=(2.0, 4.0))
    torch.cuda.synchronize()
    cases.assert_same_bits(banded.cpu().numpy(), want, "all bands")
    with pytest.raises(Exception):
        lrp.reproject_rows(im_in, im_out, 1, interp, 140, 20, rot)


@pytest.mark.parametrize("devices", [(0,), (0, 0, 0), (0,) * 8, "all"])
def test_cubemap_job_over_a_device_list(lrp, oracle, torch_cuda, devices):
    """Six 90-degree faces from one panorama (the reference: six invocations with --rotation), bicubic, RGB.  Up to six
    participants get whole faces (round-robin), eight participants get row bands of every face."""
    if devices == "all":
        devices = tuple(range(torch_cuda.cuda.device_count())) * 2  # every visible GPU, twice
    in_w, in_h, face = 768, 384, 160
    lin = lrp.LensInfo.equirectangular()
    lout = lrp.LensInfo.rectilinear(18.0, 36.0, face, face)
    src = cases.hash_noise(in_h, in_w, 3, seed=5)
    degs = [(0, 0, 0), (90, 0, 0), (180, 0, 0), (270, 0, 0), (0, 90, 0), (0, -90, 0)]
    rots = np.stack([cases.rotation(lrp, d) for d in degs])
    outs = [np.full((face, face, 3), -1.0, dtype=np.float32) for _ in degs]
    lrp.reproject_multi_gpu(lrp.Image(lin, in_w, in_h, 3, src), [lrp.Image(lout, face, face, 3, o) for o in outs], 1, 2, rots,
                            devices=devices)
    for d, o, r in zip(degs, outs, rots):
        cases.assert_same_bits(o, oracle.reproject(lin, src, lout, face, face, 1, 2, r), f"face {d} on devices {devices}")


def test_multi_gpu_jobs_on_disjoint_and_shared_device_lists_run_concurrently(lrp, oracle, torch_cuda):
    """lrp_reproject_multi locks its participants ((device, occurrence) pairs, taken in one global order), not the
    library: four host threads run cubemap jobs at once — on a one-GPU box they all name GPU 0 (they queue up on its
    participants and on the second occurrence), on a larger box the lists differ — and every job gets the oracle's bytes."""
    import threading

    n_dev = torch_cuda.cuda.device_count()
    in_w, in_h, face = 512, 256, 96
    lin = lrp.LensInfo.equirectangular()
    lout = lrp.LensInfo.rectilinear(18.0, 36.0, face, face)
    degs = [(0, 0, 0), (90, 0, 0), (0, 90, 0)]
    rots = np.stack([cases.rotation(lrp, d) for d in degs])
    lists = [(0,), (n_dev - 1, 0), (0, 0), tuple(range(n_dev))]
    srcs = [cases.hash_noise(in_h, in_w, 4, seed=100 + t) for t in range(len(lists))]
    wants = [[oracle.reproject(lin, s, lout, face, face, 1, 2, r) for r in rots] for s in srcs]
    errors = []

    def job(t):
        try:
            for _ in range(3):
                outs = [np.full((face, face, 4), -1.0, dtype=np.float32) for _ in degs]
                lrp.reproject_multi_gpu(lrp.Image(lin, in_w, in_h, 4, srcs[t]), [lrp.Image(lout, face, face, 4, o) for o in outs], 1, 2, rots,
                                        devices=lists[t])
                for o, w in zip(outs, wants[t]):
                    cases.assert_same_bits(o, w, f"job {t} on devices {lists[t]}")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=job, args=(t,)) for t in range(len(lists))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors

"""GPU parity of the image front-end: mars_yolo_letterbox / mars_hip_preprocess against the CPU
restatement of the reference's load_image() and the golden vectors the reference itself produced.  Bit-exact."""
import json
import os

import numpy as np
import pytest

import cases
import marsfile

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "golden.json")))


@pytest.mark.parametrize("case", cases.LETTERBOX_CASES, ids=lambda c: c[0])
def test_letterbox_bit_exact(gpu, orc, case):
    img = cases.letterbox_image(case)
    got = gpu.letterbox(img, case[3], case[4], case[5])
    assert cases.digest(got) == GOLD["letterbox"][case[0]]      # what the reference's load_image() produced
    want = orc.letterbox(img, case[3], case[4], case[5])
    assert np.array_equal(got, want), int((got != want).sum())


def test_letterbox_sweep(gpu, orc):
    """random geometries in both directions and layouts; the table cache is rebuilt for every new geometry"""
    rng = np.random.default_rng(9)
    for i in range(25):
        w, h = int(rng.integers(5, 400)), int(rng.integers(5, 400))
        tw, th = int(rng.integers(8, 300)), int(rng.integers(8, 300))
        s = min(tw / w, th / h)
        if min(int(w * s), int(h * s)) < 1:
            continue
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = gpu.letterbox(img, tw, th, i & 1)
        want = orc.letterbox(img, tw, th, i & 1)
        assert np.array_equal(got, want), (w, h, tw, th, i & 1, int((got != want).sum()))


def test_letterbox_extreme_ratio_uses_the_direct_kernel(gpu, orc):
    """a 40x reduction: the source region of a 16 x 16 output tile no longer fits in LDS, so the launcher falls back
    to the direct (one thread per pixel, gathers from HBM) kernel -- same bytes"""
    rng = np.random.default_rng(77)
    img = rng.integers(0, 256, (1500, 2000, 3), dtype=np.uint8)
    got = gpu.letterbox(img, 50, 50, 1)
    want = orc.letterbox(img, 50, 50, 1)
    assert np.array_equal(got, want), int((got != want).sum())


def test_preprocess_into_model_input(gpu, orc):
    """camera batch: RGB frames -> letterboxed int8 frames of the graph input in HBM -> run; the input tensor and the
    graph outputs equal the oracle's for every frame (NHWC yolov5 twin)"""
    d = gpu.synth_model(width_x16=4, input_hw=160, seed=41)
    hdr, tensors, _ = marsfile.parse(d)
    tin = hdr["inputs"][0]
    B = 3
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, (B, 90, 120, 3), dtype=np.uint8)
    m = gpu.Model(d, batch=B)
    m.preprocess(frames[:2], first_frame=0)
    m.preprocess(frames[2:], first_frame=2)
    m.run_device()
    m.download()
    for f in range(B):
        x = orc.letterbox(frames[f], 160, 160, 1)
        assert np.array_equal(m.read_tensor(tin, frame=f)[:x.size].view(np.int8), x)
        g = orc.Graph(d)
        g.set_input(0, x.tobytes())
        assert g.run() == 0
        for oi, ti in enumerate(hdr["outputs"]):
            assert np.array_equal(m.output_view(oi)[f], g.tensor(ti))
    with pytest.raises(gpu.MarsError):
        m.preprocess(frames, first_frame=1)  # frames beyond the batch
    m.close()


def test_camera_pipe_equals_preprocess_run_detect(gpu):
    """mars_hip_pipe_* in camera mode (RGB frames up, front-end on the device, graph, tail, detections back; three batches in
    flight) returns, batch for batch, what mars_hip_preprocess + run + detect give for the same frames"""
    import marsfile
    from conftest import lcg_frame
    d = gpu.synth_model(width_x16=4, input_hw=128, seed=31)
    hdr, tensors, _ = marsfile.parse(d)
    B, cw, ch = 3, 200, 152
    outputs = tuple(range(len(hdr["outputs"])))
    frames = [[lcg_frame(0xCA300 + 10 * k + f, cw * ch * 3).reshape(ch, cw, 3) for f in range(B)] for k in range(5)]
    m = gpu.Model(d, batch=B)
    want = []
    for k in range(5):
        m.preprocess(np.stack(frames[k]))
        m.run_device()
        want.append([x.tobytes() for x in m.detect(outputs=outputs, thresh=0.45)])
    m.pipe_open(download_outputs=False, detect=True, det_outputs=outputs, thresh=0.45, camera=(cw, ch))
    got = []
    for k in range(5):
        v = m.pipe_input_view(0)
        assert v.shape == (B, cw * ch * 3)
        for f in range(B):
            v[f] = frames[k][f].reshape(-1)
        m.pipe_submit()
        if k >= 2:
            got.append([x.tobytes() for x in m.pipe_wait()[1]])
            assert gpu.lib().mars_hip_pipe_camera_ms(m.p) > 0
    for _ in range(2):
        got.append([x.tobytes() for x in m.pipe_wait()[1]])
    m.pipe_close()
    m.close()
    assert got == want and sum(len(b) for w_ in want for b in w_) > 0


@pytest.mark.parametrize("form", [0, 1, 2], ids=["strips", "tiles", "per_pixel"])
def test_letterbox_kernel_forms_write_the_same_bytes(gpu, orc, form, monkeypatch):
    """round 6: the strip kernel (a workgroup streams the source rows of 8 output rows; the default wherever its tables fit LDS) against
    the 16 x 16-tile kernel and the one-thread-per-pixel kernel it replaced as the default (MARS_HIP_LETTERBOX_FORM forces them): the
    camera geometry of bench.py (1280 x 720 -> 640 x 640, bands above and below), a growing axis (Catmull-Rom), ragged row ends
    (w * 3 not a multiple of 16), bands left and right, one and four columns per thread, NHWC and planar output -- all against the oracle."""
    monkeypatch.setenv("MARS_HIP_LETTERBOX_FORM", str(form))
    rng = np.random.default_rng(606)
    for (w, h, tw, th, nhwc) in [(1280, 720, 640, 640, 1), (1280, 720, 640, 640, 0), (97, 61, 224, 160, 1), (333, 500, 320, 320, 0),
                                 (1919, 1080, 1024, 576, 1), (50, 41, 200, 216, 1), (640, 480, 640, 640, 1)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = gpu.letterbox(img, tw, th, nhwc)
        want = orc.letterbox(img, tw, th, nhwc)
        assert np.array_equal(got, want), (w, h, tw, th, nhwc, form, int((got != want).sum()))


def test_letterbox_strips_over_a_batch(gpu, orc):
    """the strip kernel over several frames at once (mars_hip_preprocess: grid = strips x frames), into an NHWC and into an NCHW-tagged
    graph input (planar output), frames at odd byte offsets from each other (w * h * 3 odd)"""
    B, w, h = 5, 211, 157
    rng = np.random.default_rng(17)
    frames = rng.integers(0, 256, (B, h, w, 3), dtype=np.uint8)
    for nchw in (False, True):
        d = gpu.synth_model(width_x16=4, input_hw=128, seed=5, nchw_int8=nchw)
        hdr, tensors, _ = marsfile.parse(d)
        tin = hdr["inputs"][0]
        m = gpu.Model(d, batch=B)
        m.preprocess(frames)
        for f in range(B):
            x = orc.letterbox(frames[f], 128, 128, 0 if nchw else 1)
            assert np.array_equal(m.read_tensor(tin, frame=f)[:x.size].view(np.int8), x), (nchw, f)
        m.close()

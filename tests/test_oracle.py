"""The CPU restatement (oracle/restate) against (a) the reference compiled in place
(oracle/_ref; only where it was built) and (b) the committed golden vectors the
reference produced (everywhere).  No GPU involved."""
import json
import os

import numpy as np
import pytest

import cases
import marsfile
from conftest import lcg_frame, pattern_input

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "golden.json")))


def model_bytes(name):
    return open(os.path.join(HERE, "golden", "models", name + ".mars"), "rb").read()


def model_input(tin, tag):
    nb = marsfile.tensor_nbytes(tin)
    if tag == "pattern":
        return pattern_input(tin["dtype"], nb)
    if tin["dtype"] == 0:
        return cases.f32(0x5EED0000, nb // 4, 0.0, 1.0).view(np.uint8)
    return lcg_frame(0x5EED0000, nb)


@pytest.mark.parametrize("name", cases.SHIPPED)
@pytest.mark.parametrize("tag", ["pattern", "lcg"])
def test_restatement_matches_golden_models(orc, name, tag):
    """every activation tensor of every shipped model, as the reference left it"""
    if name == "yolov5n_int8" and tag == "lcg":
        pytest.skip("one full-size run is enough for the CPU suite")
    d = model_bytes(name)
    hdr, tensors, _ = marsfile.parse(d)
    g = orc.Graph(d)
    g.set_input(0, model_input(tensors[hdr["inputs"][0]], tag).tobytes())
    want = GOLD["models"][name][tag]
    assert g.run() == want["rc"]
    out = g.tensor(hdr["outputs"][0])
    assert [int(v) for v in out[:32]] == want["out_head"]
    assert cases.digest(out) == want["out"]
    for ti, dg in want["tensors"].items():
        assert cases.digest(g.tensor(int(ti))) == dg, "tensor %s" % ti
    if "o1_out" in want:  # hazard-free graphs: the reference's public API gives the same bytes
        assert want["o1_out"] == want["out"]


@pytest.mark.parametrize("case", cases.CONV_I8_CASES, ids=lambda c: c[0])
def test_conv_i8_golden(orc, case):
    assert cases.digest(cases.conv_i8_call(orc.conv2d_int8, case)) == GOLD["conv_i8"][case[0]]


@pytest.mark.parametrize("fam", sorted(cases.CONV_I8_FAMILIES))
def test_conv_i8_family_golden(orc, fam):
    """the shape families the GPU tests force launch forms with: the restatement reproduces the digests the reference's own
    conv2d_int8_nhwc_mxu produced for them (so on the GPU box kernel, restatement and reference are three witnesses)"""
    seed, cs = cases.family_cases(fam)
    for c in cs:
        assert cases.digest(cases.conv_i8_call(orc.conv2d_int8, c, seed)) == GOLD["conv_i8_family"][c[0]], c[0]


@pytest.mark.parametrize("case", cases.CONV_F32_CASES, ids=lambda c: c[0])
def test_conv_f32_golden(orc, case):
    assert cases.digest(cases.conv_f32_call(orc.conv2d_f32, case)) == GOLD["conv_f32"][case[0]]


@pytest.mark.parametrize("case", cases.YOLO_CASES, ids=lambda c: c[0])
def test_yolo_golden(orc, case):
    pred, npred, scale = cases.yolo_pred(case)
    raw = orc.parse_output(pred, npred, scale)
    kept = orc.nms(raw, 0.45)
    want = GOLD["yolo"][case[0]]
    assert (len(raw), len(kept)) == (want["raw"], want["kept"])
    assert cases.digest(raw) == want["raw_digest"] and cases.digest(kept) == want["kept_digest"]


def test_trunc_x86(orc):
    """SURVEY.md appendix B.2 probe values"""
    assert orc.trunc_x86(0.5 + 0.5) == 1 and orc.trunc_x86(-0.5 - 0.5) == -1
    assert orc.trunc_x86(4e10) == -2147483648 and orc.trunc_x86(-4e10) == -2147483648
    assert orc.trunc_x86(float("nan")) == -2147483648
    assert orc.trunc_x86(2147483520.0) == 2147483520 and orc.trunc_x86(-2147483648.0) == -2147483648


# ---------------------------------------------------------------- live reference (build container only)
@pytest.mark.parametrize("name,kw", cases.SYNTH, ids=lambda v: v if isinstance(v, str) else "")
def test_restatement_vs_reference_synthetic(orc, ref, marsrt, name, kw):
    d = marsrt.synth_model(**kw)
    hdr, tensors, _ = marsfile.parse(d)
    tin = tensors[hdr["inputs"][0]]
    x = model_input(tin, "lcg")
    g, m = orc.Graph(d, slack_mult=2), ref.O2Model(d, slack_mult=2)
    g.set_input(0, x.tobytes())
    m.set_input(0, x.tobytes())
    assert g.run() == 0 and m.run() == 0
    for ti in range(len(tensors)):
        if tensors[ti]["size"] == 0:
            assert np.array_equal(g.tensor(ti), m.tensor(ti)), "tensor %d" % ti
    # the synthetic graphs are hazard- and overrun-free: outputs are not degenerate
    out = g.tensor(hdr["outputs"][0])
    assert len(np.unique(out)) > 16


@pytest.mark.parametrize("seed", [2, 3])
@pytest.mark.parametrize("case", cases.CONV_I8_CASES, ids=lambda c: c[0])
def test_conv_i8_vs_reference(orc, ref, case, seed):
    a = cases.conv_i8_call(orc.conv2d_int8, case, seed)
    b = cases.conv_i8_call(ref.conv2d_int8, case, seed)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("fam", ["split", "patch", "stem"])
def test_conv_f32_family_golden(orc, fam):
    """the float shape lists of the GPU tests (tests/f32shapes.py: one convolution + SIGMOID / MUL (+ ADD) graphs): the restatement
    reproduces the digests the REFERENCE left for every input frame (golden.json "conv_f32_family", made by make_golden.py)"""
    import f32shapes
    shapes, build = {f: (s_, b) for f, s_, b in f32shapes.FAMILIES}[fam]
    for shape in shapes:
        got = f32shapes.reference_digests(build(shape), orc.Graph, cases.digest)
        assert got == GOLD["conv_f32_family"][f32shapes.shape_id(fam, shape)], shape


@pytest.mark.parametrize("fam", sorted(cases.CONV_I8_FAMILIES))
def test_conv_i8_family_vs_reference(orc, ref, fam):
    """... and element-wise against the live reference, at a second seed the goldens do not hold"""
    seed, cs = cases.family_cases(fam)
    for c in cs:
        a = cases.conv_i8_call(orc.conv2d_int8, c, seed + 100)
        b = cases.conv_i8_call(ref.conv2d_int8, c, seed + 100)
        assert np.array_equal(a, b), c[0]


@pytest.mark.parametrize("case", cases.CONV_F32_CASES, ids=lambda c: c[0])
def test_conv_f32_vs_reference(orc, ref, case):
    a = cases.conv_f32_call(orc.conv2d_f32, case, 5)
    b = cases.conv_f32_call(ref.conv2d_f32, case, 5)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))  # bit-identical, not just close


def _layer_graph(kind):
    """one graph per non-conv layer kind, int8 and f32, incl. the no-op and failing kinds"""
    G = marsfile.Graph()
    rng_w = cases.f32(77, 8, 0.5, 1.5)
    if kind in ("sigmoid", "relu", "relu6", "leaky"):
        a = G.tensor([1, 5, 7, 8], scale=0.05)
        o = G.tensor([1, 5, 7, 8], scale=1 / 127.0 if kind == "sigmoid" else 0.05)
        G.layer(dict(sigmoid=marsfile.SIGMOID, relu=marsfile.RELU, relu6=marsfile.RELU6, leaky=marsfile.LEAKY)[kind], [a], [o])
        return G.serialise([a], [o])
    if kind in ("mul", "add"):
        a = G.tensor([1, 5, 7, 8], scale=0.03)
        c = G.tensor([1, 5, 7, 8], scale=0.0079)
        s = G.tensor([1, 5, 7, 8], scale=0.031)  # second operand produced by a sigmoid so it is an activation too
        o = G.tensor([1, 5, 7, 8], scale=0.02)
        G.layer(marsfile.SIGMOID, [a], [c])
        G.layer(marsfile.MUL if kind == "mul" else marsfile.ADD, [a, c], [o])
        return G.serialise([a], [o])
    if kind == "mul_const":  # second operand lives in the weight blob
        a = G.tensor([1, 4, 4, 8], scale=0.03)
        k = G.tensor([1, 4, 4, 8], scale=0.01, data=cases.i8(5, 128))
        o = G.tensor([1, 4, 4, 8], scale=0.002)
        G.layer(marsfile.MUL, [a, k], [o])
        return G.serialise([a], [o])
    if kind == "maxpool":
        a = G.tensor([1, 9, 11, 8], scale=0.05)
        o = G.tensor([1, 9, 11, 8], scale=0.05)
        G.pool(a, o, (5, 5), (1, 1))
        return G.serialise([a], [o])
    if kind == "maxpool_s2_c3":
        a = G.tensor([1, 9, 10, 3], scale=0.05)
        o = G.tensor([1, 5, 5, 3], scale=0.05)
        G.pool(a, o, (2, 2), (2, 2))
        return G.serialise([a], [o])
    if kind == "concat":
        a = G.tensor([1, 6, 5, 16], scale=0.05)
        b = G.tensor([1, 6, 5, 16], scale=0.05)
        c = G.tensor([1, 6, 5, 5], scale=0.05)
        o = G.tensor([1, 6, 5, 37], scale=0.05)
        G.layer(marsfile.RELU, [a], [b])
        G.pool(a, c, (1, 1), (1, 1))  # reads channels with ITS OWN (input) channel count
        G.concat([a, b, c], o)
        return G.serialise([a], [o])
    if kind == "upsample":
        a = G.tensor([1, 4, 5, 16], scale=0.05)
        o = G.tensor([1, 8, 10, 16], scale=0.05)
        G.upsample(a, o, 2, 2)
        return G.serialise([a], [o])
    if kind == "upsample_auto_c3":
        a = G.tensor([1, 3, 4, 3], scale=0.05)
        o = G.tensor([1, 10, 9, 3], scale=0.05)
        G.upsample(a, o, 0, 0)  # scale taken from out/in, rows clamped
        return G.serialise([a], [o])
    if kind == "batchnorm":
        a = G.tensor([1, 8, 5, 6], fmt=marsfile.NCHW, scale=0.05)
        s = G.tensor([8], dtype=marsfile.F32, fmt=marsfile.D1, data=rng_w)
        b = G.tensor([8], dtype=marsfile.F32, fmt=marsfile.D1, data=cases.f32(78, 8, -1, 1))
        o = G.tensor([1, 8, 5, 6], fmt=marsfile.NCHW, scale=0.04)
        G.layer(marsfile.BATCHNORM, [a, s, b], [o])
        return G.serialise([a], [o])
    if kind == "f32_chain":
        a = G.tensor([1, 3, 6, 6], dtype=marsfile.F32, fmt=marsfile.NCHW)
        c = G.tensor([1, 3, 6, 6], dtype=marsfile.F32, fmt=marsfile.NCHW)
        e = G.tensor([1, 3, 6, 6], dtype=marsfile.F32, fmt=marsfile.NCHW)
        f = G.tensor([1, 3, 6, 6], dtype=marsfile.F32, fmt=marsfile.NCHW)
        s = G.tensor([3], dtype=marsfile.F32, fmt=marsfile.D1, data=cases.f32(79, 3, 0.5, 2))
        o = G.tensor([1, 3, 6, 6], dtype=marsfile.F32, fmt=marsfile.NCHW)
        G.layer(marsfile.LEAKY, [a], [c])
        G.layer(marsfile.ADD, [a, c], [e])
        G.layer(marsfile.MUL, [e, c], [f])
        G.layer(marsfile.BATCHNORM, [f, s, marsfile.NONE], [o])
        return G.serialise([a], [o])
    if kind == "noops":
        a = G.tensor([1, 4, 4, 8], scale=0.05)
        o = G.tensor([1, 4, 4, 8], scale=0.05)
        for t in (marsfile.DWCONV, marsfile.AVGPOOL, marsfile.SILU, marsfile.RESHAPE, marsfile.TRANSPOSE, marsfile.SOFTMAX):
            G.layer(t, [a], [o])
        return G.serialise([a], [o])
    if kind in ("fc", "gap", "unknown"):
        a = G.tensor([1, 4, 4, 8], scale=0.05)
        b = G.tensor([1, 4, 4, 8], scale=0.05)
        o = G.tensor([1, 4, 4, 8], scale=0.05)
        G.layer(marsfile.RELU, [a], [b])
        G.layer(dict(fc=marsfile.FC, gap=marsfile.GAP, unknown=77)[kind], [b], [o])
        return G.serialise([a], [o])
    if kind == "missing_tensor":
        a = G.tensor([1, 4, 4, 8], scale=0.05)
        o = G.tensor([1, 4, 4, 8], scale=0.05)
        G.layer(marsfile.SIGMOID, [99], [o])
        return G.serialise([a], [o])
    if kind == "conv_relu_valid":
        a = G.tensor([1, 9, 9, 16], scale=0.02)
        w = G.tensor([24, 3, 3, 16], fmt=marsfile.OHWI, scale=0.004, data=cases.i8(81, 24 * 9 * 16))
        o = G.tensor([1, 9, 9, 24], scale=0.05)  # shape assumes padding, EXPLICIT runs unpadded
        G.conv(a, o, w, marsfile.NONE, pad=marsfile.PAD_EXPLICIT, act=1)
        return G.serialise([a], [o])
    raise KeyError(kind)


LAYER_KINDS = ["sigmoid", "relu", "relu6", "leaky", "mul", "add", "mul_const", "maxpool", "maxpool_s2_c3", "concat",
               "upsample", "upsample_auto_c3", "batchnorm", "f32_chain", "noops", "fc", "gap", "unknown",
               "missing_tensor", "conv_relu_valid"]


@pytest.mark.parametrize("kind", LAYER_KINDS)
def test_layers_vs_reference(orc, ref, kind):
    d = _layer_graph(kind)
    hdr, tensors, _ = marsfile.parse(d)
    tin = tensors[hdr["inputs"][0]]
    x = model_input(tin, "lcg")
    g, m = orc.Graph(d), ref.O2Model(d)
    g.set_input(0, x.tobytes())
    m.set_input(0, x.tobytes())
    assert g.run() == m.run()
    for ti in range(len(tensors)):
        if tensors[ti]["size"] == 0:
            ext = marsfile.tensor_nbytes(tensors[ti]) + 64
            assert np.array_equal(g.tensor(ti, extent=ext), m.tensor(ti, extent=ext)), "tensor %d" % ti


@pytest.mark.parametrize("case", cases.YOLO_CASES, ids=lambda c: c[0])
def test_yolo_vs_reference(orc, ref, case):
    pred, npred, scale = cases.yolo_pred(case)
    a, b = orc.parse_output(pred, npred, scale), ref.parse_output(pred, npred, scale)
    assert a.tobytes() == b.tobytes()
    assert orc.nms(a, 0.45).tobytes() == ref.nms(b, 0.45).tobytes()
    assert orc.nms(a, 0.1).tobytes() == ref.nms(b, 0.1).tobytes()


def test_run_frames_threads(orc, marsrt):
    """frames are independent: 1 thread == 3 threads == one graph per frame"""
    d = marsrt.synth_model(tiny=True, input_hw=24, seed=9)
    hdr, tensors, _ = marsfile.parse(d)
    nb, ob = marsfile.tensor_nbytes(tensors[hdr["inputs"][0]]), marsfile.tensor_nbytes(tensors[hdr["outputs"][0]])
    x = np.stack([lcg_frame(0x5EED0000 + f, nb) for f in range(5)])
    a = orc.run_frames(d, x, ob, nthreads=1)
    b = orc.run_frames(d, x, ob, nthreads=3)
    assert np.array_equal(a, b)
    g = orc.Graph(d)
    g.set_input(0, x[3].tobytes())
    assert g.run() == 0 and np.array_equal(g.tensor(hdr["outputs"][0]), a[3])


# ---------------------------------------------------------------- image front-end
@pytest.mark.parametrize("case", cases.LETTERBOX_CASES, ids=lambda c: c[0])
def test_letterbox_vs_golden(orc, case):
    """the restated stb_image_resize + letterbox (oracle/restate/orc_resize.c) against what the reference's own
    load_image() produced (golden.json, generated through oracle/_ref)"""
    out = orc.letterbox(cases.letterbox_image(case), case[3], case[4], case[5])
    assert cases.digest(out) == GOLD["letterbox"][case[0]]
    assert (out == -17).any() or case[1] * case[4] == case[2] * case[3]  # padding present unless aspect ratios match


def test_letterbox_vs_reference_sweep(orc, ref):
    """many geometries (growing, shrinking, identical, extreme aspect ratios, odd sizes), both layouts"""
    rng = np.random.default_rng(5)
    for i in range(40):
        w, h = int(rng.integers(5, 300)), int(rng.integers(5, 300))
        tw, th = int(rng.integers(8, 200)), int(rng.integers(8, 200))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if min(int(w * min(tw / w, th / h)), int(h * min(tw / w, th / h))) < 1:
            continue
        nhwc = i & 1
        a = ref.load_image(img, tw, th, nhwc)
        b = orc.letterbox(img, tw, th, nhwc)
        assert np.array_equal(a, b), (w, h, tw, th, nhwc, int((a != b).sum()))

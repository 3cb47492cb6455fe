"""ONNX -> .mars compile step (include/mars_compile.h).

PARITY UNPINNED against the reference's Rust compiler (it cannot run in this image and the reference ships no ONNX / .mars
pair made by it).  What these tests pin instead: the file layout the runtime shares with the reference (include/mars.h),
the operator table (mars-compiler/src/main.rs:76-103), the quantisation arithmetic restated in numpy
(main.rs:663-676), the scale rules (:849-874, :991-998, :1160-1175, :1240-1252, :312-405), and -- in the GPU test --
that a compiled file runs bit-identically on the oracle and on the device.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

import marsfile
import onnxmini as ox

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_ONNX = "/root/reference/mgk-decompiler/yolov5s_t41.onnx"
NONE = 0xFFFFFFFF
F32 = np.float32


def quantise_ref(w):
    """main.rs:663-676: scale = max|w| / 127 (1 when all zero); q = clamp(round_half_away(w / scale), -127, 127)."""
    w = np.asarray(w, np.float32).ravel()
    max_abs = F32(np.max(np.abs(w))) if w.size else F32(0)
    scale = F32(max_abs / F32(127.0)) if max_abs > 0 else F32(1.0)
    r = (w / scale).astype(np.float32).astype(np.float64)  # the division is an f32 one; the rounding below is then exact
    q = np.trunc(r + np.copysign(0.5, r))
    return np.clip(q, -127, 127).astype(np.int8), scale


def words(layer, n):
    return struct.unpack_from("<%dI" % n, layer["params"], 0)


def blob(file_bytes, hdr, t):
    return file_bytes[hdr["woff"] + t["off"]: hdr["woff"] + t["off"] + t["size"]]


def names(file_bytes, hdr):
    return [file_bytes[76 + 124 * i + 4: 76 + 124 * i + 64].split(b"\0")[0].decode() for i in range(hdr["tensors"])]


def small_graph(rng, with_value_info, bias_bits=None):
    """conv3x3 s2 (+bias) -> SiLU as Sigmoid + Mul -> conv1x1 -> MaxPool 2x2 -> Resize x2 -> Concat(C) -> Add"""
    w0 = rng.standard_normal((8, 3, 3, 3)).astype(np.float32)
    b0 = (rng.standard_normal(8).astype(np.float32) if bias_bits is None else np.asarray(bias_bits, np.int32).view(np.float32))
    w1 = rng.standard_normal((16, 8, 1, 1)).astype(np.float32)
    scales = np.array([1, 1, 2, 2], np.float32)
    nodes = [
        ox.node("Conv", ["images", "w0", "b0"], ["c0"], strides=[2, 2], pads=[1, 1, 1, 1], kernel_shape=[3, 3], dilations=[1, 1], group=1),
        ox.node("Sigmoid", ["c0"], ["s0"]),
        ox.node("Mul", ["c0", "s0"], ["m0"]),
        ox.node("Conv", ["m0", "w1"], ["c1"], strides=[1, 1], pads=[0, 0, 0, 0]),
        ox.node("MaxPool", ["c1"], ["p1"], kernel_shape=[2, 2], strides=[2, 2]),
        ox.node("Resize", ["p1", "", "scales"], ["u1"], mode="nearest"),
        ox.node("Concat", ["u1", "c1"], ["cat"], axis=1),
        ox.node("Add", ["cat", "cat"], ["out"]),
    ]
    inits = [ox.tensor("w0", w0), ox.tensor("b0", b0), ox.tensor("w1", w1), ox.tensor("scales", scales)]
    vi = []
    if with_value_info:
        vi = [ox.value_info(n, d) for n, d in (("c0", [1, 8, 8, 8]), ("s0", [1, 8, 8, 8]), ("m0", [1, 8, 8, 8]), ("c1", [1, 16, 8, 8]),
                                               ("p1", [1, 16, 4, 4]), ("u1", [1, 16, 8, 8]), ("cat", [1, 32, 8, 8]))]
    m = ox.model(nodes, inits, [ox.value_info("images", [1, 3, 16, 16])], [ox.value_info("out", [1, 32, 8, 8])], vi)
    return m, dict(w0=w0, b0=b0, w1=w1)


def test_layout_weights_and_scales_nchw(marsrt):
    """no value_info: every shape comes from the per-op formulas; int8, NCHW / OIHW"""
    onnx, p = small_graph(np.random.default_rng(3), with_value_info=False)
    d = marsrt.compile_onnx(onnx)
    hdr, T, L = marsfile.parse(d)
    assert d[:4] == b"MARS" and struct.unpack_from("<HHI", d, 4) == (1, 0, 0)
    assert hdr["tensors"] == 12 and hdr["layers"] == 8
    assert hdr["woff"] == 76 + 124 * 12 + 112 * 8 and len(d) == hdr["woff"] + hdr["wsz"]  # no padding before the blob
    assert names(d, hdr) == ["images", "w0", "b0", "c0", "s0", "m0", "w1", "c1", "p1", "u1", "cat", "out"]
    assert [t["id"] for t in T] == list(range(12))
    assert hdr["inputs"] == (0,) and hdr["outputs"] == (11,)
    assert struct.unpack_from("<4I", d, 44) == (0, NONE, NONE, NONE) and struct.unpack_from("<4I", d, 60) == (11, NONE, NONE, NONE)
    # shapes
    shapes = {n: t["shape"] for n, t in zip(names(d, hdr), T)}
    assert shapes["images"] == (1, 3, 16, 16) and shapes["c0"] == (1, 8, 8, 8) and shapes["s0"] == (1, 8, 8, 8)
    assert shapes["c1"] == (1, 16, 8, 8) and shapes["p1"] == (1, 16, 4, 4) and shapes["u1"] == (1, 16, 8, 8)
    assert shapes["cat"] == (1, 32, 8, 8) and shapes["out"] == (1, 32, 8, 8)
    assert shapes["w0"] == (8, 3, 3, 3) and shapes["b0"] == (8,) and shapes["w1"] == (16, 8, 1, 1)
    # dtypes / formats: features int8 NCHW, weights int8 OIHW, bias float32 with the default NHWC tag
    for n, t in zip(names(d, hdr), T):
        if n in ("w0", "w1"):
            assert (t["dtype"], t["fmt"]) == (marsfile.I8, marsfile.OIHW)
        elif n == "b0":
            assert (t["dtype"], t["fmt"]) == (marsfile.F32, marsfile.NHWC)
        else:
            assert (t["dtype"], t["fmt"]) == (marsfile.I8, marsfile.NCHW) and t["size"] == 0
    # weights: max-abs / 127, halves away from zero, OIHW kept; bias bytes verbatim; 4-byte alignment in the blob
    q0, s0 = quantise_ref(p["w0"])
    q1, s1 = quantise_ref(p["w1"])
    assert blob(d, hdr, T[1]) == q0.tobytes() and T[1]["scale"] == s0
    assert blob(d, hdr, T[6]) == q1.tobytes() and T[6]["scale"] == s1
    assert blob(d, hdr, T[2]) == p["b0"].tobytes()
    assert (T[1]["off"], T[1]["size"]) == (0, 216) and (T[2]["off"], T[2]["size"]) == (216, 32) and (T[6]["off"], T[6]["size"]) == (248, 128)
    assert hdr["wsz"] == 376
    # scales: input 1/255; conv = in * w * fan_in (f32, in that order); sigmoid 1/127; mul = min of two live scales;
    # pool / resize keep; concat = max; add = max
    s_in = F32(1.0) / F32(255.0)
    s_c0 = F32(F32(s_in * s0) * F32(27.0))
    assert T[0]["scale"] == s_in and T[3]["scale"] == s_c0 and T[4]["scale"] == F32(1.0) / F32(127.0)
    s_m0 = min(s_c0, F32(1.0) / F32(127.0))
    assert T[5]["scale"] == s_m0
    s_c1 = F32(F32(s_m0 * s1) * F32(8.0))
    assert T[7]["scale"] == s_c1 and T[8]["scale"] == s_c1 and T[9]["scale"] == s_c1 and T[10]["scale"] == s_c1 and T[11]["scale"] == s_c1
    # layers
    assert [l["type"] for l in L] == [marsfile.CONV2D, marsfile.SIGMOID, marsfile.MUL, marsfile.CONV2D, marsfile.MAXPOOL,
                                      marsfile.UPSAMPLE, marsfile.CONCAT, marsfile.ADD]
    assert [l["id"] for l in L] == list(range(8))
    assert L[0]["ins"] == (0,) and L[0]["outs"] == (3,)
    #                       kh kw sh sw dh dw pad       t  b  l  r  g  act w  b
    assert words(L[0], 15) == (3, 3, 2, 2, 1, 1, marsfile.PAD_EXPLICIT, 1, 1, 1, 1, 1, 0, 1, 2)
    assert words(L[3], 15) == (1, 1, 1, 1, 1, 1, marsfile.PAD_VALID, 0, 0, 0, 0, 1, 0, 6, NONE)
    assert L[2]["ins"] == (3, 4) and L[2]["outs"] == (5,)
    assert words(L[4], 9) == (2, 2, 2, 2, marsfile.PAD_VALID, 0, 0, 0, 0)
    assert words(L[5], 3) == (2, 2, 0)
    assert L[6]["ins"] == (9, 7) and words(L[6], 2) == (1, 2)
    assert L[7]["ins"] == (10, 10) and L[7]["params"] == b"\0" * 64
    # unused id slots of a layer are 0xFFFFFFFF
    assert struct.unpack_from("<4I", d, 76 + 124 * 12 + 16) == (0, NONE, NONE, NONE)


def test_nhwc_shapes_weights_and_axis(marsrt):
    onnx, p = small_graph(np.random.default_rng(4), with_value_info=True)
    d = marsrt.compile_onnx(onnx, nhwc=True)
    hdr, T, L = marsfile.parse(d)
    shapes = {n: t["shape"] for n, t in zip(names(d, hdr), T)}
    assert shapes["images"] == (1, 16, 16, 3) and shapes["c0"] == (1, 8, 8, 8) and shapes["c1"] == (1, 8, 8, 16)
    assert shapes["p1"] == (1, 4, 4, 16) and shapes["u1"] == (1, 8, 8, 16) and shapes["cat"] == (1, 8, 8, 32)
    assert shapes["w0"] == (8, 3, 3, 3)  # the weight record keeps [O, I, KH, KW] while the bytes are O-H-W-I
    for n, t in zip(names(d, hdr), T):
        if n in ("w0", "w1"):
            assert t["fmt"] == marsfile.OHWI
        elif n != "b0":
            assert t["fmt"] == marsfile.NHWC
    q0, s0 = quantise_ref(p["w0"])
    assert blob(d, hdr, T[1]) == q0.reshape(8, 3, 3, 3).transpose(0, 2, 3, 1).tobytes() and T[1]["scale"] == s0
    assert words(L[6], 2) == (3, 2)  # Concat axis 1 (C of NCHW) -> 3
    # the same graph without --nhwc differs only in shapes / formats / weight order: same scales
    hdr2, T2, _ = marsfile.parse(marsrt.compile_onnx(onnx))
    assert [t["scale"] for t in T] == [t["scale"] for t in T2]


def test_float32_mode_keeps_weights(marsrt):
    onnx, p = small_graph(np.random.default_rng(5), with_value_info=True)
    d = marsrt.compile_onnx(onnx, float32=True, nhwc=True)  # float weights stay OIHW even under --nhwc (main.rs:762-765)
    hdr, T, L = marsfile.parse(d)
    assert blob(d, hdr, T[1]) == p["w0"].tobytes() and (T[1]["dtype"], T[1]["fmt"], T[1]["scale"]) == (marsfile.F32, marsfile.OIHW, 1.0)
    assert all(t["dtype"] == marsfile.F32 for t in T)
    assert all(t["scale"] == 1.0 for t in T)  # no scale is ever set without quantisation


def test_rounding_clamp_and_degenerate_weights(marsrt):
    """exact halves round away from zero; an all-zero tensor takes scale 1; NaN quantises to 0; fp16 weights are widened first"""
    def conv_with(w, dtype=None):
        m = ox.model([ox.node("Conv", ["x", "w"], ["y"])], [ox.tensor("w", w, dtype=dtype)], [ox.value_info("x", [1, w.shape[1], 4, 4])],
                     [ox.value_info("y", [1, w.shape[0], 4, 4])])
        d = marsrt.compile_onnx(m)
        hdr, T, _ = marsfile.parse(d)
        return np.frombuffer(blob(d, hdr, T[1]), np.int8), T[1]["scale"]

    w = np.array([127.0, 0.5, -0.5, 1.5, -1.5, 2.5, 126.5, -126.5, 0.49999997, -127.0, 0.0, 63.5], np.float32).reshape(12, 1, 1, 1)
    q, s = conv_with(w)
    assert s == 1.0 and list(q) == [127, 1, -1, 2, -2, 3, 127, -127, 0, -127, 0, 64]
    q, s = conv_with(np.zeros((4, 2, 1, 1), np.float32))
    assert s == 1.0 and not q.any()
    w = np.array([1.0, np.nan, -2.0, 0.25], np.float32).reshape(4, 1, 1, 1)
    q, s = conv_with(w)
    assert s == F32(2.0) / F32(127.0) and list(q) == [64, 0, -127, 16]
    rng = np.random.default_rng(11)
    w16 = (rng.standard_normal((6, 5, 3, 3)) * 3).astype(np.float16)
    w16.ravel()[:3] = [np.float16(6e-8), np.float16(-6e-8), np.float16(65504)]  # subnormals and the largest half
    q, s = conv_with(w16)
    qr, sr = quantise_ref(w16.astype(np.float32))
    assert s == sr and np.array_equal(q, qr)
    for seed in range(20):  # seeded floats at several magnitudes against the numpy restatement
        rng = np.random.default_rng(seed)
        w = (rng.standard_normal((7, 3, 3, 3)) * 10.0 ** rng.integers(-6, 4)).astype(np.float32)
        q, s = conv_with(w)
        qr, sr = quantise_ref(w)
        assert s == sr and np.array_equal(q, qr), seed


def test_operator_table(marsrt):
    """main.rs:76-103, with the compiler's own numbering (mars_format.rs:50-71: Transpose 15, Softmax 17)"""
    x = ox.value_info("x", [1, 4, 8, 8])
    table = [("MaxPool", 2), ("AveragePool", 3), ("GlobalAveragePool", 3), ("Relu", 5), ("LeakyRelu", 7), ("Sigmoid", 9),
             ("Resize", 13), ("Upsample", 13), ("Reshape", 14), ("Transpose", 15), ("Softmax", 17), ("BatchNormalization", 18)]
    for op, code in table:
        d = marsrt.compile_onnx(ox.model([ox.node(op, ["x"], ["y"])], [], [x], [ox.value_info("y", [1, 4, 8, 8])]))
        hdr, T, L = marsfile.parse(d)
        assert hdr["layers"] == 1 and L[0]["type"] == code, op
    for op, code in (("Add", 11), ("Mul", 12)):
        d = marsrt.compile_onnx(ox.model([ox.node(op, ["x", "x"], ["y"])], [], [x], [ox.value_info("y", [1, 4, 8, 8])]))
        assert marsfile.parse(d)[2][0]["type"] == code
    for op in ("Constant", "Shape", "Gather", "Slice", "Split", "Sub", "Div", "Unsqueeze", "Pow", "QuantizeLinear", "DequantizeLinear",
               "Identity", "Gemm", "SomethingElse"):
        d = marsrt.compile_onnx(ox.model([ox.node(op, ["x"], ["y"])], [], [x], [ox.value_info("y", [1, 4, 8, 8])]))
        hdr, T, L = marsfile.parse(d)
        assert hdr["layers"] == 0 and hdr["tensors"] == 1 and hdr["outputs"] == (NONE,), op  # the output is never produced
    # the depthwise type needs group == weight dims[1] == weight dims[0] (main.rs:877): a real depthwise weight is [C, 1, k, k],
    # so it stays a CONV2D with groups = C; only a [C, C, k, k] weight with group = C gets the DEPTHWISE code
    for wshape, code in (((4, 1, 3, 3), marsfile.CONV2D), ((4, 4, 3, 3), marsfile.DWCONV)):
        d = marsrt.compile_onnx(ox.model([ox.node("Conv", ["x", "w"], ["y"], group=4, pads=[1, 1, 1, 1])],
                                         [ox.tensor("w", np.ones(wshape, np.float32))], [x], [ox.value_info("y", [1, 4, 8, 8])]))
        hdr, T, L = marsfile.parse(d)
        assert L[0]["type"] == code and words(L[0], 12)[11] == 4 and T[1]["shape"] == wshape


def test_per_op_params(marsrt):
    x = ox.value_info("x", [1, 4, 8, 6])

    def one(node, inits=(), nhwc=False, out_dims=None):
        outs = [ox.value_info("y", out_dims)] if out_dims else [ox.value_info("y", [], shape=False)]
        d = marsrt.compile_onnx(ox.model([node], list(inits), [x], outs), nhwc=nhwc)
        hdr, T, L = marsfile.parse(d)
        return hdr, T, L, d

    # Reshape: int64 target from raw bytes or from int64_data (packed or not); -1 kept as written
    for kw in (dict(raw=True), dict(raw=False, typed=7), dict(raw=False, typed=7, packed=False)):
        hdr, T, L, d = one(ox.node("Reshape", ["x", "shape"], ["y"]), [ox.tensor("shape", np.array([1, -1, 12], np.int64), **kw)])
        assert struct.unpack_from("<6iI", L[0]["params"]) == (1, -1, 12, 0, 0, 0, 3) and T[1]["shape"] == (1, -1, 12, 1)
    # Transpose: perm words + count; shape permuted
    hdr, T, L, d = one(ox.node("Transpose", ["x"], ["y"], perm=[0, 2, 3, 1]))
    assert words(L[0], 7) == (0, 2, 3, 1, 0, 0, 4) and T[1]["shape"] == (1, 8, 6, 4)
    # Softmax: negative axis counted from 4 dims; scale 1/127
    hdr, T, L, d = one(ox.node("Softmax", ["x"], ["y"], axis=-1))
    assert words(L[0], 1) == (3,) and T[1]["scale"] == F32(1.0) / F32(127.0)
    # Concat: negative axis, more than 4 inputs capped at 4
    hdr, T, L, d = one(ox.node("Concat", ["x", "x", "x", "x", "x"], ["y"], axis=-3))
    assert words(L[0], 2) == (1, 4) and L[0]["ins"] == (0, 0, 0, 0) and T[1]["shape"] == (1, 16, 8, 6)
    # pooling: attributes, explicit pads (top, bottom, left, right order in the record), defaults 2x2 / 2
    hdr, T, L, d = one(ox.node("MaxPool", ["x"], ["y"], kernel_shape=[3, 3], strides=[1, 1], pads=[1, 2, 3, 4]))
    assert words(L[0], 9) == (3, 3, 1, 1, marsfile.PAD_EXPLICIT, 1, 3, 2, 4) and T[1]["shape"] == (1, 4, 10, 10)
    hdr, T, L, d = one(ox.node("GlobalAveragePool", ["x"], ["y"]))
    assert words(L[0], 9) == (2, 2, 2, 2, 0, 0, 0, 0, 0) and T[1]["shape"] == (1, 4, 4, 3)
    # under --nhwc the pool formula still reads positions 2 and 3 (main.rs:942-944): W and C of the NHWC shape
    hdr, T, L, d = one(ox.node("MaxPool", ["x"], ["y"], kernel_shape=[2, 2], strides=[2, 2]), nhwc=True)
    assert T[0]["shape"] == (1, 8, 6, 4) and T[1]["shape"] == (1, 8, 3, 2)
    # Resize: factors from input 2; bilinear flag; Upsample-9 (factors at input 1) falls back to 2x2
    sc = ox.tensor("sc", np.array([1, 1, 3, 2], np.float32))
    hdr, T, L, d = one(ox.node("Resize", ["x", "", "sc"], ["y"], mode="linear"), [sc])
    assert words(L[0], 3) == (3, 2, 1) and T[1]["shape"] == (1, 4, 24, 12)
    hdr, T, L, d = one(ox.node("Upsample", ["x", "sc"], ["y"]), [sc])
    assert words(L[0], 3) == (2, 2, 0)
    # attributes without AttributeProto.type are not seen (onnx_parser.rs:317-329): defaults apply
    n = ox.node("MaxPool", ["x"], ["y"], kernel_shape=ox.attr("kernel_shape", [3, 3], with_type=False))
    hdr, T, L, d = one(n)
    assert words(L[0], 2) == (2, 2)
    # BatchNormalization: fused scale / bias tensors named after the node, three inputs, channel count from shape[1]
    g, b, mu, var = (np.array(v, np.float32) for v in ([1, 2, 3, 4], [0.5, 0, -1, 2], [0.1, 0.2, 0.3, 0.4], [1, 4, 9, 16]))
    n = ox.node("BatchNormalization", ["x", "g", "b", "mu", "var"], ["y"], name="bn0", epsilon=1e-3)
    hdr, T, L, d = one(n, [ox.tensor("g", g), ox.tensor("b", b), ox.tensor("mu", mu), ox.tensor("var", var)])
    fs = (g * (F32(1.0) / np.sqrt(var + F32(1e-3)))).astype(np.float32)
    fb = (b - mu * fs).astype(np.float32)
    assert names(d, hdr) == ["x", "y", "bn0_scale", "bn0_bias"] and L[0]["ins"] == (0, 2, 3) and L[0]["outs"] == (1,)
    assert blob(d, hdr, T[2]) == fs.tobytes() and blob(d, hdr, T[3]) == fb.tobytes()
    assert T[2]["shape"] == (4,) and T[2]["dtype"] == marsfile.F32
    assert T[1]["scale"] == F32(T[0]["scale"] * max(np.max(np.abs(fs)), F32(0.1)))


def test_symbolic_and_missing_dimensions(marsrt):
    """a symbolic dimension (dim_param) reads as -1 and is stored as 1; a value_info without a shape is not an input at all"""
    m = ox.model([ox.node("Relu", ["x"], ["y"])], [], [ox.value_info("x", [None, 3, 8, 8])], [ox.value_info("y", [None, 3, 8, 8])])
    hdr, T, L = marsfile.parse(marsrt.compile_onnx(m))
    assert T[0]["shape"] == (1, 3, 8, 8) and T[1]["shape"] == (1, 3, 8, 8) and hdr["inputs"] == (0,)
    m = ox.model([ox.node("Relu", ["x"], ["y"])], [], [ox.value_info("x", [], shape=False)], [ox.value_info("y", [1, 3, 8, 8])])
    hdr, T, L = marsfile.parse(marsrt.compile_onnx(m))
    assert hdr["inputs"] == () and T[0]["shape"] == (0, 0, 0, 0) and T[0]["scale"] == 1.0
    # initialisers listed among the graph inputs (old exporters) are not inputs
    w = np.ones((2, 3, 1, 1), np.float32)
    m = ox.model([ox.node("Conv", ["x", "w"], ["y"])], [ox.tensor("w", w)], [ox.value_info("x", [1, 3, 8, 8]), ox.value_info("w", [2, 3, 1, 1])],
                 [ox.value_info("y", [1, 2, 8, 8])])
    hdr, T, L = marsfile.parse(marsrt.compile_onnx(m))
    assert hdr["inputs"] == (0,) and struct.unpack_from("<I", marsrt.compile_onnx(m), 20) == (1,)


def qdq_graph(rng):
    """images -Q-> -DQ-> Conv(w -DQ->) -> Q -> DQ -> MaxPool -> Q(shared scale) -> DQ -> output0 (via _QuantizeLinear_Input)"""
    wq = rng.integers(-127, 128, (4, 3, 3, 3), dtype=np.int8)
    zp = ox.tensor("zp", np.zeros((), np.int8))
    inits = [
        ox.tensor("images_scale", np.array(0.0125, np.float32)),
        ox.tensor("w_scale", np.array([0.004], np.float32), raw=False, typed=4),  # a scale kept in float_data
        ox.tensor("conv_out_scale", np.array(0.031, np.float16)),  # a 2-byte scale is read as a half
        ox.tensor("w_quantized", wq), zp,
        ox.tensor("bias", np.array([3, -2, 0, 7], np.int32).view(np.float32)),
    ]
    nodes = [
        ox.node("QuantizeLinear", ["images", "images_scale", "zp"], ["images_QuantizeLinear_Output"]),
        ox.node("DequantizeLinear", ["images_QuantizeLinear_Output", "images_scale", "zp"], ["images_DequantizeLinear_Output"]),
        ox.node("DequantizeLinear", ["w_quantized", "w_scale", "zp"], ["w_DequantizeLinear_Output"]),
        ox.node("Conv", ["images_DequantizeLinear_Output", "w_DequantizeLinear_Output", "bias"], ["conv_out_QuantizeLinear_Input"], pads=[1, 1, 1, 1]),
        ox.node("QuantizeLinear", ["conv_out_QuantizeLinear_Input", "conv_out_scale", "zp"], ["conv_out_QuantizeLinear_Output"]),
        ox.node("DequantizeLinear", ["conv_out_QuantizeLinear_Output", "conv_out_scale", "zp"], ["conv_out_DequantizeLinear_Output"]),
        ox.node("MaxPool", ["conv_out_DequantizeLinear_Output"], ["output0_QuantizeLinear_Input"], kernel_shape=[2, 2], strides=[2, 2]),
        ox.node("QuantizeLinear", ["output0_QuantizeLinear_Input", "conv_out_scale", "zp"], ["output0_QuantizeLinear_Output"]),
        ox.node("DequantizeLinear", ["output0_QuantizeLinear_Output", "conv_out_scale", "zp"], ["output0"]),
    ]
    vi = [ox.value_info("conv_out", [1, 4, 8, 8])]
    return ox.model(nodes, inits, [ox.value_info("images", [1, 3, 8, 8])], [ox.value_info("output0", [1, 4, 4, 4])], vi), wq


def test_qdq_scales_and_prequantised_weights(marsrt):
    onnx, wq = qdq_graph(np.random.default_rng(9))
    d = marsrt.compile_onnx(onnx, nhwc=True)
    hdr, T, L = marsfile.parse(d)
    nm = names(d, hdr)
    assert nm == ["images", "images_DequantizeLinear_Output", "w_quantized", "bias", "conv_out_QuantizeLinear_Input",
                  "conv_out_DequantizeLinear_Output", "output0_QuantizeLinear_Input"]
    t = dict(zip(nm, T))
    assert t["images"]["scale"] == F32(0.0125) and t["images_DequantizeLinear_Output"]["scale"] == F32(0.0125)
    assert t["w_quantized"]["scale"] == F32(0.004) and t["w_quantized"]["fmt"] == marsfile.OHWI
    assert blob(d, hdr, t["w_quantized"]) == wq.transpose(0, 2, 3, 1).tobytes()  # already int8: bytes pass through, re-ordered
    half = F32(np.float16(0.031))
    assert t["conv_out_QuantizeLinear_Input"]["scale"] == half and t["conv_out_DequantizeLinear_Output"]["scale"] == half
    assert t["conv_out_QuantizeLinear_Input"]["shape"] == (1, 8, 8, 4)  # shape found under the name without the QDQ suffix
    assert t["output0_QuantizeLinear_Input"]["scale"] == half           # shared scale, mapped through the QuantizeLinear node
    assert [l["type"] for l in L] == [marsfile.CONV2D, marsfile.MAXPOOL]
    # the conv reads the dequantised image tensor (a feature nobody produces) and the pool the dequantised conv output
    assert L[0]["ins"] == (1,) and L[0]["outs"] == (4,) and L[1]["ins"] == (5,) and L[1]["outs"] == (6,)
    assert hdr["outputs"] == (6,)  # "output0" resolved through "output0_QuantizeLinear_Input" (main.rs:1491-1494)


def test_scales_reach_tensors_made_before_their_producer(marsrt):
    """nodes out of topological order: the pool's output scale is still the default after the node pass; the fixed-point
    pass (main.rs:312-405) gives it the producer's"""
    w = np.random.default_rng(2).standard_normal((4, 3, 1, 1)).astype(np.float32)
    nodes = [ox.node("MaxPool", ["c"], ["p"], kernel_shape=[2, 2], strides=[2, 2]),
             ox.node("Reshape", ["p", "shape"], ["r"]),
             ox.node("Conv", ["x", "w"], ["c"])]
    m = ox.model(nodes, [ox.tensor("w", w), ox.tensor("shape", np.array([1, -1], np.int64))], [ox.value_info("x", [1, 3, 8, 8])],
                 [ox.value_info("r", [1, 64])], [ox.value_info("c", [1, 4, 8, 8]), ox.value_info("p", [1, 4, 4, 4])])
    d = marsrt.compile_onnx(m)
    hdr, T, L = marsfile.parse(d)
    t = dict(zip(names(d, hdr), T))
    assert t["c"]["scale"] != 1.0 and t["p"]["scale"] == t["c"]["scale"] and t["r"]["scale"] == t["c"]["scale"]


def test_malformed_input_is_refused(marsrt):
    onnx, _ = small_graph(np.random.default_rng(1), True)
    for bad in (b"", onnx[:len(onnx) // 2], b"\x3a\xff\xff\xff\xff\x0f" + b"x", onnx[:-3], b"\x0b\x00"):
        with pytest.raises(ValueError):
            marsrt.compile_onnx(bad)
    with pytest.raises(ValueError, match="no graph"):
        marsrt.compile_onnx(ox.f_varint(1, 8))
    m = ox.model([ox.node("Conv", ["x", "w"], ["y"])], [], [ox.value_info("x", [1, 3, 8, 8])], [ox.value_info("y", [1, 2, 8, 8])])
    with pytest.raises(ValueError, match="weight not found"):
        marsrt.compile_onnx(m)
    m = ox.model([ox.node("Add", ["x"], ["y"])], [], [ox.value_info("x", [1, 3, 8, 8])], [ox.value_info("y", [1, 3, 8, 8])])
    with pytest.raises(ValueError, match="missing input B"):
        marsrt.compile_onnx(m)
    w = np.ones((2, 3, 1, 1), np.float32)
    m = ox.model([ox.node("Conv", ["x", "w"], ["y"], strides=[0, 1])], [ox.tensor("w", w)], [ox.value_info("x", [1, 3, 8, 8])],
                 [ox.value_info("y", [1, 2, 8, 8])])
    with pytest.raises(ValueError, match="stride"):
        marsrt.compile_onnx(m)
    # a protobuf `string` that is not UTF-8 (here an op_type) fails the decode, as in the reference's decoder
    m = ox.model([ox.f_bytes(1, "x") + ox.f_bytes(2, "y") + ox.f_bytes(4, b"Re\xfflu")], [], [ox.value_info("x", [1, 3, 8, 8])],
                 [ox.value_info("y", [1, 3, 8, 8])])
    with pytest.raises(ValueError, match="UTF-8"):
        marsrt.compile_onnx(m)
    # every truncation point of a valid file either compiles or is refused -- never crashes
    for cut in range(0, len(onnx), 37):
        try:
            marsrt.compile_onnx(onnx[:cut])
        except ValueError:
            pass


def test_weights_with_few_dims_or_no_inline_payload_compile_like_the_reference(marsrt):
    """the payload check guards the allocation of the --nhwc re-order of quantised weights, nothing else.  A weight with
    three dims (kh / kw then default to 3, as in the reference's `dims.get(2).unwrap_or(3)`) and an initializer without an inline
    payload (external data: the reference reads raw_data only and gets an empty blob) compile wherever that re-order does not run;
    under --nhwc a payload shorter than the o x i x 3 x 3 bytes the re-order walks is refused instead of asking for memory."""
    x, y = [ox.value_info("x", [1, 3, 8, 8])], [ox.value_info("y", [1, 2, 8, 8])]
    w3 = ox.tensor("w", np.ones((2, 3, 3), np.float32))  # Conv1d-style: [O, I, K]
    d = marsrt.compile_onnx(ox.model([ox.node("Conv", ["x", "w"], ["y"])], [w3], x, y))
    hdr, T, L = marsfile.parse(d)
    wt = [t for t in T if t["size"]][0]
    assert list(wt["shape"][:4]) == [2, 3, 3, 3] and wt["size"] == 18  # the 18 values present, quantised; dims as the reference writes them
    d = marsrt.compile_onnx(ox.model([ox.node("Conv", ["x", "w"], ["y"])], [w3], x, y), float32=True)
    assert [t for t in marsfile.parse(d)[1] if t["size"]][0]["size"] == 72
    ext = ox.tensor("w", None, dims=[2, 3, 3, 3], dtype=ox.FLOAT)  # no raw_data, no typed field
    d = marsrt.compile_onnx(ox.model([ox.node("Conv", ["x", "w"], ["y"])], [ext], x, y), float32=True)
    assert len(marsfile.parse(d)[2]) == 1
    d = marsrt.compile_onnx(ox.model([ox.node("Conv", ["x", "w"], ["y"])], [w3], x, y), nhwc=True)  # 72 payload bytes cover the 54 it walks
    assert [t for t in marsfile.parse(d)[1] if t["size"]][0]["size"] == 54
    big = ox.tensor("w", None, dims=[1 << 12, 1 << 12, 3, 3], dtype=ox.INT8)  # 150 M elements declared, none present
    with pytest.raises(ValueError, match="shorter than its dims"):
        marsrt.compile_onnx(ox.model([ox.node("Conv", ["x", "w"], ["y"])], [big], x, y), nhwc=True)


def test_long_names_are_cut_to_59_bytes(marsrt):
    long = "n" * 100
    m = ox.model([ox.node("Relu", ["x"], [long])], [], [ox.value_info("x", [1, 3, 4, 4])], [ox.value_info(long, [1, 3, 4, 4])])
    d = marsrt.compile_onnx(m)
    hdr, T, L = marsfile.parse(d)
    assert names(d, hdr)[1] == "n" * 59 and hdr["outputs"] == (1,)


def test_command_line_tool_writes_the_same_file(marsrt, tmp_path):
    onnx, _ = small_graph(np.random.default_rng(6), True)
    (tmp_path / "m.onnx").write_bytes(onnx)
    exe = os.path.join(ROOT, "thingino-accel_amd", "lib", "mars")
    r = subprocess.run([exe, "--input", str(tmp_path / "m.onnx"), "-o", str(tmp_path / "m.mars"), "--nhwc"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "m.mars").read_bytes() == marsrt.compile_onnx(onnx, nhwc=True)
    r = subprocess.run([exe, "-i", str(tmp_path / "m.onnx"), "-o", str(tmp_path / "f.mars"), "-f"], capture_output=True, text=True)
    assert r.returncode == 0 and (tmp_path / "f.mars").read_bytes() == marsrt.compile_onnx(onnx, float32=True)
    r = subprocess.run([exe, "-i", str(tmp_path / "missing.onnx"), "-o", str(tmp_path / "x.mars")], capture_output=True, text=True)
    assert r.returncode == 1 and "Failed to read ONNX file" in r.stderr
    assert subprocess.run([exe], capture_output=True).returncode == 2


def runnable_graph(rng, float_bias=False):
    """a graph whose compiled form the executor can run: 1x1 / 3x3 convolutions with int32 bias bit patterns (the runtime
    reads the bias bytes as int32), SiLU pairs, a residual Add, MaxPool, Resize, Concat"""
    def w(o, i, k):
        return rng.standard_normal((o, i, k, k)).astype(np.float32)

    def bias(n):
        if float_bias:
            return rng.standard_normal(n).astype(np.float32)
        return rng.integers(-40, 40, n).astype(np.int32).view(np.float32)

    inits = [ox.tensor("w0", w(16, 3, 3)), ox.tensor("b0", bias(16)), ox.tensor("w1", w(32, 16, 1)), ox.tensor("b1", bias(32)),
             ox.tensor("w2", w(32, 32, 3)), ox.tensor("w3", w(32, 32, 1)), ox.tensor("b3", bias(32)), ox.tensor("w4", w(24, 64, 1)),
             ox.tensor("sc", np.array([1, 1, 2, 2], np.float32))]
    nodes = [
        ox.node("Conv", ["images", "w0", "b0"], ["c0"], strides=[2, 2], pads=[1, 1, 1, 1]),
        ox.node("Sigmoid", ["c0"], ["g0"]), ox.node("Mul", ["c0", "g0"], ["a0"]),
        ox.node("Conv", ["a0", "w1", "b1"], ["c1"]),
        ox.node("Sigmoid", ["c1"], ["g1"]), ox.node("Mul", ["c1", "g1"], ["a1"]),
        ox.node("Conv", ["a1", "w2"], ["c2"], pads=[1, 1, 1, 1]),
        ox.node("Relu", ["c2"], ["a2"]),
        ox.node("Add", ["a1", "a2"], ["r2"]),
        ox.node("MaxPool", ["r2"], ["p2"], kernel_shape=[2, 2], strides=[2, 2]),
        ox.node("Conv", ["p2", "w3", "b3"], ["c3"]),
        ox.node("Resize", ["c3", "", "sc"], ["u3"], mode="nearest"),
        ox.node("Concat", ["u3", "r2"], ["cat"], axis=1),
        ox.node("Conv", ["cat", "w4"], ["out"]),
    ]
    dims = dict(c0=[1, 16, 16, 16], g0=[1, 16, 16, 16], a0=[1, 16, 16, 16], c1=[1, 32, 16, 16], g1=[1, 32, 16, 16], a1=[1, 32, 16, 16],
                c2=[1, 32, 16, 16], a2=[1, 32, 16, 16], r2=[1, 32, 16, 16], p2=[1, 32, 8, 8], c3=[1, 32, 8, 8], u3=[1, 32, 16, 16],
                cat=[1, 64, 16, 16])
    return ox.model(nodes, inits, [ox.value_info("images", [1, 3, 32, 32])], [ox.value_info("out", [1, 24, 16, 16])],
                    [ox.value_info(k, v) for k, v in dims.items()])


def test_compiled_file_loads_and_runs_on_the_oracle(marsrt):
    import orcbind
    d = marsrt.compile_onnx(runnable_graph(np.random.default_rng(21)), nhwc=True)
    hdr, T, L = marsfile.parse(d)
    g = orcbind.Graph(d)
    x = np.random.default_rng(5).integers(0, 256, (32, 32, 3), dtype=np.uint8)
    g.set_input(0, x.tobytes())
    assert g.run() == 0
    out = g.tensor(hdr["outputs"][0])
    assert out.size == 16 * 16 * 24 and len(np.unique(out)) > 8  # a live result, not a saturated or all-zero one


def test_transpose_and_softmax_numbering_is_kept_and_announced(marsrt, capfd):
    """the reference's compiler writes Transpose = 15 / Softmax = 17 while the runtime header reads 15 as SOFTMAX
    and 17 as TRANSPOSE (a mismatch inside the reference, kept byte for byte).  A file with either op therefore runs a
    DIFFERENT op once loaded: the compile step says so on stderr, and the oracle's runtime (the reference's dispatch) treats the
    compiled Transpose as its no-op SOFTMAX kind and the compiled Softmax as its no-op TRANSPOSE kind -- nothing is written."""
    import orcbind
    x = ox.value_info("x", [1, 4, 8, 6])
    for op, written, reads, attrs in (("Transpose", 15, "SOFTMAX", dict(perm=[0, 2, 3, 1])), ("Softmax", 17, "TRANSPOSE", dict(axis=-1))):
        d = marsrt.compile_onnx(ox.model([ox.node(op, ["x"], ["y"], **attrs)], [], [x], [ox.value_info("y", [], shape=False)]))
        err = capfd.readouterr().err
        assert "warning: %s is written as layer type %d" % (op, written) in err and reads in err
        hdr, T, L = marsfile.parse(d)
        assert L[0]["type"] == written
        g = orcbind.Graph(d)
        g.set_input(0, bytes(range(192)))
        assert g.run() == 0  # both kinds are no-ops in the reference's runtime (mars_runtime.c:1168-1213): the output stays zero
        assert not np.any(g.tensor(L[0]["outs"][0]))


@pytest.mark.skipif(not os.path.exists(REF_ONNX), reason="reference tree not present (GPU box)")
def test_reference_tree_onnx_compiles(marsrt):
    """the yolov5s export that ships in the reference tree (read here only; nothing of it is committed)"""
    onnx = open(REF_ONNX, "rb").read()
    d = marsrt.compile_onnx(onnx, nhwc=True)
    hdr, T, L = marsfile.parse(d)
    types = [l["type"] for l in L]
    assert types.count(marsfile.CONV2D) == 67 and types.count(marsfile.CONCAT) == 2 and len(L) == 69
    assert len(d) == hdr["woff"] + hdr["wsz"]
    weights = [t for t in T if t["size"] and t["dtype"] == marsfile.I8]
    assert len(weights) == 67 and all(t["fmt"] == marsfile.OHWI for t in weights)
    for t in weights:  # max-abs scaling: the extreme weight lands on +-127; an all-zero tensor (this export has them) takes scale 1
        q = np.frombuffer(blob(d, hdr, t), np.int8)
        assert (np.abs(q.astype(np.int32)).max() == 127 and 0 < t["scale"]) or (not q.any() and t["scale"] == 1.0)


@pytest.mark.gpu
def test_compiled_file_runs_bit_identically_on_the_device(marsrt):
    import orcbind
    marsrt.nna_init()
    for nhwc, seed in ((True, 31), (True, 32), (False, 33)):
        d = marsrt.compile_onnx(runnable_graph(np.random.default_rng(seed)), nhwc=nhwc)
        hdr, T, L = marsfile.parse(d)
        m = marsrt.Model(d, batch=3)
        rng = np.random.default_rng(seed + 100)
        x = rng.integers(0, 256, m.input_view(0).shape, dtype=np.uint8)
        m.input_view(0)[:] = x
        m.run()  # raises on a non-zero mars_run
        got = m.output_view(0).copy()
        for f in range(3):
            g = orcbind.Graph(d)
            g.set_input(0, x[f].tobytes())
            assert g.run() == 0
            want = g.tensor(hdr["outputs"][0])
            assert np.array_equal(want.ravel().view(np.uint8), got[f].ravel().view(np.uint8)), (nhwc, seed, f)
        m.close()
    # --float32: float weights and NCHW features; real float biases this time (the float path reads them as floats).  With the
    # reference's summation order forced (f32_mfma = 0, per model) the device is bit-identical to the oracle here too
    rng = np.random.default_rng(77)
    onnx = runnable_graph(rng, float_bias=True)
    d = marsrt.compile_onnx(onnx, float32=True)
    hdr, T, L = marsfile.parse(d)
    m = marsrt.Model(d, batch=2)
    m.set_tuning("f32_mfma", 0)
    x = rng.random((2, 3 * 32 * 32), dtype=np.float32)
    m.input_view(0).reshape(2, -1)[:] = x.view(np.uint8).reshape(2, -1)
    m.run()
    got = m.output_view(0).copy()
    for f in range(2):
        g = orcbind.Graph(d)
        g.set_input(0, x[f].tobytes())
        assert g.run() == 0
        want = g.tensor(hdr["outputs"][0])
        assert np.array_equal(want.ravel().view(np.uint8), got[f].ravel().view(np.uint8)), f
    m.close()

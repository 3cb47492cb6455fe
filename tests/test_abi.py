"""Host-side checks that need no GPU: the C-ABI library loads, exports every symbol the
headers under include/ declare, keeps the reference's struct layout, validates files
before touching a device, and refuses to run without one (no CPU fallback)."""
import ctypes as C
import os
import re
import struct
import subprocess

import numpy as np
import pytest

import marsfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def test_library_exports_every_declared_symbol(marsrt):
    L = marsrt.lib()
    for hdr, names in marsrt.EXPORTS.items():
        text = open(os.path.join(INC, hdr)).read()
        for n in names:
            assert re.search(r"\b%s\s*\(" % n, text), "%s not declared in %s" % (n, hdr)
            assert hasattr(L, n), "%s not exported" % n
    # and the other way round: every function declared in a public header is in the table
    for hdr in os.listdir(INC):
        text = re.sub(r"/\*.*?\*/", "", open(os.path.join(INC, hdr)).read(), flags=re.S)
        for m in re.finditer(r"^\s*(?:const\s+)?[\w\*\s]+?\b(\w+)\s*\([^;{]*\)\s*;", text, flags=re.M):
            name = m.group(1)
            if name.startswith(("nna_", "mars_", "mxu_", "conv2d_")):
                assert any(name in v for v in marsrt.EXPORTS.values()), "%s (%s) missing from EXPORTS" % (name, hdr)


def test_struct_abi_matches_reference_sizes(tmp_path):
    """sizeof header/tensor/layer/conv_params = 76/124/112/60 and the
    public runtime structs keep the reference's field order"""
    src = tmp_path / "abi.c"
    src.write_text("""
#include <stdio.h>
#include <stddef.h>
#include "mars_runtime.h"
#include "mars_hip.h"
#include "nna.h"
#include "nna_memory.h"
#include "nna_tensor.h"
#include "mxu_ops.h"
int main(void){
 printf("%zu %zu %zu %zu %zu ", sizeof(mars_header_t), sizeof(mars_tensor_t), sizeof(mars_layer_t), sizeof(mars_conv_params_t), sizeof(mars_det_t));
 printf("%zu %zu %zu %zu ", offsetof(mars_runtime_tensor_t, vaddr), offsetof(mars_runtime_tensor_t, paddr), offsetof(mars_runtime_tensor_t, alloc_size), sizeof(mars_runtime_tensor_t));
 printf("%zu %zu %zu %zu %zu\\n", offsetof(mars_model_t, tensors), offsetof(mars_model_t, ddr_base), offsetof(mars_model_t, weights), offsetof(mars_model_t, total_inference_us), sizeof(nna_tensor_t));
 return 0; }
""")
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", INC, str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert out[:5] == ["76", "124", "112", "60", "24"]
    assert out[5:9] == ["128", "136", "144", "160"]
    assert out[9:] == ["80", "96", "144", "160", "56"]


def test_error_strings(marsrt):
    L = marsrt.lib()
    want = ["OK", "Invalid magic number", "Version mismatch", "Memory allocation failed", "Invalid file format",
            "NNA initialization failed", "Layer execution failed", "Invalid tensor", "Invalid layer"]
    for i, s in enumerate(want):
        assert L.mars_get_error_string(-i).decode() == s
    assert L.mars_get_error_string(-99).decode() == "Unknown error"
    assert L.nna_get_version().decode() == "0.1.0-dev"


def _load(marsrt, data):
    p = C.POINTER(marsrt.MarsModel)()
    buf = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    return marsrt.lib().mars_load_memory(buf.ctypes.data, buf.size, C.byref(p))


def test_loader_validates_before_touching_the_device(marsrt):
    G = marsfile.Graph()
    a = G.tensor([1, 4, 4, 8])
    o = G.tensor([1, 4, 4, 8])
    G.layer(marsfile.RELU, [a], [o])
    good = G.serialise([a], [o])
    assert _load(marsrt, good[:40]) == marsrt.MARS_ERR_INVALID_FILE
    assert _load(marsrt, G.serialise([a], [o], magic=0x12345678)) == marsrt.MARS_ERR_INVALID_MAGIC
    assert _load(marsrt, G.serialise([a], [o], major=2)) == marsrt.MARS_ERR_VERSION_MISMATCH
    assert _load(marsrt, good[:100]) == marsrt.MARS_ERR_INVALID_FILE  # tables run past the end
    bad = bytearray(good)
    struct.pack_into("<I", bad, 16, 1 << 30)  # absurd tensor count
    assert _load(marsrt, bad) == marsrt.MARS_ERR_INVALID_FILE
    assert marsrt.lib().mars_load_memory(None, 0, None) == marsrt.MARS_ERR_INVALID_FILE
    assert marsrt.lib().mars_run(None) == marsrt.MARS_ERR_INVALID_FILE


def _has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.mark.skipif(_has_gpu(), reason="this check is about the GPU-less container")
def test_no_cpu_fallback_without_a_gpu(marsrt):
    """without a device the product refuses: nna_init fails, a valid file does not load,
    nna_malloc returns NULL, the direct kernels leave their output untouched"""
    L = marsrt.lib()
    assert L.nna_init() != marsrt.NNA_SUCCESS
    assert L.nna_is_ready() == 0
    assert _load(marsrt, marsrt.synth_model(tiny=True, input_hw=16)) == marsrt.MARS_ERR_NNA_INIT_FAILED
    assert not L.nna_malloc(1024)
    hw = marsrt.HwInfo()
    assert L.nna_get_hw_info(C.byref(hw)) == -1 and L.nna_get_hw_info(None) == -4
    out = np.full(4 * 4 * 32, 77, dtype=np.int8)
    x, w = np.ones(4 * 4 * 16, np.int8), np.ones(32 * 16, np.int8)
    L.conv2d_int8_nhwc_mxu(x.ctypes.data, 4, 4, 16, w.ctypes.data, 32, 1, 1, None, out.ctypes.data, 4, 4, 1, 1, 0, 0,
                           C.c_float(1), C.c_float(1), C.c_float(1))
    assert (out == 77).all()


def test_tensor_handles_are_host_only(marsrt):
    L = marsrt.lib()

    class Shape(C.Structure):
        _fields_ = [("dims", C.c_int32 * 4), ("ndim", C.c_int32)]

    class Tensor(C.Structure):
        _fields_ = [("data", C.c_void_p), ("shape", Shape), ("dtype", C.c_int), ("format", C.c_int),
                    ("bytes", C.c_size_t), ("owns_data", C.c_int)]

    L.nna_shape_make.restype = Shape
    L.nna_shape_make.argtypes = [C.c_int32] * 4
    L.nna_tensor_from_data.restype = C.POINTER(Tensor)
    L.nna_tensor_from_data.argtypes = [C.c_void_p, C.POINTER(Shape), C.c_int, C.c_int]
    L.nna_tensor_numel.restype = C.c_size_t
    L.nna_tensor_numel.argtypes = [C.POINTER(Tensor)]
    L.nna_tensor_bytes.restype = C.c_size_t
    L.nna_tensor_bytes.argtypes = [C.POINTER(Tensor)]
    L.nna_tensor_reshape.argtypes = [C.POINTER(Tensor), C.POINTER(Shape)]
    L.nna_tensor_destroy.argtypes = [C.POINTER(Tensor)]
    L.nna_tensor_data.restype = C.c_void_p
    L.nna_tensor_data.argtypes = [C.POINTER(Tensor)]
    s = L.nna_shape_make(1, 224, 224, 3)
    assert list(s.dims) == [1, 224, 224, 3] and s.ndim == 4
    buf = np.zeros(224 * 224 * 3 * 2, np.uint8)
    t = L.nna_tensor_from_data(buf.ctypes.data, C.byref(s), 4, 1)  # INT16
    assert t and t.contents.owns_data == 0 and L.nna_tensor_numel(t) == 150528 and L.nna_tensor_bytes(t) == 301056
    assert L.nna_tensor_data(t) == buf.ctypes.data
    s2 = L.nna_shape_make(1, 112, 448, 3)
    assert L.nna_tensor_reshape(t, C.byref(s2)) == 0
    s3 = L.nna_shape_make(1, 1, 1, 3)
    assert L.nna_tensor_reshape(t, C.byref(s3)) == -4
    L.nna_tensor_destroy(t)  # does not free the borrowed buffer
    assert not L.nna_tensor_from_data(None, C.byref(s), 3, 1)
    assert L.nna_tensor_numel(None) == 0 and not L.nna_tensor_data(None)


def test_synth_models(marsrt):
    """the synthetic yolov5 twins have the layer census of the shipped yolov5n"""
    d = marsrt.synth_model(width_x16=8, input_hw=640, seed=1)
    assert d == marsrt.synth_model(width_x16=8, input_hw=640, seed=1)  # deterministic
    assert d != marsrt.synth_model(width_x16=8, input_hw=640, seed=2)
    hdr, tensors, layers = marsfile.parse(d)
    kinds = {}
    for l in layers:
        kinds[l["type"]] = kinds.get(l["type"], 0) + 1
    assert kinds[marsfile.CONV2D] == 60 and kinds[marsfile.SIGMOID] == 57 and kinds[marsfile.MUL] == 57
    assert kinds[marsfile.ADD] == 7 and kinds[marsfile.CONCAT] == 13 and kinds[marsfile.MAXPOOL] == 3
    assert kinds[marsfile.UPSAMPLE] == 2
    outs = [tensors[i]["shape"] for i in hdr["outputs"]]
    assert outs == [(1, 80, 80, 255), (1, 40, 40, 255), (1, 20, 20, 255)]
    assert sum(s[1] * s[2] * 3 for s in outs) == 25200  # rows of 85 for the detection tail
    wbytes = sum(t["size"] for t in tensors if t["dtype"] == marsfile.I8 and t["size"])
    assert 7.0e6 < wbytes < 7.5e6  # "weights ~7.2 M int8" 
    n = marsfile.parse(marsrt.synth_model(width_x16=4, input_hw=640))[1]
    assert sum(t["size"] for t in n if t["dtype"] == marsfile.I8 and t["size"]) < 2.0e6
    assert marsrt.lib().mars_synth_model(None, None, 0) == 0
    with pytest.raises(ValueError):
        marsrt.synth_model(input_hw=100)  # not a multiple of 32


def test_half_step_lut_guard(marsrt):
    """mhip_conv_i8_lut2_ok(cs): the 4-instruction requantisation (index trunc(2*acc*cs) into a half-step LUT) equals
    the reference's round-half-away for every float except |acc*cs| == 0x3EFFFFFF; the guard must refuse exactly the
    scales for which some int32 accumulator produces that float (host code: no GPU needed)"""
    import ctypes as C
    L = marsrt.lib()
    L.mhip_conv_i8_lut2_ok.argtypes = [C.c_float]
    quirk = np.frombuffer(np.uint32(0x3EFFFFFF).tobytes(), dtype=np.float32)[0]

    def brute(cs):
        a = abs(np.float32(cs))
        if not (a >= np.float32(1e-6)) or not (a < np.float32(0.99)):
            return 0
        a0 = int(float(quirk) / float(a))
        for acc in range(max(a0 - 6, 1), a0 + 7):
            if np.float32(np.float32(acc) * a) == quirk:
                return 0
        return 1

    rng = np.random.default_rng(4)
    refused = 0
    for cs in np.exp(rng.uniform(np.log(2e-6), np.log(0.9), 4000)).astype(np.float32):
        want = brute(cs)
        assert L.mhip_conv_i8_lut2_ok(float(cs)) == want, float(cs)
        refused += 1 - want
    # constructed offenders: scales a few ulps around quirk / acc
    hits = 0
    for acc in (1, 3, 5, 7, 11, 100, 12345):
        base = np.float32(float(quirk) / acc)
        for ulps in range(-4, 5):
            cs = np.frombuffer((base.view(np.uint32) + np.uint32(ulps % (1 << 32))).astype(np.uint32).tobytes(), dtype=np.float32)[0]
            want = brute(cs)
            assert L.mhip_conv_i8_lut2_ok(float(cs)) == want, (acc, ulps)
            hits += 1 - want
    assert hits > 0  # the guard does fire
    assert L.mhip_conv_i8_lut2_ok(float("nan")) == 0 and L.mhip_conv_i8_lut2_ok(1.5) == 0 and L.mhip_conv_i8_lut2_ok(1e-9) == 0


def test_tensor_byte_size_matches_reference(marsrt):
    """row a3: the format-aware tensor size (reference mars_runtime.c:80-124) against what the reference's own
    static function returned for bare descriptors and for every tensor of every shipped model (goldens made by
    tests/golden/make_golden.py through oracle/_ref)"""
    import ctypes as C
    import json
    import cases
    import marsfile
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.json")))
    L = marsrt.lib()

    def tbs(dtype, fmt, shape):
        d = marsrt.MarsTensorDesc()
        d.dtype, d.format, d.ndims = dtype, fmt, len(shape)
        for i, v in enumerate(shape):
            d.shape[i] = v
        return int(L.mars_hip_tensor_byte_size(C.byref(d)))

    got = [tbs(*c) for c in cases.TBS_CASES]
    assert got == gold["tensor_byte_size"]
    for name in cases.SHIPPED:
        data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "models", name + ".mars"), "rb").read()
        _, tensors, _ = marsfile.parse(data)
        want = gold["models"][name]["pattern"]["tensor_byte_size"]
        assert [tbs(t["dtype"], t["fmt"], list(t["shape"])) for t in tensors] == want, name
    assert L.mars_hip_tensor_byte_size(None) == 0

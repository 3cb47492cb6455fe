"""ctypes bindings for oracle/liboracle.so (this repo's CPU restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- as the checker, never as the product.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None

DET_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("w", "<f4"), ("h", "<f4"),
                      ("conf", "<f4"), ("cls", "<i4")])


class Geom(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("in_h", "in_w", "in_c", "out_h", "out_w", "out_c", "kh",
                                        "kw", "stride_h", "stride_w", "pad_top", "pad_left")]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "restate"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO, mode=os.RTLD_LOCAL)
        L.orc_graph_open.restype = C.c_void_p
        L.orc_graph_open.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]
        L.orc_graph_tensor.restype = C.c_void_p
        L.orc_graph_tensor.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
        L.orc_graph_tensor_bytes.restype = C.c_size_t
        L.orc_graph_tensor_bytes.argtypes = [C.c_void_p, C.c_int]
        L.orc_graph_set_input.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        for n in ("orc_graph_run", "orc_graph_close", "orc_graph_num_tensors", "orc_graph_num_layers",
                  "orc_graph_zero_activations"):
            getattr(L, n).argtypes = [C.c_void_p]
        L.orc_graph_close.restype = None
        L.orc_graph_zero_activations.restype = None
        L.orc_graph_run_range.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_graph_input_id.argtypes = [C.c_void_p, C.c_int]
        L.orc_graph_output_id.argtypes = [C.c_void_p, C.c_int]
        L.orc_graph_conv_macs.restype = C.c_double
        L.orc_graph_conv_macs.argtypes = [C.c_void_p]
        L.orc_run_frames.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.c_size_t, C.c_int, C.c_int, C.c_int]
        L.orc_trunc_x86.argtypes = [C.c_float]
        L.orc_trunc_x86.restype = C.c_int32
        L.orc_parse_output.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int]
        L.orc_nms.argtypes = [C.c_void_p, C.c_int, C.c_float]
        _lib = L
    return _lib


class Graph:
    def __init__(self, file_bytes, slack_mult=8, slack_add=1 << 16):
        L = lib()
        self._buf = np.frombuffer(bytes(file_bytes), dtype=np.uint8).copy()
        err = C.c_int(0)
        self.h = L.orc_graph_open(self._buf.ctypes.data, self._buf.size, slack_mult, slack_add, C.byref(err))
        self.err = err.value
        if not self.h:
            raise RuntimeError("orc_graph_open failed: %d" % err.value)

    num_tensors = property(lambda s: lib().orc_graph_num_tensors(s.h))
    num_layers = property(lambda s: lib().orc_graph_num_layers(s.h))

    def input_id(self, i=0):
        return lib().orc_graph_input_id(self.h, i)

    def output_id(self, i=0):
        return lib().orc_graph_output_id(self.h, i)

    def set_input(self, idx, data):
        a = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        if lib().orc_graph_set_input(self.h, idx, a.ctypes.data, a.size) != 0:
            raise RuntimeError("set_input")

    def run(self, first=None, last=None):
        if first is None:
            return lib().orc_graph_run(self.h)
        return lib().orc_graph_run_range(self.h, first, last)

    def zero(self):
        lib().orc_graph_zero_activations(self.h)

    def tensor_bytes(self, idx):
        return lib().orc_graph_tensor_bytes(self.h, idx)

    def tensor(self, idx, extent=None):
        al = C.c_size_t()
        p = lib().orc_graph_tensor(self.h, idx, C.byref(al))
        n = self.tensor_bytes(idx) if extent is None else min(extent, al.value)
        if not p or n == 0:
            return np.zeros(0, dtype=np.uint8)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)).copy()

    def conv_macs(self):
        return lib().orc_graph_conv_macs(self.h)

    def close(self):
        if self.h:
            lib().orc_graph_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_frames(file_bytes, inputs, out_bytes, out_index=0, nthreads=1):
    """inputs: uint8 array [nframes, in_bytes]; returns uint8 [nframes, out_bytes]."""
    L = lib()
    fb = np.frombuffer(bytes(file_bytes), dtype=np.uint8).copy()
    inputs = np.ascontiguousarray(inputs).view(np.uint8).reshape(inputs.shape[0], -1)
    out = np.zeros((inputs.shape[0], out_bytes), dtype=np.uint8)
    rc = L.orc_run_frames(fb.ctypes.data, fb.size, inputs.ctypes.data, inputs.shape[1],
                          out.ctypes.data, out_bytes, out_index, inputs.shape[0], nthreads)
    if rc != 0:
        raise RuntimeError("orc_run_frames: %d" % rc)
    return out


def _geom(in_h, in_w, in_c, out_h, out_w, out_c, kh, kw, sh, sw, pt, pl):
    return Geom(in_h, in_w, in_c, out_h, out_w, out_c, kh, kw, sh, sw, pt, pl)


def conv2d_int8(nhwc, x, in_h, in_w, in_c, w, out_c, kh, kw, bias, out_h, out_w, sh, sw, pt, pl,
                in_scale, w_scale, out_scale):
    L = lib()
    fn = L.orc_conv_i8_nhwc if nhwc else L.orc_conv_i8_nchw
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Geom), C.c_float,
                   C.c_float, C.c_float]
    x = np.ascontiguousarray(x, dtype=np.int8)
    w = np.ascontiguousarray(w, dtype=np.int8)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.int32)
    out = np.zeros(out_h * out_w * out_c, dtype=np.int8)
    g = _geom(in_h, in_w, in_c, out_h, out_w, out_c, kh, kw, sh, sw, pt, pl)
    fn(x.ctypes.data, w.ctypes.data, None if b is None else b.ctypes.data, out.ctypes.data,
       C.byref(g), in_scale, w_scale, out_scale)
    return out


def conv2d_f32(x, in_h, in_w, in_c, w, out_c, kh, kw, bias, out_h, out_w, sh, sw, pt, pl):
    L = lib()
    fn = L.orc_conv_f32_nchw
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Geom)]
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    out = np.zeros(out_h * out_w * out_c, dtype=np.float32)
    g = _geom(in_h, in_w, in_c, out_h, out_w, out_c, kh, kw, sh, sw, pt, pl)
    fn(x.ctypes.data, w.ctypes.data, None if b is None else b.ctypes.data, out.ctypes.data, C.byref(g))
    return out


def parse_output(pred_i8, npred, scale, maxd=1000):
    p = np.ascontiguousarray(pred_i8, dtype=np.int8)
    dets = np.zeros(maxd, dtype=DET_DTYPE)
    n = lib().orc_parse_output(p.ctypes.data, npred, scale, dets.ctypes.data, maxd)
    return dets[:n].copy()


def nms(dets, thresh=0.45):
    d = np.ascontiguousarray(dets, dtype=DET_DTYPE).copy()
    n = lib().orc_nms(d.ctypes.data, len(d), thresh)
    return d[:n].copy()


def trunc_x86(x):
    return lib().orc_trunc_x86(float(np.float32(x)))


def letterbox(rgb, tw, th, nhwc):
    """orc_letterbox: RGB uint8 [h][w][3] -> int8 letterboxed frame (reference load_image minus the decode)"""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[:2]
    out = np.zeros(tw * th * 3, dtype=np.int8)
    L = lib()
    L.orc_letterbox.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rc = L.orc_letterbox(rgb.ctypes.data, w, h, tw, th, int(bool(nhwc)), out.ctypes.data)
    if rc != 0:
        raise RuntimeError("orc_letterbox failed")
    return out

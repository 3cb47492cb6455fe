"""ctypes bindings for oracle/_ref (the reference's own sources compiled in place).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never imported by the product path.
oracle/_ref exists only where /root/reference was present at build time (or the
prebuilt .so travelled with the tree); `available()` says which.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(_HERE, "_ref")
_libs = {}


def _lib(name):
    if name not in _libs:
        path = os.path.join(_REF, name)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        _libs[name] = C.CDLL(path, mode=os.RTLD_LOCAL)
    return _libs[name]


def available():
    return all(os.path.exists(os.path.join(_REF, n)) for n in ("libref_o1.so", "libref_o2.so"))


def fnv1a64(data):
    h = 0xCBF29CE484222325
    for b in bytes(data):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


# ---------------------------------------------------------------- O1
def o1_run_file(path, input_bytes, out_cap=64 << 20):
    lib = _lib("libref_o1.so")
    lib.ref_o1_run_file.restype = C.c_long
    lib.ref_o1_run_file.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    inp = np.frombuffer(bytes(input_bytes), dtype=np.uint8).copy()
    out = np.zeros(out_cap, dtype=np.uint8)
    rc = lib.ref_o1_run_file(path.encode(), inp.ctypes.data, inp.size, out.ctypes.data, out.size)
    if rc <= 0:
        raise RuntimeError("reference mars_run failed: %d" % rc)
    return out[:rc].copy()


def o1_io_alloc(path):
    """(input 0, output 0) alloc_size as the reference's mars_load_file leaves them"""
    lib = _lib("libref_o1.so")
    lib.ref_o1_io_alloc.restype = C.c_long
    lib.ref_o1_io_alloc.argtypes = [C.c_char_p, C.POINTER(C.c_size_t)]
    out = (C.c_size_t * 2)()
    rc = lib.ref_o1_io_alloc(path.encode(), out)
    if rc != 0:
        raise RuntimeError("reference mars_load_file failed: %d" % rc)
    return int(out[0]), int(out[1])


# ---------------------------------------------------------------- O2
class O2Model:
    """Reference layer functions on a private non-aliased arena."""

    def __init__(self, file_bytes, slack_mult=8, slack_add=1 << 16, fast=False):
        self.lib = _lib("libref_o2_fast.so" if fast else "libref_o2.so")
        L = self.lib
        L.ref_o2_open.restype = C.c_void_p
        L.ref_o2_open.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]
        L.ref_o2_tensor.restype = C.c_void_p
        L.ref_o2_tensor.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.ref_o2_set_input.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.ref_o2_run.argtypes = [C.c_void_p]
        L.ref_o2_run_range.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ref_o2_close.argtypes = [C.c_void_p]
        L.ref_o2_num_tensors.argtypes = [C.c_void_p]
        L.ref_o2_num_layers.argtypes = [C.c_void_p]
        L.ref_o2_tensor_byte_size.restype = C.c_size_t
        L.ref_o2_tensor_byte_size.argtypes = [C.c_void_p, C.c_int]
        self._buf = np.frombuffer(bytes(file_bytes), dtype=np.uint8).copy()
        self.h = L.ref_o2_open(self._buf.ctypes.data, self._buf.size, slack_mult, slack_add)
        if not self.h:
            raise RuntimeError("ref_o2_open failed")

    def set_input(self, idx, data):
        a = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        if self.lib.ref_o2_set_input(self.h, idx, a.ctypes.data, a.size) != 0:
            raise RuntimeError("set_input")

    def run(self, first=None, last=None):
        if first is None:
            rc = self.lib.ref_o2_run(self.h)
        else:
            rc = self.lib.ref_o2_run_range(self.h, first, last)
        return rc

    def tensor(self, idx, extent=None):
        sb, al = C.c_size_t(), C.c_size_t()
        p = self.lib.ref_o2_tensor(self.h, idx, C.byref(sb), C.byref(al))
        n = sb.value if extent is None else min(extent, al.value)
        if not p or n == 0:
            return np.zeros(0, dtype=np.uint8)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)).copy()

    def tensor_byte_size(self, idx):
        return self.lib.ref_o2_tensor_byte_size(self.h, idx)

    def close(self):
        if self.h:
            self.lib.ref_o2_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------- direct kernels (mxu_conv.c externs)
def _conv_args():
    return [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
            C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]


def conv2d_int8(nhwc, x, in_h, in_w, in_c, w, out_c, kh, kw, bias, out_h, out_w,
                sh, sw, pt, pl, in_scale, w_scale, out_scale):
    lib = _lib("libref_o2.so")
    fn = lib.conv2d_int8_nhwc_mxu if nhwc else lib.conv2d_int8_mxu
    fn.restype = None
    fn.argtypes = _conv_args() + [C.c_float, C.c_float, C.c_float]
    x = np.ascontiguousarray(x, dtype=np.int8)
    w = np.ascontiguousarray(w, dtype=np.int8)
    out = np.zeros(out_h * out_w * out_c, dtype=np.int8)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.int32)
    fn(x.ctypes.data, in_h, in_w, in_c, w.ctypes.data, out_c, kh, kw,
       None if b is None else b.ctypes.data, out.ctypes.data, out_h, out_w, sh, sw, pt, pl,
       in_scale, w_scale, out_scale)
    return out


def conv2d_f32(x, in_h, in_w, in_c, w, out_c, kh, kw, bias, out_h, out_w, sh, sw, pt, pl):
    lib = _lib("libref_o2.so")
    fn = lib.conv2d_float32_mxu
    fn.restype = None
    fn.argtypes = _conv_args() + [C.c_void_p]
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    out = np.zeros(out_h * out_w * out_c, dtype=np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    fn(x.ctypes.data, in_h, in_w, in_c, w.ctypes.data, out_c, kh, kw,
       None if b is None else b.ctypes.data, out.ctypes.data, out_h, out_w, sh, sw, pt, pl, None)
    return out


# ---------------------------------------------------------------- O3
DET_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("w", "<f4"), ("h", "<f4"),
                      ("conf", "<f4"), ("cls", "<i4")])


def parse_output(pred_i8, npred, scale, maxd=1000):
    lib = _lib("libref_o2.so")
    assert lib.ref_o3_sizeof_det() == DET_DTYPE.itemsize
    lib.ref_o3_parse_output.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int]
    p = np.ascontiguousarray(pred_i8, dtype=np.int8)
    dets = np.zeros(maxd, dtype=DET_DTYPE)
    n = lib.ref_o3_parse_output(p.ctypes.data, npred, scale, dets.ctypes.data, maxd)
    return dets[:n].copy()


def nms(dets, thresh=0.45):
    lib = _lib("libref_o2.so")
    lib.ref_o3_nms.argtypes = [C.c_void_p, C.c_int, C.c_float]
    d = np.ascontiguousarray(dets, dtype=DET_DTYPE).copy()
    n = lib.ref_o3_nms(d.ctypes.data, len(d), thresh)
    return d[:n].copy()


def load_image(rgb, tw, th, nhwc):
    """the reference's load_image() (mars_yolo_test.c:40-77) on an RGB uint8 array [h][w][3]: the frame is written
    as a binary PPM, decoded by the reference's stb_image, letterbox-resized by its stb_image_resize and converted"""
    import tempfile
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[:2]
    lib = _lib("libref_o2.so")
    lib.ref_o3_load_image.restype = C.c_int
    lib.ref_o3_load_image.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    out = np.zeros(tw * th * 3, dtype=np.int8)
    with tempfile.NamedTemporaryFile(suffix=".ppm", delete=False) as fh:
        fh.write(b"P6\n%d %d\n255\n" % (w, h))
        fh.write(rgb.tobytes())
        path = fh.name
    try:
        sw, sh = C.c_int(), C.c_int()
        # the reference prints two lines per image; keep the test log quiet
        fd = os.dup(1)
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
        try:
            rc = lib.ref_o3_load_image(path.encode(), tw, th, int(bool(nhwc)), out.ctypes.data, C.byref(sw), C.byref(sh))
        finally:
            C.CDLL(None).fflush(None)  # the reference's printf output is still in stdio's buffer
            os.dup2(fd, 1)
            os.close(fd)
            os.close(devnull)
    finally:
        os.unlink(path)
    if rc != 0 or (sw.value, sh.value) != (w, h):
        raise RuntimeError("reference load_image failed")
    return out

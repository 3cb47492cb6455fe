"""ctypes view of libnna_mars.so -- the C-ABI the reference's callers bind.

Host-side mirror of the reference's C interface for this path (same function
names, argument meaning and error codes: include/nna.h, nna_memory.h,
nna_tensor.h, mars_runtime.h, mxu_ops.h) plus the additive mars_hip_* calls of
include/mars_hip.h.  There is no compute in this file and no fallback: if the
shared library or the GPU is missing, calls fail.

The package directory is called ``thingino-accel_amd`` (not importable by name);
load this module by path::

    import importlib.util, os
    spec = importlib.util.spec_from_file_location("marsrt", ".../thingino-accel_amd/marsrt.py")
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libnna_mars.so")

MARS_OK = 0
MARS_ERR_INVALID_MAGIC = -1
MARS_ERR_VERSION_MISMATCH = -2
MARS_ERR_ALLOC_FAILED = -3
MARS_ERR_INVALID_FILE = -4
MARS_ERR_NNA_INIT_FAILED = -5
MARS_ERR_LAYER_FAILED = -6
MARS_ERR_INVALID_TENSOR = -7
MARS_ERR_INVALID_LAYER = -8
NNA_SUCCESS = 0

DET_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("w", "<f4"), ("h", "<f4"),
                      ("conf", "<f4"), ("cls", "<i4")])
MAX_DET = 1000


# ---- struct mirrors (include/mars.h, include/mars_runtime.h) -----------------
class MarsHeader(C.Structure):
    _pack_ = 1
    _fields_ = [("magic", C.c_uint32), ("version_major", C.c_uint16), ("version_minor", C.c_uint16),
                ("flags", C.c_uint32), ("num_layers", C.c_uint32), ("num_tensors", C.c_uint32),
                ("num_inputs", C.c_uint32), ("num_outputs", C.c_uint32), ("weights_offset", C.c_uint64),
                ("weights_size", C.c_uint64), ("input_tensor_ids", C.c_uint32 * 4),
                ("output_tensor_ids", C.c_uint32 * 4)]


class MarsTensorDesc(C.Structure):
    _pack_ = 1
    _fields_ = [("id", C.c_uint32), ("name", C.c_char * 60), ("dtype", C.c_uint32), ("format", C.c_uint32),
                ("ndims", C.c_uint32), ("shape", C.c_int32 * 6), ("data_offset", C.c_uint64),
                ("data_size", C.c_uint64), ("scale", C.c_float), ("zero_point", C.c_int32)]


class MarsRuntimeTensor(C.Structure):
    _fields_ = [("desc", MarsTensorDesc), ("vaddr", C.c_void_p), ("paddr", C.c_void_p),
                ("alloc_size", C.c_size_t), ("is_external", C.c_bool)]


class MarsRuntimeLayer(C.Structure):
    _fields_ = [("desc", C.c_uint8 * 112), ("is_executed", C.c_bool)]


class MarsModel(C.Structure):
    _fields_ = [("header", MarsHeader), ("tensors", C.POINTER(MarsRuntimeTensor)),
                ("layers", C.POINTER(MarsRuntimeLayer)), ("ddr_base", C.c_void_p), ("ddr_paddr", C.c_void_p),
                ("ddr_size", C.c_size_t), ("oram_base", C.c_void_p), ("oram_paddr", C.c_void_p),
                ("oram_size", C.c_size_t), ("weights", C.c_void_p), ("weights_size", C.c_size_t),
                ("total_inference_us", C.c_uint64), ("inference_count", C.c_uint32)]


class HwInfo(C.Structure):
    _fields_ = [("oram_vbase", C.c_uint32), ("oram_pbase", C.c_uint32), ("oram_size", C.c_uint32),
                ("version", C.c_uint32)]


class SynthOpts(C.Structure):
    _fields_ = [("width_x16", C.c_int), ("depth_x3", C.c_int), ("input_hw", C.c_int), ("float32", C.c_int),
                ("nchw_int8", C.c_int), ("seed", C.c_uint), ("tiny", C.c_int), ("vary_scales", C.c_int)]


class PipeOpts(C.Structure):
    _fields_ = [("download_outputs", C.c_int), ("detect", C.c_int), ("det_outputs", C.c_int * 4), ("n_det_outputs", C.c_int),
                ("nms_thresh", C.c_float), ("camera_w", C.c_int), ("camera_h", C.c_int)]


assert C.sizeof(MarsHeader) == 76 and C.sizeof(MarsTensorDesc) == 124


class CompileOpts(C.Structure):  # mars_compile_opts_t (include/mars_compile.h)
    _fields_ = [("float32", C.c_int), ("nhwc", C.c_int), ("verbose", C.c_int)]

# every symbol the headers under include/ declare (checked by tests/test_abi.py)
EXPORTS = {
    "nna.h": ["nna_init", "nna_deinit", "nna_get_hw_info", "nna_is_ready", "nna_get_version", "nna_lock",
              "nna_unlock"],
    "nna_memory.h": ["nna_malloc", "nna_memalign", "nna_calloc", "nna_free", "nna_oram_malloc", "nna_oram_free",
                     "nna_oram_get_stats", "nna_cache_flush", "nna_cache_invalidate"],
    "nna_tensor.h": ["nna_tensor_create", "nna_tensor_from_data", "nna_tensor_destroy", "nna_tensor_data",
                     "nna_tensor_shape", "nna_tensor_dtype", "nna_tensor_numel", "nna_tensor_bytes",
                     "nna_tensor_reshape", "nna_shape_make"],
    "nna_model.h": ["nna_model_load", "nna_model_load_from_memory", "nna_model_get_info", "nna_model_get_input",
                    "nna_model_get_input_by_name", "nna_model_get_output", "nna_model_get_output_by_name",
                    "nna_model_run", "nna_model_unload"],
    "mars_runtime.h": ["mars_load_file", "mars_load_memory", "mars_free", "mars_get_input", "mars_get_output",
                       "mars_run", "mars_get_error_string", "mars_get_num_inputs", "mars_get_num_outputs",
                       "mars_print_summary"],
    "mxu_ops.h": ["mxu_init", "mxu_is_initialized", "mxu_mul_f32", "mxu_add_f32", "mxu_sub_f32", "mxu_relu_f32",
                  "conv2d_int8_mxu", "conv2d_int8_nhwc_mxu", "conv2d_float32_mxu"],
    "mars_hip.h": ["mars_hip_set_batch", "mars_hip_get_batch", "mars_hip_upload_inputs", "mars_hip_run_device",
                   "mars_hip_run_device_async", "mars_hip_download_outputs", "mars_hip_sync",
                   "mars_hip_tensor_device", "mars_hip_tensor_row_pitch", "mars_hip_read_tensor", "mars_hip_write_tensor", "mars_hip_set_fusion",
                   "mars_hip_set_profiling", "mars_hip_num_ops", "mars_hip_op_info", "mars_hip_stream",
                   "mars_hip_load_memory_ex", "mars_hip_describe_plan", "mars_hip_param_arena", "mars_yolo_parse_output", "mars_yolo_nms",
                   "mars_hip_detect", "mars_hip_detect_device", "mars_synth_model", "mars_hip_set_tuning", "mars_hip_autotune", "mars_yolo_letterbox",
                   "mars_hip_preprocess", "mars_hip_preprocess_device", "mars_hip_tensor_frame_bytes", "mars_hip_tensor_byte_size", "mars_hip_pipe_open",
                   "mars_hip_pipe_input", "mars_hip_pipe_submit", "mars_hip_pipe_wait", "mars_hip_pipe_close", "mars_hip_pipe_camera_ms", "mars_hip_set_output_mode",
                   "mars_hip_get_tuning", "mars_hip_model_set_tuning", "mars_hip_model_get_tuning"],
    "mars_compile.h": ["mars_compile_onnx", "mars_compile_file", "mars_compile_last_error"],
}

_lib = None


def lib():
    """dlopen the C-ABI library and declare prototypes.  Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    L = C.CDLL(LIB_PATH, mode=os.RTLD_LOCAL)  # never interpose the oracle's same-named symbols
    P = C.POINTER
    L.nna_get_version.restype = C.c_char_p
    L.nna_get_hw_info.argtypes = [P(HwInfo)]
    L.nna_malloc.restype = C.c_void_p
    L.nna_malloc.argtypes = [C.c_size_t]
    L.nna_calloc.restype = C.c_void_p
    L.nna_calloc.argtypes = [C.c_size_t, C.c_size_t]
    L.nna_memalign.restype = C.c_void_p
    L.nna_memalign.argtypes = [C.c_size_t, C.c_size_t]
    L.nna_free.argtypes = [C.c_void_p]
    L.nna_free.restype = None
    L.nna_oram_malloc.restype = C.c_void_p
    L.nna_oram_malloc.argtypes = [C.c_size_t]
    L.nna_oram_get_stats.argtypes = [P(C.c_size_t)] * 3
    L.mars_load_file.argtypes = [C.c_char_p, P(P(MarsModel))]
    L.mars_load_memory.argtypes = [C.c_void_p, C.c_size_t, P(P(MarsModel))]
    L.mars_hip_load_memory_ex.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, P(P(MarsModel))]
    L.mars_hip_describe_plan.restype = C.c_size_t
    L.mars_hip_describe_plan.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_char_p, C.c_size_t]
    L.mars_free.argtypes = [P(MarsModel)]
    L.mars_free.restype = None
    L.mars_get_input.restype = P(MarsRuntimeTensor)
    L.mars_get_input.argtypes = [P(MarsModel), C.c_int]
    L.mars_get_output.restype = P(MarsRuntimeTensor)
    L.mars_get_output.argtypes = [P(MarsModel), C.c_int]
    L.mars_get_num_inputs.argtypes = [P(MarsModel)]
    L.mars_get_num_outputs.argtypes = [P(MarsModel)]
    L.mars_run.argtypes = [P(MarsModel)]
    L.mars_get_error_string.restype = C.c_char_p
    L.mars_get_error_string.argtypes = [C.c_int]
    L.mars_print_summary.argtypes = [P(MarsModel)]
    L.mars_hip_tensor_frame_bytes.restype = C.c_size_t
    L.mars_hip_tensor_frame_bytes.argtypes = [P(MarsModel), C.c_int]
    L.mars_hip_tensor_byte_size.restype = C.c_size_t
    L.mars_hip_tensor_byte_size.argtypes = [P(MarsTensorDesc)]
    L.mars_hip_pipe_open.argtypes = [P(MarsModel), P(PipeOpts)]
    L.mars_hip_pipe_input.restype = C.c_void_p
    L.mars_hip_pipe_input.argtypes = [P(MarsModel), C.c_int]
    L.mars_hip_pipe_submit.argtypes = [P(MarsModel)]
    L.mars_hip_pipe_camera_ms.restype = C.c_float
    L.mars_hip_pipe_camera_ms.argtypes = [P(MarsModel)]
    L.mars_hip_pipe_wait.argtypes = [P(MarsModel), P(C.c_void_p), P(C.c_void_p), P(C.c_void_p)]
    L.mars_hip_pipe_close.argtypes = [P(MarsModel)]
    L.mars_hip_pipe_close.restype = None
    L.mars_hip_set_output_mode.argtypes = [P(MarsModel), C.c_int]
    for n in ("mars_hip_upload_inputs", "mars_hip_run_device", "mars_hip_run_device_async",
              "mars_hip_download_outputs", "mars_hip_get_batch", "mars_hip_num_ops"):
        getattr(L, n).argtypes = [P(MarsModel)]
    L.mars_hip_set_batch.argtypes = [P(MarsModel), C.c_int]
    L.mars_hip_set_fusion.argtypes = [P(MarsModel), C.c_int]
    L.mars_hip_set_tuning.argtypes = [C.c_char_p, C.c_int]
    L.mars_hip_get_tuning.argtypes = [C.c_char_p, P(C.c_int)]
    L.mars_hip_model_set_tuning.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.mars_hip_model_get_tuning.argtypes = [C.c_void_p, C.c_char_p, P(C.c_int)]
    L.mars_hip_autotune.argtypes = [P(MarsModel), C.c_int]
    L.mars_yolo_letterbox.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.mars_hip_preprocess.argtypes = [P(MarsModel), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.mars_hip_preprocess_device.argtypes = [P(MarsModel), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.mars_hip_set_profiling.argtypes = [P(MarsModel), C.c_int]
    L.mars_hip_set_profiling.restype = None
    L.mars_hip_tensor_device.restype = C.c_void_p
    L.mars_hip_tensor_device.argtypes = [P(MarsModel), C.c_int, P(C.c_size_t)]
    L.mars_hip_read_tensor.argtypes = [P(MarsModel), C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.mars_hip_tensor_row_pitch.argtypes = [P(MarsModel), C.c_int, P(C.c_int)]
    L.mars_hip_write_tensor.argtypes = [P(MarsModel), C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.mars_hip_op_info.argtypes = [P(MarsModel), C.c_int, P(C.c_int), P(C.c_int), P(C.c_double), P(C.c_double),
                                   P(C.c_float)]
    L.mars_hip_stream.restype = C.c_void_p
    L.mars_hip_param_arena.restype = C.c_void_p
    L.mars_hip_param_arena.argtypes = [P(MarsModel), P(C.c_size_t)]
    L.mars_yolo_parse_output.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int]
    L.mars_yolo_nms.argtypes = [C.c_void_p, C.c_int, C.c_float]
    L.mars_hip_detect.argtypes = [P(MarsModel), P(C.c_int), C.c_int, C.c_float, C.c_void_p, P(C.c_int)]
    L.mars_hip_detect_device.argtypes = [P(MarsModel), P(C.c_int), C.c_int, C.c_float]
    L.mars_compile_onnx.restype = C.c_size_t
    L.mars_compile_onnx.argtypes = [C.c_char_p, C.c_size_t, P(CompileOpts), C.c_void_p, C.c_size_t]
    L.mars_compile_file.argtypes = [C.c_char_p, C.c_char_p, P(CompileOpts)]
    L.mars_compile_last_error.restype = C.c_char_p
    L.mars_synth_model.restype = C.c_size_t
    L.mars_synth_model.argtypes = [P(SynthOpts), C.c_void_p, C.c_size_t]
    conv_args = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                 C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    for n in ("conv2d_int8_mxu", "conv2d_int8_nhwc_mxu"):
        getattr(L, n).argtypes = conv_args + [C.c_float, C.c_float, C.c_float]
        getattr(L, n).restype = None
    L.conv2d_float32_mxu.argtypes = conv_args + [C.c_void_p]
    L.conv2d_float32_mxu.restype = None
    for n in ("mxu_mul_f32", "mxu_add_f32", "mxu_sub_f32"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        getattr(L, n).restype = None
    L.mxu_relu_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.mxu_relu_f32.restype = None
    _lib = L
    return L


class MarsError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = lib().mars_get_error_string(code).decode()
        super().__init__("%s: %s (%d)" % (what, msg, code))


def set_tuning(key, value):
    """Launch-policy knob of the conv kernels (mars_hip_set_tuning); results never depend on it."""
    if lib().mars_hip_set_tuning(key.encode(), int(value)) != 0:
        raise KeyError(key)


def get_tuning(key):
    v = C.c_int(0)
    if lib().mars_hip_get_tuning(key.encode(), C.byref(v)) != 0:
        raise KeyError(key)
    return v.value


def letterbox(rgb, tw, th, nhwc=True):
    """mars_yolo_letterbox: uint8 RGB [h][w][3] -> int8 letterboxed frame, on the GPU."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[:2]
    out = np.zeros(tw * th * 3, dtype=np.int8)
    if lib().mars_yolo_letterbox(rgb.ctypes.data, w, h, tw, th, int(bool(nhwc)), out.ctypes.data) != 0:
        raise RuntimeError("mars_yolo_letterbox failed")
    return out


def synth_model(width_x16=8, depth_x3=1, input_hw=640, float32=False, nchw_int8=False, seed=1, tiny=False, vary_scales=False):
    """Bytes of a synthetic well-formed .mars graph (mars_synth_model)."""
    o = SynthOpts(width_x16, depth_x3, input_hw, int(float32), int(nchw_int8), seed, int(tiny), int(vary_scales))
    n = lib().mars_synth_model(C.byref(o), None, 0)
    if n == 0:
        raise ValueError("mars_synth_model rejected the options")
    buf = (C.c_uint8 * n)()
    assert lib().mars_synth_model(C.byref(o), buf, n) == n
    return bytes(buf)


def describe_plan(file_bytes, flags=0):
    """the launch plan of a .mars file as a list of text lines (mars_hip_describe_plan; host only: works without a GPU)"""
    b = bytes(file_bytes)
    n = lib().mars_hip_describe_plan(b, len(b), flags, None, 0)
    if n == 0:
        raise ValueError("the loader rejects the file")
    buf = C.create_string_buffer(n + 1)
    lib().mars_hip_describe_plan(b, len(b), flags, buf, n + 1)
    return buf.value.decode().splitlines()


def compile_onnx(onnx_bytes, float32=False, nhwc=False, verbose=False):
    """ONNX model bytes -> .mars file bytes (mars_compile_onnx; host-only, needs no GPU)."""
    o = CompileOpts(int(float32), int(nhwc), int(verbose))
    n = lib().mars_compile_onnx(onnx_bytes, len(onnx_bytes), C.byref(o), None, 0)
    if n == 0:
        raise ValueError(lib().mars_compile_last_error().decode())
    buf = C.create_string_buffer(n)
    assert lib().mars_compile_onnx(onnx_bytes, len(onnx_bytes), C.byref(o), buf, n) == n
    return buf.raw


def nna_init():
    rc = lib().nna_init()
    if rc != NNA_SUCCESS:
        raise RuntimeError("nna_init failed (%d): no usable MI355X; this library has no CPU path" % rc)


class Model:
    """mars_load_memory / mars_run / mars_free with numpy views of the pinned I/O staging."""

    def __init__(self, file_bytes, batch=1, fusion=None, flags=0):
        L = lib()
        self._bytes = np.frombuffer(bytes(file_bytes), dtype=np.uint8).copy()
        self.p = C.POINTER(MarsModel)()
        rc = L.mars_hip_load_memory_ex(self._bytes.ctypes.data, self._bytes.size, flags, C.byref(self.p))
        if rc != MARS_OK:
            self.p = None
            raise MarsError(rc, "mars_load_memory")
        if fusion is not None:
            self.set_fusion(fusion)
        if batch != 1:
            self.set_batch(batch)

    # -- reference API
    @property
    def header(self):
        return self.p.contents.header

    def tensor_desc(self, idx):
        return self.p.contents.tensors[idx].desc

    def input(self, i=0):
        return lib().mars_get_input(self.p, i)

    def output(self, i=0):
        return lib().mars_get_output(self.p, i)

    def _view(self, rt, tid):
        # batch * frame bytes by shape; alloc_size is larger for a single frame (the reference's working-buffer size)
        t = rt.contents
        n = lib().mars_hip_tensor_frame_bytes(self.p, tid) * self.batch
        return np.ctypeslib.as_array(C.cast(t.vaddr, C.POINTER(C.c_uint8)), shape=(n,))

    def input_view(self, i=0):
        """uint8 view [batch, frame_bytes] of mars_get_input(i)->vaddr."""
        return self._view(self.input(i), self.header.input_tensor_ids[i]).reshape(self.batch, -1)

    def output_view(self, i=0):
        return self._view(self.output(i), self.header.output_tensor_ids[i]).reshape(self.batch, -1)

    def run(self):
        rc = lib().mars_run(self.p)
        if rc != MARS_OK:
            raise MarsError(rc, "mars_run")

    # -- extensions
    @property
    def batch(self):
        return lib().mars_hip_get_batch(self.p)

    def set_batch(self, n):
        rc = lib().mars_hip_set_batch(self.p, n)
        if rc != MARS_OK:
            raise MarsError(rc, "mars_hip_set_batch")

    def set_fusion(self, level):
        rc = lib().mars_hip_set_fusion(self.p, level)
        if rc != MARS_OK:
            raise MarsError(rc, "mars_hip_set_fusion")

    def upload(self):
        rc = lib().mars_hip_upload_inputs(self.p)
        if rc != MARS_OK:
            raise MarsError(rc, "upload")

    def run_device(self, sync=True):
        rc = (lib().mars_hip_run_device if sync else lib().mars_hip_run_device_async)(self.p)
        if rc != MARS_OK:
            raise MarsError(rc, "run_device")

    def download(self):
        rc = lib().mars_hip_download_outputs(self.p)
        if rc != MARS_OK:
            raise MarsError(rc, "download")

    def read_tensor(self, idx, frame=0, nbytes=None):
        if nbytes is None:
            d = self.tensor_desc(idx)
            n = 1
            for k in range(d.ndims):
                n *= max(d.shape[k], 0)
            nbytes = n * (4 if d.dtype in (0, 1) else 2 if d.dtype == 2 else 1)
        out = np.zeros(nbytes, dtype=np.uint8)
        rc = lib().mars_hip_read_tensor(self.p, idx, frame, out.ctypes.data, nbytes)
        if rc != MARS_OK:
            raise MarsError(rc, "read_tensor %d" % idx)
        return out

    def set_profiling(self, on):
        lib().mars_hip_set_profiling(self.p, int(on))

    def preprocess(self, rgb_frames, first_frame=0, input_index=0):
        """uint8 RGB frames [n][h][w][3] -> letterboxed int8 frames of the graph input, in HBM"""
        a = np.ascontiguousarray(rgb_frames, dtype=np.uint8)
        n, h, w = a.shape[:3]
        rc = lib().mars_hip_preprocess(self.p, input_index, a.ctypes.data, w, h, first_frame, n)
        if rc != MARS_OK:
            raise MarsError(rc, "mars_hip_preprocess")

    def set_tuning(self, key, value):
        """per-model override of a launch-policy knob (mars_hip_model_set_tuning)"""
        if lib().mars_hip_model_set_tuning(self.p, key.encode(), int(value)) != 0:
            raise KeyError(key)

    def get_tuning(self, key):
        v = C.c_int(0)
        if lib().mars_hip_model_get_tuning(self.p, key.encode(), C.byref(v)) != 0:
            raise KeyError(key)
        return v.value

    def autotune(self, reps=3):
        rc = lib().mars_hip_autotune(self.p, reps)
        if rc != 0:
            raise MarsError(rc, "mars_hip_autotune")

    def ops(self):
        n = lib().mars_hip_num_ops(self.p)
        res = []
        for i in range(n):
            layer, kind = C.c_int(), C.c_int()
            macs, byt, ms = C.c_double(), C.c_double(), C.c_float()
            lib().mars_hip_op_info(self.p, i, C.byref(layer), C.byref(kind), C.byref(macs), C.byref(byt), C.byref(ms))
            res.append(dict(layer=layer.value, kind=kind.value, macs=macs.value, bytes=byt.value, ms=ms.value))
        return res

    def param_arena(self):
        n = C.c_size_t()
        p = lib().mars_hip_param_arena(self.p, C.byref(n))
        return p, n.value

    def detect(self, outputs=(0,), thresh=0.45):
        idx = (C.c_int * len(outputs))(*outputs)
        dets = np.zeros((self.batch, MAX_DET), dtype=DET_DTYPE)
        counts = np.zeros(self.batch, dtype=np.int32)
        rc = lib().mars_hip_detect(self.p, idx, len(outputs), thresh, dets.ctypes.data,
                                   counts.ctypes.data_as(C.POINTER(C.c_int)))
        if rc != MARS_OK:
            raise MarsError(rc, "mars_hip_detect")
        return [dets[f, :counts[f]].copy() for f in range(self.batch)]

    def detect_device(self, outputs=(0,), thresh=0.45):
        idx = (C.c_int * len(outputs))(*outputs)
        rc = lib().mars_hip_detect_device(self.p, idx, len(outputs), thresh)
        if rc != MARS_OK:
            raise MarsError(rc, "mars_hip_detect_device")

    # -- pipelined host I/O (mars_hip_pipe_*)
    def pipe_open(self, download_outputs=True, detect=False, det_outputs=(0,), thresh=0.45, camera=None):
        """camera = (w, h): input 0 is fed from uint8 RGB camera frames, the letterbox front-end runs on the device behind the upload"""
        cw, ch = camera if camera else (0, 0)
        o = PipeOpts(int(download_outputs), int(detect), (C.c_int * 4)(*(list(det_outputs) + [0] * (4 - len(det_outputs)))),
                     len(det_outputs) if detect else 0, thresh, int(cw), int(ch))
        rc = lib().mars_hip_pipe_open(self.p, C.byref(o))
        if rc != MARS_OK:
            raise MarsError(rc, "mars_hip_pipe_open")
        self._pipe = (bool(download_outputs), bool(detect))
        self._pipe_camera = (int(cw), int(ch)) if camera else None

    def pipe_input_view(self, i=0):
        """uint8 view [batch, frame_bytes] of the staging buffer the NEXT pipe_submit() uploads (camera mode, input 0:
        [batch, h * w * 3] RGB bytes)"""
        ptr = lib().mars_hip_pipe_input(self.p, i)
        if i == 0 and getattr(self, "_pipe_camera", None):
            n = self._pipe_camera[0] * self._pipe_camera[1] * 3 * self.batch
        else:
            n = lib().mars_hip_tensor_frame_bytes(self.p, self.header.input_tensor_ids[i]) * self.batch
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n,)).reshape(self.batch, -1)

    def pipe_submit(self):
        rc = lib().mars_hip_pipe_submit(self.p)
        if rc != MARS_OK:
            raise MarsError(rc, "mars_hip_pipe_submit")

    def pipe_wait(self, copy=True):
        """-> (outputs: list of uint8 [batch, frame_bytes] or None, dets: list of record arrays per frame or None).
        copy=False returns views into the pipe's pinned result buffers: valid until the SECOND pipe_submit() after this
        call (include/mars_hip.h, mars_hip_pipe_wait)."""
        nout = self.header.num_outputs
        outs = (C.c_void_p * max(nout, 1))()
        dets, counts = C.c_void_p(), C.c_void_p()
        rc = lib().mars_hip_pipe_wait(self.p, outs, C.byref(dets), C.byref(counts))
        if rc != MARS_OK:
            raise MarsError(rc, "mars_hip_pipe_wait")
        want_out, want_det = self._pipe
        ro = rd = None
        if want_out:
            ro = []
            for i in range(nout):
                n = lib().mars_hip_tensor_frame_bytes(self.p, self.header.output_tensor_ids[i]) * self.batch
                a = np.ctypeslib.as_array(C.cast(outs[i], C.POINTER(C.c_uint8)), shape=(n,)).reshape(self.batch, -1)
                ro.append(a.copy() if copy else a)
        if want_det:
            d = np.ctypeslib.as_array(C.cast(dets.value, C.POINTER(C.c_uint8)),
                                      shape=(self.batch * MAX_DET * DET_DTYPE.itemsize,)).view(DET_DTYPE).reshape(self.batch, MAX_DET)
            c = np.ctypeslib.as_array(C.cast(counts.value, C.POINTER(C.c_int32)), shape=(self.batch,))
            rd = [d[f, :c[f]].copy() for f in range(self.batch)] if copy else (d, c)
        return ro, rd

    def pipe_close(self):
        lib().mars_hip_pipe_close(self.p)

    def close(self):
        if self.p:
            lib().mars_free(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- direct kernels with host pointers (include/mxu_ops.h) --------------------
def conv2d_int8(nhwc, x, in_h, in_w, in_c, w, out_c, kh, kw, bias, out_h, out_w, sh, sw, pt, pl,
                in_scale, w_scale, out_scale):
    L = lib()
    fn = L.conv2d_int8_nhwc_mxu if nhwc else L.conv2d_int8_mxu
    x = np.ascontiguousarray(x, dtype=np.int8)
    w = np.ascontiguousarray(w, dtype=np.int8)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.int32)
    out = np.zeros(out_h * out_w * out_c, dtype=np.int8)
    fn(x.ctypes.data, in_h, in_w, in_c, w.ctypes.data, out_c, kh, kw, None if b is None else b.ctypes.data,
       out.ctypes.data, out_h, out_w, sh, sw, pt, pl, in_scale, w_scale, out_scale)
    return out


def conv2d_f32(x, in_h, in_w, in_c, w, out_c, kh, kw, bias, out_h, out_w, sh, sw, pt, pl):
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    out = np.zeros(out_h * out_w * out_c, dtype=np.float32)
    lib().conv2d_float32_mxu(x.ctypes.data, in_h, in_w, in_c, w.ctypes.data, out_c, kh, kw,
                             None if b is None else b.ctypes.data, out.ctypes.data, out_h, out_w, sh, sw, pt, pl, None)
    return out


def parse_output(pred_i8, npred, scale, maxd=1000):
    p = np.ascontiguousarray(pred_i8, dtype=np.int8)
    dets = np.zeros(maxd, dtype=DET_DTYPE)
    n = lib().mars_yolo_parse_output(p.ctypes.data, npred, scale, dets.ctypes.data, maxd)
    if n < 0:
        raise RuntimeError("mars_yolo_parse_output failed")
    return dets[:n].copy()


def nms(dets, thresh=0.45):
    d = np.ascontiguousarray(dets, dtype=DET_DTYPE).copy()
    n = lib().mars_yolo_nms(d.ctypes.data, len(d), thresh)
    if n < 0:
        raise RuntimeError("mars_yolo_nms failed")
    return d[:n].copy()

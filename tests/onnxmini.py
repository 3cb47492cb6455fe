"""Tiny ONNX (protobuf) writer for the compile-step tests: just the messages and fields the compiler reads
(ModelProto.graph -> node / initializer / input / output / value_info; onnx.proto3 field numbers).  Test-side only."""
import struct

import numpy as np

FLOAT, UINT8, INT8, INT32, INT64, FLOAT16 = 1, 2, 3, 6, 7, 10
A_FLOAT, A_INT, A_STRING, A_TENSOR, A_FLOATS, A_INTS = 1, 2, 3, 4, 6, 7


def varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def key(no, wt):
    return varint(no << 3 | wt)


def f_varint(no, v):
    return key(no, 0) + varint(v)


def f_bytes(no, b):
    if isinstance(b, str):
        b = b.encode()
    return key(no, 2) + varint(len(b)) + bytes(b)


def f_f32(no, v):
    return key(no, 5) + struct.pack("<f", v)


def tensor(name, arr=None, dims=None, dtype=None, raw=True, typed=None, packed=True):
    """TensorProto.  raw=True: raw_data; else the typed repeated field (`typed` = 4 float_data, 5 int32_data, 7 int64_data)."""
    out = b""
    if arr is not None:
        arr = np.asarray(arr)
        dims = list(arr.shape) if dims is None else dims
        if dtype is None:
            dtype = {np.dtype("float32"): FLOAT, np.dtype("int8"): INT8, np.dtype("int64"): INT64, np.dtype("int32"): INT32,
                     np.dtype("float16"): FLOAT16, np.dtype("uint8"): UINT8}[arr.dtype]
    for d in dims or []:
        out += f_varint(1, d)
    out += f_varint(2, dtype or 0)
    if arr is not None:
        if raw:
            out += f_bytes(9, arr.tobytes())
        elif typed == 4:
            vals = arr.astype(np.float32).ravel()
            out += f_bytes(4, vals.tobytes()) if packed else b"".join(f_f32(4, float(v)) for v in vals)
        else:
            vals = [int(v) for v in arr.ravel()]
            out += f_bytes(typed, b"".join(varint(v) for v in vals)) if packed else b"".join(f_varint(typed, v) for v in vals)
    out += f_bytes(8, name)
    return out


def attr(name, value, atype=None, with_type=True):
    out = f_bytes(1, name)
    if atype is None:
        if isinstance(value, float):
            atype = A_FLOAT
        elif isinstance(value, int):
            atype = A_INT
        elif isinstance(value, (str, bytes)):
            atype = A_STRING
        elif isinstance(value, (list, tuple)) and value and isinstance(value[0], float):
            atype = A_FLOATS
        else:
            atype = A_INTS
    if atype == A_FLOAT:
        out += f_f32(2, value)
    elif atype == A_INT:
        out += f_varint(3, value)
    elif atype == A_STRING:
        out += f_bytes(4, value)
    elif atype == A_FLOATS:
        out += b"".join(f_f32(7, v) for v in value)
    elif atype == A_INTS:
        out += b"".join(f_varint(8, v) for v in value)
    if with_type:
        out += f_varint(20, atype)
    return out


def node(op, inputs, outputs, name="", **attrs):
    out = b"".join(f_bytes(1, i) for i in inputs) + b"".join(f_bytes(2, o) for o in outputs)
    out += f_bytes(3, name or "%s_%s" % (op, outputs[0] if outputs else "")) + f_bytes(4, op)
    for k, v in attrs.items():
        out += f_bytes(5, v if isinstance(v, bytes) else attr(k, v))
    return out


def value_info(name, dims, elem=FLOAT, shape=True):
    """ValueInfoProto; a None dimension is symbolic (dim_param), shape=False leaves the shape message out."""
    tt = f_varint(1, elem)
    if shape:
        sh = b""
        for d in dims:
            sh += f_bytes(1, f_varint(1, d) if d is not None else f_bytes(2, "N"))
        tt += f_bytes(2, sh)
    return f_bytes(1, name) + f_bytes(2, f_bytes(1, tt))


def model(nodes, inits, inputs, outputs, value_infos=(), name="g", opset=13, producer="onnxmini"):
    g = b"".join(f_bytes(1, n) for n in nodes) + f_bytes(2, name)
    g += b"".join(f_bytes(5, t) for t in inits)
    g += b"".join(f_bytes(11, v) for v in inputs) + b"".join(f_bytes(12, v) for v in outputs)
    g += b"".join(f_bytes(13, v) for v in value_infos)
    return f_varint(1, 8) + f_bytes(2, producer) + f_bytes(7, g) + f_bytes(8, f_bytes(1, "") + f_varint(2, opset))

"""Tiny .mars writer for ad-hoc test graphs (format: include/mars.h; reference mars.h:103-221)."""
import struct

import numpy as np

F32, I32, I16, I8, U8 = 0, 1, 2, 3, 4
NCHW, NDHWC32, OHWI, NHWC, OIHW, D1 = 0, 1, 6, 7, 8, 5
CONV2D, DWCONV, MAXPOOL, AVGPOOL, GAP, RELU, RELU6, LEAKY, SILU, SIGMOID, CONCAT, ADD, MUL, UPSAMPLE, RESHAPE, \
    SOFTMAX, FC, TRANSPOSE, BATCHNORM = range(19)
PAD_VALID, PAD_SAME, PAD_EXPLICIT = 0, 1, 2
NONE = 0xFFFFFFFF


class Graph:
    def __init__(self):
        self.tensors, self.layers, self.blob = [], [], bytearray()

    def tensor(self, shape, dtype=I8, fmt=NHWC, scale=1.0, data=None, name=None, tid=None):
        t = dict(id=len(self.tensors) if tid is None else tid, name=name or "t%d" % len(self.tensors),
                 dtype=dtype, fmt=fmt, shape=list(shape), scale=scale, off=0, size=0)
        if data is not None:
            raw = np.ascontiguousarray(data).tobytes()
            while len(self.blob) % 4:
                self.blob.append(0)
            t["off"], t["size"] = len(self.blob), len(raw)
            self.blob += raw
        self.tensors.append(t)
        return len(self.tensors) - 1

    def layer(self, ltype, ins, outs, params=b""):
        self.layers.append(dict(type=ltype, ins=list(ins), outs=list(outs), params=bytes(params)))

    def conv(self, x, out, w, b=NONE, k=(3, 3), s=(1, 1), pad=PAD_SAME, act=0):
        p = struct.pack("<15I", k[0], k[1], s[0], s[1], 1, 1, pad, 0, 0, 0, 0, 1, act, w, b)
        self.layer(CONV2D, [x], [out], p)

    def pool(self, x, out, k, s):
        self.layer(MAXPOOL, [x], [out], struct.pack("<9I", k[0], k[1], s[0], s[1], 0, 0, 0, 0, 0))

    def upsample(self, x, out, sh, sw):
        self.layer(UPSAMPLE, [x], [out], struct.pack("<3I", sh, sw, 0))

    def concat(self, xs, out, axis=3):
        self.layer(CONCAT, xs, [out], struct.pack("<2I", axis, len(xs)))

    def serialise(self, inputs, outputs, magic=0x5352414D, major=1):
        nt, nl = len(self.tensors), len(self.layers)
        off = 76 + 124 * nt + 112 * nl
        off = (off + 63) & ~63
        ins = list(inputs) + [0] * (4 - len(inputs))
        outs = list(outputs) + [0] * (4 - len(outputs))
        out = bytearray(struct.pack("<IHHIIIIIQQ4I4I", magic, major, 0, 0, nl, nt, len(inputs), len(outputs), off,
                                    len(self.blob), *ins, *outs))
        for t in self.tensors:
            shape = t["shape"] + [0] * (6 - len(t["shape"]))
            out += struct.pack("<I60sIII6iQQfi", t["id"], t["name"].encode()[:59], t["dtype"], t["fmt"],
                               len(t["shape"]), *shape, t["off"], t["size"], t["scale"], 0)
        for i, l in enumerate(self.layers):
            ins = l["ins"][:4] + [0] * (4 - min(len(l["ins"]), 4))
            outs = l["outs"][:4] + [0] * (4 - min(len(l["outs"]), 4))
            out += struct.pack("<IIII4I4I", i, l["type"], len(l["ins"]), len(l["outs"]), *ins, *outs)
            out += l["params"][:64] + b"\0" * (64 - min(len(l["params"]), 64))
        out += b"\0" * (off - len(out))
        out += self.blob
        return bytes(out)


def parse(file_bytes):
    """-> (header dict, tensor dicts, layer dicts) for assertions in tests."""
    d = file_bytes
    magic, vmaj, vmin, flags, nl, nt, ni, no, woff, wsz = struct.unpack_from("<IHHIIIIIQQ", d, 0)
    ins = struct.unpack_from("<4I", d, 44)[:ni]
    outs = struct.unpack_from("<4I", d, 60)[:no]
    tensors, layers = [], []
    for i in range(nt):
        o = 76 + 124 * i
        tid, = struct.unpack_from("<I", d, o)
        dtype, fmt, nd = struct.unpack_from("<III", d, o + 64)
        shape = struct.unpack_from("<6i", d, o + 76)[:nd]
        doff, dsz = struct.unpack_from("<QQ", d, o + 100)
        scale, = struct.unpack_from("<f", d, o + 116)
        tensors.append(dict(id=tid, dtype=dtype, fmt=fmt, shape=shape, off=doff, size=dsz, scale=scale))
    for i in range(nl):
        o = 76 + 124 * nt + 112 * i
        lid, typ, n_in, n_out = struct.unpack_from("<IIII", d, o)
        layers.append(dict(id=lid, type=typ, ins=struct.unpack_from("<4I", d, o + 16)[:n_in],
                           outs=struct.unpack_from("<4I", d, o + 32)[:n_out], params=d[o + 48:o + 112]))
    return dict(layers=nl, tensors=nt, inputs=ins, outputs=outs, woff=woff, wsz=wsz), tensors, layers


def tensor_nbytes(t):
    n = 1
    for s in t["shape"]:
        n *= max(s, 0)
    return n * (4 if t["dtype"] in (F32, I32) else 2 if t["dtype"] == I16 else 1)

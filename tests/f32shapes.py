"""The float32 shape lists the GPU tests force conv_f32_split / conv_f32_patch / conv_f32_stem with, and the graphs they run -- shared
with tests/golden/make_golden.py, which pins every one of them to the REFERENCE's own output (golden.json "conv_f32_family": digests of
the output tensor the reference's conv2d_float32_mxu + float SIGMOID / MUL / ADD leave for each input frame), so that the restatement is
not the only witness of these shapes: the GPU's bit-identical mode (f32_mfma = 0) must reproduce the digests, and
tests/test_oracle.py checks the restatement against them on the CPU.  All data from numpy's default_rng with fixed seeds."""
import numpy as np

import marsfile

SPLIT = [
    # h, w, in_c, out_c, k, stride, pad, batch, silu       which path of conv_f32_split (round 4)
    (7, 9, 3, 5, 3, 1, "same", 3, True),       # odd map width: one dword per tap (GATHER 0), a 32-channel tile 5 channels full
    (16, 20, 8, 40, 3, 1, "same", 5, True),    # stride 1, 16-byte gathers (4 pixels x 1 tap), 64-channel tile
    (17, 18, 4, 130, 3, 2, "same", 2, False),  # stride 2, odd output width: not a split shape -> falls back to conv_f32_mfma
    (32, 32, 16, 128, 3, 2, "same", 4, True),  # stride 2, kernel rows padded to 4 taps, 2 pixels x 2 taps per load, 128-channel tile
    (48, 48, 8, 16, 3, 1, "same", 64, True),   # 576 pixel tiles on 512 slots: every workgroup walks two tiles (the K pipeline
                                               # runs through the tile boundary), the last ones one
    (64, 64, 3, 32, 6, 2, "same", 9, True),    # the stem's geometry: 6 x 6, stride 2, 3 channels
    (12, 12, 24, 200, 1, 1, "same", 7, False), # 1 x 1, two 128-channel tiles (the second 72 channels full), K = 24 < one step
    (20, 24, 6, 12, 3, 1, "valid", 3, True),   # no padding: no tap ever starts left of the image
    (9, 16, 5, 7, 5, 1, "same", 2, True),      # 5 x 5, pad 2: more than one column left of the image -> GATHER 0
]

PATCH = [
    # h, w, in_c, out_c, k, stride, batch, silu, add          conv_f32_patch (round 5): which geometry
    (20, 20, 32, 16, 3, 1, 3, True, False),    # whole-row tiles that run on into the next frame; 32-channel tile half full; 4 dummy units
    (12, 40, 64, 72, 3, 1, 5, True, True),     # 40-wide map, 128-channel tile 72 channels full, fused residual Add
    (24, 160, 32, 64, 3, 1, 2, False, False),  # wide map: 32-column strips (2-D tiles), 64-channel tile
    (40, 40, 32, 130, 3, 2, 3, True, False),   # stride 2 (de-interleaved patch columns), two 128-channel tiles (the second 2 channels full)
    (32, 160, 64, 32, 3, 2, 2, True, False),   # stride 2 onto an 80-wide map: 16-column strips, 34 patch rows
    (16, 24, 32, 20, 5, 1, 4, False, False),   # 5 x 5, pad 2: 25 taps per chunk
    (9, 20, 96, 128, 3, 1, 37, True, True),    # short frames: several frame boundaries per tile; 12 chunks; 27 tiles
    (80, 80, 64, 64, 3, 1, 9, True, False),    # 225 tiles: every workgroup walks a run of tiles (the ring runs through tile boundaries)
]

STEM = [
    # h, w, in_c, out_c, k, pad-as-SAME, batch, silu        conv_f32_stem (round 5)
    (64, 64, 3, 32, 6, 9, True),     # the twins' first layer at 64 x 64: 2 x 1 tiles per frame, 9 frames
    (128, 192, 3, 32, 6, 5, True),   # 4 x 3 tiles per frame: interior tiles and every edge
    (32, 64, 3, 20, 6, 3, False),    # 20 of the 32 channel rows
    (96, 64, 1, 32, 6, 4, True),     # one channel
    (96, 64, 3, 32, 4, 2, True),     # 4 x 4 under SAME padding has pad 1: odd, declined (conv_f32_split takes it)
    (64, 128, 4, 16, 6, 7, True),    # four channels (every slot real)
    (32, 64, 2, 32, 8, 2, True),     # 8 x 8 would be 32 units: declined (conv_f32_split takes it)
]


def shape_id(fam, shape):
    return fam + ":" + "x".join(str(q) for q in shape)


def _conv_chain(rng, h, w, ic, oc, k, st, pad, silu, add):
    """conv (+ SIGMOID / MUL) (+ ADD with a second graph input) on NCHW floats -> (file bytes, output tensor, out_h, out_w)"""
    G = marsfile.Graph()
    F, N = marsfile.F32, marsfile.NCHW
    if pad == "same":
        oh, ow = (h + st - 1) // st, (w + st - 1) // st
    else:
        oh, ow = (h - k) // st + 1, (w - k) // st + 1
    x = G.tensor([1, ic, h, w], dtype=F, fmt=N)
    a = G.tensor([1, oc, oh, ow], dtype=F, fmt=N)
    amp = 1.7 / (k * k * ic) ** 0.5
    wt = G.tensor([oc, ic, k, k], dtype=F, fmt=marsfile.OIHW, data=((rng.random((oc, ic, k, k), dtype=np.float32) * 2 - 1) * amp).astype(np.float32))
    b = G.tensor([oc], dtype=F, fmt=marsfile.D1, data=((rng.random(oc, dtype=np.float32) * 2 - 1) * 0.1).astype(np.float32))
    G.conv(x, a, wt, b, (k, k), (st, st), pad=marsfile.PAD_SAME if pad == "same" else marsfile.PAD_VALID)
    out = a
    if silu:
        g_, o_ = G.tensor([1, oc, oh, ow], dtype=F, fmt=N), G.tensor([1, oc, oh, ow], dtype=F, fmt=N)
        G.layer(marsfile.SIGMOID, [a], [g_])
        G.layer(marsfile.MUL, [a, g_], [o_])
        out = o_
    ins = [x]
    if add:  # the C3 shortcut: Add(conv-chain result, another tensor of the same shape) folded into the convolution's epilogue
        r_ = G.tensor([1, oc, oh, ow], dtype=F, fmt=N)
        s_ = G.tensor([1, oc, oh, ow], dtype=F, fmt=N)
        G.layer(marsfile.ADD, [out, r_], [s_])
        ins.append(r_)
        out = s_
    return G.serialise(ins, [out]), out, oh, ow


def split_case(shape):
    """-> dict(d, out, B, xs = input frames (float32 arrays, frame f takes xs[f % len(xs)]), rs = None)"""
    h, w, ic, oc, k, st, pad, B, silu = shape
    rng = np.random.default_rng(h * 1000 + w * 10 + k)
    d, out, oh, ow = _conv_chain(rng, h, w, ic, oc, k, st, pad, silu, False)
    xs = [(rng.random(ic * h * w, dtype=np.float32) * 2 - 1).astype(np.float32) for _ in range(min(B, 4))]
    return dict(d=d, out=out, B=B, xs=xs, rs=None, oh=oh, ow=ow)


def patch_case(shape):
    h, w, ic, oc, k, st, B, silu, add = shape
    rng = np.random.default_rng(h * 1000 + w * 10 + k + st)
    d, out, oh, ow = _conv_chain(rng, h, w, ic, oc, k, st, "same", silu, add)
    nx = min(B, 4)
    xs = [(rng.random(ic * h * w, dtype=np.float32) * 2 - 1).astype(np.float32) for _ in range(nx)]
    rs = [(rng.random(oc * oh * ow, dtype=np.float32) * 2 - 1).astype(np.float32) for _ in range(nx)]
    return dict(d=d, out=out, B=B, xs=xs, rs=rs if add else None, oh=oh, ow=ow)


def stem_case(shape):
    h, w, ic, oc, k, B, silu = shape
    rng = np.random.default_rng(h * 1000 + w * 10 + k + ic)
    d, out, oh, ow = _conv_chain(rng, h, w, ic, oc, k, 2, "same", silu, False)
    xs = [(rng.random(ic * h * w, dtype=np.float32) * 2 - 1).astype(np.float32) for _ in range(min(B, 3))]
    return dict(d=d, out=out, B=B, xs=xs, rs=None, oh=oh, ow=ow)


FAMILIES = (("split", SPLIT, split_case), ("patch", PATCH, patch_case), ("stem", STEM, stem_case))


def reference_digests(case, make_runner, digest):
    """digest of the output tensor for every input frame of a case, through `make_runner(file_bytes)` (refbind.O2Model / orcbind.Graph)"""
    res = []
    for i, q in enumerate(case["xs"]):
        g = make_runner(case["d"])
        g.set_input(0, q.tobytes())
        if case["rs"] is not None:
            g.set_input(1, case["rs"][i].tobytes())
        assert g.run() == 0
        res.append(digest(g.tensor(case["out"])))
        g.close()
    return res

"""Seeded test cases shared by the golden generator, the oracle tests and the GPU parity tests.

All data derive from conftest.lcg_frame (a plain 32-bit LCG), so they are
identical on every machine and numpy version.
"""
import hashlib

import numpy as np

from conftest import lcg_frame


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).tobytes()).hexdigest()[:24]


def i8(seed, n):
    return lcg_frame(seed, n).view(np.int8)


def f32(seed, n, lo=-1.0, hi=1.0):
    u = lcg_frame(seed, n).astype(np.float32) / np.float32(255.0)
    return (u * np.float32(hi - lo) + np.float32(lo)).astype(np.float32)


# (name, nhwc, in_h, in_w, in_c, out_c, kh, kw, sh, sw, pad_top, pad_left, out_h, out_w, in_s, w_s, out_s, bias)
CONV_I8_CASES = [
    ("k3_c16", 1, 12, 14, 16, 32, 3, 3, 1, 1, 1, 1, 12, 14, 0.02, 0.004, 0.05, True),
    ("k3_c64_s2", 1, 17, 19, 64, 64, 3, 3, 2, 2, 0, 0, 9, 10, 0.03, 0.002, 0.08, True),
    ("k1_c32_o255", 1, 9, 7, 32, 255, 1, 1, 1, 1, 0, 0, 9, 7, 0.03, 0.01, 0.05, True),   # head conv, N tail
    ("k1_c128_o16", 1, 6, 6, 128, 16, 1, 1, 1, 1, 0, 0, 6, 6, 0.03, 0.003, 0.05, False),
    ("stem_k6_c3", 1, 32, 32, 3, 16, 6, 6, 2, 2, 2, 2, 16, 16, 0.0157, 0.003, 0.04, True),  # K=108, row pad
    ("k3_c3_valid", 1, 20, 20, 3, 16, 3, 3, 1, 1, 0, 0, 18, 18, 1.0, 0.0026, 0.9, True),   # tiny_160 layer 0 shape
    ("k5_c24_pad", 1, 11, 13, 24, 40, 5, 5, 1, 1, 2, 2, 11, 13, 0.02, 0.003, 0.07, True),  # in_c % 16 != 0
    ("k3_c48_asym", 1, 10, 10, 48, 48, 3, 1, 1, 2, 1, 0, 10, 5, 0.02, 0.004, 0.06, True),
    ("ties_half", 1, 8, 8, 16, 32, 1, 1, 1, 1, 0, 0, 8, 8, 1.0, 0.5, 16.0, False),          # acc*cs hits x.5 often
    ("overflow", 1, 8, 8, 16, 32, 3, 3, 1, 1, 1, 1, 8, 8, 1.0, 13272.3, 1e-6, True),        # +/-overflow -> -128
    ("nan_scale", 1, 4, 4, 16, 32, 1, 1, 1, 1, 0, 0, 4, 4, 0.0, 1.0, 0.0, True),            # 0/0 -> NaN -> -128
    ("neg_scale", 1, 6, 6, 32, 32, 3, 3, 1, 1, 1, 1, 6, 6, 0.02, -0.004, 0.05, True),
    ("big_k", 1, 5, 5, 512, 64, 3, 3, 1, 1, 1, 1, 5, 5, 0.03, 0.0007, 0.05, True),          # K = 4608
    ("out_gt_in", 1, 5, 5, 16, 32, 3, 3, 1, 1, 0, 0, 7, 7, 0.03, 0.004, 0.05, True),        # output larger than valid
    ("nchw_k3", 0, 12, 14, 3, 16, 3, 3, 1, 1, 0, 0, 10, 12, 1.0, 0.0026, 0.5, True),
    ("nchw_k3_c16_s2", 0, 15, 15, 16, 32, 3, 3, 2, 2, 1, 1, 8, 8, 0.03, 0.004, 0.05, True),
    ("nchw_k1_c40", 0, 7, 9, 40, 24, 1, 1, 1, 1, 0, 0, 7, 9, 0.03, 0.006, 0.05, False),
]

# (name, in_h, in_w, in_c, out_c, kh, kw, sh, sw, pt, pl, out_h, out_w, bias)
CONV_F32_CASES = [
    ("f_k3", 10, 12, 3, 16, 3, 3, 1, 1, 0, 0, 8, 10, True),
    ("f_k3_pad_s2", 13, 13, 8, 12, 3, 3, 2, 2, 1, 1, 7, 7, True),
    ("f_k1", 6, 6, 32, 20, 1, 1, 1, 1, 0, 0, 6, 6, False),
    ("f_k5", 9, 9, 4, 6, 5, 5, 1, 1, 2, 2, 9, 9, True),
]


# Shape families the GPU tests use to FORCE particular launch forms (tests/test_gpu_kernels.py): SAME-padded NHWC layers at
# sizes the reference's scalar C finishes in under a second.  Listed here -- not inside the GPU tests -- so that
# tests/golden/make_golden.py pins every one of them to the reference's own output (golden.json "conv_i8_family") and
# tests/test_oracle.py checks the restatement against it on the CPU: a GPU kernel and the restatement can then not share a bug
# on these shapes.  name -> (seed, [(in_h, in_w, in_c, out_c, kh, kw, stride)]).
CONV_I8_FAMILIES = {
    "big": (4, [(40, 40, 64, 128, 3, 3, 2), (20, 20, 256, 256, 1, 1, 1), (23, 17, 128, 64, 3, 3, 1), (16, 16, 512, 255, 1, 1, 1),
                (64, 64, 32, 32, 1, 1, 1), (48, 48, 3, 32, 6, 6, 2)]),
    "stem": (6, [(37, 53, 3, 32, 6, 6, 2), (70, 41, 3, 16, 3, 3, 1), (9, 3, 3, 32, 3, 3, 1), (33, 6, 3, 48, 6, 6, 2), (18, 50, 1, 32, 5, 5, 2),
                 (11, 13, 4, 64, 3, 3, 1), (21, 35, 2, 16, 6, 6, 2), (130, 131, 3, 32, 6, 6, 2),
                 (16, 128, 3, 32, 7, 7, 4), (40, 90, 2, 48, 8, 8, 3)]),  # the last two: patches wider than the staged kernel holds
    "walk": (5, [(64, 64, 32, 32, 1, 1, 1), (40, 40, 64, 128, 3, 3, 2), (37, 29, 128, 64, 1, 1, 1), (23, 17, 128, 64, 3, 3, 1),
                 (33, 31, 256, 256, 1, 1, 1), (50, 50, 64, 64, 1, 1, 1), (80, 80, 16, 48, 3, 3, 1),
                 # ragged channel runs (unaligned rows, 8+4+2+1-byte tail stores): the 255-channel heads and odd widths
                 (16, 16, 512, 255, 1, 1, 1), (21, 19, 64, 81, 1, 1, 1), (20, 20, 32, 7, 3, 3, 1), (17, 23, 128, 131, 1, 1, 1)]),
    "wres": (9, [(64, 64, 32, 32, 1, 1, 1), (40, 40, 64, 128, 3, 3, 2), (37, 29, 128, 64, 1, 1, 1), (23, 17, 128, 64, 3, 3, 1),
                 (33, 31, 256, 256, 1, 1, 1), (50, 50, 64, 64, 1, 1, 1), (16, 16, 512, 255, 1, 1, 1), (21, 19, 64, 81, 1, 1, 1),
                 (20, 20, 32, 7, 3, 3, 1), (17, 23, 128, 131, 1, 1, 1)]),
    "wide": (7, [(40, 40, 128, 128, 3, 3, 1), (23, 17, 256, 256, 3, 3, 1), (33, 31, 128, 256, 1, 1, 1), (20, 20, 512, 255, 1, 1, 1),
                 (19, 21, 64, 128, 3, 3, 2), (16, 16, 1024, 128, 1, 1, 1), (21, 19, 128, 128, 3, 3, 2), (9, 11, 256, 128, 5, 5, 1)]),
    # conv_i8_rows (launch variant 20) takes 3x3 stride-1 layers on 20- / 40-wide maps with in_c a multiple of 128 and an even
    # number of 64-channel chunks; every shape below is one it accepts (a launch counter in the test proves it ran)
    "rows": (9, [(40, 40, 128, 128, 3, 3, 1), (20, 20, 256, 256, 3, 3, 1), (23, 40, 128, 256, 3, 3, 1), (9, 20, 512, 128, 3, 3, 1),
                 (7, 40, 256, 128, 3, 3, 1), (31, 20, 128, 128, 3, 3, 1)]),
    "patch": (6, [(48, 48, 64, 64, 3, 3, 1), (64, 64, 32, 64, 3, 3, 2), (47, 45, 32, 32, 3, 3, 1), (61, 63, 64, 32, 3, 3, 2),
                  (32, 48, 32, 48, 5, 5, 1), (40, 32, 64, 16, 3, 1, 1), (64, 64, 32, 32, 1, 3, 2), (33, 31, 64, 128, 3, 3, 1),
                  # 16 input channels (round 6: the yolov5n models' second layer; one 16-byte unit per pixel)
                  (64, 64, 16, 32, 3, 3, 2), (48, 47, 16, 16, 3, 3, 1), (32, 48, 16, 64, 5, 5, 1), (63, 61, 16, 48, 3, 3, 2)]),
}


def family_cases(name):
    """the CONV_I8_CASES-style tuples of one family: NHWC, SAME-style padding (the split the GPU tests have always used), bias,
    weight scale shrinking with K so that the int8 outputs spread over the whole range"""
    seed, shapes = CONV_I8_FAMILIES[name]
    out = []
    for i, (h, w, ic, oc, kh, kw, s) in enumerate(shapes):
        oh, ow = (h + s - 1) // s, (w + s - 1) // s
        ph = max((oh - 1) * s + kh - h, 0) // 2
        pw = max((ow - 1) * s + kw - w, 0) // 2
        out.append(("%s%d" % (name, i), 1, h, w, ic, oc, kh, kw, s, s, ph, pw, oh, ow, 0.03, 0.003 / (kh * kw * ic) ** 0.5 * 8, 0.05, True))
    return seed, out


def conv_i8_inputs(case, seed=1):
    (name, nhwc, in_h, in_w, in_c, out_c, kh, kw, sh, sw, pt, pl, out_h, out_w, in_s, w_s, out_s, has_b) = case
    x = i8(seed * 7919 + 1, in_h * in_w * in_c)
    w = i8(seed * 7919 + 2, out_c * kh * kw * in_c)
    b = None
    if has_b:
        b = (lcg_frame(seed * 7919 + 3, out_c * 4).view(np.int32) >> 18).astype(np.int32)
    return x, w, b


def conv_i8_call(fn, case, seed=1):
    (name, nhwc, in_h, in_w, in_c, out_c, kh, kw, sh, sw, pt, pl, out_h, out_w, in_s, w_s, out_s, has_b) = case
    x, w, b = conv_i8_inputs(case, seed)
    return fn(nhwc, x, in_h, in_w, in_c, w, out_c, kh, kw, b, out_h, out_w, sh, sw, pt, pl,
              np.float32(in_s), np.float32(w_s), np.float32(out_s))


def conv_f32_call(fn, case, seed=1):
    (name, in_h, in_w, in_c, out_c, kh, kw, sh, sw, pt, pl, out_h, out_w, has_b) = case
    x = f32(seed * 104729 + 1, in_h * in_w * in_c)
    w = f32(seed * 104729 + 2, out_c * in_c * kh * kw, -0.5, 0.5)
    b = f32(seed * 104729 + 3, out_c) if has_b else None
    return fn(x, in_h, in_w, in_c, w, out_c, kh, kw, b, out_h, out_w, sh, sw, pt, pl)


# detection tail: (name, npred, scale, seed, transform)
YOLO_CASES = [
    ("rand_0p05", 25200, 0.05, 11, "raw"),          # the survey probe: cap of 1000 reached
    ("sparse", 6300, 0.04, 12, "sparse"),           # few candidates
    ("ties", 4000, 0.1, 13, "ties"),                # many equal confidences -> exchange-sort permutation
    ("clustered", 3000, 0.05, 14, "cluster"),       # overlapping same-class boxes -> suppression
    ("none", 500, 0.05, 15, "none"),                # no candidate at all
]


def yolo_pred(case):
    name, npred, scale, seed, kind = case
    p = i8(seed, npred * 85).reshape(npred, 85).copy()
    if kind == "sparse":
        p[:, 4] = np.where(np.arange(npred) % 97 == 0, p[:, 4] | 64, -100).astype(np.int8)
    elif kind == "ties":
        p[:, 4] = np.int8(40)
        p[:, 5:] = (p[:, 5:] // 32 * 32).astype(np.int8)          # few distinct logits -> equal conf
        p[:, 0:4] = (np.abs(p[:, 0:4].astype(np.int16)) // 2).astype(np.int8)
    elif kind == "cluster":
        p[:, 4] = np.int8(60)
        p[:, 5:] = np.int8(-90)
        p[np.arange(npred), 5 + (np.arange(npred) % 3)] = (40 + (np.arange(npred) % 50)).astype(np.int8)
        p[:, 0] = (np.arange(npred) % 7 * 3).astype(np.int8)
        p[:, 1] = (np.arange(npred) // 7 % 5 * 3).astype(np.int8)
        p[:, 2] = np.int8(40)
        p[:, 3] = np.int8(36)
    elif kind == "none":
        p[:, 4] = np.int8(-120)
    return np.ascontiguousarray(p.reshape(-1)), npred, np.float32(scale)


SHIPPED = ["test_model", "test_simple", "tiny_160_int8", "tiny_160_f32", "yolov5n_int8", "yolov5nu", "yolov5n"]
# synthetic graphs small enough for the CPU oracle: (name, synth kwargs)
SYNTH = [
    ("v5n_64", dict(width_x16=4, input_hw=64, seed=1)),
    ("v5s_96", dict(width_x16=8, input_hw=96, seed=2)),
    ("tiny_40", dict(tiny=True, input_hw=40, seed=3)),
    ("v5n_64_nchw", dict(width_x16=4, input_hw=64, nchw_int8=True, seed=4)),
    ("tiny_32_f32", dict(tiny=True, input_hw=32, float32=True, seed=5)),
    ("v5n_64_f32", dict(width_x16=4, input_hw=64, float32=True, seed=6)),   # config 5 topology (yolov5 f32), small
    # the same topology at a size where EVERY map width is a multiple of 4 (64 ... 4), as at config 5's 640 x 640 (160 ... 20): the
    # reference's CONCAT copies runs of shape[3] = W BYTES at byte offsets that are multiples of W (mars_runtime.c:977-996), so with
    # W % 4 != 0 (the 2-wide map of the 64 x 64 twin, the 10- / 5-wide ones of a 160 x 160 twin) floats are cut between bytes and
    # 1e38-sized values appear; with W % 4 == 0 whole floats move and every tensor stays O(0.1) -- the WELL-CONDITIONED float twin
    ("v5n_128_f32", dict(width_x16=4, input_hw=128, float32=True, seed=9)),
    # per-convolution scales: no two fused tables (or folded-Add factors) are equal, so a launch that picks up a
    # neighbour's table shows (the plain twins share one scale per tensor kind and once hid exactly that in a pair)
    ("v5n_64_vs", dict(width_x16=4, input_hw=64, seed=7, vary_scales=True)),
    ("v5s_96_vs", dict(width_x16=8, input_hw=96, seed=8, vary_scales=True)),
]


# image front-end: (name, src_w, src_h, target_w, target_h, nhwc, seed, kind)
#   kind "noise" = uniform random bytes, "smooth" = low-frequency gradients + a few hard edges (exercises the
#   negative lobes of both kernels against saturation), "flat" = constant colour
LETTERBOX_CASES = [
    ("down_4x3", 64, 48, 32, 32, 1, 1, "noise"), ("up_3x4", 48, 64, 128, 128, 0, 2, "noise"),
    ("down_odd", 100, 75, 64, 64, 1, 3, "smooth"), ("up_odd", 33, 57, 96, 80, 0, 4, "smooth"),
    ("vga_160", 640, 480, 160, 160, 1, 5, "noise"), ("tiny_up", 37, 41, 320, 320, 1, 6, "noise"),
    ("same", 64, 64, 64, 64, 0, 7, "noise"), ("almost_same", 65, 64, 64, 64, 1, 8, "smooth"),
    ("tall", 31, 200, 96, 96, 1, 9, "noise"), ("wide", 200, 31, 96, 96, 0, 10, "noise"),
    ("primes", 17, 19, 23, 29, 1, 11, "noise"), ("flat", 50, 40, 64, 64, 1, 12, "flat"),
    ("hd_640", 1280, 720, 640, 640, 1, 13, "smooth"),
]


def letterbox_image(case):
    name, w, h, tw, th, nhwc, seed, kind = case
    if kind == "noise":
        return lcg_frame(0x1AA60000 + seed, w * h * 3).reshape(h, w, 3).copy()
    if kind == "flat":
        return np.full((h, w, 3), 200, dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255) // max(w - 1, 1), (yy * 255) // max(h - 1, 1), ((xx + yy) * 7) % 256], axis=2)
    img[h // 3: h // 3 + 2, :, :] = 255   # hard edges: ringing of the cubic kernels must clamp like the reference
    img[:, w // 2: w // 2 + 1, :] = 0
    return img.astype(np.uint8)


# reference tensor_byte_size() (mars_runtime.c:80-124): (dtype, format tag, shape); formats: 0 NCHW, 1 NDHWC32, 2 HWIO,
# 3 NMHWSOIB2, 4 NMC32, 5 D1, 6 OHWI, 7 NHWC, 8 OIHW; dtypes: 0 f32, 1 i32, 2 i16, 3 i8, 4 u8, 5 u4
TBS_CASES = [
    (3, 7, [1, 640, 640, 3]), (3, 0, [1, 3, 640, 640]), (0, 0, [1, 3, 160, 160]), (1, 5, [255]), (2, 7, [1, 7, 9, 5]),
    (3, 1, [1, 3, 160, 160]), (3, 1, [1, 64, 154, 154]), (3, 1, [1, 33, 5, 7]), (0, 1, [1, 16, 80, 80]), (3, 1, [2, 32, 4, 4]),
    (3, 1, [1, 25200, 85]),            # NDHWC32 with 3 dims: the plain product
    (3, 3, [16, 3, 3, 3]), (3, 3, [255, 128, 1, 1]), (3, 3, [33, 65, 3, 3]), (0, 3, [64, 32, 3, 3]),
    (5, 7, [1, 5, 5, 3]), (5, 7, [1, 4, 4, 4]), (5, 1, [1, 40, 6, 6]), (4, 8, [16, 3, 3, 3]), (7, 7, [1, 2, 3, 4]),
    (3, 7, [0, 0, 0, 0]), (3, 7, []), (0, 6, [16, 6, 6, 3]), (3, 4, [1, 100, 32]),
]

"""can the epilogue's float requantisation  k = trunc(fl(fl(acc) * c2))  (c2 = 2 * cs; the index of the
half-step table, conv_i8_common.hpp) be replaced by ONE integer instruction  k' = (acc * M) >> 32  (v_mul_hi_i32_i24: 24-bit
signed operands) with a per-layer proof?

For every int8 convolution of a model this script
  1. bounds the accumulator: B = max over output channels of (sum |w| * 128 + |bias|)   (inputs are int8);
  2. builds the EXACT thresholds of the float form by bisection with numpy float32 arithmetic (it is monotone in acc):
     T[k] = the smallest acc >= 0 with trunc(f32(acc) * c2) >= k, k = 1 .. 255 (the form is odd, negatives mirror);
  3. searches M (all 24-bit candidates around c2 * 2^32 / 2^s for the pre-shifts s that keep acc << s inside 24 bits) such
     that floor(acc * M / 2^32) steps at exactly the same accumulators for every threshold below B -- two monotone step
     functions that agree at every step of one of them and have no other steps are equal on [-B, B].
Prints, per layer: cs, B, bits needed, whether acc fits 24 bits at all, and whether ANY (s, M) passes.

Result on the yolov5s twin (profiles/r04_requant_int_study.txt): see DESIGN.md section 6.

    python tools/requant_int_study.py [--width 8] [--hw 640]   (CPU only: reads the synthetic twin through the host library)"""
import argparse
import os
import struct
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "thingino-accel_amd"))
import marsfile  # noqa: E402
import marsrt as M  # noqa: E402

F = np.float32


def float_k(acc, c2):
    return np.trunc(acc.astype(F) * F(c2)).astype(np.int64)


def thresholds(c2, bound):
    """T[k] for k = 1..255 (np.inf when beyond the bound)"""
    T = []
    for k in range(1, 256):
        lo, hi = 0, int(bound) + 1  # invariant: f(lo) < k <= f(hi) (if f(hi) < k: unreachable)
        if float_k(np.array([hi], dtype=np.int64), c2)[0] < k:
            T.append(None)
            continue
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if float_k(np.array([mid], dtype=np.int64), c2)[0] >= k:
                hi = mid
            else:
                lo = mid
        T.append(hi)
    return T


def try_integer(c2, bound, T):
    """-> (shift, M) or None"""
    if bound >= 1 << 23:
        return None, "accumulator exceeds 24 bits"
    best = None
    for s in range(0, 24):
        if (int(bound) << s) >= (1 << 23):
            break
        m0 = float(np.float64(F(c2))) * 2.0 ** (32 - s)
        if m0 >= (1 << 23):
            continue  # the multiplier does not fit 24 bits at this pre-shift
        for dm in range(-3, 4):
            Mi = int(round(m0)) + dm
            if Mi <= 0 or Mi >= (1 << 23):
                continue
            ok = True
            for k, t in enumerate(T, start=1):
                if t is None:
                    break
                # floor(((t << s) * M) >> 32) must be >= k at t and < k at t - 1
                if (((t << s) * Mi) >> 32) < k or ((((t - 1) << s) * Mi) >> 32) >= k:
                    ok = False
                    break
            if ok:
                best = (s, Mi)
                break
        if best:
            break
    return best, ("no 24-bit multiplier reproduces every float threshold" if best is None else "ok")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=8)
    ap.add_argument("--hw", type=int, default=640)
    ap.add_argument("--vary-scales", action="store_true")
    a = ap.parse_args()
    d = M.synth_model(width_x16=a.width, input_hw=a.hw, seed=1, vary_scales=a.vary_scales)
    hdr, tensors, layers = marsfile.parse(d)
    blob = d[hdr["weights_offset"]:] if "weights_offset" in hdr else None
    npass = nfit = n = 0
    print("layer  K      cs            bound      T[255]    fits24  integer form")
    for L in layers:
        if L["type"] != marsfile.CONV2D:
            continue
        tin, tout = tensors[L["ins"][0]], tensors[L["outs"][0]]
        # mars_conv_params_t (include/mars.h): 13 words of geometry, then weight / bias tensor ids
        cp = struct.unpack_from("<15I", L["params"], 0)
        byid = {t["id"]: t for t in tensors}
        wt, bt = byid[cp[13]], byid.get(cp[14])

        def data(t):
            return d[hdr["woff"] + t["off"]: hdr["woff"] + t["off"] + t["size"]]
        w = np.frombuffer(data(wt), dtype=np.int8).reshape(wt["shape"][0], -1).astype(np.int64)
        b = np.frombuffer(data(bt), dtype=np.int32).astype(np.int64) if bt else np.zeros(w.shape[0], np.int64)
        cs = F(F(tin["scale"]) * F(wt["scale"])) / F(tout["scale"])
        c2 = F(cs) * F(2.0)
        bound = int((np.abs(w).sum(axis=1) * 128 + np.abs(b[:w.shape[0]])).max())
        T = thresholds(c2, bound)
        res, why = try_integer(c2, bound, T)
        n += 1
        nfit += bound < (1 << 23)
        npass += res is not None
        t255 = T[-1] if T[-1] is not None else max(t for t in T if t is not None)
        print("%-5d %-6d %-13.6g %-10d %-9d %-7s %s" % (L["id"], w.shape[1], cs, bound, t255, bound < (1 << 23),
                                                       ("shift %d, M %d" % res) if res else why))
    print("\n%d convolutions: %d with accumulators inside 24 bits, %d with a proven one-instruction integer form" % (n, nfit, npass))


if __name__ == "__main__":
    main()

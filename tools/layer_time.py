"""Time ONE convolution (+ fused SiLU, optionally + fused residual Add) of the yolov5s twin's shapes at batch 256 under
several launch configurations, and check that every configuration writes the same bytes (the first one also against the
CPU oracle on frame 0).  The harness behind the per-layer numbers in DESIGN.md / profiles/*_experiments.md.

    python tools/layer_time.py D40 D20 --cfg default --cfg variant=20 --cfg variant=13
    python tools/layer_time.py 40,40,128,128,3,1,0 --cfg variant=20,persist_slots=64
    BATCH=64 LIB=path/to/other/libnna_mars.so python tools/layer_time.py L3

A layer is a name from LAYERS or h,w,in_c,out_c,k,stride,add.  A configuration is `default` or key=value[,key=value...]
with the keys of mars_hip_set_tuning.  Times are the sum of the per-launch HIP-event times of the graph (one launch for a
fused layer), best of 5."""
import argparse
import os
import sys
import zlib

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "thingino-accel_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import marsfile  # noqa: E402
import marsrt as M  # noqa: E402

LAYERS = {  # yolov5s twin at 640 x 640: h, w, in_c, out_c, k, stride, fused Add
    "L0": (640, 640, 3, 32, 6, 2, False),
    "L3": (320, 320, 32, 64, 3, 2, False),
    "L15": (160, 160, 32, 32, 3, 1, True),
    "L23": (160, 160, 64, 128, 3, 2, False),
    "L35": (80, 80, 64, 64, 3, 1, True),
    "L44": (80, 80, 64, 64, 3, 1, False),
    "D40": (40, 40, 128, 128, 3, 1, False),
    "D40a": (40, 40, 128, 128, 3, 1, True),
    "D80s2": (80, 80, 128, 256, 3, 2, False),
    "D80": (80, 80, 128, 128, 3, 1, False),
    "D20": (20, 20, 256, 256, 3, 1, False),
    "D20a": (20, 20, 256, 256, 3, 1, True),
    "D40s2": (40, 40, 256, 512, 3, 2, False),
    "D20b": (20, 20, 512, 512, 3, 1, False),
    "D160s2": (160, 160, 64, 128, 3, 2, False),
    "P40": (40, 40, 512, 256, 1, 1, False),
    "P20": (20, 20, 1024, 512, 1, 1, False),
}


def build_f32(h, w, ic, oc, k, s, add=False, chain=False):
    """the float32 form of the same layer: NCHW / OIHW float32 convolution + SIGMOID + MUL (folded into the conv's epilogue).  chain: a
    1 x 1 convolution ic -> ic (+ SIGMOID + MUL) in front of it (a C3 bottleneck's pair: the planner keeps the tensor between the two in
    record format under f32_mfma = 3; `norec=1` in a --cfg switches that off) and, with add, the shortcut Add(result, x)"""
    G = marsfile.Graph()
    rng = np.random.default_rng(1)
    F, N = marsfile.F32, marsfile.NCHW
    x = G.tensor([1, ic, h, w], dtype=F, fmt=N)
    x_in = x
    if chain:
        a0 = G.tensor([1, ic, h, w], dtype=F, fmt=N)
        g0 = G.tensor([1, ic, h, w], dtype=F, fmt=N)
        o0 = G.tensor([1, ic, h, w], dtype=F, fmt=N)
        w0 = G.tensor([ic, ic, 1, 1], dtype=F, fmt=marsfile.OIHW, data=((rng.random((ic, ic, 1, 1), dtype=np.float32) * 2 - 1) * (1.7 / ic ** 0.5)).astype(np.float32))
        b0 = G.tensor([ic], dtype=F, fmt=marsfile.D1, data=((rng.random(ic, dtype=np.float32) * 2 - 1) * 0.1).astype(np.float32))
        G.conv(x, a0, w0, b0, (1, 1), (1, 1))
        G.layer(marsfile.SIGMOID, [a0], [g0])
        G.layer(marsfile.MUL, [a0, g0], [o0])
        x = o0
    oh, ow = (h + s - 1) // s, (w + s - 1) // s
    a = G.tensor([1, oc, oh, ow], dtype=F, fmt=N)
    g = G.tensor([1, oc, oh, ow], dtype=F, fmt=N)
    o = G.tensor([1, oc, oh, ow], dtype=F, fmt=N)
    amp = 1.7 / (k * k * ic) ** 0.5
    wt = G.tensor([oc, ic, k, k], dtype=F, fmt=marsfile.OIHW, data=((rng.random((oc, ic, k, k), dtype=np.float32) * 2 - 1) * amp).astype(np.float32))
    b = G.tensor([oc], dtype=F, fmt=marsfile.D1, data=((rng.random(oc, dtype=np.float32) * 2 - 1) * 0.1).astype(np.float32))
    G.conv(x, a, wt, b, (k, k), (s, s))
    G.layer(marsfile.SIGMOID, [a], [g])
    G.layer(marsfile.MUL, [a, g], [o])
    if chain and add and s == 1 and ic == oc:
        o2 = G.tensor([1, oc, oh, ow], dtype=F, fmt=N)
        G.layer(marsfile.ADD, [o, x_in], [o2])
        o = o2
    return G.serialise([x_in], [o])


def build(h, w, ic, oc, k, s, add):
    G = marsfile.Graph()
    rng = np.random.default_rng(1)
    x = G.tensor([1, h, w, ic], scale=4 / 127)
    oh, ow = (h + s - 1) // s, (w + s - 1) // s
    a = G.tensor([1, oh, ow, oc], scale=0.03125)
    g = G.tensor([1, oh, ow, oc], scale=1 / 127)
    o = G.tensor([1, oh, ow, oc], scale=4 / 127)
    wt = G.tensor([oc, k, k, ic], scale=0.0005, data=rng.integers(-127, 128, (oc, k, k, ic), dtype=np.int8))
    b = G.tensor([oc], dtype=marsfile.I32, scale=1.0, data=rng.integers(-500, 500, oc, dtype=np.int32))
    G.conv(x, a, wt, b, (k, k), (s, s))
    G.layer(marsfile.SIGMOID, [a], [g])
    G.layer(marsfile.MUL, [a, g], [o])
    outs = [o]
    if add:  # the C3 bottleneck's shortcut: out = Add(SiLU(conv(x)), x)
        assert s == 1 and ic == oc
        o2 = G.tensor([1, oh, ow, oc], scale=6 / 127)
        G.layer(marsfile.ADD, [o, x], [o2])
        outs = [o2]
    return G.serialise([x], outs)


def parse_cfg(text):
    if text == "default":
        return {}
    out = {}
    for item in text.split(","):
        k, v = item.split("=")
        out[k] = int(v)
    return out


def apply_tune(tune):
    for kk, vv in tune.items():
        if kk == "norec":  # planner switch, read from the environment when the model is loaded
            if vv:
                os.environ["MARS_HIP_NO_REC"] = "1"
            else:
                os.environ.pop("MARS_HIP_NO_REC", None)
        else:
            M.set_tuning(kk, vv)


def run(name, cfg, configs, batch, oracle, f32=False, chain=False):
    h, w, ic, oc, k, s, add = cfg
    d = build_f32(h, w, ic, oc, k, s, add, chain) if f32 else build(*cfg)
    rng = np.random.default_rng(7)
    ref = None
    inp = None
    print("%s: %dx%d %d->%d k%d s%d add=%s batch %d" % ((name,) + tuple(cfg) + (batch,)), flush=True)
    for text in configs:
        tune = parse_cfg(text)
        apply_tune(tune)
        m = M.Model(d, batch=batch)
        if inp is None:
            inp = (rng.random(m.input_view(0).shape[0] * (m.input_view(0).shape[1] // 4), dtype=np.float32).view(np.uint8).reshape(m.input_view(0).shape)
                   if f32 else rng.integers(0, 256, m.input_view(0).shape, dtype=np.uint8))
        m.input_view(0)[:] = inp
        m.upload()
        m.run_device()
        m.set_profiling(True)
        best = 1e9
        nlaunch = 0
        per = []
        for _ in range(5):
            m.run_device()
            ops = m.ops()
            if sum(op["ms"] for op in ops) < best:
                best = sum(op["ms"] for op in ops)
                per = [op["ms"] for op in ops if op["ms"] > 0]
            nlaunch = sum(1 for op in ops if op["ms"] > 0)
        m.download()
        out = m.output_view(0)
        crc = zlib.crc32(out.tobytes())
        verdict = "same bytes"
        if ref is None:
            ref = crc
            verdict = "reference"
            if oracle:
                import orcbind
                g = orcbind.Graph(d)
                g.set_input(0, inp[0].tobytes())
                assert g.run() == 0
                hdr, _, _ = marsfile.parse(d)
                want = g.tensor(hdr["outputs"][0])
                if f32:
                    a_, b_ = out[0].view(np.float32).astype(np.float64), want.view(np.float32).astype(np.float64)
                    verdict = "oracle frame 0: worst |a-b|/max(1,|b|) = %.2g" % float(np.max(np.abs(a_ - b_) / np.maximum(1.0, np.abs(b_))))
                else:
                    verdict = "oracle frame 0: " + ("OK" if np.array_equal(want, out[0]) else "MISMATCH")
        elif crc != ref:
            verdict = "different bytes (expected between float32 modes)" if f32 else "DIFFERENT BYTES"
        oh, ow = (h + s - 1) // s, (w + s - 1) // s
        macs = oh * ow * oc * ic * k * k * batch
        byts = (h * w * ic + oh * ow * oc * (2 if add else 1)) * batch * (4 if f32 else 1)
        print("   %-40s %7.1f us  %5.0f TOP/s  %5.0f GB/s  %d launch(es)  %s" %
              (text, best * 1e3, 2 * macs / best / 1e9, byts / best / 1e6, nlaunch, verdict) +
              ("  [" + " + ".join("%.1f" % (v * 1e3) for v in per) + "]" if len(per) > 1 else ""), flush=True)
        m.close()
        apply_tune({kk: (80 if kk == "patch_lds_kb" else 1 if kk == "f32_mfma" else 0) for kk in tune})


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("layers", nargs="*", default=["D40", "D20"])
    ap.add_argument("--cfg", action="append", default=None, help="default | key=value[,key=value...]; repeatable")
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--f32", action="store_true", help="the float32 (NCHW) form of the layer; pick the kernel with --cfg f32_mfma=0|2|3")
    ap.add_argument("--chain", action="store_true", help="(--f32) a 1 x 1 convolution in_c -> in_c in front of the layer, the shortcut Add behind it where the layer has one: a C3 bottleneck")
    args = ap.parse_args()
    if os.environ.get("LIB"):
        M.LIB_PATH = os.path.abspath(os.environ["LIB"])
    M.nna_init()
    batch = int(os.environ.get("BATCH", "256"))
    for name in args.layers:
        cfg = LAYERS[name] if name in LAYERS else tuple(int(v) for v in name.split(","))
        cfg = tuple(cfg[:6]) + (bool(cfg[6]),)
        run(name, cfg, args.cfg or ["default"], batch, not args.no_oracle, args.f32, args.chain)


if __name__ == "__main__":
    main()

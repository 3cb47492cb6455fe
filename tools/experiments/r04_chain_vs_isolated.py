"""Does a layer run slower inside a chain than alone?  (round 4: layer 3 of the yolov5s twin reads 450 us in the step's
per-launch table and in the rocprofv3 trace, 280 us in tools/layer_time.py and in mars_hip_autotune's back-to-back reps.)
Builds stem -> L3 [-> 1x1 32] as ONE graph, prints per-launch HIP-event times, then the same with variants of the order.
    python tools/experiments/r04_chain_vs_isolated.py"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "thingino-accel_amd"))
import marsfile  # noqa: E402
import marsrt as M  # noqa: E402


def chain(spec, h=640, w=640, c=3):
    """spec: list of (out_c, k, stride)"""
    G = marsfile.Graph()
    rng = np.random.default_rng(1)
    x = G.tensor([1, h, w, c], scale=1 / 255 if c == 3 else 4 / 127)
    x0 = x
    for oc, k, s in spec:
        oh, ow = (h + s - 1) // s, (w + s - 1) // s
        a = G.tensor([1, oh, ow, oc], scale=0.03125)
        g = G.tensor([1, oh, ow, oc], scale=1 / 127)
        o = G.tensor([1, oh, ow, oc], scale=4 / 127)
        wt = G.tensor([oc, k, k, c], scale=0.0005, data=rng.integers(-127, 128, (oc, k, k, c), dtype=np.int8))
        b = G.tensor([oc], dtype=marsfile.I32, scale=1.0, data=rng.integers(-500, 500, oc, dtype=np.int32))
        G.conv(x, a, wt, b, (k, k), (s, s))
        G.layer(marsfile.SIGMOID, [a], [g])
        G.layer(marsfile.MUL, [a, g], [o])
        x, h, w, c = o, oh, ow, oc
    return G.serialise([x0], [x])


def times(d, batch, reps=6, gap_s=0.0):
    import time
    m = M.Model(d, batch=batch)
    m.input_view(0)[:] = np.random.default_rng(7).integers(0, 256, m.input_view(0).shape, dtype=np.uint8)
    m.upload(); m.run_device()
    m.set_profiling(True)
    best = None
    for _ in range(reps):
        if gap_s:
            time.sleep(gap_s)
        m.run_device()
        t = [op["ms"] * 1e3 for op in m.ops() if op["ms"] > 0]
        best = t if best is None else [min(a, b) for a, b in zip(best, t)]
    m.set_profiling(False)
    m.close()
    return best


M.nna_init()
B = int(os.environ.get("BATCH", "256"))
for k, v in [a.split("=") for a in sys.argv[1:]]:
    M.set_tuning(k, int(v))
fmt = lambda t: "  ".join("%7.1f" % v for v in t)
print("stem alone (640x640x3 -> 320x320x32 k6 s2)            ", fmt(times(chain([(32, 6, 2)]), B)), flush=True)
print("L3 alone (320x320x32 -> 160x160x64 k3 s2)             ", fmt(times(chain([(64, 3, 2)], 320, 320, 32), B)), flush=True)
print("stem -> L3                                            ", fmt(times(chain([(32, 6, 2), (64, 3, 2)]), B)), flush=True)
print("stem -> L3 -> 1x1 32                                  ", fmt(times(chain([(32, 6, 2), (64, 3, 2), (32, 1, 1)]), B)), flush=True)
print("1x1 32->32 @320 -> L3                                 ", fmt(times(chain([(32, 1, 1), (64, 3, 2)], 320, 320, 32), B)), flush=True)
print("L3 -> L3-like (160x160x64 -> 80x80x128 k3 s2)         ", fmt(times(chain([(64, 3, 2), (128, 3, 2)], 320, 320, 32), B)), flush=True)

"""Layer 3 of the yolov5s twin inside the full step vs after an idle gap (round 4; see r04_chain_vs_isolated.py).
    python tools/experiments/r04_l3_in_step.py [key=value ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "thingino-accel_amd"))
import marsrt as M  # noqa: E402

M.nna_init()
M.set_tuning("dual_stream_min_batch", 0)
for k, v in [a.split("=") for a in sys.argv[1:]]:
    M.set_tuning(k, int(v))
B = int(os.environ.get("BATCH", "256"))
d = M.synth_model(width_x16=8, input_hw=640, seed=1, vary_scales=False, float32=False)
m = M.Model(d, batch=B)
m.input_view(0)[:] = np.random.default_rng(7).integers(0, 256, m.input_view(0).shape, dtype=np.uint8)
m.upload()
for _ in range(3):
    m.run_device()
m.set_profiling(True)


def show(tag):
    ops = m.ops()
    t = [op["ms"] * 1e3 for op in ops]
    print("%-44s L0 %6.1f  L3 %6.1f  L6+9 %6.1f  L12 %6.1f  L15 %6.1f  L20 %6.1f  L23 %6.1f | all %7.1f" %
          (tag, t[0], t[1], t[2], t[4], t[5], t[6], t[7], sum(t)), flush=True)


m.run_device(); show("profiled step")
m.run_device(); show("profiled step again")
time.sleep(1.0)
m.run_device(); show("after 1 s idle")
for _ in range(30):
    m.run_device()
show("30th of 30 back to back")
m.set_profiling(False)
for _ in range(50):
    m.run_device()
m.set_profiling(True)
m.run_device(); show("after 50 unprofiled steps")
m.close()

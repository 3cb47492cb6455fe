#!/usr/bin/env python3
"""The letterbox front-end alone at the camera leg's geometry (256 frames of 1280 x 720 RGB -> 640 x 640 int8), for a kernel trace:
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lb -o lb -- python3 tools/letterbox_time.py [form]
form: 0 strips (default), 1 16 x 16 tiles, 2 one thread per pixel (MARS_HIP_LETTERBOX_FORM)."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    os.environ["MARS_HIP_LETTERBOX_FORM"] = sys.argv[1]
spec = importlib.util.spec_from_file_location("marsrt", os.path.join(ROOT, "thingino-accel_amd", "marsrt.py"))
M = importlib.util.module_from_spec(spec)
spec.loader.exec_module(M)
M.nna_init()
B, w, h = 256, 1280, 720
d = M.synth_model(width_x16=4, input_hw=640, seed=1)
m = M.Model(d, batch=B)
rng = np.random.default_rng(1)
frames = rng.integers(0, 256, (B, h, w, 3), dtype=np.uint8)
for _ in range(6):
    m.preprocess(frames)
M.lib().mars_hip_sync()
m.close()
print("done")

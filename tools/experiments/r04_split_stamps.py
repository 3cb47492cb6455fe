"""Where the halves of conv_f32_split's ping-pong step go (diagnostic build: tools/stamps_build.sh splitstamps; GPU box):
    python tools/split_stamps.py D40 P40
s_memtime stamps of wave lane 0, summed per wave group (early = waves 0-3, late = waves 4-7), first step of every iteration."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import layer_time as LT  # noqa: E402

M = LT.M
M.LIB_PATH = os.path.abspath(os.environ.get("LIB", os.path.join(HERE, "..", "thingino-accel_amd", "lib", "diag", "lib_stamps_split.so")))
M.nna_init()
L = M.lib()
L.mhip_split_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
M.set_tuning("f32_mfma", 3)
for name in [a for a in sys.argv[1:] if a in LT.LAYERS] or ["D40"]:
    h, w, ic, oc, k, s, _ = LT.LAYERS[name]
    m = M.Model(LT.build_f32(h, w, ic, oc, k, s), batch=int(os.environ.get("BATCH", "256")))
    iv = m.input_view(0)
    iv[:] = np.random.default_rng(7).random(iv.shape[0] * (iv.shape[1] // 4), dtype=np.float32).view(np.uint8).reshape(iv.shape)
    m.upload(); m.run_device(); m.run_device()
    L.mhip_split_stamps(None, 1)
    m.run_device()
    out = (C.c_ulonglong * 16)()
    L.mhip_split_stamps(out, 1)
    print(name)
    for g, base in (("early", 0), ("late", 8)):
        v = [float(out[base + i]) for i in range(8)]
        n = max(v[7], 1.0)
        print("   %-5s per wave: MFMA half %8.0f  barrier %8.0f | commit %8.0f  fetch %8.0f  barrier %8.0f | second step of the iteration + tile epilogues %8.0f   (waves %d)"
              % (g, v[0] / n, v[1] / n, v[2] / n, v[3] / n, v[4] / n, v[5] / n, n))
    m.close()

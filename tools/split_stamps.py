"""Where a K step of conv_f32_split goes (diagnostic build: tools/stamps_build.sh splitstamps; GPU box):
    python tools/split_stamps.py D40 P40 L15
s_memtime stamps of every wave's lane 0, summed over a run: cycles per wave and K step in the fetch (address arithmetic + load
issue), the step's body (fragment reads, MFMAs, split + LDS writes), the barrier, and the rest (loop overhead, tile epilogues)."""
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
M.set_tuning("f32_mfma", int(os.environ.get("MODE", "3")))
for name in [a for a in sys.argv[1:] if a in LT.LAYERS] or ["D40"]:
    h, w, ic, oc, k, s, _ = LT.LAYERS[name]
    m = M.Model(LT.build_f32(h, w, ic, oc, k, s), batch=int(os.environ.get("BATCH", "256")))
    iv = m.input_view(0)
    iv[:] = np.random.default_rng(7).random(iv.shape[0] * (iv.shape[1] // 4), dtype=np.float32).view(np.uint8).reshape(iv.shape)
    m.upload(); m.run_device(); m.run_device()
    L.mhip_split_stamps(None, 1)
    m.run_device()
    out = (C.c_ulonglong * 8)()
    L.mhip_split_stamps(out, 1)
    v = [float(out[i]) for i in range(5)]
    n = max(v[4], 1.0)
    print("%-6s per wave and K step: fetch %6.0f  body %6.0f  barrier %6.0f  rest %6.0f  = %6.0f cycles   (wave-steps %.0f)"
          % (name, v[0] / n, v[1] / n, v[2] / n, v[3] / n, sum(v[:4]) / n, n))
    m.close()

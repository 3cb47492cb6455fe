"""Where a K step of conv_i8_rows spends its cycles (diagnostic build: tools/stamps_build.sh rows, run on the GPU box):
    python tools/rows_stamps.py D40 [D20 ...]
Per wave class (waves 0-3 run their epilogue slice before the step's MFMAs, waves 4-7 after): mean shader cycles per K step
from `s_memtime` stamps -- wait (vmcnt + barrier) | reads, slice, DMA issue up to the first MFMA | MFMA issue | after."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import layer_time as LT  # noqa: E402

M = LT.M
M.LIB_PATH = os.path.abspath(os.environ.get("LIB", os.path.join(HERE, "..", "thingino-accel_amd", "lib", "diag", "lib_stamps_rows.so")))
M.nna_init()
L = M.lib()
L.mhip_rows_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
batch = int(os.environ.get("BATCH", "256"))
M.set_tuning("variant", 20)
for name in sys.argv[1:] or ["D40"]:
    cfg = LT.LAYERS[name]
    d = LT.build(*cfg)
    m = M.Model(d, batch=batch)
    m.input_view(0)[:] = np.random.default_rng(7).integers(0, 256, m.input_view(0).shape, dtype=np.uint8)
    m.upload(); m.run_device(); m.run_device()
    L.mhip_rows_stamps(None, 1)
    m.set_profiling(True)
    m.run_device()
    ms = sum(op["ms"] for op in m.ops())
    out = (C.c_ulonglong * 16)()
    L.mhip_rows_stamps(out, 1)
    print("%s %s: %.1f us" % (name, cfg, ms * 1e3))
    for o, who in ((0, "waves 0-3 (slice first)"), (8, "waves 4-7 (slice last) ")):
        n = max(float(out[o + 4]), 1.0)
        w, pre, mf, post = [float(out[o + j]) / n for j in range(4)]
        print("   %s: per step  wait %.0f | pre %.0f | mfma issue %.0f | post %.0f | total %.0f cycles  (%.0f steps)" %
              (who, w, pre, mf, post, w + pre + mf + post, n))
    m.close()

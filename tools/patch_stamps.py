"""Per-tile cycle shares of the patch-staged kernel conv_i8_patch (diagnostic build: tools/stamps_build.sh patch; GPU box):
    python tools/patch_stamps.py L3 L15 L44 [--cfg variant=10,patch_ring=2 ...]
s_memtime stamps summed over all waves: wait for the patch DMA | DMA issue | K loop | epilogue, per wave and tile."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import layer_time as LT  # noqa: E402

M = LT.M
M.LIB_PATH = os.path.abspath(os.environ.get("LIB", os.path.join(HERE, "..", "thingino-accel_amd", "lib", "diag", "lib_stamps_patch.so")))
M.nna_init()
L = M.lib()
L.mhip_patch_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
batch = int(os.environ.get("BATCH", "256"))
names = [a for a in sys.argv[1:] if a in LT.LAYERS] or ["L3", "L15", "L44"]
cfgs = [a for a in sys.argv[1:] if "=" in a or a == "default"] or ["default"]
for name in names:
    cfg = LT.LAYERS[name]
    d = LT.build(*cfg)
    print(name, cfg, flush=True)
    for text in cfgs:
        tune = LT.parse_cfg(text)
        for kk, vv in tune.items():
            M.set_tuning(kk, vv)
        m = M.Model(d, batch=batch)
        m.input_view(0)[:] = np.random.default_rng(7).integers(0, 256, m.input_view(0).shape, dtype=np.uint8)
        m.upload(); m.run_device(); m.run_device()
        L.mhip_patch_stamps(None, 1)
        m.set_profiling(True)
        m.run_device()
        ms = sum(op["ms"] for op in m.ops())
        out = (C.c_ulonglong * 8)()
        L.mhip_patch_stamps(out, 1)
        w, i, k, e, n = [float(out[j]) for j in range(5)]
        n = max(n, 1.0)
        tot = w + i + k + e
        print("   %-28s %7.1f us | per wave and tile: wait %.0f  issue %.0f  K loop %.0f  epilogue %.0f  = %.0f cycles | wave-tiles %.0f"
              % (text, ms * 1e3, w / n, i / n, k / n, e / n, tot / n, n), flush=True)
        m.close()
        for kk in tune:
            M.set_tuning(kk, 80 if kk == "patch_lds_kb" else 0)

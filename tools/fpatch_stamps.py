"""Where a K step of conv_f32_patch goes (diagnostic build: tools/stamps_build.sh fpatch; GPU box):
    python tools/fpatch_stamps.py D40 D20 L15 L3
Cycle-counter stamps of every wave's lane 0, summed over a run: cycles per wave and K step in the chunk commit (when the schedule names
one: the wait for its loads, the split, 8 LDS writes), the load issue (weights of the step, the next chunk), the step's body (table
read, fragment reads, MFMAs, next weights into LDS), the barrier, and the rest (loop overhead, tile setup, epilogue)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import layer_time as LT  # noqa: E402

M = LT.M
M.LIB_PATH = os.path.abspath(os.environ.get("LIB", os.path.join(HERE, "..", "thingino-accel_amd", "lib", "diag", "lib_stamps_fpatch.so")))
M.nna_init()
L = M.lib()
L.mhip_fpatch_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
M.set_tuning("f32_mfma", 3)
M.set_tuning("dual_stream_min_batch", 0)
CHAIN = "--chain" in sys.argv  # a 1 x 1 in front (layer_time.build_f32 chain): the k x k layer then reads records (conv_f32_prec) -- its stamps are
#                                0 weight registers -> LDS, 1 patch DMA + next weight loads issued, 2 fragment reads (waited for), 3 MFMAs, 4 DMA waits, 5 barriers, 6 rest
for name in [a for a in sys.argv[1:] if a in LT.LAYERS] or ["D40"]:
    h, w, ic, oc, k, s, add_ = LT.LAYERS[name]
    m = M.Model(LT.build_f32(h, w, ic, oc, k, s, add_, CHAIN), batch=int(os.environ.get("BATCH", "256")))
    iv = m.input_view(0)
    iv[:] = np.random.default_rng(7).random(iv.shape[0] * (iv.shape[1] // 4), dtype=np.float32).view(np.uint8).reshape(iv.shape)
    m.upload(); m.run_device(); m.run_device()
    L.mhip_fpatch_stamps(None, 1)
    m.set_profiling(1)
    m.run_device()
    ms = sum(op["ms"] for op in m.ops())
    out = (C.c_ulonglong * 8)()
    L.mhip_fpatch_stamps(out, 1)
    if CHAIN:
        v = [float(out[i]) for i in range(8)]
        n = max(v[7], 1.0)
        print("%-6s %7.1f us (both launches)  conv_f32_prec per wave and K step: weight LDS writes %5.0f  patch DMA + weight load issue %5.0f  fragment reads %5.0f  MFMAs %5.0f  DMA waits %5.0f  barriers %5.0f  rest %5.0f  = %6.0f cycles   (wave-steps %.0f)"
              % ((name, ms * 1e3) + tuple(x / n for x in v[:7]) + (sum(v[:7]) / n, n)))
        m.close()
        continue
    v = [float(out[i]) for i in range(6)]
    n = max(v[5], 1.0)
    print("%-6s %7.1f us   per wave and K step: commit %6.0f  issue %6.0f  body %6.0f  barrier %6.0f  rest %6.0f  = %6.0f cycles   (wave-steps %.0f)"
          % (name, ms * 1e3, v[0] / n, v[1] / n, v[2] / n, v[3] / n, v[4] / n, sum(v[:5]) / n, n))
    m.close()

"""Soak test (GPU box, through gpurun): random int8 convolution shapes through the C-ABI host entry point (conv2d_int8),
default launch policy or a forced launch variant, vs the oracle.
  python tests/soak/fuzz_convs.py SEED N [VARIANT]      (FUZZ_IC=16: that input channel count only, maps around multiples of 16)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "thingino-accel_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import cases, marsrt as gpu, orcbind as orc
gpu.nna_init()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 150
if len(sys.argv) > 3:
    gpu.set_tuning("variant", int(sys.argv[3]))
for i in range(N):
    ic = int(rng.choice([1, 3, 4, 5, 16, 24, 32, 48, 64, 96, 128, 256, 512]))
    oc = int(rng.choice([7, 16, 32, 48, 64, 81, 128, 255, 256]))
    k = int(rng.choice([1, 2, 3, 5, 7])) if ic > 4 else int(rng.choice([1, 3, 6, 8]))
    s = int(rng.choice([1, 2, 3]))
    h, w = int(rng.integers(5, 70)), int(rng.integers(5, 70))
    if ic >= 128 and k >= 5: k = 3
    if os.environ.get("FUZZ_IC"):  # one channel count, map sizes around multiples of 16 (the patch-staged kernel wants mostly full 16-wide tiles)
        ic = int(os.environ["FUZZ_IC"])
        k = int(rng.choice([2, 3, 5, 7])); s = int(rng.choice([1, 2, 2]))
        h, w = int(rng.choice([16, 31, 32, 33, 48, 64, 80])), int(rng.choice([16, 32, 47, 48, 64, 96]))
    oh, ow = (h + s - 1) // s, (w + s - 1) // s
    ph = max((oh - 1) * s + k - h, 0) // 2; pw = max((ow - 1) * s + k - w, 0) // 2
    case = ("fz%d" % i, 1, h, w, ic, oc, k, k, s, s, ph, pw, oh, ow, 0.03, 0.003 / (k * k * ic) ** 0.5 * 8, 0.05, bool(rng.integers(0, 2)))
    a = cases.conv_i8_call(gpu.conv2d_int8, case, 100 + i)
    b = cases.conv_i8_call(orc.conv2d_int8, case, 100 + i)
    if not np.array_equal(a, b):
        bad += 1
        print("MISMATCH", case, int((a != b).sum()), flush=True)
print("fuzz done:", N, "cases,", bad, "mismatches")

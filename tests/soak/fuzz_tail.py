"""Soak test (GPU box, through gpurun): the detection tail on random candidate lists with MANY ties -- confidences drawn
from a handful of values, few classes, overlapping boxes, 0..1000 candidates -- mars_yolo_nms (exchange-sort
permutation + greedy class-wise suppression) and mars_yolo_parse_output vs the oracle, byte for byte.
  python tests/soak/fuzz_tail.py SEED N"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "thingino-accel_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import marsrt as gpu, orcbind as orc
gpu.nna_init()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for it in range(N):
    n = int(rng.choice([0, 1, 2, 3, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 500, 999, 1000, int(rng.integers(0, 1001))]))
    nvals = int(rng.choice([1, 2, 3, 5, 17, 300]))
    vals = np.sort(rng.random(nvals).astype(np.float32))
    d = np.zeros(n, dtype=gpu.DET_DTYPE)
    d["conf"] = vals[rng.integers(0, nvals, n)]
    d["cls"] = rng.integers(0, int(rng.choice([1, 2, 5, 80])), n)
    spread = float(rng.choice([20.0, 100.0, 640.0]))
    d["x"] = rng.random(n).astype(np.float32) * spread
    d["y"] = rng.random(n).astype(np.float32) * spread
    d["w"] = (rng.random(n).astype(np.float32) * 60 + 1)
    d["h"] = (rng.random(n).astype(np.float32) * 60 + 1)
    thr = float(rng.choice([0.3, 0.45, 0.7]))
    got = gpu.nms(d, thr)
    want = orc.nms(d.astype(orc.DET_DTYPE), thr)
    if got.tobytes() != want.tobytes():
        bad += 1
        print("NMS MISMATCH", it, n, nvals, thr, len(got), len(want), flush=True)
    # decode: random int8 predictions, few distinct objectness values
    npred = int(rng.choice([1, 85, 300, 1200, 4800]))
    pred = rng.integers(-128, 128, (npred, 85), dtype=np.int8)
    pred[:, 4] = rng.choice(np.array([-128, -40, -3, 0, 5, 60, 127], dtype=np.int8), npred)
    scale = np.float32(rng.choice([0.02, 0.05, 0.11]))
    a = gpu.parse_output(pred.reshape(-1), npred, scale)
    b = orc.parse_output(pred.reshape(-1), npred, scale)
    if a.tobytes() != b.tobytes():
        bad += 1
        print("DECODE MISMATCH", it, npred, float(scale), len(a), len(b), flush=True)
print("tail fuzz done:", N, "rounds,", bad, "mismatches")

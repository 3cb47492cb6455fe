"""Soak test (GPU box, through gpurun): YOLOv5 twins with per-convolution scales over seeds / widths / input sizes / batches
(1..67: the two-stream execution included) / fusion levels 1 and 2; graph outputs and detections vs the oracle.
  python tests/soak/fuzz_twins.py SEED N"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "thingino-accel_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import marsfile, marsrt as gpu, orcbind as orc
from conftest import lcg_frame
gpu.nna_init()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    width = int(rng.choice([4, 8])); hw = int(rng.choice([64, 96, 128, 160, 224, 256, 320])); seed = int(rng.integers(1, 1 << 20)); B = int(rng.choice([1, 2, 3, 5, 7, 64, 67])) if hw <= 160 else int(rng.integers(1, 6)); level = int(rng.choice([1, 2]))
    d = gpu.synth_model(width_x16=width, input_hw=hw, seed=seed, vary_scales=True)
    hdr, tensors, _ = marsfile.parse(d)
    nb = marsfile.tensor_nbytes(tensors[hdr["inputs"][0]])
    m = gpu.Model(d, batch=B, fusion=level)
    xs = [lcg_frame(seed * 16 + f, nb) for f in range(B)]
    for f in range(B): m.input_view(0)[f] = xs[f]
    m.run()
    dets = m.detect(outputs=(0, 1, 2), thresh=0.45)
    t0 = time.time()
    for f in sorted(set([0, B - 1, B // 2])):
        g = orc.Graph(d); g.set_input(0, xs[f].tobytes()); assert g.run() == 0
        for oi, ti in enumerate(hdr["outputs"]):
            if not np.array_equal(m.output_view(oi)[f], g.tensor(ti)):
                bad += 1; print("MISMATCH", width, hw, seed, B, f, oi, flush=True)
        parts = []
        for ti in hdr["outputs"]:  # every head with its own scale; the candidate list is capped at 1000 in order
            pred = g.tensor(ti).view(np.int8)
            parts.append(orc.parse_output(pred, len(pred) // 85, np.float32(tensors[ti]["scale"])))
        raw = np.concatenate(parts)[:1000]
        want = orc.nms(raw, 0.45)
        if dets[f].tobytes() != want.tobytes():
            bad += 1; print("DET MISMATCH", width, hw, seed, B, f, len(raw), flush=True)
    print("ok", width, hw, seed, B, "level", level, "oracle %.1fs" % (time.time() - t0), flush=True)
    m.close()
print("graph fuzz done,", bad, "mismatches")

"""Soak test (GPU box, through gpurun): ONE model driven through a random sequence of state changes -- batch sizes
(1 .. 65: HIP-graph replay, one stream, two streams), fusion levels 0 / 1 / 2, launch-policy knobs, per-launch profiling,
autotuning -- and run through mars_run, the resident path, the detection tail and the pipelined I/O path; after every
step the outputs of every frame must equal the oracle's (frames are a fixed pool, so batch b holds frames 0 .. b-1).
  python tests/soak/fuzz_api_states.py SEED STEPS"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "thingino-accel_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import marsfile, marsrt as gpu, orcbind as orc
from conftest import lcg_frame
gpu.nna_init()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
d = gpu.synth_model(width_x16=4, input_hw=96, seed=5, vary_scales=True)
hdr, tensors, _ = marsfile.parse(d)
nb = marsfile.tensor_nbytes(tensors[hdr["inputs"][0]])
POOL = 65
xs = [lcg_frame(0xA91 * 64 + f, nb) for f in range(POOL)]
want, wdet = [], []
for f in range(POOL):
    g = orc.Graph(d); g.set_input(0, xs[f].tobytes()); assert g.run() == 0
    want.append([g.tensor(ti).copy() for ti in hdr["outputs"]])
    parts = [orc.parse_output(g.tensor(ti).view(np.int8), len(g.tensor(ti)) // 85, np.float32(tensors[ti]["scale"])) for ti in hdr["outputs"]]
    wdet.append(orc.nms(np.concatenate(parts)[:1000], 0.45).tobytes())
m = gpu.Model(d, batch=1)
B = 1
bad = 0


def fill():
    for f in range(B):
        m.input_view(0)[f] = xs[f]


def check(tag):
    global bad
    for f in range(B):
        for oi in range(3):
            if not np.array_equal(m.output_view(oi)[f], want[f][oi]):
                bad += 1
                print("MISMATCH after", tag, "batch", B, "frame", f, "output", oi, flush=True)
                return


for step in range(STEPS):
    op = str(rng.choice(["batch", "fusion", "tune", "prof", "autotune", "run", "run", "resident", "detect", "pipe"]))
    if op == "batch":
        B = int(rng.choice([1, 2, 3, 8, 9, 33, 64, 65])); m.set_batch(B)
    elif op == "fusion":
        m.set_fusion(int(rng.choice([0, 1, 2])))
    elif op == "tune":
        k, v = [("graph_max_batch", int(rng.choice([0, 8, 100]))), ("dual_stream_min_batch", int(rng.choice([0, 2, 64]))),
                ("small_batch", int(rng.choice([0, 1]))), ("persist_slots", int(rng.choice([0, 3, 7]))),
                ("rgb_direct", int(rng.choice([0, 1]))), ("run_chunk", int(rng.choice([0, 1, 2, 128])))][int(rng.integers(0, 6))]
        gpu.set_tuning(k, v); op = "tune %s=%d" % (k, v)
    elif op == "prof":
        m.set_profiling(int(rng.choice([0, 1, 2])))
    elif op == "autotune":
        fill(); m.upload(); m.run_device(); m.autotune(1)
    elif op == "run":
        fill(); m.output_view(0)[:] = 0; m.run(); check(op)
    elif op == "resident":
        fill(); m.upload()
        for _ in range(int(rng.integers(1, 4))):
            m.run_device(sync=bool(rng.integers(0, 2)))
        m.output_view(0)[:] = 0; m.download(); check(op)
    elif op == "detect":
        fill(); m.run()
        dets = m.detect(outputs=(0, 1, 2), thresh=0.45)
        for f in range(B):
            if dets[f].tobytes() != wdet[f]:
                bad += 1; print("DET MISMATCH batch", B, "frame", f, flush=True); break
        if rng.integers(0, 2):  # tail left pending on the auxiliary stream, then another run
            m.run_device(sync=False); m.detect_device(outputs=(0, 1, 2), thresh=0.45); m.run_device(sync=False)
            assert gpu.lib().mars_hip_sync() == 0
            m.download(); check("detect+run")
    else:
        dl = bool(rng.integers(0, 2))
        m.pipe_open(download_outputs=dl, detect=True, det_outputs=(0, 1, 2), thresh=0.45)
        nbat = int(rng.integers(1, 5)); inflight = 0
        for k in range(nbat):
            iv = m.pipe_input_view(0)
            for f in range(B):
                iv[f] = xs[f]
            m.pipe_submit(); inflight += 1
            if inflight == 3 or rng.integers(0, 2):
                outs, dets = m.pipe_wait(); inflight -= 1
                for f in range(B):
                    if dets[f].tobytes() != wdet[f] or (dl and not np.array_equal(outs[0][f], want[f][0])):
                        bad += 1; print("PIPE MISMATCH batch", B, "frame", f, flush=True); break
        while inflight:
            outs, dets = m.pipe_wait(); inflight -= 1
            for f in range(B):
                if dets[f].tobytes() != wdet[f]:
                    bad += 1; print("PIPE MISMATCH (drain) batch", B, "frame", f, flush=True); break
        m.pipe_close()
    print("step", step, op, "batch", B, flush=True)
m.close()
for k, v in (("graph_max_batch", 8), ("dual_stream_min_batch", 64), ("small_batch", 1), ("persist_slots", 0), ("rgb_direct", 1), ("run_chunk", 128)):
    gpu.set_tuning(k, v)
print("api state fuzz done:", STEPS, "steps,", bad, "mismatches")

"""Soak test (GPU box, through gpurun): small-channel convolutions of random geometry -- 1..4 input channels, kernels 1..9,
strides 1..4, with and without the fused SiLU table, several frames -- GPU vs the oracle.
  python tests/soak/fuzz_small_channel_convs.py SEED N"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "thingino-accel_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import marsfile, marsrt as gpu, orcbind as orc
from conftest import lcg_frame
gpu.nna_init()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    h = int(rng.integers(3, 140)); w = int(rng.integers(3, 200)); kh = int(rng.integers(1, 9)); kw = int(rng.integers(1, 10))
    sh = int(rng.choice([1, 2, 2, 3, 4])); sw = int(rng.choice([1, 2, 2, 3, 4])); oc = int(rng.choice([16, 32, 48, 64, 7, 24])); B = int(rng.integers(1, 5)); ic = int(rng.choice([3, 3, 1, 2, 4]))
    silu = bool(rng.integers(0, 4))
    oh, ow = (h + sh - 1) // sh, (w + sw - 1) // sw
    G = marsfile.Graph()
    x = G.tensor([1, h, w, ic], scale=0.02)
    a = G.tensor([1, oh, ow, oc], scale=0.05)
    wt = G.tensor([oc, kh, kw, ic], scale=0.004, data=rng.integers(-127, 128, (oc, kh, kw, ic), dtype=np.int8))
    b = G.tensor([oc], dtype=marsfile.I32, scale=1.0, data=rng.integers(-2000, 2000, oc, dtype=np.int32))
    G.conv(x, a, wt, b, (kh, kw), (sh, sw))
    outs = [a]
    if silu:
        sg = G.tensor([1, oh, ow, oc], scale=1.0 / 256); o = G.tensor([1, oh, ow, oc], scale=0.04)
        G.layer(marsfile.SIGMOID, [a], [sg]); G.layer(marsfile.MUL, [a, sg], [o])
        outs = [o]
    d = G.serialise([x], outs)
    hdr, tensors, _ = marsfile.parse(d)
    print('case', (h, w, ic, kh, kw, sh, sw, oc, B, silu), flush=True)
    m = gpu.Model(d, batch=B)
    xs = [lcg_frame(0x57E3 * 64 + 8 * it + f, h * w * ic) for f in range(B)]
    for f in range(B): m.input_view(0)[f] = xs[f]
    m.run()
    for f in range(B):
        g = orc.Graph(d); g.set_input(0, xs[f].tobytes()); assert g.run() == 0
        want = g.tensor(hdr["outputs"][0]); got = m.output_view(0)[f]
        if not np.array_equal(got, want):
            bad += 1; print("MISMATCH", (h, w, kh, kw, sh, sw, oc, B, silu), f, int((got != want).sum()), flush=True)
    m.close()
print("stem fuzz done,", bad, "mismatches")

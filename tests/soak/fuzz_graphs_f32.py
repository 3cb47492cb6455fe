"""Soak test (GPU box, through gpurun): random float32 NCHW graphs -- convolutions (1x1 / 3x3 / 5x5, stride 1 / 2, fused
ReLU byte clamp, conv -> sigmoid -> mul chains), byte-wise max-pools, LeakyReLU / ReLU, sigmoid, add / mul, batchnorm --
vs the oracle: bit for bit with f32_mfma = 0 (the reference's summation order), and within 1e-4 * max(1, |b|) under the
default policy (matrix cores wherever no byte-wise consumer follows), where a value that the reference itself lets run
away (|b| > 1e6 or non-finite) ends the comparison of that output.
  python tests/soak/fuzz_graphs_f32.py SEED N"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "thingino-accel_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import marsfile, marsrt as gpu, orcbind as orc
gpu.nna_init()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
bad = 0
F, NC = marsfile.F32, marsfile.NCHW


def build():
    G = marsfile.Graph()
    c, h, w = int(rng.choice([3, 8, 16, 32])), int(rng.integers(6, 30)), int(rng.integers(6, 30))
    x = G.tensor([1, c, h, w], dtype=F, fmt=NC)
    avail = [(x, c, h, w)]
    desc = []
    for _ in range(int(rng.integers(3, 9))):
        t, tc, th, tw = avail[int(rng.integers(0, len(avail)))]
        op = str(rng.choice(["conv", "conv", "conv", "pool", "act", "bin", "bn"]))
        if op == "conv":
            k = int(rng.choice([1, 3, 5])); s = int(rng.choice([1, 1, 2])); oc = int(rng.choice([8, 16, 24, 64, 100]))
            oh, ow = (th + s - 1) // s, (tw + s - 1) // s
            a0 = 1.7 / (k * k * tc) ** 0.5
            wt = G.tensor([oc, tc, k, k], dtype=F, fmt=marsfile.OIHW, data=((rng.random((oc, tc, k, k)) * 2 - 1) * a0).astype(np.float32))
            b = G.tensor([oc], dtype=F, fmt=marsfile.D1, data=((rng.random(oc) * 2 - 1) * 0.1).astype(np.float32))
            a = G.tensor([1, oc, oh, ow], dtype=F, fmt=NC)
            silu = bool(rng.integers(0, 2))
            G.conv(t, a, wt, b, (k, k), (s, s), act=0 if silu else int(rng.integers(0, 2)))
            out = a
            if silu:
                sg = G.tensor([1, oc, oh, ow], dtype=F, fmt=NC); o = G.tensor([1, oc, oh, ow], dtype=F, fmt=NC)
                G.layer(marsfile.SIGMOID, [a], [sg]); G.layer(marsfile.MUL, [a, sg], [o])
                out = o
            avail.append((out, oc, oh, ow)); desc.append(("conv", k, s, tc, oc, silu))
        elif op == "pool":  # indexes shape[1..3] as H, W, C over BYTES: stride 1 keeps the shape
            k = int(rng.choice([2, 3, 5]))
            o = G.tensor([1, tc, th, tw], dtype=F, fmt=NC)
            G.pool(t, o, (k, k), (1, 1))
            avail.append((o, tc, th, tw)); desc.append(("pool", k))
        elif op == "act":
            kind = int(rng.choice([marsfile.RELU, marsfile.LEAKY, marsfile.SIGMOID]))
            o = G.tensor([1, tc, th, tw], dtype=F, fmt=NC)
            G.layer(kind, [t], [o])
            avail.append((o, tc, th, tw)); desc.append(("act", kind))
        elif op == "bin":
            same = [q for q in avail if q[1:] == (tc, th, tw) and q[0] != t]
            if not same:
                continue
            o = G.tensor([1, tc, th, tw], dtype=F, fmt=NC)
            G.layer(int(rng.choice([marsfile.ADD, marsfile.MUL])), [t, same[int(rng.integers(0, len(same)))][0]], [o])
            avail.append((o, tc, th, tw)); desc.append(("bin",))
        else:
            sc = G.tensor([tc], dtype=F, fmt=marsfile.D1, data=(rng.random(tc) + 0.5).astype(np.float32))
            bi = G.tensor([tc], dtype=F, fmt=marsfile.D1, data=(rng.random(tc) - 0.5).astype(np.float32))
            o = G.tensor([1, tc, th, tw], dtype=F, fmt=NC)
            G.layer(marsfile.BATCHNORM, [t, sc, bi], [o])
            avail.append((o, tc, th, tw)); desc.append(("bn",))
    outs = [q[0] for q in avail[1:]][-3:]
    if not outs:
        return None
    return G.serialise([x], outs), desc


for it in range(N):
    r = build()
    if r is None:
        continue
    d, desc = r
    hdr, tensors, _ = marsfile.parse(d)
    n_in = int(np.prod(tensors[hdr["inputs"][0]]["shape"]))
    B = int(rng.integers(1, 3))
    xs = [((np.random.default_rng(1000 * it + f).random(n_in) * 2 - 1) * 2).astype(np.float32) for f in range(B)]
    want = []
    for f in range(B):
        g = orc.Graph(d); g.set_input(0, xs[f].tobytes()); rc = g.run()
        assert rc == 0, (rc, desc)
        want.append([g.tensor(ti).copy() for ti in hdr["outputs"]])
    for mode in (0, 1):
        gpu.set_tuning("f32_mfma", mode)
        for level in (0, 1):
            m = gpu.Model(d, batch=B, fusion=level)
            for f in range(B):
                m.input_view(0)[f] = xs[f].view(np.uint8)
            try:
                m.run()
            except gpu.MarsError as e:
                bad += 1; print("RUN FAILED", it, mode, level, e, desc, flush=True); m.close(); continue
            for f in range(B):
                for oi in range(len(hdr["outputs"])):
                    a = m.output_view(oi)[f].view(np.float32); b = want[f][oi].view(np.float32)
                    if mode == 0:
                        ok = np.array_equal(a.view(np.uint32), b.view(np.uint32))
                    else:
                        fin = np.isfinite(b) & (np.abs(b) < 1e6)
                        ok = bool(np.all(np.abs(a[fin] - b[fin]) <= 1e-4 * np.maximum(1.0, np.abs(b[fin])))) if fin.all() else True
                    if not ok:
                        bad += 1
                        print("MISMATCH graph", it, "mode", mode, "level", level, "frame", f, "output", oi, desc, flush=True)
            m.close()
gpu.set_tuning("f32_mfma", 1)
print("f32 graph fuzz done:", N, "graphs,", bad, "mismatches")

"""Soak test (GPU box, through gpurun): random int8 NHWC graphs -- convolutions (1x1 / 3x3 / 5x5, stride 1 / 2, fused
ReLU, conv -> sigmoid -> mul chains), max-pools, ReLU family, sigmoid, add / mul, concats, 2x upsampling, with shared
inputs and several readers per tensor -- at fusion levels 0, 1 and 2, several frames, every graph output vs the oracle.
  python tests/soak/fuzz_graphs.py SEED N"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "thingino-accel_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import marsfile, marsrt as gpu, orcbind as orc
from conftest import lcg_frame
gpu.nna_init()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
bad = 0


def build():
    G = marsfile.Graph()
    h, w = int(rng.integers(6, 40)), int(rng.integers(6, 40))
    c = int(rng.choice([3, 16, 32, 64]))
    x = G.tensor([1, h, w, c], scale=float(rng.choice([0.02, 0.04])))
    avail = [(x, h, w, c)]
    desc = []
    for _ in range(int(rng.integers(3, 11))):
        t, th, tw, tc = avail[int(rng.integers(0, len(avail)))]
        op = str(rng.choice(["conv", "conv", "conv", "pool", "act", "bin", "concat", "up"]))
        if op == "conv":
            k = int(rng.choice([1, 3, 5])) if tc > 4 else int(rng.choice([3, 6]))
            s = int(rng.choice([1, 1, 2]))
            oc = int(rng.choice([16, 24, 32, 64, 128]))
            oh, ow = (th + s - 1) // s, (tw + s - 1) // s
            wt = G.tensor([oc, k, k, tc], fmt=marsfile.OHWI, scale=0.003 / (k * k * tc) ** 0.5 * 8,
                          data=rng.integers(-127, 128, (oc, k, k, tc), dtype=np.int8))
            b = G.tensor([oc], dtype=marsfile.I32, fmt=marsfile.D1, scale=1.0, data=rng.integers(-3000, 3000, oc, dtype=np.int32))
            a = G.tensor([1, oh, ow, oc], scale=float(rng.choice([0.04, 0.06])))
            silu = bool(rng.integers(0, 2))
            G.conv(t, a, wt, b, (k, k), (s, s), act=0 if silu else int(rng.integers(0, 2)))
            out = a
            if silu:
                sg = G.tensor([1, oh, ow, oc], scale=1.0 / 256)
                o = G.tensor([1, oh, ow, oc], scale=float(rng.choice([0.03, 0.05])))
                G.layer(marsfile.SIGMOID, [a], [sg])
                G.layer(marsfile.MUL, [a, sg], [o])
                out = o
            avail.append((out, oh, ow, oc)); desc.append(("conv", k, s, tc, oc, silu))
        elif op == "pool":
            k = int(rng.choice([2, 3, 5])); s = int(rng.choice([1, 2]))
            oh, ow = (th + s - 1) // s, (tw + s - 1) // s
            o = G.tensor([1, oh, ow, tc], scale=G.tensors[t]["scale"])
            G.pool(t, o, (k, k), (s, s))
            avail.append((o, oh, ow, tc)); desc.append(("pool", k, s))
        elif op == "act":
            kind = int(rng.choice([marsfile.RELU, marsfile.RELU6, marsfile.LEAKY, marsfile.SIGMOID]))
            o = G.tensor([1, th, tw, tc], scale=float(rng.choice([0.01, 0.03])))
            G.layer(kind, [t], [o])
            avail.append((o, th, tw, tc)); desc.append(("act", kind))
        elif op == "bin":
            same = [q for q in avail if q[1:] == (th, tw, tc) and q[0] != t]
            if not same:
                continue
            u = same[int(rng.integers(0, len(same)))][0]
            o = G.tensor([1, th, tw, tc], scale=float(rng.choice([0.03, 0.06])))
            G.layer(int(rng.choice([marsfile.ADD, marsfile.MUL])), [t, u], [o])
            avail.append((o, th, tw, tc)); desc.append(("bin",))
        elif op == "concat":
            same = [q for q in avail if q[1] == th and q[2] == tw and q[0] != t and q[3] % 16 == 0]
            if not same or tc % 16:
                continue
            parts = [t] + [q[0] for q in same[:int(rng.integers(1, 3))]]
            cs = tc + sum(q[3] for q in same[:len(parts) - 1])
            o = G.tensor([1, th, tw, cs], scale=G.tensors[t]["scale"])
            G.concat(parts, o)
            avail.append((o, th, tw, cs)); desc.append(("concat", len(parts)))
        else:
            if th * tw > 600:
                continue
            o = G.tensor([1, th * 2, tw * 2, tc], scale=G.tensors[t]["scale"])
            G.upsample(t, o, 2, 2)
            avail.append((o, th * 2, tw * 2, tc)); desc.append(("up",))
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
    nb = marsfile.tensor_nbytes(tensors[hdr["inputs"][0]])
    B = int(rng.integers(1, 4))
    xs = [lcg_frame(0xF00D * 64 + 4 * it + f, nb) for f in range(B)]
    want = []
    for f in range(B):
        g = orc.Graph(d); g.set_input(0, xs[f].tobytes()); rc = g.run()
        assert rc == 0, (rc, desc)
        want.append([g.tensor(ti).copy() for ti in hdr["outputs"]])
    for level in (0, 1, 2):
        m = gpu.Model(d, batch=B, fusion=level)
        for f in range(B):
            m.input_view(0)[f] = xs[f]
        try:
            m.run()
        except gpu.MarsError as e:
            bad += 1
            print("RUN FAILED graph", it, "level", level, "batch", B, e, desc, flush=True)
            m.close()
            continue
        for f in range(B):
            for oi in range(len(hdr["outputs"])):
                if not np.array_equal(m.output_view(oi)[f], want[f][oi]):
                    bad += 1
                    print("MISMATCH graph", it, "level", level, "frame", f, "output", oi, desc, flush=True)
        m.close()
print("graph fuzz done:", N, "graphs,", bad, "mismatches")

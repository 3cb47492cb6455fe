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
    """tensors are kept as (id, d1, d2, d3) = their shape[1..3].  NHWC graphs: (H, W, C).  NCHW-tagged graphs: (C, H, W) --
    convolutions take their dims by the tag, while pools / concats / upsampling index shape[1..3] as H, W, C whatever the
    tag says (the reference's behaviour), so there a pool halves "C" and "H" and a concat joins along "W" """
    G = marsfile.Graph()
    nchw = bool(rng.integers(0, 4) == 0) if os.environ.get("FUZZ_NCHW") is None else os.environ["FUZZ_NCHW"] == "1"
    fmt = marsfile.NCHW if nchw else marsfile.NHWC
    h, w = int(rng.integers(6, 40)), int(rng.integers(6, 40))
    c = int(rng.choice([3, 5, 8, 16, 24, 32, 40, 64]))
    dims = (c, h, w) if nchw else (h, w, c)
    x = G.tensor([1, *dims], fmt=fmt, scale=float(rng.choice([0.02, 0.04])))
    avail = [(x, *dims)]
    desc = [("nchw" if nchw else "nhwc", dims)]

    def mkconv(t, tc, th, tw, k, s, oc, pad, silu, act):
        oh, ow = (th + s - 1) // s, (tw + s - 1) // s  # the output tensor's shape; VALID still fills it (unpadded window)
        wshape = (oc, tc, k, k) if nchw else (oc, k, k, tc)
        wt = G.tensor(list(wshape), fmt=marsfile.OIHW if nchw else marsfile.OHWI, scale=0.003 / (k * k * tc) ** 0.5 * 8,
                      data=rng.integers(-127, 128, wshape, dtype=np.int8))
        b = G.tensor([oc], dtype=marsfile.I32, fmt=marsfile.D1, scale=1.0, data=rng.integers(-3000, 3000, oc, dtype=np.int32)) \
            if rng.integers(0, 5) else marsfile.NONE
        od = (oc, oh, ow) if nchw else (oh, ow, oc)
        a = G.tensor([1, *od], fmt=fmt, scale=float(rng.choice([0.04, 0.06])))
        G.conv(t, a, wt, b, (k, k), (s, s), pad=pad, act=act)
        out = a
        if silu:
            sg = G.tensor([1, *od], fmt=fmt, scale=1.0 / 256)
            o = G.tensor([1, *od], fmt=fmt, scale=float(rng.choice([0.03, 0.05])))
            G.layer(marsfile.SIGMOID, [a], [sg])
            G.layer(marsfile.MUL, [a, sg], [o])
            out = o
        return out, od

    for _ in range(int(rng.integers(3, 11))):
        t, d1, d2, d3 = avail[int(rng.integers(0, len(avail)))]
        op = str(rng.choice(["conv", "conv", "conv", "pool", "act", "bin", "concat", "up", "c3", "sppf"]))
        if op == "conv":
            tc, th, tw = (d1, d2, d3) if nchw else (d3, d1, d2)
            if th * tw > 4000 or tc > 300:
                continue
            k = int(rng.choice([1, 3, 5, 7])) if tc > 4 else int(rng.choice([3, 6]))
            if tc >= 64 and k == 7: k = 3
            s = int(rng.choice([1, 1, 2, 3]))
            oc = int(rng.choice([7, 16, 24, 32, 64, 81, 128]))
            pad = int(rng.choice([marsfile.PAD_SAME, marsfile.PAD_SAME, marsfile.PAD_SAME, marsfile.PAD_VALID]))
            silu = bool(rng.integers(0, 2))
            out, od = mkconv(t, tc, th, tw, k, s, oc, pad, silu, 0 if silu else int(rng.integers(0, 2)))
            avail.append((out, *od)); desc.append(("conv", k, s, tc, oc, silu, pad))
        elif op == "pool":
            k = int(rng.choice([2, 3, 5])); s = int(rng.choice([1, 2]))
            od = ((d1 + s - 1) // s, (d2 + s - 1) // s, d3)
            o = G.tensor([1, *od], fmt=fmt, scale=G.tensors[t]["scale"])
            G.pool(t, o, (k, k), (s, s))
            avail.append((o, *od)); desc.append(("pool", k, s))
        elif op == "act":
            kind = int(rng.choice([marsfile.RELU, marsfile.RELU6, marsfile.LEAKY, marsfile.SIGMOID]))
            o = G.tensor([1, d1, d2, d3], fmt=fmt, scale=float(rng.choice([0.01, 0.03])))
            G.layer(kind, [t], [o])
            avail.append((o, d1, d2, d3)); desc.append(("act", kind))
        elif op == "bin":
            same = [q for q in avail if q[1:] == (d1, d2, d3) and q[0] != t]
            if not same:
                continue
            u = same[int(rng.integers(0, len(same)))][0]
            o = G.tensor([1, d1, d2, d3], fmt=fmt, scale=float(rng.choice([0.03, 0.06])))
            G.layer(int(rng.choice([marsfile.ADD, marsfile.MUL])), [t, u], [o])
            avail.append((o, d1, d2, d3)); desc.append(("bin",))
        elif op == "concat":
            same = [q for q in avail if q[1] == d1 and q[2] == d2 and q[0] != t]
            if not same:
                continue
            extra = same[:int(rng.integers(1, 3))]
            parts = [t] + [q[0] for q in extra]
            cs = d3 + sum(q[3] for q in extra)
            o = G.tensor([1, d1, d2, cs], fmt=fmt, scale=G.tensors[t]["scale"])
            G.concat(parts, o)
            avail.append((o, d1, d2, cs)); desc.append(("concat", [d3] + [q[3] for q in extra]))
        elif op in ("c3", "sppf"):
            # the detectors' motifs, which the planner has passes for (virtual_concat / virtual_concat_q, pairs, pool chains, the byte-wise
            # layers on the internal layout): C3 = two 1x1 convolutions of one tensor (one of them through a bottleneck with a shortcut)
            # -> concat -> 1x1;  SPPF = 1x1 -> three chained 5x5 stride-1 pools -> concat of the four -> 1x1
            tc, th, tw = (d1, d2, d3) if nchw else (d3, d1, d2)
            if th * tw > 2500 or tc > 200:
                continue
            oc = int(rng.choice([16, 32, 64]))
            pad = marsfile.PAD_SAME
            silu = bool(rng.integers(0, 2))
            if op == "c3":
                a, od = mkconv(t, tc, th, tw, 1, 1, oc, pad, silu, 0)
                b, _ = mkconv(t, tc, th, tw, 1, 1, oc, pad, silu, 0)
                if rng.integers(0, 2):
                    m1, _ = mkconv(a, oc, th, tw, 1, 1, oc, pad, silu, 0)
                    m2, _ = mkconv(m1, oc, th, tw, 3, 1, oc, pad, silu, 0)
                    a2 = G.tensor([1, *od], fmt=fmt, scale=float(rng.choice([0.05, 0.08])))
                    G.layer(marsfile.ADD, [a, m2], [a2])
                    a = a2
                parts = [a, b]
            else:
                a, od = mkconv(t, tc, th, tw, 1, 1, oc, pad, silu, 0)
                parts = [a]
                for _i in range(3):
                    o = G.tensor([1, *od], fmt=fmt, scale=G.tensors[parts[-1]]["scale"])
                    G.pool(parts[-1], o, (5, 5), (1, 1))
                    parts.append(o)
            # (NCHW-tagged: the exporter's form, [1, sum C, H, W] along axis 1 -- which the reference's byte-wise CONCAT turns into a shift
            #  of the last input by N - 1 map rows, reading past the inputs' ends into the arena)
            cd = (od[0] * len(parts), od[1], od[2]) if nchw else (od[0], od[1], od[2] * len(parts))
            cat = G.tensor([1, *cd], fmt=fmt, scale=G.tensors[parts[0]]["scale"])
            if nchw:
                G.concat(parts, cat, axis=1)
            else:
                G.concat(parts, cat)
            cc, ch, cw = cd if nchw else (cd[2], cd[0], cd[1])
            roc = int(rng.choice([16, 32, 48]))
            out, od2 = mkconv(cat, cc, ch, cw, 1, 1, roc, pad, silu, 0)
            avail.append((parts[0], *od)); avail.append((out, *od2)); desc.append((op, tc, oc, silu))
            if rng.integers(0, 3) == 0:  # a second reader of the same shape: the head C3s' cv1 + cv2 over a concat (one paired launch)
                out2, _ = mkconv(cat, cc, ch, cw, 1, 1, roc, pad, silu, 0)
                avail.append((out2, *od2))
        else:
            if d1 * d2 > 600:
                continue
            f = int(rng.choice([2, 2, 3]))
            o = G.tensor([1, d1 * f, d2 * f, d3], fmt=fmt, scale=G.tensors[t]["scale"])
            G.upsample(t, o, f if rng.integers(0, 2) else 0, f if rng.integers(0, 2) else 0)
            avail.append((o, d1 * f, d2 * f, d3)); desc.append(("up", f))
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

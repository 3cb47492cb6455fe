"""Soak test (GPU box, through gpurun): random float32 NCHW graphs around the reference's byte-wise CONCAT of equal maps -- 2 to 4 branches
(1x1 / 3x3 convolutions, chained byte-wise max-pools as in SPPF, the graph input itself) -> CONCAT in the exporter's form [1, sum C, H, W] -> one
1x1 convolution or two of the same shape (a C3's cv1 + cv2: one paired launch) -> sometimes a k x k convolution behind (record-format pairs) --
under the split-bf16 modes (f32_mfma 3 / 4), where the planner cuts the readers' K loops (zero_tail_f32) and, for map widths that are multiples
of 4, never materialises the concat (virtual_concat_f32: a view of the last input + a head launch).  Every graph output against the oracle
within 1e-4 * max(1, |b|), with the passes on and with MARS_HIP_NO_VCONCAT_F32, wherever the plain plan (fusion level 0) is inside it; several frames.
  python tests/soak/fuzz_vcat_f32.py SEED N"""
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
views = 0
F, NC = marsfile.F32, marsfile.NCHW


def build():
    G = marsfile.Graph()
    h = int(rng.integers(4, 26))
    w = int(rng.choice([4, 8, 12, 16, 20, 24, 40, 6, 10]))  # (6, 10: the concat's runs are not whole floats -- it stays materialised)
    c_in = int(rng.choice([8, 16, 32]))
    x = G.tensor([1, c_in, h, w], dtype=F, fmt=NC)

    def conv(t, tc, th, tw, k, s, oc, silu):
        oh, ow = (th + s - 1) // s, (tw + s - 1) // s
        a0 = 1.7 / (k * k * tc) ** 0.5
        wt = G.tensor([oc, tc, k, k], dtype=F, fmt=marsfile.OIHW, data=((rng.random((oc, tc, k, k)) * 2 - 1) * a0).astype(np.float32))
        b = G.tensor([oc], dtype=F, fmt=marsfile.D1, data=((rng.random(oc) * 2 - 1) * 0.1).astype(np.float32)) if rng.integers(0, 4) else marsfile.NONE
        a = G.tensor([1, oc, oh, ow], dtype=F, fmt=NC)
        G.conv(t, a, wt, b, (k, k), (s, s), act=0)
        if not silu:
            return a, oh, ow
        sg = G.tensor([1, oc, oh, ow], dtype=F, fmt=NC); o = G.tensor([1, oc, oh, ow], dtype=F, fmt=NC)
        G.layer(marsfile.SIGMOID, [a], [sg]); G.layer(marsfile.MUL, [a, sg], [o])
        return o, oh, ow

    nb = int(rng.integers(2, 5))
    oc = int(rng.choice([8, 16, 32, 64]))
    silu = bool(rng.integers(0, 2))
    kind = str(rng.choice(["convs", "convs", "sppf", "with_input"]))
    parts = []
    if kind == "sppf":
        a, _, _ = conv(x, c_in, h, w, 1, 1, oc, silu)
        parts = [a]
        for _ in range(nb - 1):
            o = G.tensor([1, oc, h, w], dtype=F, fmt=NC)
            G.pool(parts[-1], o, (5, 5), (1, 1))
            parts.append(o)
    else:
        for q in range(nb):
            if kind == "with_input" and q == 0 and c_in == oc:
                parts.append(x)
                continue
            a, _, _ = conv(x, c_in, h, w, int(rng.choice([1, 1, 3])), 1, oc, silu)
            parts.append(a)
        rng.shuffle(parts)
    cat = G.tensor([1, oc * nb, h, w], dtype=F, fmt=NC)
    G.concat([int(p) for p in parts], cat, axis=1)
    outs = []
    roc = int(rng.choice([8, 16, 40, 64, 128, 7, 255]))  # (7, 255: channel counts that are not multiples of the head kernel's 4-channel lanes)
    readers = 2 if rng.integers(0, 3) == 0 else 1
    for _ in range(readers):
        r, _, _ = conv(cat, oc * nb, h, w, 1, 1, roc, silu)
        outs.append(r)
    if rng.integers(0, 3) == 0 and roc % 8 == 0:  # a k x k convolution behind the reader: a record-format pair where the shapes allow
        r2, _, _ = conv(outs[0], roc, h, w, 3, int(rng.choice([1, 2])), int(rng.choice([16, 32])), silu)
        outs = [r2] + outs[1:]
    return G.serialise([x], outs[:4]), (kind, nb, oc, roc, readers, h, w, c_in, silu)


def run_all(d, xs, want, tag, fusion):
    """-> mismatching (frame, output) pairs, or -1 when the run failed"""
    hdr = marsfile.parse(d)[0]
    m = gpu.Model(d, batch=len(xs), fusion=fusion)
    for f in range(len(xs)):
        m.input_view(0)[f] = xs[f].view(np.uint8)
    try:
        m.run()
    except gpu.MarsError as e:
        print("RUN FAILED", tag, e, flush=True); m.close(); return -1
    n = 0
    for f in range(len(xs)):
        for oi in range(len(hdr["outputs"])):
            a = m.output_view(oi)[f].view(np.float32); b = want[f][oi].view(np.float32)
            fin = np.isfinite(b) & (np.abs(b) < 1e6)
            if fin.all() and not bool(np.all(np.abs(a - b) <= 1e-4 * np.maximum(1.0, np.abs(b)))):
                n += 1
                if fusion:
                    print("MISMATCH", tag, "frame", f, "output", oi, "worst", float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))), flush=True)
    m.close()
    return n


ill = 0
for it in range(N):
    d, desc = build()
    hdr, tensors, _ = marsfile.parse(d)
    n_in = int(np.prod(tensors[hdr["inputs"][0]]["shape"]))
    B = int(rng.integers(1, 4))
    xs = [((np.random.default_rng(1000 * it + f).random(n_in) * 2 - 1) * 2).astype(np.float32) for f in range(B)]
    want = []
    for f in range(B):
        g = orc.Graph(d); g.set_input(0, xs[f].tobytes()); rc = g.run()
        assert rc == 0, (rc, desc)
        want.append([g.tensor(ti).copy() for ti in hdr["outputs"]])
    for mode in (3, 4):
        gpu.set_tuning("f32_mfma", mode)
        os.environ.pop("MARS_HIP_NO_VCONCAT_F32", None)
        # the plain plan first (fusion level 0: one launch per layer, every concat copied, full K loops).  Where THAT is outside the tolerance the
        # graph is ill-conditioned for the split-bf16 arithmetic (floats spliced or max-ed byte-wise reach 1e38 and cancel: runs of 6 or 10 bytes,
        # the pools of the SPPF form) and says nothing about the passes: counted, not compared
        r0 = run_all(d, xs, want, ("graph", it, "mode", mode, "unfused", desc), 0)
        if r0 != 0:
            bad += r0 < 0
            ill += r0 > 0
            continue
        views += sum(" view=-" in l for l in gpu.describe_plan(d))
        r = run_all(d, xs, want, ("graph", it, "mode", mode, "virtual", desc), 1)
        bad += abs(r)
        os.environ["MARS_HIP_NO_VCONCAT_F32"] = "1"
        r = run_all(d, xs, want, ("graph", it, "mode", mode, "copied", desc), 1)
        bad += abs(r)
        os.environ.pop("MARS_HIP_NO_VCONCAT_F32", None)
gpu.set_tuning("f32_mfma", 1)
print("f32 concat fuzz done:", N, "graphs x 2 modes,", ill, "ill-conditioned (skipped),", views, "launches on a view,", bad, "mismatches")

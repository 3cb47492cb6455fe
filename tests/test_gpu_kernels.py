"""GPU parity, kernel level: the HIP kernels called through the C-ABI's host-pointer entry
points (include/mxu_ops.h, mars_yolo_* of include/mars_hip.h) against the CPU oracle and the
golden vectors the reference produced.  Bit-exact for every int8 / index result; the f32
convolution keeps the reference's summation order and is compared bit-exactly too (the task's
tolerance is 1e-4; asserted as well, with the formula written out)."""
import json
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "golden.json")))


def check_family(gpu, orc, name, note=None):
    """every case of a shape family (cases.CONV_I8_FAMILIES) under the launch policy currently forced: against the oracle
    (element-wise, so a failure says how many bytes differ) AND against the digest the reference itself produced for the same
    inputs (golden.json "conv_i8_family", tests/golden/make_golden.py): the restatement is not the only witness"""
    seed, cs = cases.family_cases(name)
    for case in cs:
        a = cases.conv_i8_call(gpu.conv2d_int8, case, seed)
        b = cases.conv_i8_call(orc.conv2d_int8, case, seed)
        assert np.array_equal(a, b), (case[0], note, int((a != b).sum()))
        assert len(np.unique(a)) > 32  # not a saturated / all-zero comparison
        assert cases.digest(a) == GOLD["conv_i8_family"][case[0]], (case[0], note)


@pytest.mark.parametrize("case", cases.CONV_I8_CASES, ids=lambda c: c[0])
def test_conv_i8_bit_exact(gpu, orc, case):
    got = cases.conv_i8_call(gpu.conv2d_int8, case)
    assert cases.digest(got) == GOLD["conv_i8"][case[0]]          # what the reference itself produced
    for seed in (2, 3):
        a = cases.conv_i8_call(gpu.conv2d_int8, case, seed)
        b = cases.conv_i8_call(orc.conv2d_int8, case, seed)
        assert np.array_equal(a, b), "mismatches: %d" % int((a != b).sum())


@pytest.mark.parametrize("case", cases.CONV_F32_CASES, ids=lambda c: c[0])
def test_conv_f32(gpu, orc, case):
    got = cases.conv_f32_call(gpu.conv2d_f32, case)
    want = cases.conv_f32_call(orc.conv2d_f32, case)
    tol = 1e-4  # north_star: |a-b| <= 1e-4 * max(1, |b|)
    assert np.all(np.abs(got - want) <= tol * np.maximum(1.0, np.abs(want)))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))  # same order, same roundings
    assert cases.digest(got) == GOLD["conv_f32"][case[0]]


def test_conv_i8_larger_shapes(gpu, orc):
    """yolov5s-shaped layers at sizes the oracle still finishes quickly: M tails, all N tiles"""
    check_family(gpu, orc, "big")


def test_conv_i8_stem_edges(gpu, orc):
    """the small-channel (RGB stem) kernel: tiles hanging over every image edge (its edge units load at a clamped
    column and shift), widths below one 4-pixel unit and 1 / 2 / 4 channels (the per-byte path), several tiles per
    workgroup, strides 1 and 2"""
    check_family(gpu, orc, "stem")


@pytest.mark.parametrize("slots,stages", [(1, 2), (3, 2), (5, 2), (0, 2)])  # (the three-stage walker was pruned in round 4)
def test_conv_i8_persistent_tile_walk(gpu, orc, slots, stages):
    """the persistent kernel walking SEVERAL pixel tiles per workgroup (cross-tile prefetch, counted vmcnt
    across the epilogue's buffer stores): force few workgroups so that small inputs exercise it; both ring
    depths; K loops of 1, 2, 4, 9 and 18 steps; pixel counts that are not a multiple of the tile"""
    try:
        gpu.set_tuning("persist", 1)
        gpu.set_tuning("persist_maxk", 1 << 20)
        gpu.set_tuning("persist_slots", slots)
        gpu.set_tuning("persist_stages", stages)
        gpu.set_tuning("variant", 2 if stages == 2 else 6)  # the tile-walking form wherever a layer has it
        check_family(gpu, orc, "walk", (slots, stages))
        # a whole graph (fused SiLU LUT epilogues, zero-copy concat slices, strided outputs), several frames
        import marsfile
        from conftest import lcg_frame
        d = gpu.synth_model(width_x16=4, input_hw=96, seed=21)
        hdr, tensors, _ = marsfile.parse(d)
        nb = marsfile.tensor_nbytes(tensors[hdr["inputs"][0]])
        m = gpu.Model(d, batch=3)
        xs = [lcg_frame(0xAB0000 + f, nb) for f in range(3)]
        for f in range(3):
            m.input_view(0)[f] = xs[f]
        m.run()
        for f in range(3):
            g = orc.Graph(d)
            g.set_input(0, xs[f].tobytes())
            assert g.run() == 0
            for oi, ti in enumerate(hdr["outputs"]):
                assert np.array_equal(m.output_view(oi)[f], g.tensor(ti))
        m.close()
    finally:
        gpu.set_tuning("variant", 0)
        gpu.set_tuning("persist_slots", 0)
        gpu.set_tuning("persist_stages", 2)
        gpu.set_tuning("persist_maxk", 8)


@pytest.mark.parametrize("slots,variant", [(1, 14), (3, 15), (0, 14), (0, 15)])
def test_conv_i8_tile_walk_resident_weights(gpu, orc, slots, variant):
    """variants 14 / 15: the tile walker with the weights of its channel tile fetched once and kept in LDS (the ring
    carries pixel tiles only); few workgroups so that every one walks several tiles; K loops of 1..18 steps; ragged
    rows; then a whole graph, whose never-materialised concat inputs take the same form"""
    try:
        gpu.set_tuning("persist_slots", slots)
        gpu.set_tuning("variant", variant)
        check_family(gpu, orc, "wres", (slots, variant))
        import marsfile
        from conftest import lcg_frame
        d = gpu.synth_model(width_x16=4, input_hw=96, seed=22)
        hdr, tensors, _ = marsfile.parse(d)
        nb = marsfile.tensor_nbytes(tensors[hdr["inputs"][0]])
        m = gpu.Model(d, batch=3)
        xs = [lcg_frame(0xAC0000 + f, nb) for f in range(3)]
        for f in range(3):
            m.input_view(0)[f] = xs[f]
        m.run()
        for f in range(3):
            g = orc.Graph(d)
            g.set_input(0, xs[f].tobytes())
            assert g.run() == 0
            for oi, ti in enumerate(hdr["outputs"]):
                assert np.array_equal(m.output_view(oi)[f], g.tensor(ti))
        m.close()
    finally:
        gpu.set_tuning("variant", 0)
        gpu.set_tuning("persist_slots", 0)


@pytest.mark.parametrize("variant,slots", [(12, 0), (13, 0), (18, 0), (19, 0)])
def test_conv_i8_wide_one_tile_forms(gpu, orc, variant, slots):
    """variant 12 (two K slices per ring stage), 13 (256 x 128 tile on an 8-wave workgroup), 18 / 19 (128-byte K steps:
    whole-line DMA requests, 128-byte LDS rows, 4 / 8 waves; layers with fewer than 128 input channels fall back to the
    default there): deep K loops, 128 / 256 output channels, pixel counts that are not multiples of the tile, aligned and
    ragged (255) rows.  (Variants 16, 17 and 20 -- patch-staged input with streamed weights, the two-team strip kernel, the
    persistent 128-byte-step tile -- were measured in round 2, never chosen by the policy or the tuner on a BASELINE
    workload, and removed in round 3: DESIGN.md section 5.)"""
    try:
        gpu.set_tuning("variant", variant)
        gpu.set_tuning("persist_slots", slots)
        check_family(gpu, orc, "wide", variant)
    finally:
        gpu.set_tuning("variant", 0)
        gpu.set_tuning("persist_slots", 0)


@pytest.mark.parametrize("slots", [0, 2, 5])
def test_conv_i8_rows_persistent_deep(gpu, orc, slots):
    """variant 20 (round 4, conv_i8_rows): deep 3x3 stride-1 layers as one persistent 8-wave workgroup per CU over tiles of
    256 consecutive pixels -- patch-staged input (64-channel chunks, two buffers), streamed weights in K-step blocks, the K
    stream in pairs of steps with waves 4-7 one phase behind waves 0-3, the epilogue in line after a tile's last pair,
    hand-counted vmcnt.  Map widths 20 / 40 (the instantiated patch pitches; the kernel declines others), heights that make
    tiles straddle rows and FRAMES (the zero rows between stacked frames), 128 / 256 / 512 input channels (2 / 4 / 8 chunks),
    128 / 256 output channels, partial last tiles; `slots` forces 2 or 5 workgroups so that each walks many tiles.  A launch
    counter proves that every case took this kernel (a forced variant falls back silently where the geometry is declined).
    Direct calls (no table, one frame) and whole graphs with the fused SiLU table and several frames."""
    count = gpu.lib().mhip_conv_i8_rows_launches
    count.restype = __import__("ctypes").c_ulong
    try:
        gpu.set_tuning("variant", 20)
        gpu.set_tuning("persist_slots", slots)
        n0 = count()
        check_family(gpu, orc, "rows", slots)
        assert count() - n0 == len(cases.CONV_I8_FAMILIES["rows"][1])  # one launch per case: nothing fell back
        # declined shapes keep working through the default policy: an 80-wide map, an odd chunk count, a short map whose
        # tiles would hold too many frame boundaries
        n0 = count()
        for i, (h, w, ic, oc) in enumerate([(7, 80, 128, 128), (20, 20, 192, 128), (3, 40, 256, 128)]):
            case = ("rowsx%d" % i, 1, h, w, ic, oc, 3, 3, 1, 1, 1, 1, h, w, 0.03, 0.003 / (9 * ic) ** 0.5 * 8, 0.05, True)
            a = cases.conv_i8_call(gpu.conv2d_int8, case, 9)
            b = cases.conv_i8_call(orc.conv2d_int8, case, 9)
            assert np.array_equal(a, b), (case[0], slots, int((a != b).sum()))
        assert count() == n0
        import marsfile
        for (h, w, ic, oc, frames) in [(20, 20, 256, 256, 7), (13, 40, 128, 128, 5), (5, 80, 128, 128, 3)]:  # (the 80-wide one: default policy)
            G = marsfile.Graph()
            rng = np.random.default_rng(h * 100 + w)
            x = G.tensor([1, h, w, ic], scale=4 / 127)
            a_ = G.tensor([1, h, w, oc], scale=0.03125)
            g_ = G.tensor([1, h, w, oc], scale=1 / 127)
            o = G.tensor([1, h, w, oc], scale=4 / 127)
            wt = G.tensor([oc, 3, 3, ic], scale=0.0005, data=rng.integers(-127, 128, (oc, 3, 3, ic), dtype=np.int8))
            bt = G.tensor([oc], dtype=marsfile.I32, scale=1.0, data=rng.integers(-500, 500, oc, dtype=np.int32))
            G.conv(x, a_, wt, bt, (3, 3), (1, 1))
            G.layer(marsfile.SIGMOID, [a_], [g_])
            G.layer(marsfile.MUL, [a_, g_], [o])
            d = G.serialise([x], [o])
            hdr, tensors, _ = marsfile.parse(d)
            m = gpu.Model(d, batch=frames)
            xs = rng.integers(0, 256, m.input_view(0).shape, dtype=np.uint8)
            m.input_view(0)[:] = xs
            m.run()
            for f in range(frames):
                gr = orc.Graph(d)
                gr.set_input(0, xs[f].tobytes())
                assert gr.run() == 0
                want = gr.tensor(hdr["outputs"][0])
                assert np.array_equal(m.output_view(0)[f], want), (h, w, ic, oc, f, slots)
            m.close()
    finally:
        gpu.set_tuning("variant", 0)
        gpu.set_tuning("persist_slots", 0)


@pytest.mark.parametrize("variant,ring", [(9, 0), (10, 0), (11, 0), (10, 1), (9, 3), (11, 4), (10, 4)])
def test_conv_i8_patch_staged(gpu, orc, variant, ring):
    """the patch-staged kernel (input patch of a tile staged once in LDS, weights resident, taps fed from LDS):
    3x3 / 5x5 / 3x1 kernels, stride 1 and 2 (de-interleaved patch columns), in_c 16 (round 6) / 32 / 64, partial tiles at the
    right and bottom edges, SAME padding on every side, several tiles per workgroup.  `ring` forces the depth of the patch
    ring (1 = one buffer, no prefetch; 3 / 4 = two / three patches in flight behind the one being computed, with an LDS
    budget that holds them), so that the hand-counted vector-memory waits are exercised at every depth, on interior tiles
    (buffer-addressed LDS-DMA) and edge tiles alike, with and without the wave-private residual staging of a folded Add"""
    try:
        gpu.set_tuning("variant", variant)
        gpu.set_tuning("patch_ring", ring)
        gpu.set_tuning("patch_lds_kb", 160 if ring > 2 else 80)
        for slots in (0, 3):
            gpu.set_tuning("persist_slots", slots)
            check_family(gpu, orc, "patch", (variant, ring, slots))
        gpu.set_tuning("persist_slots", 0)
        import marsfile
        from conftest import lcg_frame
        d = gpu.synth_model(width_x16=4, input_hw=256, seed=22)  # 128x128 / 64x64 maps: eligible layers exist
        hdr, tensors, _ = marsfile.parse(d)
        nb = marsfile.tensor_nbytes(tensors[hdr["inputs"][0]])
        m = gpu.Model(d, batch=2)
        xs = [lcg_frame(0xAC0000 + f, nb) for f in range(2)]
        for f in range(2):
            m.input_view(0)[f] = xs[f]
        m.run()
        g = orc.Graph(d)
        g.set_input(0, xs[1].tobytes())
        assert g.run() == 0
        for oi, ti in enumerate(hdr["outputs"]):
            assert np.array_equal(m.output_view(oi)[1], g.tensor(ti))
        m.close()
    finally:
        gpu.set_tuning("variant", 0)
        gpu.set_tuning("persist_slots", 0)
        gpu.set_tuning("patch_ring", 0)
        gpu.set_tuning("patch_lds_kb", 80)


def test_mxu_f32_elementwise(gpu):
    L = gpu.lib()
    n = 100003
    a, b = cases.f32(1, n, -3, 3), cases.f32(2, n, -3, 3)
    out = np.zeros(n, np.float32)
    L.mxu_mul_f32(out.ctypes.data, a.ctypes.data, b.ctypes.data, n)
    assert np.array_equal(out, a * b)
    L.mxu_add_f32(out.ctypes.data, a.ctypes.data, b.ctypes.data, n)
    assert np.array_equal(out, a + b)
    L.mxu_sub_f32(out.ctypes.data, a.ctypes.data, b.ctypes.data, n)
    assert np.array_equal(out, a - b)
    L.mxu_relu_f32(out.ctypes.data, a.ctypes.data, n)
    assert np.array_equal(out, np.where(a > 0, a, a * np.float32(0)))
    L.mxu_init(None)
    assert L.mxu_is_initialized() == 1


@pytest.mark.parametrize("case", cases.YOLO_CASES, ids=lambda c: c[0])
def test_decode_and_nms_bit_exact(gpu, orc, case):
    pred, npred, scale = cases.yolo_pred(case)
    raw = gpu.parse_output(pred, npred, scale)
    want_raw = orc.parse_output(pred, npred, scale)
    assert raw.tobytes() == want_raw.tobytes()
    kept = gpu.nms(raw, 0.45)
    want = orc.nms(want_raw, 0.45)
    assert len(kept) == len(want)
    assert kept.tobytes() == want.tobytes()                      # same boxes in the same order (ties included)
    g = GOLD["yolo"][case[0]]
    assert (len(raw), len(kept)) == (g["raw"], g["kept"])
    assert cases.digest(raw) == g["raw_digest"] and cases.digest(kept) == g["kept_digest"]
    k2 = gpu.nms(raw, 0.1)
    assert k2.tobytes() == orc.nms(want_raw, 0.1).tobytes()


@pytest.mark.parametrize("n,levels,seed", [(1, 1, 0), (2, 1, 1), (3, 2, 2), (64, 3, 3), (65, 2, 4), (257, 7, 5), (513, 40, 6), (999, 2, 7),
                                            (1000, 1, 8), (1000, 5, 9), (1000, 90, 10), (1000, 700, 11), (1000, 100000, 12), (777, 13, 13)])
def test_sort_tie_permutation_closed_form(gpu, orc, n, levels, seed):
    """the reference's exchange sort is not stable; the device derives the permutation it leaves in closed form (bitonic sort +
    a queue replay per tie group, yolo_tail.hip "sort").  Candidate lists with few distinct confidences -- groups of 2 up to
    all 1000 -- against the oracle's literal double loop; boxes far apart and in distinct classes so that nothing is suppressed
    and the kept list IS the sorted list."""
    rng = np.random.default_rng(seed)
    d = np.zeros(n, dtype=gpu.DET_DTYPE)
    vals = (0.25 + 0.7 * rng.random(levels)).astype(np.float32)
    d["conf"] = vals[rng.integers(0, levels, n)]
    d["x"] = np.arange(n, dtype=np.float32) * 100.0  # disjoint boxes
    d["y"] = rng.random(n).astype(np.float32)
    d["w"] = 1.0
    d["h"] = 1.0
    d["cls"] = np.arange(n) % 80
    want = orc.nms(d, 0.45)
    got = gpu.nms(d, 0.45)
    assert len(want) == n and got.tobytes() == want.tobytes()
    # and with suppression on top: overlapping boxes in few classes
    d["x"] = rng.integers(0, 20, n).astype(np.float32)
    d["cls"] = rng.integers(0, 3, n)
    assert gpu.nms(d, 0.3).tobytes() == orc.nms(d, 0.3).tobytes()


def test_sort_with_nan_confidence_follows_the_reference_loop(gpu, orc):
    """`>` does not order NaN: the device falls back to the literal double loop for such a frame"""
    rng = np.random.default_rng(3)
    n = 200
    d = np.zeros(n, dtype=gpu.DET_DTYPE)
    d["conf"] = (0.25 + 0.7 * rng.random(n)).astype(np.float32)
    d["conf"][[3, 50, 51, 199]] = np.nan
    d["x"] = np.arange(n, dtype=np.float32) * 100.0
    d["w"] = 1.0
    d["h"] = 1.0
    d["cls"] = np.arange(n) % 80
    want, got = orc.nms(d, 0.45), gpu.nms(d, 0.45)
    assert len(want) == len(got) and got.tobytes() == want.tobytes()


@pytest.mark.parametrize("scale", [0.05, -0.05, 0.0, 1e8, 3e9, 1e-45, 1e-40, float("nan"), float("inf"), 1e-3])
def test_decode_odd_scales(gpu, orc, scale):
    """the class argmax compares int8 bytes when value[q] = q * scale is strictly increasing, and walks the float table
    otherwise (negative / zero / NaN / underflowing scales); huge scales exercise the reference's `top = -1e9` start"""
    rng = np.random.default_rng(17)
    npred = 3000
    p = rng.integers(-128, 128, (npred, 85), dtype=np.int8)
    p[::3, 5:] = rng.integers(-128, -9, (len(p[::3]), 80), dtype=np.int8)  # rows whose classes all sit below -1e9 at scale 1e8
    p[1::7, 5:] = -128
    p[:, 4] = rng.integers(-20, 128, npred, dtype=np.int8)
    raw = gpu.parse_output(p.ravel(), npred, scale)
    want = orc.parse_output(p.ravel(), npred, scale)
    assert len(raw) == len(want) and np.array_equal(raw["cls"], want["cls"])
    for k in ("x", "y", "w", "h", "conf"):  # bit for bit, except that a NaN's sign and payload are nobody's contract
        a, b = raw[k], want[k]
        assert np.array_equal(np.isnan(a), np.isnan(b)) and a[~np.isnan(a)].tobytes() == b[~np.isnan(b)].tobytes(), k


def test_decode_cap_and_maxd(gpu, orc):
    pred, npred, scale = cases.yolo_pred(cases.YOLO_CASES[0])
    a = gpu.parse_output(pred, npred, scale, maxd=37)
    b = orc.parse_output(pred, npred, scale, maxd=37)
    assert len(a) == 37 and a.tobytes() == b.tobytes()



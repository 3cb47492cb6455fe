"""Host-side packers of the HIP library, on the CPU (no device call): the float32 -> bf16 split of conv_f32_split's weights."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "thingino-accel_amd"))
import marsrt  # noqa: E402


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


@pytest.mark.parametrize("out_c,in_c,kh,kw,stride", [(5, 3, 3, 3, 1), (40, 8, 3, 3, 2), (130, 16, 1, 1, 1), (32, 3, 6, 6, 2), (7, 5, 5, 5, 1)])
def test_conv_f32_split_pack_is_an_exact_split(out_c, in_c, kh, kw, stride):
    """mhip_conv_f32_split_pack (csrc/hip/conv_f32_split.hip, host code): three bf16 planes [oc_pad][k_pad] with
    w == hi + mid + lo EXACTLY for every weight (hi = bf16(w) and mid = bf16(w - hi) rounded to nearest, so also
    |w - hi - mid| <= 2^-16 |w|: what the three-product mode drops), zeros in every padding position, and -- for an odd kernel
    width under stride 2 -- one zero column appended to every kernel row."""
    L = marsrt.lib()
    f = L.mhip_conv_f32_split_pack
    f.restype = C.c_size_t
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(out_c * 100 + kw)
    w = ((rng.random((out_c, in_c, kh, kw), dtype=np.float32) * 2 - 1) * np.float32(10.0) ** rng.integers(-6, 3, (out_c, 1, 1, 1))).astype(np.float32)
    w[0, 0, 0, 0] = 0.0
    w[-1, -1, -1, -1] = np.float32(1.0) + np.float32(2.0) ** -23  # needs all three pieces
    n = f(out_c, in_c, kh, kw, stride, 3, None, None)
    assert f(out_c, in_c, kh, kw, stride, 2, None, None) == n // 3 * 2  # mode 3 reads two planes: only those are packed for it
    kwp = kw + 1 if (stride == 2 and kw > 1 and kw % 2) else kw
    K = in_c * kh * kwp
    kp = (K + 63) // 64 * 64 + 64
    ocp = (out_c + 127) // 128 * 128
    assert n == 3 * ocp * kp * 2
    buf = np.zeros(n // 2, dtype=np.uint16)
    assert f(out_c, in_c, kh, kw, stride, 3, w.ctypes.data, buf.ctypes.data) == n
    planes = buf.reshape(3, ocp, kp)
    hi, mid, lo = (bf16_to_f32(planes[i]) for i in range(3))
    got = np.zeros((out_c, in_c, kh, kwp), dtype=np.float64)
    for pl in (hi, mid, lo):
        got += pl[:out_c, :K].reshape(out_c, in_c, kh, kwp).astype(np.float64)
    assert np.array_equal(got[..., :kw], w.astype(np.float64))          # exact, in real arithmetic
    assert not got[..., kw:].any()                                      # the appended column
    for pl in planes:
        assert not pl[out_c:].any() and not pl[:, K:].any()             # padding rows and taps
    two = hi[:out_c, :K].reshape(out_c, in_c, kh, kwp)[..., :kw].astype(np.float64) + mid[:out_c, :K].reshape(out_c, in_c, kh, kwp)[..., :kw].astype(np.float64)
    assert (np.abs(w.astype(np.float64) - two) <= np.abs(w.astype(np.float64)) * 2.0 ** -16).all()


# ---------------------------------------------------------------------------------------------------------------------
# conv_f32_patch (csrc/hip/conv_f32_patch.hip): everything index-shaped in that kernel is host code -- the layer geometry
# (strip order, patch rows / pitch), the unit table, the chunk schedule, the weights' K order.  The emulation below walks the
# kernel's control flow (prologue, per-step patch action, ring slots, the four units of a step, tile -> pixel mapping) in numpy
# with those tables and must reproduce a direct convolution; a slot read before its chunk was committed, a row outside the
# patch or a wrong tap offset shows as a wrong sum.

GEOM_FIELDS = ("s kh kw pad C nchunk U SW nstrips H_in W_in H_out W_out HV PR PWP PWH dx slotpix nsteps ngrp nitems BM kp oc_pad "
               "tab_ints ndummy cpi bn woff poff lds_bytes nb rec ndma").split()


def patch_geom(L, out_c, in_c, kh, kw, s, pad, in_h, in_w, out_h, out_w, rec=0):
    f = L.mhip_conv_f32_patch_geom2
    f.restype = C.c_int
    f.argtypes = [C.c_int] * 11 + [C.c_void_p, C.c_int]
    v = np.zeros(64, dtype=np.int32)
    n = f(out_c, in_c, kh, kw, s, pad, in_h, in_w, out_h, out_w, rec, v.ctypes.data, 64)
    return dict(zip(GEOM_FIELDS, (int(x) for x in v[:n]))) if n else None


def patch_pack(L, w, s, pad, in_h, in_w, out_h, out_w, rec=0):
    out_c, in_c, kh, kw = w.shape
    f = L.mhip_conv_f32_patch_pack2
    f.restype = C.c_size_t
    f.argtypes = [C.c_int] * 11 + [C.c_void_p, C.c_void_p]
    n = f(out_c, in_c, kh, kw, s, pad, in_h, in_w, out_h, out_w, rec, None, None)
    buf = np.zeros(n, dtype=np.uint8)
    assert f(out_c, in_c, kh, kw, s, pad, in_h, in_w, out_h, out_w, rec, w.ctypes.data, buf.ctypes.data) == n
    return buf


def direct_conv(x, w, s, pad, out_h, out_w):
    frames, C_, H, W = x.shape
    out_c, _, kh, kw = w.shape
    xp = np.zeros((frames, C_, H + 2 * pad + 8, W + 2 * pad + 8), dtype=np.float64)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    out = np.zeros((frames, out_c, out_h, out_w), dtype=np.float64)
    for ky in range(kh):
        for kx in range(kw):
            win = xp[:, :, ky:ky + (out_h - 1) * s + 1:s, kx:kx + (out_w - 1) * s + 1:s]
            out += np.einsum("fchw,oc->fohw", win, w[:, :, ky, kx].astype(np.float64))
    return out


def emulate_patch_kernel(g, tabs, wk, x, frames, t_first, t_end, recin=False):
    """the workgroup that walks tiles [t_first, t_end): returns {tile: acc[256, oc_rows]} following conv_f32_patch's control flow.
    recin: the record-input form (conv_f32_patch<..., RECIN>): a thread's items are the 16-byte half records tid + 512 q of the slot
    (position = item / 2 -> (patch row, column) as the kernel derives it), copied through registers at the same schedule"""
    s, pad, PWP, PWH, SW, HV = g["s"], g["pad"], g["PWP"], g["PWH"], g["SW"], g["HV"]
    nchunk, nsteps, slotpix = g["nchunk"], g["nsteps"], g["slotpix"]
    dutab, sched = tabs[:nsteps * 4].reshape(nsteps, 4), tabs[nsteps * 4:nsteps * 5]
    BN = g["bn"]
    total = frames * g["H_out"] * g["W_out"]
    ntiles = (total + BN - 1) // BN
    nsegs = frames * g["nstrips"]
    slots = np.full((2 * slotpix, 8), np.nan)  # NaN = never written: a read of it poisons the sum
    cmap = (lambda v: (v >> 1) + (v & 1) * PWH) if s == 2 else (lambda v: v)

    def tile_v0(t):
        R0 = (t * BN) // SW
        seg0 = R0 // g["H_out"]
        return seg0 * HV + (R0 - seg0 * g["H_out"]) * s

    def rowtab(t):
        rows = []
        for r in range(g["PR"]):
            V = tile_v0(t) + r
            seg, iy = V // HV, V % HV - pad
            f, st = seg // g["nstrips"], seg % g["nstrips"]
            ok = t < ntiles and seg < nsegs and 0 <= iy < g["H_in"]
            rows.append(((f, iy) if ok else None, st * SW * s - pad - g["dx"]))
        return rows

    tables = {}

    cpi = g["cpi"]
    nsub = 8 // cpi  # items per (row, 4-column group): each fetches cpi of the chunk's 8 channels
    assert g["nitems"] == g["PR"] * g["ngrp"] * nsub

    def fetch_rec(t, chunk):
        vals = np.zeros((slotpix, 8))
        for pos in range(slotpix):
            r, cp = divmod(pos, PWP)
            if r >= g["PR"]:
                continue
            v = (2 * cp if cp < PWH else 2 * (cp - PWH) + 1) if s == 2 else cp
            src, xal = tables[t & 1][r]
            xx = xal + v
            if src is not None and 0 <= xx < g["W_in"]:
                vals[pos] = x[src[0], chunk * 8:chunk * 8 + 8, src[1], xx]
        return vals

    def commit_rec(slot, vals):
        slots[slot * slotpix:(slot + 1) * slotpix] = vals

    def fetch(t, chunk):
        if recin:
            return fetch_rec(t, chunk)
        vals = np.zeros((g["nitems"], 4, cpi))
        for it in range(g["nitems"]):
            cell, ich = divmod(it, nsub)
            ir, igq = divmod(cell, g["ngrp"])
            src, xal = tables[t & 1][ir]
            xx = xal + 4 * igq
            if src is not None and 0 <= xx < g["W_in"]:
                f, iy = src
                c0 = chunk * 8 + ich * cpi
                vals[it] = x[f, c0:c0 + cpi, iy, xx:xx + 4].T
        return vals

    def commit(slot, vals):
        if recin:
            return commit_rec(slot, vals)
        for it in range(g["nitems"]):
            cell, ich = divmod(it, nsub)
            ir, igq = divmod(cell, g["ngrp"])
            for i in range(4):
                slots[slot * slotpix + ir * PWP + cmap(4 * igq + i), ich * cpi:(ich + 1) * cpi] = vals[it, i]

    # prologue
    tables[t_first & 1] = rowtab(t_first)
    breg = fetch(t_first, 0)
    for i in range(nsteps):
        if sched[i] > nchunk:
            cc = sched[i] - 1 - nchunk
            commit(cc & 1, breg)
            breg = fetch(t_first, cc + 1)
    out = {}
    for t in range(t_first, t_end):
        V0 = tile_v0(t)
        q = t * BN + np.arange(BN)
        R, xs = q // SW, q % SW
        seg, y = R // g["H_out"], R % g["H_out"]
        pbase = np.where(q < total, (seg * HV + y * s - V0) * PWP + xs, 0)
        acc = np.zeros((BN, wk.shape[0]))
        for ks in range(nsteps):
            if ks == 0:
                tables[(t + 1) & 1] = rowtab(t + 1)
            if sched[ks]:
                cc = sched[ks] - 1
                commit(cc & 1, breg)
                nx = cc + 1
                breg = fetch(t + 1, nx - nchunk) if nx >= nchunk else fetch(t, nx)
            for fc in range(4):
                e = dutab[ks, fc]  # byte offset into the patch ring (32 bytes per patch pixel), or -1
                if e < 0:
                    continue  # dummy unit: zeros
                assert e % 32 == 0
                P = pbase + e // 32
                assert (P >= 0).all() and (P < 2 * slotpix).all()
                xv = slots[P]  # [256, 8]
                acc += xv @ wk[:, (ks * 4 + fc) * 8:(ks * 4 + fc) * 8 + 8].T
        out[t] = (q, acc)
    return out


def emulate_prec_kernel(g, tabs, wk, x, frames, t_first, t_end):
    """conv_f32_prec (record-format input): a chunk's slot is filled by LDS-DMA in the step the schedule names -- block i of a slot =
    patch positions 32 i .. + 31, position -> (patch row, column) as the kernel derives it, zeros outside the image -- and read from
    there; the ring has g['nb'] slots.  Returns {tile: (pixels, acc)} like emulate_patch_kernel."""
    s, pad, PWP, PWH, SW, HV = g["s"], g["pad"], g["PWP"], g["PWH"], g["SW"], g["HV"]
    nchunk, nsteps, slotpix, nb, U = g["nchunk"], g["nsteps"], g["slotpix"], g["nb"], g["U"]
    dutab, sched = tabs[:nsteps * 4].reshape(nsteps, 4), tabs[nsteps * 4:nsteps * 5]
    BN = g["bn"]
    total = frames * g["H_out"] * g["W_out"]
    ntiles = (total + BN - 1) // BN
    nsegs = frames * g["nstrips"]
    assert slotpix % 32 == 0 and g["ndma"] == slotpix // 32 and nb in (2, 4) and nchunk % nb == 0
    slots = np.full((nb * slotpix, 8), np.nan)

    def tile_v0(t):
        R0 = (t * BN) // SW
        seg0 = R0 // g["H_out"]
        return seg0 * HV + (R0 - seg0 * g["H_out"]) * s

    def rowtab(t):
        rows = []
        for r in range(g["PR"]):
            V = tile_v0(t) + r
            seg, iy = V // HV, V % HV - pad
            f, st = seg // g["nstrips"], seg % g["nstrips"]
            ok = t < ntiles and seg < nsegs and 0 <= iy < g["H_in"]
            rows.append(((f, iy) if ok else None, st * SW * s - pad - g["dx"]))
        return rows

    tables = {}
    pending = []  # chunks issued and not yet read for the first time (the kernel's FIFO of ages)

    def dma(t, c):
        slot = c % nb
        for pos in range(slotpix):
            r, cp = divmod(pos, PWP)
            rec = np.zeros(8)
            if r < g["PR"]:
                v = (2 * cp if cp < PWH else 2 * (cp - PWH) + 1) if s == 2 else cp
                src, xal = tables[t & 1][r]
                xx = xal + v
                if src is not None and 0 <= xx < g["W_in"]:
                    rec = x[src[0], c * 8:c * 8 + 8, src[1], xx]
            slots[slot * slotpix + pos] = rec
        pending.append(c)

    first_read = {(c * U) // 4: c for c in range(nchunk)}
    tables[t_first & 1] = rowtab(t_first)
    for i in range(nsteps):
        sc = int(sched[i]) & 0xffff
        if sc > nchunk:
            dma(t_first, sc - 1 - nchunk)
    assert pending and pending[0] == 0
    pending.pop(0)  # chunk 0: waited for by the prologue
    out = {}
    for t in range(t_first, t_end):
        V0 = tile_v0(t)
        q = t * BN + np.arange(BN)
        R, xs = q // SW, q % SW
        seg, y = R // g["H_out"], R % g["H_out"]
        pbase = np.where(q < total, (seg * HV + y * s - V0) * PWP + xs, 0)
        acc = np.zeros((BN, wk.shape[0]))
        for ks in range(nsteps):
            if ks == 0:
                tables[(t + 1) & 1] = rowtab(t + 1)
            w_ = int(sched[ks])
            sc = w_ & 0xffff
            if sc:
                rel = sc - 1
                assert rel < nchunk or ks >= 2, "no DMA for the next tile before step 2 (its row table)"
                if rel >= nchunk:
                    dma(t + 1, rel - nchunk)
                else:
                    dma(t, rel)
            assert len(pending) <= nb
            for fc in range(4):
                e = dutab[ks, fc]
                if e < 0:
                    continue
                assert e % 32 == 0
                P = pbase + e // 32
                assert (P >= 0).all() and (P < nb * slotpix).all()
                acc += slots[P] @ wk[:, (ks * 4 + fc) * 8:(ks * 4 + fc) * 8 + 8].T
            # bit 16: a chunk is read for the first time in the next step -- it must be the oldest pending one
            nxt = (ks + 1) % nsteps
            assert bool(w_ >> 16) == (nxt in first_read), (ks, w_)
            if w_ >> 16:
                assert pending and pending[0] == first_read[nxt], (ks, pending)
                pending.pop(0)
        out[t] = (q, acc)
    return out


PATCH_SHAPES = [
    # out_c, in_c, k, s, pad, in_h, in_w, frames
    (16, 32, 3, 1, 1, 20, 20, 3),    # whole-row tiles running on into the next frame (the 20-wide maps)
    (24, 64, 3, 1, 1, 12, 40, 2),    # 40-wide
    (8, 32, 3, 1, 1, 24, 160, 1),    # wide map: 2-D tiles out of 32-column strips
    (8, 32, 3, 2, 1, 40, 40, 2),     # stride 2: de-interleaved patch columns, 20-wide output
    (8, 64, 3, 2, 1, 32, 160, 1),    # stride 2, 80-wide output: 16-column strips
    (8, 32, 5, 1, 2, 16, 24, 2),     # 25 taps
    (8, 32, 6, 2, 2, 32, 32, 1),     # the stem's kernel geometry on 32 channels
    (40, 96, 3, 1, 1, 9, 20, 5),     # short frames: several frame boundaries inside one tile; 12 chunks
    (200, 64, 3, 1, 1, 16, 40, 3),   # two 128-channel tiles: 256-pixel tiles (the cases above with <= 64 channels take 512-pixel tiles)
    (64, 32, 3, 2, 1, 64, 320, 1),   # the twin's second layer's geometry: a 512-pixel tile's patch does not fit, 256 it is
]


@pytest.mark.parametrize("rec", [0, 1])
@pytest.mark.parametrize("shape", PATCH_SHAPES)
def test_conv_f32_patch_tables_reproduce_a_direct_convolution(shape, rec):
    out_c, in_c, k, s, pad, in_h, in_w, frames = shape
    out_h, out_w = (in_h + s - 1) // s, (in_w + s - 1) // s
    L = marsrt.lib()
    g = patch_geom(L, out_c, in_c, k, k, s, pad, in_h, in_w, out_h, out_w, rec)
    if rec and g is None:  # the record form wants four ring slots: the large stride-2 patches leave no room (those layers keep the register-staged form)
        assert s == 2 and patch_geom(L, out_c, in_c, k, k, s, pad, in_h, in_w, out_h, out_w, 0) is not None
        pytest.skip("no record form for this shape (four slots do not fit)")
    assert g is not None, "the kernel must take this shape"
    assert g["rec"] == rec and g["nb"] == (4 if rec else 2)
    assert g["bn"] == (512 if out_c <= 64 and s == 1 else 256)  # (stride 2: the 512-pixel tile's patch exceeds the fetch items / LDS)
    assert g["nsteps"] % 2 == 0 and g["nitems"] <= 512 and g["lds_bytes"] <= 160 * 1024 and g["PWP"] % 8 == 0 and g["slotpix"] % 32 == 0
    rng = np.random.default_rng(k * 1000 + in_c + s)
    w = (rng.random((out_c, in_c, k, k), dtype=np.float32) - np.float32(0.5)).astype(np.float32)
    x = rng.random((frames, in_c, in_h, in_w), dtype=np.float32).astype(np.float64)
    img = patch_pack(L, w, s, pad, in_h, in_w, out_h, out_w, rec)
    assert len(img) == len(patch_pack(L, w, s, pad, in_h, in_w, out_h, out_w, 0))  # (the planner repacks in place: same size)
    tabb = (g["tab_ints"] * 4 + 255) & ~255
    tabs = img[:g["tab_ints"] * 4].view(np.int32)
    planes = img[tabb:].view(np.uint16).reshape(2, g["oc_pad"], g["kp"])
    wk = (bf16_to_f32(planes[0]).astype(np.float64) + bf16_to_f32(planes[1]).astype(np.float64))[:out_c]
    # the two planes hold w to 2^-16 relative, in the kernel's K order (checked through the convolution below); padding is zero
    assert not planes[:, out_c:].any() and not planes[:, :, g["nchunk"] * g["U"] * 8:].any()
    want = direct_conv(x, w, s, pad, out_h, out_w)
    total = frames * out_h * out_w
    ntiles = (total + g["bn"] - 1) // g["bn"]
    runs = [(0, ntiles)] if ntiles < 3 else [(0, ntiles), (1, 3), (ntiles - 1, ntiles)]  # different first tiles: the prologue's state
    for t0, t1 in runs:
        res = (emulate_prec_kernel if rec else emulate_patch_kernel)(g, tabs, wk, x, frames, t0, t1)
        for t, (q, acc) in res.items():
            ok = q < total
            R, xs = q // g["SW"], q % g["SW"]
            seg, y = R // out_h, R % out_h
            f, st = seg // g["nstrips"], seg % g["nstrips"]
            ref = want[f[ok], :, y[ok], st[ok] * g["SW"] + xs[ok]]
            got = acc[ok]
            assert not np.isnan(got).any(), "tile %d read a slot before its chunk was committed" % t
            assert np.abs(got - ref).max() <= 1e-3 * max(1.0, np.abs(ref).max()), (shape, t)


@pytest.mark.parametrize("shape", [s_ for s_ in PATCH_SHAPES if s_[3] == 2])
def test_conv_f32_patch_record_input_through_registers(shape):
    """the stride-2 shapes whose patch leaves no room for conv_f32_prec's four ring slots read records through registers
    (conv_f32_patch<..., RECIN>): the plain image and schedule, a chunk's staging a copy of half records -- emulated against a direct
    convolution like the float form"""
    out_c, in_c, k, s, pad, in_h, in_w, frames = shape
    out_h, out_w = (in_h + s - 1) // s, (in_w + s - 1) // s
    L = marsrt.lib()
    form = L.mhip_conv_f32_patch_rec_form
    form.restype = C.c_int
    form.argtypes = [C.c_int] * 10
    f = form(out_c, in_c, k, k, s, pad, in_h, in_w, out_h, out_w)
    assert f in (1, 2)
    if f == 1:
        pytest.skip("four slots fit: conv_f32_prec takes this shape")
    g = patch_geom(L, out_c, in_c, k, k, s, pad, in_h, in_w, out_h, out_w, 0)
    assert g["bn"] == 256 and (g["slotpix"] * 2 + 511) // 512 <= 8
    rng = np.random.default_rng(k * 100 + in_c)
    w = (rng.random((out_c, in_c, k, k), dtype=np.float32) - np.float32(0.5)).astype(np.float32)
    x = rng.random((frames, in_c, in_h, in_w), dtype=np.float32).astype(np.float64)
    img = patch_pack(L, w, s, pad, in_h, in_w, out_h, out_w, 0)
    tabb = (g["tab_ints"] * 4 + 255) & ~255
    tabs = img[:g["tab_ints"] * 4].view(np.int32)
    planes = img[tabb:].view(np.uint16).reshape(2, g["oc_pad"], g["kp"])
    wk = (bf16_to_f32(planes[0]).astype(np.float64) + bf16_to_f32(planes[1]).astype(np.float64))[:out_c]
    want = direct_conv(x, w, s, pad, out_h, out_w)
    total = frames * out_h * out_w
    ntiles = (total + g["bn"] - 1) // g["bn"]
    for t0, t1 in ([(0, ntiles)] if ntiles < 3 else [(0, ntiles), (1, 3)]):
        for t, (q, acc) in emulate_patch_kernel(g, tabs, wk, x, frames, t0, t1, recin=True).items():
            ok = q < total
            R, xs = q // g["SW"], q % g["SW"]
            seg, y = R // out_h, R % out_h
            f_, st = seg // g["nstrips"], seg % g["nstrips"]
            ref = want[f_[ok], :, y[ok], st[ok] * g["SW"] + xs[ok]]
            assert not np.isnan(acc[ok]).any() and np.abs(acc[ok] - ref).max() <= 1e-3 * max(1.0, np.abs(ref).max()), (shape, t)


def test_conv_f32_record_forms_of_the_twins_layers():
    """mhip_conv_f32_patch_rec_form: which kernel reads a layer's input when it arrives as records -- the bottlenecks' 3 x 3 (four ring
    slots fit: conv_f32_prec, 1), the stride-2 layers behind the stem / a C3 (2: through registers), shapes the patch kernels decline (0)"""
    L = marsrt.lib()
    form = L.mhip_conv_f32_patch_rec_form
    form.restype = C.c_int
    form.argtypes = [C.c_int] * 10
    for (oc, ic, k, s, h) in ((32, 32, 3, 1, 160), (64, 64, 3, 1, 80), (128, 128, 3, 1, 40), (256, 256, 3, 1, 20)):
        assert form(oc, ic, k, k, s, 1, h, h, h, h) == 1, (oc, ic, h)
    for (oc, ic, h) in ((64, 32, 320), (128, 64, 160), (256, 128, 80), (512, 256, 40)):
        assert form(oc, ic, 3, 3, 2, 1, h, h, h // 2, h // 2) == 2, (oc, ic, h)
    assert form(32, 24, 3, 3, 1, 1, 40, 40, 40, 40) == 0      # fewer than four chunks
    assert form(32, 64, 1, 1, 1, 0, 40, 40, 40, 40) == 0      # 1 x 1
    assert form(32, 3, 6, 6, 2, 2, 640, 640, 320, 320) == 0   # the stem's own shape


def test_conv_f32_patch_declines_what_it_cannot_take():
    L = marsrt.lib()
    assert patch_geom(L, 32, 3, 6, 6, 2, 2, 64, 64, 32, 32) is None      # the RGB stem: 3 channels
    assert patch_geom(L, 32, 64, 1, 1, 1, 0, 40, 40, 40, 40) is None     # 1 x 1: one tap
    assert patch_geom(L, 32, 16, 3, 3, 1, 1, 40, 40, 40, 40) is None     # fewer than four chunks
    assert patch_geom(L, 32, 64, 3, 3, 1, 1, 21, 21, 21, 21) is None     # map width not a multiple of 4
    assert patch_geom(L, 32, 64, 3, 3, 3, 1, 40, 40, 14, 14) is None     # stride 3


def test_conv_f32_stem_pack_order():
    """mhip_conv_f32_stem_pack: two bf16 planes [32][20 units][8]; element par * 4 + ch of unit ky * (kw / 2) + j is
    w[oc][ch][ky][2 j + par], hi + mid == w to 2^-16 relative, every other position zero; shapes the kernel declines give 0"""
    L = marsrt.lib()
    f = L.mhip_conv_f32_stem_pack
    f.restype = C.c_size_t
    f.argtypes = [C.c_int] * 10 + [C.c_void_p, C.c_void_p]
    out_c, in_c, k = 20, 3, 6
    rng = np.random.default_rng(5)
    w = (rng.random((out_c, in_c, k, k), dtype=np.float32) - np.float32(0.5)).astype(np.float32)
    n = f(out_c, in_c, k, k, 2, 2, 64, 64, 32, 32, None, None)
    assert n == 2 * 32 * 20 * 8 * 2
    buf = np.zeros(n // 2, dtype=np.uint16)
    assert f(out_c, in_c, k, k, 2, 2, 64, 64, 32, 32, w.ctypes.data, buf.ctypes.data) == n
    planes = buf.reshape(2, 32, 20, 2, 4)  # plane, channel row, unit, column parity, channel slot
    got = bf16_to_f32(planes[0]).astype(np.float64) + bf16_to_f32(planes[1]).astype(np.float64)
    for ky in range(k):
        for kx in range(k):
            u = ky * (k // 2) + kx // 2
            ref = w[:, :, ky, kx].astype(np.float64)
            assert np.abs(got[:out_c, u, kx & 1, :in_c] - ref).max() <= np.abs(ref).max() * 2.0 ** -16
    assert not got[out_c:].any() and not got[:, 18:].any() and not got[..., in_c:].any()
    assert f(33, 3, 6, 6, 2, 2, 64, 64, 32, 32, None, None) == 0     # more than 32 output channels
    assert f(32, 3, 6, 6, 2, 2, 72, 64, 36, 32, None, None) == 0     # output height not a multiple of 16
    assert f(32, 3, 8, 8, 2, 2, 64, 64, 32, 32, None, None) == 0     # 32 units
    assert f(32, 3, 3, 3, 2, 1, 64, 64, 32, 32, None, None) == 0     # odd kernel width / odd pad
    assert f(32, 8, 6, 6, 2, 2, 64, 64, 32, 32, None, None) == 0     # more than 4 input channels

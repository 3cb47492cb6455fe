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
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(out_c * 100 + kw)
    w = ((rng.random((out_c, in_c, kh, kw), dtype=np.float32) * 2 - 1) * np.float32(10.0) ** rng.integers(-6, 3, (out_c, 1, 1, 1))).astype(np.float32)
    w[0, 0, 0, 0] = 0.0
    w[-1, -1, -1, -1] = np.float32(1.0) + np.float32(2.0) ** -23  # needs all three pieces
    n = f(out_c, in_c, kh, kw, stride, None, None)
    kwp = kw + 1 if (stride == 2 and kw > 1 and kw % 2) else kw
    K = in_c * kh * kwp
    kp = (K + 63) // 64 * 64 + 64
    ocp = (out_c + 127) // 128 * 128
    assert n == 3 * ocp * kp * 2
    buf = np.zeros(n // 2, dtype=np.uint16)
    assert f(out_c, in_c, kh, kw, stride, w.ctypes.data, buf.ctypes.data) == n
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

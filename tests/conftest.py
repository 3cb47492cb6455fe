"""Shared fixtures.  `-m "not gpu"` runs here (no GPU); `-m gpu` runs on an MI355X box.

Only this directory, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
touch oracle/ (the CPU checker).  GPU tests call the product through its C-ABI
(thingino-accel_amd/lib/libnna_mars.so via marsrt.py).
"""
import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_marsrt():
    spec = importlib.util.spec_from_file_location("marsrt", os.path.join(ROOT, "thingino-accel_amd", "marsrt.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def marsrt():
    return load_marsrt()


@pytest.fixture(scope="session")
def orc():
    import orcbind
    orcbind.lib()
    return orcbind


@pytest.fixture(scope="session")
def ref():
    import refbind
    if not refbind.available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    return refbind


@pytest.fixture(scope="session")
def gpu(marsrt):
    """nna_init() on the real device; fails (not skips) if the library cannot reach a GPU."""
    marsrt.nna_init()
    yield marsrt
    marsrt.lib().nna_deinit()


_LCG_JUMP = None  # (mul, add): state after k + 1 steps = mul[k] * s + add[k] (mod 2^32), built once


def lcg_frame(seed, nbytes):
    """SURVEY.md section 8d synthetic frame: x[i] = (int8)(lcg >> 24)."""
    # numpy implementation of a 32-bit LCG (Numerical Recipes constants), vectorised by jumping
    global _LCG_JUMP
    a, c = np.uint64(1664525), np.uint64(1013904223)
    out = np.empty(nbytes, dtype=np.uint8)
    s = np.uint64(seed & 0xFFFFFFFF)
    mask = np.uint64(0xFFFFFFFF)
    block = 1 << 16
    if _LCG_JUMP is None:  # powers for a block jump
        mul = np.empty(block, dtype=np.uint64)
        add = np.empty(block, dtype=np.uint64)
        m, d = np.uint64(1), np.uint64(0)
        for k in range(block):
            m = (m * a) & mask
            d = (d * a + c) & mask
            mul[k], add[k] = m, d
        _LCG_JUMP = (mul, add)
    mul, add = _LCG_JUMP
    i = 0
    while i < nbytes:
        n = min(block, nbytes - i)
        vals = (mul[:n] * s + add[:n]) & mask
        out[i:i + n] = (vals >> np.uint64(24)).astype(np.uint8)
        s = vals[n - 1]
        i += n
    return out


def pattern_input(desc_dtype, nbytes):
    """Input pattern of reference src/mars/mars_test.c:75-86, over numel."""
    if desc_dtype == 0:
        n = nbytes // 4
        return ((np.arange(n) % 256).astype(np.float32) / np.float32(255.0)).astype(np.float32).view(np.uint8)
    return (np.arange(nbytes) % 127).astype(np.int8).view(np.uint8)

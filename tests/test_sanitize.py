"""CPU AddressSanitizer / UBSan build of the host side (loader, planner, tensor and memory shims) driven by
tests/c/fuzz_loader.c over the shipped and synthetic models plus the two crafted files ADVICE (round 1) reported:
an in_c that wrapped kw*in_c inside the weight packer, and an out_c that made the arena doubling loop spin."""
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "thingino-accel_amd")
HOST = os.path.join(PKG, "csrc", "host")
GOLD = os.path.join(ROOT, "tests", "golden", "models")


@pytest.fixture(scope="module")
def fuzz_bin(tmp_path_factory):
    out = tmp_path_factory.mktemp("asan") / "fuzz_loader"
    srcs = [os.path.join(HOST, f) for f in sorted(os.listdir(HOST)) if f.endswith(".c")]
    cmd = ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), "-I" + HOST,
           os.path.join(ROOT, "tests", "c", "fuzz_loader.c")] + srcs + [
           "-L" + os.path.join(PKG, "lib"), "-lnna_mars", "-Wl,-rpath," + os.path.join(PKG, "lib"), "-lm", "-o", str(out)]
    subprocess.check_call(cmd)
    return str(out)


def _run(fuzz_bin, args, timeout=240):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               MARS_HIP_DEVICE="0")
    return subprocess.run([fuzz_bin] + [str(a) for a in args], env=env, capture_output=True, text=True, timeout=timeout)


def _conv_file(in_shape, out_shape, w_shape, kh, kw, in_fmt=7, out_fmt=7):
    """one-conv .mars (int8), layout restated from reference include/mars.h:103-221"""
    def tensor(tid, name, dtype, fmt, shape, off, size, scale):
        sh = list(shape) + [0] * (6 - len(shape))
        return struct.pack("<I60sIII6iQQfi", tid, name.encode(), dtype, fmt, len(shape), *sh, off, size, scale, 0)
    wbytes = 64
    tensors = [tensor(0, "in", 3, in_fmt, in_shape, 0, 0, 0.05), tensor(1, "out", 3, out_fmt, out_shape, 0, 0, 0.05),
               tensor(2, "w", 3, 6, w_shape, 0, wbytes, 0.01)]
    conv = struct.pack("<15I", kh, kw, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 2, 0xFFFFFFFF) + b"\0" * 4
    layer = struct.pack("<4I4I4I", 0, 0, 1, 1, 0, 0, 0, 0, 1, 0, 0, 0) + conv
    assert len(layer) == 112
    woff = 76 + 124 * 3 + 112
    hdr = struct.pack("<IHHIIIIIQQ4I4I", 0x5352414D, 1, 0, 0, 1, 3, 1, 1, woff, wbytes, 0, 0, 0, 0, 1, 0, 0, 0)
    return hdr + b"".join(tensors) + layer + bytes(wbytes)


def test_crafted_files_from_advice(fuzz_bin, tmp_path):
    cases = {
        # kw * in_c wrapped to 64 in int arithmetic; the packer then wrote in_c bytes per tap into a 64-byte row
        "wrap_inc.mars": _conv_file([1, 1, 64, 0x4000001], [1, 1, 1, 16], [16, 1, 64, 0x4000001], 1, 64),
        # oc_pad = (out_c + 31) & ~31 wrapped negative; the arena's capacity doubling never reached it
        "huge_outc.mars": _conv_file([1, 4, 4, 16], [1, 1, 1, 0x7fffffff], [16, 1, 1, 16], 1, 1),
        "nchw_wrap.mars": _conv_file([1, 0x7ffffff1, 1, 1], [1, 16, 1, 1], [16, 1, 1, 1], 1, 1, in_fmt=0, out_fmt=0),
        # NDHWC32 (tag 1) / NMHWSOIB2 (tag 3) descriptors whose channel counts sit next to INT_MAX: the
        # "+ 31" of tensor_byte_size overflowed a signed int (UBSan), and the wrapped sizes asked for gigabytes of staging
        "ndhwc32_wrap.mars": _conv_file([1, 0x7ffffff0, 2, 2], [1, 16, 2, 2], [16, 1, 1, 16], 1, 1, in_fmt=1, out_fmt=1),
        "nmhwsoib2_wrap.mars": _conv_file([0x7fffffe1, 0x7ffffff5, 3, 3], [1, 16, 1, 1], [16, 1, 1, 16], 1, 1, in_fmt=3, out_fmt=3),
        "ndhwc32_out_wrap.mars": _conv_file([1, 16, 4, 4], [1, 0x7fffffff, 4, 4], [16, 1, 1, 16], 1, 1, in_fmt=7, out_fmt=1),
    }
    paths = []
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        paths.append(p)
    r = _run(fuzz_bin, [200, 7] + paths, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr


def test_fuzz_shipped_and_synthetic_models(fuzz_bin, tmp_path, marsrt):
    paths = [os.path.join(GOLD, f) for f in sorted(os.listdir(GOLD)) if f.endswith(".mars")]
    for i, kw in enumerate([dict(width_x16=4, input_hw=64, seed=3), dict(width_x16=4, input_hw=64, seed=4, float32=1),
                            dict(width_x16=4, input_hw=32, seed=5, nchw_int8=1)]):
        p = tmp_path / ("synth%d.mars" % i)
        p.write_bytes(marsrt.synth_model(**kw))
        paths.append(str(p))
    small = [p for p in paths if os.path.getsize(p) < 3_000_000]
    r = _run(fuzz_bin, [60, 11] + small)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert "fuzz_loader:" in r.stdout


def test_compile_step_under_sanitizers(tmp_path):
    """the ONNX -> .mars compile step (host-only C++) built with ASan + UBSan and driven by tests/c/fuzz_compile.cpp over
    the compile tests' graphs: whole, truncated, bit-flipped and with over-long varints spliced in"""
    import numpy as np
    import test_compile as tc
    out = tmp_path / "fuzz_compile"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "fuzz_compile.cpp"),
           os.path.join(HOST, "mars_compile.cpp"), "-o", str(out)]
    subprocess.check_call(cmd)
    files = []
    for i, data in enumerate([tc.small_graph(np.random.default_rng(1), True)[0], tc.small_graph(np.random.default_rng(2), False)[0],
                              tc.qdq_graph(np.random.default_rng(3))[0], tc.runnable_graph(np.random.default_rng(4))]):
        p = tmp_path / ("g%d.onnx" % i)
        p.write_bytes(data)
        files.append(str(p))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([str(out), "1500", "11"] + files, env=env, capture_output=True, timeout=240)
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 0, r.stdout.decode()[-2000:] + err[-4000:]
    assert "ERROR: AddressSanitizer" not in err and "runtime error" not in err and "LeakSanitizer" not in err, err[-4000:]

#!/usr/bin/env python3
"""bench.py against another build of the library, for same-box A/B runs (box-to-box spread is ~3 %, a change worth
keeping is often 0.5-1 %):  tools/ab_bench.py PATH/TO/libnna_mars.so [bench.py arguments]
Alternate the two libraries a few times inside ONE gpurun call and compare the printed values."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402

lib = os.path.abspath(sys.argv[1])
_orig = bench.load_marsrt


def _patched():
    m = _orig()
    m.LIB_PATH = lib
    return m


bench.load_marsrt = _patched
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extra-configs"] + sys.argv[2:]
bench.main()

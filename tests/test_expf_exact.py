"""csrc/expf_exact.h (used by the GPU's float32 sigmoid) against the host libm's expf over a dense
sweep of float bit patterns: every 1021st pattern of the whole 32-bit space plus every value in the
ranges where sigmoid arguments live.  Runs on the CPU (the same header is compiled for the device)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "expf_exact.h"
static unsigned long sweep(uint64_t lo, uint64_t hi, uint64_t stride) {
    unsigned long bad = 0;
    for (uint64_t u = lo; u <= hi; u += stride) {
        float x; uint32_t v = (uint32_t)u, ua, ub; memcpy(&x, &v, 4);
        float a = expf(x), b = expf_exact(x, expf_exact_tab);
        memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
        if (ua != ub && !(a != a && b != b)) bad++;
    }
    return bad;
}
int main(void) {
    unsigned long bad = sweep(0, 0xFFFFFFFFull, 1021);
    bad += sweep(0x3c000000ull, 0x42c00000ull, 3);   /* +2^-7 .. +96  */
    bad += sweep(0xbc000000ull, 0xc2c00000ull, 3);   /* -2^-7 .. -96  */
    printf("%lu\n", bad);
    return bad != 0;
}
'''


def test_expf_exact_matches_libm(tmp_path):
    src = tmp_path / "t.c"
    src.write_text(SRC)
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-std=gnu11", "-I",
                           os.path.join(ROOT, "thingino-accel_amd", "csrc"), str(src), "-o", str(exe), "-lm"])
    out = subprocess.check_output([str(exe)]).decode().strip()
    assert out == "0"

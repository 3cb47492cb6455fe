"""What plain torch copies / fills / reductions reach on this box (the practical HBM ceiling DESIGN.md section 5 quotes; bench.py's own
probe is thingino-accel_amd/lib/libmars_probe.so).  GPU box:   python tools/hbm_rate.py"""
import time

import torch

n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")


def t(f, reps=20):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


dt = t(lambda: b.copy_(a))
print("copy 1 GiB -> 1 GiB: %.1f us, %.2f TB/s (read + write)" % (dt * 1e6, 2 * n / dt / 1e12))
dt = t(lambda: a.zero_())
print("fill 1 GiB: %.1f us, %.2f TB/s" % (dt * 1e6, n / dt / 1e12))
x = a.view(torch.int32)
dt = t(lambda: x.sum())
print("read 1 GiB (sum): %.1f us, %.2f TB/s" % (dt * 1e6, n / dt / 1e12))

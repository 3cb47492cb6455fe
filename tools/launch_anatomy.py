#!/usr/bin/env python3
"""Launch anatomy of the int8 convolution family (round 6): where does every launch of one step spend its time?

Needs the diagnostic build (tools/anatomy_build.sh -> lib/diag/lib_anatomy.so): thread 0 of every workgroup of the conv_i8_* kernels stamps the
100 MHz constant clock (s_memrealtime: one time base for all CUs / XCDs) at its start, at the end of its prologue, at the end of its K stream
and at its end (conv_i8_common.hpp ANAT_*).  One step of the benchmark workload on ONE stream (every launch over the full batch, as bench.py's
roofline times them), then per launch:
  gap   first workgroup start - previous launch's last workgroup end          (the dependent-dispatch boundary)
  T     last end - first start
  head  first start -> 90 % of the launch's peak workgroup concurrency         (dispatch ramp + the first prologues have nothing to overlap)
  tail  concurrency below 90 % of peak -> last end                             (tail round: workgroups finishing while slots stand empty)
  util  sum of workgroup lifetimes / (T x peak concurrency)
  pro / K / epi   shares of the summed workgroup lifetimes: prologue (tables, LUT, resident weights, first pipeline stages issued) | K stream
                  (for tile walkers: all tiles, their per-tile epilogues included) | after it (one-tile form: the epilogue; walkers: the last
                  tile's epilogue, which no next tile's K steps hide)
Usage (GPU box):  python tools/launch_anatomy.py [--width 8 --hw 640 --batch 256] > profiles/r06_launch_anatomy.txt"""
import argparse
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=8)
ap.add_argument("--hw", type=int, default=640)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--dump", default="", help="also write the raw workgroup records (numpy .npy) here, for offline analysis")
ap.add_argument("--lib", default=os.path.join(ROOT, "thingino-accel_amd", "lib", "diag", "lib_anatomy.so"))
args = ap.parse_args()
spec = importlib.util.spec_from_file_location("marsrt", os.path.join(ROOT, "thingino-accel_amd", "marsrt.py"))
M = importlib.util.module_from_spec(spec)
spec.loader.exec_module(M)
M.LIB_PATH = os.path.abspath(args.lib)
from conftest import lcg_frame  # noqa: E402
import marsfile  # noqa: E402

M.nna_init()
L = M.lib()
M.set_tuning("dual_stream_min_batch", 0)
d = M.synth_model(width_x16=args.width, input_hw=args.hw, seed=1)
hdr, tensors, _ = marsfile.parse(d)
nb = marsfile.tensor_nbytes(tensors[hdr["inputs"][0]])
m = M.Model(d, batch=args.batch)
iv = m.input_view(0)
for f in range(args.batch):
    iv[f] = lcg_frame(0x5EED0000 + f, nb)
m.upload()
for _ in range(3):
    m.run_device()
# per-launch HIP-event durations of the SHIPPED-equivalent step (same kernels; the stamps cost a few scalar instructions per workgroup)
m.set_profiling(1)
m.run_device()
ops = m.ops()
m.set_profiling(0)

REC = np.dtype([("key", "<u4"), ("wg", "<u4"), ("hwid", "<u4"), ("xcc", "<u4"), ("t", "<u8", 4)])
CAP = 4 << 20
L.mhip_malloc.restype = C.c_void_p
L.mhip_malloc.argtypes = [C.c_size_t]
L.mhip_free.argtypes = [C.c_void_p]
L.mhip_memset_async.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
L.mhip_d2h_async.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
buf = L.mhip_malloc(CAP * REC.itemsize)
assert buf
assert L.mhip_memset_async(buf, 0, CAP * REC.itemsize) == 0 and L.mhip_sync() == 0
for name in ("i8", "patch", "stem", "rows"):
    fn = getattr(L, "mhip_anatomy_set_" + name)
    fn.argtypes = [C.c_void_p, C.c_uint]
    assert fn(buf, CAP) == 0, name
m.run_device()
hdr0 = np.zeros(1, dtype=REC)
assert L.mhip_d2h_async(hdr0.ctypes.data, buf, REC.itemsize) == 0 and L.mhip_sync() == 0
n = int(hdr0["key"][0])
assert 0 < n < CAP - 1, n
recs = np.zeros(n + 1, dtype=REC)
assert L.mhip_d2h_async(recs.ctypes.data, buf, (n + 1) * REC.itemsize) == 0 and L.mhip_sync() == 0
recs = recs[1:]
if args.dump:
    np.save(args.dump, recs)
for name in ("i8", "patch", "stem", "rows"):
    getattr(L, "mhip_anatomy_set_" + name)(None, 0)
L.mhip_free(buf)

# ---- launches: records grouped by key; groups that overlap in time are one launch (a paired launch writes two tensors)
t = recs["t"].astype(np.int64)
groups = {}
for k in np.unique(recs["key"]):
    idx = np.nonzero(recs["key"] == k)[0]
    groups[int(k)] = idx
order = sorted(groups.values(), key=lambda ix: t[ix, 0].min())
launches = []
for ix in order:
    if launches and t[ix, 0].min() < t[launches[-1], 3].max() - (t[launches[-1], 3].max() - t[launches[-1], 0].min()) // 2:
        launches[-1] = np.concatenate([launches[-1], ix])
    else:
        launches.append(ix)
conv = [(i, o) for i, o in enumerate(ops) if o["kind"] == 0 and o["ms"] > 0]
TICK = 0.01  # microseconds per 100 MHz tick
print("# launch anatomy: yolov5%s int8 twin %dx%d, batch %d, one stream; %d conv launches stamped (%d workgroup records), %d conv launches by HIP events"
      % ("s" if args.width == 8 else "n" if args.width == 4 else "?", args.hw, args.hw, args.batch, len(launches), n, len(conv)))
print("# all times in microseconds; the guide prices a dependent kernel boundary at 1.45-1.9 us (MI355X_MICROARCH.md)")
print("%3s %5s %6s %5s %8s %8s %6s %7s %7s %5s %5s %5s %5s %8s %8s" % ("#", "layer", "wgs", "conc", "T", "event", "gap", "head", "tail", "util", "pro%", "K%", "epi%", "wg_mean", "pro_mean"))
tot = dict(T=0.0, gap=0.0, head=0.0, tail=0.0, idle=0.0, pro=0.0, epi=0.0, ev=0.0, pro_mach=0.0, epi_mach=0.0)
prev_end = None
for li, ix in enumerate(launches):
    s0, s1, s2, s3 = (t[ix, q] for q in range(4))
    first, last = s0.min(), s3.max()
    T = (last - first) * TICK
    # concurrency sweep
    ev = np.concatenate([np.stack([s0, np.ones_like(s0)], 1), np.stack([s3, -np.ones_like(s3)], 1)])
    ev = ev[np.lexsort((-ev[:, 1], ev[:, 0]))]
    conc = np.cumsum(ev[:, 1])
    peak = int(conc.max())
    thr = 0.9 * peak
    above = np.nonzero(conc >= thr)[0]
    head = (ev[above[0], 0] - first) * TICK
    tail = (last - ev[above[-1] + 1, 0]) * TICK if above[-1] + 1 < len(ev) else 0.0
    life = (s3 - s0).sum() * TICK
    util = life / (T * peak) if T > 0 else 0.0
    pro = (s1 - s0).sum() * TICK
    kk = (s2 - s1).sum() * TICK
    epi = (s3 - s2).sum() * TICK
    gap = (first - prev_end) * TICK if prev_end is not None else 0.0
    prev_end = last
    evms = conv[li][1]["ms"] * 1e3 if li < len(conv) and len(conv) == len(launches) else float("nan")
    layer = conv[li][1]["layer"] if li < len(conv) and len(conv) == len(launches) else -1
    print("%3d %5d %6d %5d %8.1f %8.1f %6.2f %7.1f %7.1f %5.2f %5.1f %5.1f %5.1f %8.1f %8.1f" % (
        li, layer, len(ix), peak, T, evms, gap, head, tail, util, 100 * pro / life, 100 * kk / life, 100 * epi / life, life / len(ix), pro / len(ix)))
    tot["T"] += T; tot["gap"] += gap; tot["head"] += head; tot["tail"] += tail; tot["idle"] += T * (1 - util)
    tot["pro"] += pro; tot["epi"] += epi; tot["ev"] += 0 if evms != evms else evms
    tot["pro_mach"] += T * util * pro / life; tot["epi_mach"] += T * util * epi / life
nl = len(launches)
print("# sums over the %d launches: T %.0f us (HIP events %.0f), gaps %.1f (%.2f per boundary), head %.0f (%.1f per launch), tail %.0f (%.1f per launch), "
      "T x (1 - util) %.0f (%.1f per launch)" % (nl, tot["T"], tot["ev"], tot["gap"], tot["gap"] / max(nl - 1, 1), tot["head"], tot["head"] / nl,
                                                  tot["tail"], tot["tail"] / nl, tot["idle"], tot["idle"] / nl))
print("# machine time inside workgroups (T x util x share): prologues %.0f us (%.1f per launch), after-K-stream epilogues %.0f us (%.1f per launch)"
      % (tot["pro_mach"], tot["pro_mach"] / nl, tot["epi_mach"], tot["epi_mach"] / nl))
m.close()

#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of bench.py into per-kernel HBM traffic.

usage: pmc_summary.py --fetch DIR --write DIR --executions N [--out FILE.json]

DIR = output directory of `rocprofv3 --pmc FETCH_SIZE -d DIR -- python3 bench.py ...` (resp. WRITE_SIZE);
N   = graph executions of that bench command (warmup + steps + the one parity run).
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): counter unit is KB;
on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE as is.
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict


def read(dirname, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
                tot[k] += float(r["Counter_Value"])
                n[k] += 1
    return tot, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--executions", type=int, required=True)
    ap.add_argument("--out", default="")
    ap.add_argument("--note", default="")
    ap.add_argument("--width", type=int, default=8)
    ap.add_argument("--hw", type=int, default=640)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--family", default="conv_i8", help="kernel-name prefix of the dominant family (conv_i8 | conv_f32)")
    ap.add_argument("--dtype", default="int8")
    a = ap.parse_args()
    rd, nrd = read(a.fetch, "FETCH_SIZE")
    wr, nwr = read(a.write, "WRITE_SIZE")
    rows = []
    for k in sorted(set(rd) | set(wr), key=lambda k: -(2 * rd.get(k, 0) + wr.get(k, 0))):
        launches = max(nrd.get(k, 0), nwr.get(k, 0))
        rows.append({"kernel": k, "launches": launches, "launches_per_step": launches / a.executions,
                     "read_bytes_per_step": 2.0 * rd.get(k, 0) * 1024 / a.executions,
                     "write_bytes_per_step": wr.get(k, 0) * 1024 / a.executions})
    conv = [r for r in rows if r["kernel"].startswith(a.family)]
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = {"note": a.note, "kernel_source_sha16": bench.kernel_source_sha16(),
           "config": {"width": a.width, "hw": a.hw, "batch": a.batch}, "dtype": a.dtype, "family": a.family, "executions": a.executions, "fetch_size_doubled": True, "kernels": rows,
           "conv_i8": {"launches_per_step": sum(r["launches_per_step"] for r in conv),
                       "read_bytes_per_step": sum(r["read_bytes_per_step"] for r in conv),
                       "write_bytes_per_step": sum(r["write_bytes_per_step"] for r in conv)},
           "all": {"read_bytes_per_step": sum(r["read_bytes_per_step"] for r in rows if "fillBuffer" not in r["kernel"]),
                   "write_bytes_per_step": sum(r["write_bytes_per_step"] for r in rows if "fillBuffer" not in r["kernel"])}}
    s = json.dumps(out, indent=1)
    if a.out:
        open(a.out, "w").write(s + "\n")
    for r in rows:
        print("%-44s %6.1f launches/step  read %8.1f MB  write %8.1f MB" % (r["kernel"][:44], r["launches_per_step"],
              r["read_bytes_per_step"] / 1e6, r["write_bytes_per_step"] / 1e6))
    print("%s per step: read %.3f GB write %.3f GB" % (a.family, out["conv_i8"]["read_bytes_per_step"] / 1e9, out["conv_i8"]["write_bytes_per_step"] / 1e9))


if __name__ == "__main__":
    main()

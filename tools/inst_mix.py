#!/usr/bin/env python3
"""Instruction mix per kernel symbol from one rocprofv3 --pmc pass:  inst_mix.py PMC_DIR EXECUTIONS > table.md

Counters: SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES.
SQ_INSTS_VALU includes the MFMA instructions (they issue on the vector port), so "VALU other than MFMA per MFMA" is
(VALU - MFMA) / MFMA.  Prints a markdown table, one row per convolution kernel symbol, heaviest first.

    inst_mix.py PMC_DIR EXECUTIONS --json OUT.json [--lds LDS_PMC_DIR] [--config width,hw,batch]
additionally writes the conv_i8 family's per-step totals (MFMA / VALU / SALU / LDS instructions and, from a second pass with
SQ_LDS_IDX_ACTIVE, the cycles the LDS arrays were busy) with the hash of the kernel sources: profiles/inst_mix.json, which
bench.py turns into `roofline.pipes`."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

pmc_dir, execs = sys.argv[1], int(sys.argv[2])
opts = dict(zip(sys.argv[3::2], sys.argv[4::2]))
tot = defaultdict(lambda: defaultdict(float))
for f in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(((k, c) for k, c in tot.items() if c.get("SQ_INSTS_MFMA", 0) > 0), key=lambda kc: -kc[1]["SQ_INSTS_MFMA"])
print("| kernel | MFMA / step | (VALU - MFMA) / MFMA | SALU / MFMA | LDS / MFMA | VMEM rd / MFMA | VMEM wr / MFMA | waves / step |")
print("|---|---|---|---|---|---|---|---|")
for k, c in rows:
    m = c["SQ_INSTS_MFMA"]
    print("| `%s` | %.3g | %.2f | %.2f | %.2f | %.3f | %.3f | %.3g |" % (
        k, m / execs, (c.get("SQ_INSTS_VALU", 0) - m) / m, c.get("SQ_INSTS_SALU", 0) / m, c.get("SQ_INSTS_LDS", 0) / m,
        c.get("SQ_INSTS_VMEM_RD", 0) / m, c.get("SQ_INSTS_VMEM_WR", 0) / m, c.get("SQ_WAVES", 0) / execs))

if "--json" in opts:
    import hashlib
    import json
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    h = hashlib.sha256()
    d = os.path.join(root, "thingino-accel_amd", "csrc", "hip")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    fam = defaultdict(float)
    for k, c in tot.items():
        if k.startswith("conv_i8"):
            for name, v in c.items():
                fam[name] += v
    lds = 0.0
    if "--lds" in opts:
        for f in glob.glob(os.path.join(opts["--lds"], "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE" and re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip().startswith("conv_i8"):
                        lds += float(r["Counter_Value"])
    w, hw, b = (int(v) for v in opts.get("--config", "8,640,256").split(","))
    out = {"note": "per step (one batch), every conv_i8_* launch summed; tools/inst_mix.py", "kernel_source_sha16": h.hexdigest()[:16],
           "config": {"width": w, "hw": hw, "batch": b}, "executions": execs,
           "conv_i8_per_step": {"mfma": fam["SQ_INSTS_MFMA"] / execs, "valu": fam["SQ_INSTS_VALU"] / execs, "salu": fam["SQ_INSTS_SALU"] / execs,
                                "lds": fam["SQ_INSTS_LDS"] / execs, "vmem_rd": fam["SQ_INSTS_VMEM_RD"] / execs, "vmem_wr": fam["SQ_INSTS_VMEM_WR"] / execs,
                                "lds_idx_active_cycles": lds / execs if lds else None}}
    with open(opts["--json"], "w") as fh:
        json.dump(out, fh, indent=1)

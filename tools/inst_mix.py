#!/usr/bin/env python3
"""Instruction mix per kernel symbol from one rocprofv3 --pmc pass:  inst_mix.py PMC_DIR EXECUTIONS > table.md

Counters: SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES.
SQ_INSTS_VALU includes the MFMA instructions (they issue on the vector port), so "VALU other than MFMA per MFMA" is
(VALU - MFMA) / MFMA.  Prints a markdown table, one row per convolution kernel symbol, heaviest first."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

pmc_dir, execs = sys.argv[1], int(sys.argv[2])
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

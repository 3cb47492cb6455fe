#!/usr/bin/env python3
"""Matrix-pipe occupancy per kernel symbol from one rocprofv3 --pmc pass:  mfma_busy.py PMC_DIR TRACE_DIR EXECUTIONS

SQ_VALU_MFMA_BUSY_CYCLES counts cycles (16 per v_mfma_i32_16x16x64_i8, summed over the SIMDs); the denominator is the
time the kernel ran x the SIMDs of the chip: per kernel, busy = MFMA_BUSY / (1024 SIMDs x kernel cycles), with kernel
cycles = SQ_BUSY_CYCLES / 32 (the counter is summed over the 32 shader engines).  Wave-level counters (quad-cycles)
give the split of a wave's life: issuing / parked at s_waitcnt or s_barrier / issue-stalled.  Prints JSON."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

pmc_dir, trace_dir, execs = sys.argv[1], sys.argv[2], int(sys.argv[3])
tot = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
for f in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_BUSY_CYCLES":
                n[k] += 1
rows = []
for k, c in tot.items():
    if not k.startswith("conv_"):
        continue
    cyc = c.get("SQ_BUSY_CYCLES", 0.0) / 32.0
    wave = c.get("SQ_WAVE_CYCLES", 0.0)
    rows.append({"kernel": k, "launches_per_step": n[k] / execs,
                 "kernel_cycles_per_step": cyc / execs,
                 "mfma_busy": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc) if cyc else None,
                 "mfma_instructions_per_step": c.get("SQ_INSTS_MFMA", 0.0) / execs,
                 "wave_time_issuing": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave if wave else None,
                 "wave_time_parked_waitcnt_barrier": c.get("SQ_WAIT_ANY", 0.0) / wave if wave else None,
                 "wave_time_issue_stalled": c.get("SQ_WAIT_INST_ANY", 0.0) / wave if wave else None})
rows.sort(key=lambda r: -r["kernel_cycles_per_step"])
allc = sum(r["kernel_cycles_per_step"] for r in rows)
allb = sum((r["mfma_busy"] or 0) * r["kernel_cycles_per_step"] for r in rows)
print(json.dumps({"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ... of bench.py --timed-only; busy = MFMA busy cycles / "
                          "(1024 SIMDs x kernel cycles); kernel cycles at the clock the profiled run held",
                  "executions": execs, "conv_family_mfma_busy": allb / allc if allc else None, "kernels": rows}, indent=1))

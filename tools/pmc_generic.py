#!/usr/bin/env python3
"""Per-kernel sums of whatever counters a rocprofv3 --pmc pass collected: pmc_generic.py DIR"""
import csv, glob, os, re, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k in sorted(tot):
    print(k[:46].ljust(46), " ".join("%s=%.4g(/%d)" % (c, v, n[(k, c)]) for c, v in sorted(tot[k].items())))

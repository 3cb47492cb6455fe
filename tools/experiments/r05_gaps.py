"""Idle time between consecutive kernels of one stream in a rocprofv3 kernel trace (csv): python3 tools/experiments/r05_gaps.py DIR
Prints, for the last timed step of a `bench.py --timed-only --tune dual_stream_min_batch=0` run, the number of kernels, the sum of their
durations, the sum of the gaps between them and the largest gaps."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# steps: split at gaps > 200 us is wrong (steps are back to back); use the repeating stem kernel as the step marker
marks = [i for i, r in enumerate(rows) if "stem" in r[2] or "conv_i8_rgb" in r[2]]
if len(marks) < 2:
    sys.exit("no step marker")
a, b = marks[-2], marks[-1]
step = rows[a:b]
dur = sum(e - s for s, e, _ in step)
gaps = [(step[i + 1][0] - step[i][1], step[i][2][:50], step[i + 1][2][:50]) for i in range(len(step) - 1)]
print("kernels %d  durations %.3f ms  gaps %.3f ms (negative = overlap %.3f ms)  wall %.3f ms" % (
    len(step), dur / 1e6, sum(g for g, _, _ in gaps if g > 0) / 1e6, sum(g for g, _, _ in gaps if g < 0) / 1e6, (step[-1][1] - step[0][0]) / 1e6))
for g, x, y in sorted(gaps, reverse=True)[:8]:
    print("  %7.1f us  %s -> %s" % (g / 1e3, x, y))
import statistics
print("  median gap %.1f us" % (statistics.median(g for g, _, _ in gaps) / 1e3))

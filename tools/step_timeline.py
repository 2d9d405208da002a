#!/usr/bin/env python3
"""Per-step timeline of a GSO run from a rocprofv3 --kernel-trace directory: the launches of one steady-state step in order, each kernel's
mean duration and the mean gap in front of it (end of the previous kernel -> its start), over the run's last steps.
Usage: python tools/step_timeline.py <trace dir>"""
import csv, glob, sys
from collections import defaultdict
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not path:
    sys.exit("no kernel_trace.csv under " + sys.argv[1])
rows = list(csv.DictReader(open(path[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void ", "").replace("ld::(anonymous namespace)::", "")
    return n.split("(")[0][:44]
names = [short(r["Kernel_Name"]) for r in rows]
# a step ends with the K2 kernel
k2 = [i for i, n in enumerate(names) if n.startswith("gso_movement")]
if len(k2) < 12:
    sys.exit("fewer than 12 steps in the trace")
steps = [(k2[i - 1] + 1, k2[i] + 1) for i in range(len(k2) // 2, len(k2))]     # the second half of the run
seq = [names[i] for i in range(*steps[0])]
dur, gap, n = defaultdict(float), defaultdict(float), 0
for a, b in steps:
    if [names[i] for i in range(a, b)] != seq:
        continue
    n += 1
    for k, i in enumerate(range(a, b)):
        s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
        dur[k] += e - s
        gap[k] += s - int(rows[i - 1]["End_Timestamp"])
total = 0.0
print("# %d steps of the same %d launches; us: gap in front, duration" % (n, len(seq)))
for k, name in enumerate(seq):
    print("%-46s gap %7.2f   kernel %8.2f" % (name, gap[k] / n / 1e3, dur[k] / n / 1e3))
    total += (gap[k] + dur[k]) / n / 1e3
print("step (sum of gaps and kernels): %.1f us = %.0f steps/s" % (total, 1e6 / total))

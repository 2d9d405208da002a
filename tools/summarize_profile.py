#!/usr/bin/env python3
"""Fold the rocprofv3 CSVs written by tools/profile_round.sh into a short summary."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
KERNEL = "dfire_tiled_pairs<false"


def counters(sub):
    agg = collections.defaultdict(list)
    for p in glob.glob(os.path.join(out, sub, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(p)):
            if KERNEL in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


print("== kernel trace (rocprofv3 --kernel-trace --stats), default bench command")
for p in glob.glob(os.path.join(out, "trace", "*", "*kernel_stats.csv")):
    for i, r in enumerate(csv.DictReader(open(p))):
        if i < 6:
            print("  %-90s calls %5s avg %12.1f ns  %6s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
try:
    b = json.load(open(os.path.join(out, "bench_traced.json")))
    print("  bench (traced run): %.0f evals/s, HIP-event kernel_ms %.4f" % (b["value"], b["roofline"]["kernel_ms"]))
except Exception as e:  # noqa: BLE001
    print("  (no traced bench json: %s)" % e)
print("== PMC, mean per launch of %s (batch of the default bench command)" % KERNEL)
allc = {}
for sub in ("fetch", "write", "tcc", "sq1", "sq2"):
    c = counters(sub)
    allc.update(c)
    for k, v in sorted(c.items()):
        print("  %-28s %.6g" % (k, v))
if "FETCH_SIZE" in allc and "WRITE_SIZE" in allc:
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 64 B per 128 B
    # request for wide streaming reads (MI355X_MICROARCH.md, HBM): the upper estimate doubles it.
    lo = (allc["FETCH_SIZE"] + allc["WRITE_SIZE"]) * 1024
    hi = (2 * allc["FETCH_SIZE"] + allc["WRITE_SIZE"]) * 1024
    print("  HBM bytes per launch: %.4g (as reported) .. %.4g (FETCH_SIZE x2 gfx950 correction)" % (lo, hi))
    print("  traffic_json_value %.0f" % hi)
if "TCC_HIT_sum" in allc:
    print("  L2 hit rate %.4f" % (allc["TCC_HIT_sum"] / (allc["TCC_HIT_sum"] + allc["TCC_MISS_sum"])))
try:
    b = json.load(open(os.path.join(out, "bench.json")))
    print("== bench.py (un-profiled run)")
    print(json.dumps(b))
except Exception as e:  # noqa: BLE001
    print("(no bench json: %s)" % e)

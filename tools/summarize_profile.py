#!/usr/bin/env python3
"""Fold the rocprofv3 CSVs written by tools/profile_round.sh into a short summary (one workload)."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
KERNELS = ("dfire_bm_pairs<false", "dfire_bm_cull<false", "dfire_bm_gather", "dfire_bm_pose", "dfire_bm_plan", "dfire_bm_census", "dfire_bm_order",
           "dfire_packed_pairs<false", "dfire_tiled_pairs<false", "pose_energy_pairs<1", "pose_energy_pairs<0", "gso_movement_phase",
           "pose_energy_finish", "dfire_packed_prepare")


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


def counters():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob(os.path.join(out, "pmc*", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


print("== kernel trace (rocprofv3 --kernel-trace --stats) of: bench.py --workload %s --steps 10 --warmup 3 --no-stats (no counting launch: every kernel of the sequence runs 13 times)" % os.path.basename(out.rstrip("/")))
for p in glob.glob(os.path.join(out, "trace", "*", "*kernel_stats.csv")):
    for i, r in enumerate(csv.DictReader(open(p))):
        if i < 11:
            print("  %-84s calls %5s avg %12.1f ns  %6s %%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
    # the block-major K1 is seven kernels; bench.py's HIP events bracket all of them (`roofline.kernel_ms`)
    rows = list(csv.DictReader(open(p)))
    bm = [r for r in rows if "dfire_bm_" in r["Name"] and "<true>" not in r["Name"]]
    if bm:
        calls = max(int(r["Calls"]) for r in bm if "pairs" in r["Name"])
        print("  block-major K1 (pose + cull + plan + census + order + pairs + gather), sum of the average durations: %.1f ns over %d launches of the sequence"
              % (sum(float(r["TotalDurationNs"]) for r in bm) / calls, calls))
try:
    b = json.load(open(os.path.join(out, "bench_traced.json")))
    print("  bench (traced run): %.0f evals/s, HIP-event kernel_ms %.4f" % (b["value"], b["roofline"]["kernel_ms"]))
except Exception as e:  # noqa: BLE001
    print("  (no traced bench json: %s)" % e)
allc = counters()
for kern, c in allc.items():
    print("== PMC, mean per launch of %s" % kern)
    for k, v in sorted(c.items()):
        print("  %-30s %.6g" % (k, v))
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 64 B per 128 B
        # request for wide streaming reads (MI355X_MICROARCH.md, HBM): the upper estimate doubles it.
        lo = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        hi = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        # profiles/r04_fetch_size_calibration.txt: coalesced reads are 128-byte requests counted as 64 (x 2); the random 8-byte
        # reads of dfire_bm_gather are one request each, counted as 64 bytes: taken as reported
        best = lo if kern.startswith("dfire_bm_gather") else hi
        print("  HBM bytes per launch: %.4g (as reported) .. %.4g (FETCH_SIZE x2 gfx950 correction); calibrated for this kernel's access pattern: %.4g" % (lo, hi, best))
        print("  traffic_json hbm_bytes_per_launch %.0f" % best)
    if "SQ_INSTS_VALU" in c:
        print("  traffic_json valu_insts_per_launch %.0f" % c["SQ_INSTS_VALU"])
    if "SQ_ACTIVE_INST_SCA" in c and "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c:
        print("  scalar unit busy %.3f, vector unit busy %.3f of the wave cycles; waves waiting %.3f"
              % (c["SQ_ACTIVE_INST_SCA"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]))
    if "TCC_HIT_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        print("  L2 hit rate %.4f" % (c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
try:
    b = json.load(open(os.path.join(out, "bench.json")))
    print("== bench.py (un-profiled run)")
    print(json.dumps(b))
except Exception as e:  # noqa: BLE001
    print("(no bench json: %s)" % e)

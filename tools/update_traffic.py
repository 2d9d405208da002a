#!/usr/bin/env python3
"""Folds the summaries of a profiling round (tools/profile_round.sh -> gpurun_out/prof_<tag>/<workload>/) into
profiles/traffic.json and copies the evidence to profiles/<tag>_*.  Every entry is stamped with bench.kernel_source_hash():
bench.py reports the counters of an entry only while the kernel sources are the ones it was taken from.
Usage: python tools/update_traffic.py <tag>"""
import csv, glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
tj = os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(tj))
h = bench.kernel_source_hash()
for wdir in sorted(glob.glob(os.path.join(src, "*"))):
    w = os.path.basename(wdir)
    name = w.replace("-", "_")
    try:
        b = json.load(open(os.path.join(wdir, "bench.json")))
    except Exception as e:  # noqa: BLE001
        print("skip", w, e)
        continue
    shutil.copy(os.path.join(wdir, "summary.txt"), os.path.join(ROOT, "profiles", "%s_%s_summary.txt" % (tag, name)))
    shutil.copy(os.path.join(wdir, "bench.json"), os.path.join(ROOT, "profiles", "%s_%s_bench.json" % (tag, name)))
    for p in glob.glob(os.path.join(wdir, "trace", "*", "*kernel_stats.csv")):
        shutil.copy(p, os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, name)))
    for i, p in enumerate(sorted(glob.glob(os.path.join(wdir, "pmc*", "*", "*counter_collection.csv")))):
        rows = [r for r in csv.DictReader(open(p)) if "ld::" in r["Kernel_Name"]]
        if rows:
            with open(os.path.join(ROOT, "profiles", "%s_%s_pmc%d.csv" % (tag, name, i + 1)), "w", newline="") as f:
                wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                wr.writeheader()
                wr.writerows(rows)
    kernel = b["roofline"]["kernel"]
    units = b["config"].get("poses_per_step_per_gpu") or b["config"].get("swarms_this_rank")
    text = open(os.path.join(wdir, "summary.txt")).read()
    # the section of the dominant kernel
    # The block-major K1 is a sequence of kernels that bench.py times as one (`roofline.kernel_ms`): its counters are the sums
    # over the sequence.  Every other kernel: its own section.
    secs = re.findall(r"== PMC, mean per launch of ([^\n]*)\n(.*?)(?=\n==|\Z)", text, re.S)
    if kernel.startswith("dfire_bm"):
        mine = [body for name, body in secs if name.startswith("dfire_bm_")]
    else:
        want = kernel + "<false" if kernel.startswith("dfire") else kernel
        mine = [body for name, body in secs if name.startswith(want)][:1]
    def val(key):
        found = [re.search(r"^\s*%s\s+([0-9.e+]+)" % re.escape(key), body, re.M) for body in mine]
        found = [float(mm.group(1)) for mm in found if mm]
        return sum(found) if found else None
    entry = {"source_hash": h, "source": "profiles/%s_%s_summary.txt" % (tag, name)}
    for key, field in (("traffic_json hbm_bytes_per_launch", "hbm_bytes_per_launch"), ("SQ_INSTS_VALU", "valu_insts_per_launch"),
                       ("SQ_INSTS_SALU", "salu_insts_per_launch"), ("SQ_INSTS_VMEM_RD", "vmem_insts_per_launch"),
                       ("TCP_TOTAL_CACHE_ACCESSES_sum", "tcp_cache_accesses_per_launch"), ("TCP_TCC_READ_REQ_sum", "l2_read_requests_per_launch"),
                       ("FETCH_SIZE", "fetch_size_kib"), ("WRITE_SIZE", "write_size_kib"), ("SQ_INSTS_LDS", "lds_insts_per_launch"),
                       ("SQ_LDS_IDX_ACTIVE", "lds_busy_cycles_per_launch"), ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict_cycles_per_launch")):
        v = val(key)
        if v is not None:
            entry[field] = v
    entry["valu_source"] = "SQ_INSTS_VALU (wave-level vector instructions) per launch of %s, same file" % (
        "the block-major sequence (pose, cull, plan, census, order, pairs, gather)" if kernel.startswith("dfire_bm") else kernel)
    entry["binding"] = "valu-issue"
    if entry.get("lds_busy_cycles_per_launch"):
        entry["lds_bank_conflict_share"] = entry.get("lds_bank_conflict_cycles_per_launch", 0.0) / entry["lds_busy_cycles_per_launch"]
    if kernel.startswith("dfire_bm"):
        entry["kernels"] = {name.split("<")[0]: float(re.search(r"^\s*SQ_INSTS_VALU\s+([0-9.e+]+)", body, re.M).group(1))
                            for name, body in secs if name.startswith("dfire_bm_") and re.search(r"^\s*SQ_INSTS_VALU", body, re.M)}
    traffic["%s:%d:%s" % (w, units, kernel)] = entry
    print("%s:%d:%s" % (w, units, kernel), {k: v for k, v in entry.items() if k not in ("source", "valu_source")})
json.dump(traffic, open(tj, "w"), indent=1)

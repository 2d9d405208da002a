#!/bin/bash
# round 6: kernel trace and instruction counters of the culling kernel, the build before (prev) and after (new) the eight-poses-at-once boxing
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r6_17; mkdir -p $O
bash tools/trace_variants.sh > $O/trace.txt 2>&1; cat $O/trace.txt
for v in prev new; do
  LIGHTDOCK_HIP_VARIANT=$v timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/pmc_$v -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-stats > /dev/null 2>&1
  python3 - $O/pmc_$v $v <<'PY'
import csv,glob,collections,sys
acc=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "dfire_bm_cull" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(sys.argv[2], k, "%.4g" % (sum(v)/len(v)))
PY
done 2>&1 | tee $O/pmc.txt

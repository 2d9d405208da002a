# round 5: per-kernel times of a GSO step with 1 % of the swarms alive (1ppe, 1024 swarms)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r05tail; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 tools/gso_tail.py 1024 60 0.01 > $out/log.txt 2>&1
f=$(find $out/t -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("  %-60s calls %4s avg %10.1f us  total %8.1f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
tail -4 $out/log.txt

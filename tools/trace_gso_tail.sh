# GPU box: rocprofv3 --kernel-trace of late-stage GSO steps (tools/gso_tail.py): which kernels a step with nothing / little to
# evaluate consists of.  usage: bash tools/trace_gso_tail.sh [live share ...]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for l in ${@:-0 0.01}; do
  rm -rf /tmp/tr_tail
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_tail -- python3 tools/gso_tail.py 1024 60 $l > /tmp/tail_$l.txt 2>&1
  echo "== live share $l: $(tail -1 /tmp/tail_$l.txt)"
  python3 - <<'P'
import csv, glob
for f in glob.glob("/tmp/tr_tail/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:16]:
        print("   %-64s calls %5s avg %8.1f us  %5.1f %%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
P
done

# round 5: kernel trace of the default bench command for every prebuilt variant (or the installed library when there is none)
# usage (GPU box): bash tools/r5_trace.sh <tag>
tag=${1:-r05t}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
shopt -s nullglob
L=lightdock-rust_amd/lib
cp $L/liblightdock_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so $L/liblightdock_hip.so' EXIT INT TERM   # an interrupted run must not leave a variant installed; tools/ab6.sh never installs one
vs=$(ls $L/variants/*.so 2>/dev/null); [ -z "$vs" ] && vs=/tmp/keep.so
for v in $vs; do
  n=$(basename $v .so); cp $v $L/liblightdock_hip.so
  out=gpurun_out/$tag/$n; mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-stats > $out/bench_traced.json 2> $out/trace.log
  f=$(find $out/trace -name '*kernel_stats.csv' | head -1)
  echo "== $n"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("  %-60s calls %4s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
cp /tmp/keep.so $L/liblightdock_hip.so

# round 5: A/B of the prebuilt variants with a kernel trace of the default command each (culling / pair kernel times), then the bench lines
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
shopt -s nullglob
L=lightdock-rust_amd/lib
cp $L/liblightdock_hip.so /tmp/keep3.so
trap 'cp /tmp/keep3.so $L/liblightdock_hip.so' EXIT INT TERM   # an interrupted run must not leave a variant installed; tools/ab6.sh never installs one
for v in $L/variants/*.so; do
  n=$(basename $v .so); cp $v $L/liblightdock_hip.so
  out=gpurun_out/r05abt/$n; mkdir -p $out
  for w in ${@:-1k4c}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$w -- python3 bench.py --workload $w --steps 10 --warmup 3 --cpu-seconds 0 --no-stats > $out/bench_traced_$w.json 2> $out/trace_$w.log
  f=$(find $out/trace_$w -name '*kernel_stats.csv' | head -1)
  echo "== $n $w"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:3]:
    print("  %-60s calls %4s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
cp /tmp/keep3.so $L/liblightdock_hip.so

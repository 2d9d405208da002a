# GPU box: rocprofv3 kernel trace of one bench workload for every prebuilt library variant (top kernels by time)
# usage: bash tools/trace_variants.sh <workload>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=lightdock-rust_amd/lib
cp $L/liblightdock_hip.so /tmp/keep.so
for v in $L/variants/*.so; do
  cp $v $L/liblightdock_hip.so
  n=$(basename $v .so); out=gpurun_out/trace_$n; rm -rf $out; mkdir -p $out
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --workload $1 --steps 10 --warmup 3 --cpu-seconds 0 > $out/bench.json 2> $out/log
  echo "== $n"
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("  %-70s calls %4s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  fi
done
cp /tmp/keep.so $L/liblightdock_hip.so

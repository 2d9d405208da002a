# GPU box: rocprofv3 --kernel-trace averages of the block-major kernels for every prebuilt library variant
# usage: bash tools/trace_variants.sh [bench args]
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
shopt -s nullglob
L=lightdock-rust_amd/lib
cp $L/liblightdock_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so $L/liblightdock_hip.so' EXIT INT TERM   # an interrupted run must not leave a variant installed; tools/ab6.sh never installs one
for v in $L/variants/*.so; do
  cp $v $L/liblightdock_hip.so
  n=$(basename $v .so); rm -rf /tmp/tr_$n
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$n -- python3 bench.py --steps 10 --warmup 3 --cpu-seconds 0 "$@" > /dev/null 2>&1
  echo "== $n"
  python3 - /tmp/tr_$n <<'P'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "dfire_bm" in r["Name"] or "finish" in r["Name"]:
            print("   %-60s calls %3s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
P
done
cp /tmp/keep.so $L/liblightdock_hip.so

# A/B of prebuilt library variants on ONE box: lightdock-rust_amd/lib/variants/<name>.so
# Usage (on the GPU box): bash tools/ab.sh [bench args...]; interleaves the variants three times.
# Every run is under its own `timeout`, so a variant that hangs costs a minute, not the call.
cd "${GRAFT_REPO_ROOT:?}" || exit 1
shopt -s nullglob
L=lightdock-rust_amd/lib
cp $L/liblightdock_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so $L/liblightdock_hip.so' EXIT INT TERM   # an interrupted run must not leave a variant installed; tools/ab6.sh never installs one
for round in 1 2; do
  for v in $L/variants/*.so; do
    cp $v $L/liblightdock_hip.so
    r=$(timeout 90 python bench.py --cpu-seconds 0 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms']))" 2>&1 | tail -1)
    echo "$(basename $v) $r"
  done
done
cp /tmp/keep.so $L/liblightdock_hip.so

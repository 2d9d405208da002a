# GPU box: per-wave timers of dfire_bm_pairs (tools/bm_wave_times.py) for every prebuilt library variant
# usage: bash tools/ab_wave_times.sh [bench args]
cd "${GRAFT_REPO_ROOT:?}" || exit 1
shopt -s nullglob
L=lightdock-rust_amd/lib
cp $L/liblightdock_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so $L/liblightdock_hip.so' EXIT INT TERM   # an interrupted run must not leave a variant installed; tools/ab6.sh never installs one
for v in $L/variants/*.so; do
  cp $v $L/liblightdock_hip.so
  echo "== $(basename $v)"
  timeout 120 python tools/bm_wave_times.py "$@" 2>&1 | tail -8
done
cp /tmp/keep.so $L/liblightdock_hip.so

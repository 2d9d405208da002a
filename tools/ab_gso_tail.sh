# GPU box: tools/gso_tail.py (late-stage GSO steps) for every prebuilt library variant
# usage: bash tools/ab_gso_tail.sh [live share ...]
cd "${GRAFT_REPO_ROOT:?}" || exit 1
shopt -s nullglob
L=lightdock-rust_amd/lib
cp $L/liblightdock_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so $L/liblightdock_hip.so' EXIT INT TERM   # an interrupted run must not leave a variant installed; tools/ab6.sh never installs one
for v in $L/variants/*.so; do
  cp $v $L/liblightdock_hip.so
  for l in ${@:-0 0.01 0.1}; do
    echo "$(basename $v) live $l: $(timeout 120 python tools/gso_tail.py 1024 40 $l 2>&1 | tail -1)"
  done
done
cp /tmp/keep.so $L/liblightdock_hip.so

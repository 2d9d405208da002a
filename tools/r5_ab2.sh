# round 5: A/B of the prebuilt variants on one box: the 1k4c headline, gso-1ppe, and the 1 %-alive GSO step
cd "${GRAFT_REPO_ROOT:?}" || exit 1
shopt -s nullglob
L=lightdock-rust_amd/lib
bash tools/ab.sh
bash tools/ab.sh --workload gso-1ppe
cp $L/liblightdock_hip.so /tmp/keep2.so
trap 'cp /tmp/keep2.so $L/liblightdock_hip.so' EXIT INT TERM   # an interrupted run must not leave a variant installed; tools/ab6.sh never installs one
for round in 1 2; do
for v in $L/variants/*.so; do
  cp $v $L/liblightdock_hip.so
  echo "$(basename $v) tail: $(timeout 120 python3 tools/gso_tail.py 1024 60 0.01 2>&1 | tail -1)"
done
done
cp /tmp/keep2.so $L/liblightdock_hip.so

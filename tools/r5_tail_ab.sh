# round 5: the GSO step at 1 % / 3 % / 10 % of the swarms alive for every prebuilt variant
cd "${GRAFT_REPO_ROOT:?}" || exit 1
shopt -s nullglob
L=lightdock-rust_amd/lib
cp $L/liblightdock_hip.so /tmp/keep4.so
trap 'cp /tmp/keep4.so $L/liblightdock_hip.so' EXIT INT TERM   # an interrupted run must not leave a variant installed; tools/ab6.sh never installs one
for round in 1 2; do
for v in $L/variants/*.so; do
  cp $v $L/liblightdock_hip.so
  for live in 0.01 0.03 0.1; do
    echo "$(basename $v) live $live: $(timeout 120 python3 tools/gso_tail.py 1024 60 $live 2>&1 | tail -1)"
  done
done
done
cp /tmp/keep4.so $L/liblightdock_hip.so

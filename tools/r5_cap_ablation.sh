# round 5: why do batches get slower with longer jobs?  Wave timers of the diagnostic builds (wrong sums) at two part caps.
cd "${GRAFT_REPO_ROOT:?}" || exit 1
shopt -s nullglob
L=lightdock-rust_amd/lib
cp $L/liblightdock_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so $L/liblightdock_hip.so' EXIT INT TERM   # an interrupted run must not leave a variant installed; tools/ab6.sh never installs one
for v in $L/variants/*.so; do
  cp $v $L/liblightdock_hip.so
  for cap in 1024 1792; do
    echo "== $(basename $v) cap $cap: $(LIGHTDOCK_BM_PART_CAP=$cap timeout 120 python3 tools/bm_wave_times.py 2>&1 | grep -E 'time in batches|kernel span' | tr '\n' ' ' | cut -c1-330)"
  done
done
cp /tmp/keep.so $L/liblightdock_hip.so

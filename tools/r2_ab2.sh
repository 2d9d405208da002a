# GPU box: parity of one library variant against the oracle, then the A/B of all variants (tools/ab.sh)
# usage: bash tools/r2_ab2.sh <variant-to-check> [bench args...]
cd "${GRAFT_REPO_ROOT:?}" || exit 1
shopt -s nullglob
L=lightdock-rust_amd/lib
v=$1; shift
cp $L/liblightdock_hip.so /tmp/keep0.so
trap 'cp /tmp/keep0.so $L/liblightdock_hip.so' EXIT INT TERM   # an interrupted run must not leave a variant installed; tools/ab6.sh never installs one
cp $L/variants/$v.so $L/liblightdock_hip.so
timeout 300 python tools/debug_packed.py 1ppe 1k4c 2uuy 2>&1 | grep -v "tiled\|allpairs\|amdgpu.ids"
cp /tmp/keep0.so $L/liblightdock_hip.so
bash tools/ab.sh "$@"

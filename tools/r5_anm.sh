# round 5, VERDICT r04 item 4: the timing experiment "block-major for DFIRE + ANM".  2uuy (1615 x 415 atoms, 10 + 10 modes), 16 384 poses:
#   (a) the pose-major kernel that runs today (dfire_packed_pairs),
#   (b) the block-major kernels on the same poses AS IF the molecules were rigid (LIGHTDOCK_BM_DIAG_IGNORE_ANM=1: wrong sums) -- the floor,
#   (c) the same with the per-lane cost of flexing both molecules inside every batch (library built with -DLD_BM_DIAG_ANM_COST: 276 packed
#       multiply-adds, operands from vector registers, 80 bytes more per item; wrong sums).
# usage (GPU box): bash tools/r5_anm.sh   (needs lightdock-rust_amd/lib/variants/s_anm_cost.so)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
shopt -s nullglob
L=lightdock-rust_amd/lib
cp $L/liblightdock_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so $L/liblightdock_hip.so' EXIT INT TERM   # an interrupted run must not leave a variant installed; tools/ab6.sh never installs one
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s: %.2f M evals/s, K1 %.3f ms, kernel %s' % ('$1', d['value'] / 1e6, d['roofline']['kernel_ms'], d['roofline']['kernel']))"; }
for round in 1 2; do
  timeout 120 python bench.py --workload 2uuy --cpu-seconds 0 --no-stats 2>/dev/null | line "(a) pose-major, ANM"
  LIGHTDOCK_BM_DIAG_IGNORE_ANM=1 timeout 120 python bench.py --workload 2uuy --cpu-seconds 0 --no-stats 2>/dev/null | line "(b) block-major, rigid (floor)"
  cp $L/variants/s_anm_cost.so $L/liblightdock_hip.so
  LIGHTDOCK_BM_DIAG_IGNORE_ANM=1 timeout 120 python bench.py --workload 2uuy --cpu-seconds 0 --no-stats 2>/dev/null | line "(c) block-major + per-lane flexing cost"
  if [ -f $L/variants/t_anm_lds.so ]; then   # (d): (c) with the 480 mode components of a batch delivered from LDS (120 broadcast reads of 16 bytes)
    cp $L/variants/t_anm_lds.so $L/liblightdock_hip.so
    LIGHTDOCK_BM_DIAG_IGNORE_ANM=1 timeout 120 python bench.py --workload 2uuy --cpu-seconds 0 --no-stats 2>/dev/null | line "(d) ... with the modes read from LDS"
  fi
  cp /tmp/keep.so $L/liblightdock_hip.so
done

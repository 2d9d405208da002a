# round 5: does a pass whose affine-map table fits the L2 pay for the extra passes?  LIGHTDOCK_BM_CHUNK sweep on the 1ppe GSO loop and the 1ppe batch
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s: %.2f M evals/s, %.3f ms per step' % ('$1', d['value'] / 1e6, d['ms_per_step']))"; }
for round in 1 2; do
for c in 262144 131072 65536 32768 16384; do
  LIGHTDOCK_BM_CHUNK=$c timeout 300 python bench.py --workload gso-1ppe --steps 30 --warmup 6 --cpu-seconds 0 2>/dev/null | line "gso-1ppe chunk $c"
done
for c in 65536 32768 16384; do
  LIGHTDOCK_BM_CHUNK=$c timeout 300 python bench.py --workload 1ppe --cpu-seconds 0 2>/dev/null | line "1ppe chunk $c"
done
done

# round 5: K2's two shapes after the thread-per-glowworm kernel learnt to keep its verdicts: which one for which launch size?
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s: %.2f M evals/s, %.3f ms per step' % ('$1', d['value'] / 1e6, d['ms_per_step']))"; }
for round in 1 2; do
for s in 1024 512 256 128 32; do
  for k2 in single phased; do
    LIGHTDOCK_GSO_K2=$k2 timeout 300 python bench.py --workload gso-1ppe --swarms $s --steps 40 --warmup 6 --cpu-seconds 0 2>/dev/null | line "gso-1ppe $s swarms, K2 $k2"
  done
done
for k2 in single phased; do
  LIGHTDOCK_GSO_K2=$k2 timeout 300 python bench.py --workload gso-1k4c --cpu-seconds 0 2>/dev/null | line "gso-1k4c 64 swarms, K2 $k2"
done
done

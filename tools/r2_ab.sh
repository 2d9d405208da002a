cd $GRAFT_REPO_ROOT
for sp in 1 2 3 4; do
  for w in 1ppe gso-1ppe; do
  echo "== split $sp $w: $(LIGHTDOCK_TILED_SPLIT=$sp timeout 200 python bench.py --workload $w --cpu-seconds 0 --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms']))")"
  done
done
for c in 1 2; do echo "== cells $c 1k4c: $(LIGHTDOCK_PACKED_CELLS=$c timeout 200 python bench.py --cpu-seconds 0 --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms']))")"; done

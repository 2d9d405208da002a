cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2_pytest.log
for w in 1k4c gso-1ppe gso-1k4c; do
  echo "== $w"; timeout 400 python bench.py --workload $w --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s kernel %.3f ms step %.3f ms' % (d['value'], d['roofline']['kernel_ms'], d['ms_per_step']), d['config'].get('k1_k2_split'))"
done
timeout 100 python bench.py --workload 1ppe --cpu-seconds 0 2>&1 | tail -1 | cut -c1-150

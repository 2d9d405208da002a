cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2_pytest.log
for env in "LIGHTDOCK_PACKED_CELLS=2" "LIGHTDOCK_PACKED_CELLS=1" "LIGHTDOCK_DFIRE_KERNEL=tiled"; do
  for i in 1 2; do
  echo "== $env"; env $env timeout 120 python bench.py --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s kernel %.3f ms blocks %s name %s' % (d['value'], d['roofline']['kernel_ms'], d['config']['mean_8x8_blocks_evaluated'], d['roofline']['kernel']))"
  done
done
env LIGHTDOCK_PACKED_CELLS=2 timeout 120 python bench.py --workload 1ppe --batch 65536 --cpu-seconds 0 2>&1 | tail -1 | cut -c1-300
env LIGHTDOCK_DFIRE_KERNEL=tiled timeout 120 python bench.py --workload 1ppe --batch 65536 --cpu-seconds 0 2>&1 | tail -1 | cut -c1-300

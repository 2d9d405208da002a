cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_pytest.log
timeout 200 python bench.py 2>&1 | tail -1
timeout 100 python bench.py --workload 1ppe --batch 65536 --cpu-seconds 0 2>&1 | tail -1 | cut -c1-200

cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r2_pytest.log | cut -c1-250

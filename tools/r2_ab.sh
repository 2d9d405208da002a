cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2_pytest.log

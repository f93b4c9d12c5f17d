cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r3_tests.log 2>&1
grep -E "^E |^FAILED|passed|failed" gpurun_out/r3_tests.log | head -20

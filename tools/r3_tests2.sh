cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -k "work_counters or equal_distance or schedule_knobs" > gpurun_out/r3_tests2.log 2>&1
grep -E "^E |^FAILED|passed|failed" gpurun_out/r3_tests2.log | head -20

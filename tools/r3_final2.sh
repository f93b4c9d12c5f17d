cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu > gpurun_out/r3_final2_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3_final2_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 5 300 python tools/photon_build_probe.py 2>&1 | tail -3

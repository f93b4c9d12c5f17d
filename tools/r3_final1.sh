cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu > gpurun_out/r3_final_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3_final_tests.log
bash tools/profile_r03.sh r03_v2_c4 teapot 1920 1080 256 200000 > gpurun_out/r3_prof_v2_c4.log 2>&1; tail -12 gpurun_out/r3_prof_v2_c4.log
timeout -k 5 300 python tools/stripe_probe.py 1 2 4 8 > gpurun_out/r3_stripe.log 2>&1; cat gpurun_out/r3_stripe.log

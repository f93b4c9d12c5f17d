cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_r03.sh r03_v1 > gpurun_out/r3_prof_c3.log 2>&1
tail -25 gpurun_out/r3_prof_c3.log

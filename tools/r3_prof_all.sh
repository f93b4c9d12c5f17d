cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_r03.sh r03_v2 > gpurun_out/r3_prof_v2_c3.log 2>&1
bash tools/profile_r03.sh r03_v2_c2 cornell 512 512 64 0 > gpurun_out/r3_prof_v2_c2.log 2>&1
tail -12 gpurun_out/r3_prof_v2_c3.log
tail -12 gpurun_out/r3_prof_v2_c2.log

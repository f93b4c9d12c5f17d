cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_r03.sh r03_v7 > gpurun_out/r3_prof_v7_c3.log 2>&1
bash tools/profile_r03.sh r03_v7_c2 cornell 512 512 64 0 > gpurun_out/r3_prof_v7_c2.log 2>&1
bash tools/profile_r03.sh r03_v7_c4 teapot 1920 1080 256 200000 > gpurun_out/r3_prof_v7_c4.log 2>&1
tail -11 gpurun_out/r3_prof_v7_c3.log
tail -4 gpurun_out/r3_prof_v7_c2.log
tail -4 gpurun_out/r3_prof_v7_c4.log
timeout -k 5 300 python tools/stripe_probe.py 1 2 4 8 > gpurun_out/r3_stripe.log 2>&1; cat gpurun_out/r3_stripe.log | grep -v amdgpu
timeout -k 5 600 python bench.py > gpurun_out/r3_bench_full.json 2> gpurun_out/r3_bench_full.err; tail -c 600 gpurun_out/r3_bench_full.json

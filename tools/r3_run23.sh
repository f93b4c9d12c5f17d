cd $GRAFT_REPO_ROOT
export GI_EXPERIMENTAL=1 GI_LIB_PATH=$PWD/exp/gprobe.so
timeout -k 5 300 python bench.py --steps 1 --warmup 0 --no-cpu --no-others --no-executed 2> gpurun_out/r3_gprobe.err > /dev/null
grep gprobe gpurun_out/r3_gprobe.err

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GI_DEBUG_STAGES=1 timeout -k 5 300 python bench.py --steps 1 --warmup 0 --no-cpu --no-others --no-executed 2> gpurun_out/r3_dbg26.err > /dev/null
grep "^\[stage\]" gpurun_out/r3_dbg26.err | head -24 | tr '\n' ';'

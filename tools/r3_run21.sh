cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GI_DEBUG_STAGES=1 GI_DEBUG_WF=1 timeout -k 5 300 python bench.py --steps 1 --warmup 0 --no-cpu --no-others --no-executed > gpurun_out/r3_dbg21.json 2> gpurun_out/r3_dbg21.err
grep -c stage gpurun_out/r3_dbg21.err
tail -150 gpurun_out/r3_dbg21.err | cut -c1-200

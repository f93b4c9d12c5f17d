cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GI_DEBUG_STAGES=1 GI_DEBUG_WF=1 timeout -k 5 300 python tools/stripe_probe.py 8 > gpurun_out/r3_tail8.log 2>&1
GI_DEBUG_STAGES=1 GI_DEBUG_WF=1 timeout -k 5 300 python tools/stripe_probe.py 1 > gpurun_out/r3_tail1.log 2>&1
tail -3 gpurun_out/r3_tail8.log

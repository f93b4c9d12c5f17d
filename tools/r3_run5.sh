cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GI_DEBUG_WF=1 timeout -k 5 300 python tools/count_probe.py caustics 1920 1080 32 200000 > gpurun_out/r3_cnt_c3.log 2>&1
GI_DEBUG_WF=1 timeout -k 5 300 python tools/count_probe.py cornell 512 512 64 0 > gpurun_out/r3_cnt_c2.log 2>&1
GI_DEBUG_WF=1 timeout -k 5 300 python tools/count_probe.py teapot 1920 1080 8 200000 > gpurun_out/r3_cnt_c4.log 2>&1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -rP -k "work_counters" 2>&1 | grep -E "executed|passed|failed|Error|assert" 
tail -1 gpurun_out/r3_cnt_c3.log gpurun_out/r3_cnt_c2.log gpurun_out/r3_cnt_c4.log

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GI_DEBUG_STAGES=1 timeout -k 5 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-others --no-executed 2> gpurun_out/r3_dbg26.err | python -c "import json,sys; d=json.load(sys.stdin); print('c3', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
grep "^\[stage\]" gpurun_out/r3_dbg26.err | tail -160 | head -45 | tr '\n' ';'

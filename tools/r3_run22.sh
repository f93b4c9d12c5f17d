cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gather or render_matches_oracle or schedule_knobs or baseline" > gpurun_out/r3_t22.log 2>&1
grep -E "^E |passed|failed" gpurun_out/r3_t22.log | head
GI_DEBUG_STAGES=1 timeout -k 5 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-others --no-executed 2> gpurun_out/r3_dbg22.err | python -c "import json,sys; d=json.load(sys.stdin); print('c3', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
grep "^\[stage\] 4" gpurun_out/r3_dbg22.err | tail -10 | tr '\n' ' '
timeout -k 5 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-others --no-executed --scene teapot | python -c "import json,sys; d=json.load(sys.stdin); print('c4', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
timeout -k 5 300 python tools/stripe_probe.py 1 8 2>&1 | grep -v amdgpu

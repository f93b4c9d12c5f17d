cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quick_descent or descent_variants or schedule_knobs" > gpurun_out/r3_t20.log 2>&1
grep -E "^E |passed|failed" gpurun_out/r3_t20.log | head
run() {
  tag=$1
  timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --no-executed | python -c "import json,sys; d=json.load(sys.stdin); print('$tag c3', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
  timeout -k 5 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-others --no-executed --scene teapot | python -c "import json,sys; d=json.load(sys.stdin); print('$tag c4', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
}
run jump
GI_DESCENT_JUMP=0 run nojump

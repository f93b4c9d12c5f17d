cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "schedule_knobs or photon or gather or render_matches_oracle or baseline" 2>&1 | tail -3
for fd in 1 0; do
  export GI_FAST_DESCENT=$fd
  timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --no-executed | python -c "import json,sys; d=json.load(sys.stdin); print('fd $fd c3', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
  timeout -k 5 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-others --no-executed --scene teapot | python -c "import json,sys; d=json.load(sys.stdin); print('fd $fd c4', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
done

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "schedule_knobs or work_counters or trace_matches or visible_matches or render_matches_oracle" > gpurun_out/r3_t9.log 2>&1
grep -E "^E |passed|failed" gpurun_out/r3_t9.log | head
for eb in 1 0; do
  export GI_ENTITY_BOXES=$eb
  timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --no-executed | python -c "import json,sys; d=json.load(sys.stdin); print('eb $eb c3', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
  timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --no-executed --scene cornell --width 512 --height 512 --spp 64 --photons 0 | python -c "import json,sys; d=json.load(sys.stdin); print('eb $eb c2', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
  timeout -k 5 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-others --no-executed --scene teapot | python -c "import json,sys; d=json.load(sys.stdin); print('eb $eb c4', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
done

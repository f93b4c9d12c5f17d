cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  tag=$1
  timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --no-executed | python -c "import json,sys; d=json.load(sys.stdin); print('$tag c3', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
  timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --no-executed --scene cornell --width 512 --height 512 --spp 64 --photons 0 | python -c "import json,sys; d=json.load(sys.stdin); print('$tag c2', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
  timeout -k 5 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-others --no-executed --scene teapot | python -c "import json,sys; d=json.load(sys.stdin); print('$tag c4', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
}
GI_SORT_SHADE=1 run lo0
GI_SORT_SHADE=1 GI_SORT_SHADE_LO=5 run lo5
GI_SORT_SHADE=1 GI_SORT_SHADE_LO=9 run lo9
GI_SORT_SHADE=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "render_matches_oracle or schedule_knobs" 2>&1 | tail -2

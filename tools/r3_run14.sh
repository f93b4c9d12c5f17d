cd $GRAFT_REPO_ROOT
for os in 1 0; do
  export GI_OWN_SORT=$os
  timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --no-executed | python -c "import json,sys; d=json.load(sys.stdin); print('own $os c3', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
  timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --no-executed --scene cornell --width 512 --height 512 --spp 64 --photons 0 | python -c "import json,sys; d=json.load(sys.stdin); print('own $os c2', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
  timeout -k 5 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-others --no-executed --scene teapot | python -c "import json,sys; d=json.load(sys.stdin); print('own $os c4', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
done
timeout -k 5 300 python tools/stripe_probe.py 8 2>&1 | tail -1
GI_OWN_SORT=1 timeout -k 5 300 python tools/stripe_probe.py 8 2>&1 | tail -1

cd $GRAFT_REPO_ROOT
for sh in 0.6 0.8 0.95 2.0; do
  export GI_CUSE_SHARE=$sh
  timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --no-executed | python -c "import json,sys; d=json.load(sys.stdin); print('share $sh c3', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
  timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --no-executed --scene cornell --width 512 --height 512 --spp 64 --photons 0 | python -c "import json,sys; d=json.load(sys.stdin); print('share $sh c2', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
  timeout -k 5 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-others --no-executed --scene teapot | python -c "import json,sys; d=json.load(sys.stdin); print('share $sh c4', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
done

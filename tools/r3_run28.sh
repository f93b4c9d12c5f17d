cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quick_descent or schedule_knobs" 2>&1 | tail -2
for so in pj5.so base pj7.so; do
  if [ "$so" = base ]; then unset GI_LIB_PATH; else export GI_EXPERIMENTAL=1 GI_LIB_PATH=$PWD/exp/$so; fi
  timeout -k 5 150 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-others --no-executed | python -c "import json,sys; d=json.load(sys.stdin); print('$so c3', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
  timeout -k 5 250 python3 bench.py --steps 1 --warmup 1 --no-cpu --no-others --no-executed --scene teapot | python -c "import json,sys; d=json.load(sys.stdin); print('$so c4', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
done

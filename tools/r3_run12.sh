cd $GRAFT_REPO_ROOT
bash tools/exp_bench2.sh base shpf.so base shpf.so
for so in base shpf.so; do
  if [ "$so" = base ]; then unset GI_LIB_PATH; else export GI_EXPERIMENTAL=1 GI_LIB_PATH=$PWD/exp/$so; fi
  timeout -k 5 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-others --no-executed --scene teapot | python -c "import json,sys; d=json.load(sys.stdin); print('$so c4', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
done

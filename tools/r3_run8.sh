cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "schedule_knobs or render_matches_oracle or baseline_configs or eight_bit or small_pool" 2>&1 | tail -3
for ct in 1 0; do
  export GI_CAM_TABLES=$ct
  timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --no-executed | python -c "import json,sys; d=json.load(sys.stdin); print('cam $ct', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
  timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --no-executed --scene cornell --width 512 --height 512 --spp 64 --photons 0 | python -c "import json,sys; d=json.load(sys.stdin); print('cam $ct', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
done
GI_DEBUG_STAGES=1 timeout -k 5 300 python tools/stripe_probe.py 1 2>&1 | grep "^\[stage\] 1 " | tail -12 | head -2

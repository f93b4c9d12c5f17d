set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -rP -k "descent or work_counters or schedule_knobs" > gpurun_out/r3_t3.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t3.log
grep -E "executed|passed|failed|rc=|Error|assert|^[a-z_0-9]+ \(" gpurun_out/r3_t3.log | head -40
for lz in 1 0; do
  export GI_LAZY_DESCENT=$lz
  timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others > gpurun_out/r3_lz${lz}_c3.json 2>> gpurun_out/r3_lz.err
  timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --scene cornell --width 512 --height 512 --spp 64 --photons 0 > gpurun_out/r3_lz${lz}_c2.json 2>> gpurun_out/r3_lz.err
  timeout -k 5 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-others --scene teapot > gpurun_out/r3_lz${lz}_c4.json 2>> gpurun_out/r3_lz.err
done
python - <<'PY'
import json
for lz in (1,0):
  for c in ('c3','c2','c4'):
    try:
      d=json.load(open(f'gpurun_out/r3_lz{lz}_{c}.json'))
      print('lazy',lz,c, round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})
    except Exception as e: print('lazy',lz,c,'failed',e)
PY

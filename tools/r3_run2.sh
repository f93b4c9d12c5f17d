set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -rP -k "work_counters or baseline_configs or 256_spp" > gpurun_out/r3_t2.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t2.log
grep -E "rmse|executed|passed|failed|rc=|Error|assert" gpurun_out/r3_t2.log | head -40
timeout -k 5 400 python bench.py --steps 3 --warmup 1 > gpurun_out/r3_bench0.json 2> gpurun_out/r3_bench0.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench0.json'))
print(d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})
for o in d.get('other_configs',[]): print(o['workload'][:30], round(o['value'],1), {k:round(v,1) for k,v in o['roofline']['stage_ms'].items()})
PY

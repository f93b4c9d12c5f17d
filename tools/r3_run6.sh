cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu > gpurun_out/r3_t6.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t6.log
tail -4 gpurun_out/r3_t6.log
timeout -k 5 500 python bench.py --steps 3 --warmup 1 > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench1.json'))
r=d['roofline']
print(d['value'], d['ms_per_step'], {k:round(v,1) for k,v in r['stage_ms'].items()})
print('dominant', r['kernel'], r['bound'], r['frac'], 'lanes', r['lanes_per_valu'], 'hbm', r['hbm_frac'])
print('executed vs ref', r.get('executed_work',{}).get('vs_reference'))
for o in d.get('other_configs',[]): print(o['workload'][:30], round(o['value'],1), {k:round(v,1) for k,v in o['roofline']['stage_ms'].items()}, o['roofline'].get('executed_work',{}).get('vs_reference'))
PY

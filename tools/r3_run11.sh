cd $GRAFT_REPO_ROOT
bash tools/exp_bench2.sh base fullsec.so base fullsec.so
timeout -k 5 500 python bench.py --steps 3 --warmup 1 > gpurun_out/r3_bench2.json 2> gpurun_out/r3_bench2.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench2.json'))
r=d['roofline']
print(d['value'], d['ms_per_step'], {k:round(v,1) for k,v in r['stage_ms'].items()})
print('dominant', r['kernel'], r['bound'], r['frac'], 'lanes', r['lanes_per_valu'], 'hbm', r['hbm_frac'], r['evidence']['sq_counters'])
print('executed', {k:round(v,2) for k,v in r['executed_work']['per_sample'].items()}, r['executed_work'].get('vs_reference'))
print('cpu', d['cpu_baseline']['value'], d['config'].get('rmse_vs_oracle_on_cpu_rows'))
for o in d.get('other_configs',[]): print(o['workload'][:30], round(o['value'],1), {k:round(v,2) for k,v in o['roofline']['executed_work']['per_sample'].items()}, o['roofline']['executed_work'].get('vs_reference'), o['cpu_baseline']['value'], o['rmse_vs_oracle_on_cpu_rows'], o['roofline']['frac'], o['roofline']['lanes_per_valu'])
PY

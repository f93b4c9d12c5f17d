cd $GRAFT_REPO_ROOT
GI_BENCH_REHEARSAL=1 timeout -k 5 600 python bench.py --gpus 2 --steps 2 --warmup 1 --spp 32 > gpurun_out/r3_rehearsal.json 2> gpurun_out/r3_rehearsal.err; echo "rehearsal rc=$?"
tail -3 gpurun_out/r3_rehearsal.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_rehearsal.json'))
print({k:d[k] for k in ('metric','value','n_gpus','steps','scaling')}, d['config']['sharding'], d['roofline'].get('bound'), d['roofline'].get('frac'), 'executed' in str(d['roofline'].keys()))
PY

cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/rs_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rs_prof -o rs -- python3 bench.py --steps 2 --warmup 0 --no-cpu --no-others --no-executed > gpurun_out/rs_prof/bench.json 2> gpurun_out/rs_prof/err.txt
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/rs_prof/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if n.startswith('k_rs') or 'k_st_compact' in n: print(n[:40], r['Calls'], float(r['TotalDurationNs'])/1e6/2, 'ms/frame', float(r['AverageNs'])/1e3, 'us avg', float(r['MaxNs'])/1e6,'ms max')
PY
rm -rf gpurun_out/rs_prof

#!/bin/bash
# tuning aid: the benchmark frame for several hand-over points of the finisher (GI_FINISH_THRESHOLD = paths left)
for r in "$@"; do
  export GI_FINISH_THRESHOLD=$r
  timeout -k 5 150 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-others --no-executed > gpurun_out/fin_$r.json 2> gpurun_out/fin_$r.err
  python3 -c "
import json
d=json.load(open('gpurun_out/fin_$r.json')); print('finish_threshold $r', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
done

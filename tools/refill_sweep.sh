#!/bin/bash
# tuning aid: the benchmark frame and the closed-box frame for several refill thresholds of k_st_trace (GI_REFILL_MIN)
for r in "$@"; do
  export GI_REFILL_MIN=$r
  timeout -k 5 150 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-others --no-executed > gpurun_out/refill_$r.json 2> gpurun_out/refill_$r.err
  timeout -k 5 150 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-others --no-executed --scene cornell --width 512 --height 512 --spp 64 --photons 0 > gpurun_out/refill_c_$r.json 2>> gpurun_out/refill_$r.err
  python3 -c "
import json
for f in ('gpurun_out/refill_$r.json','gpurun_out/refill_c_$r.json'):
    d=json.load(open(f)); print('refill_min $r', d['config']['workload'][:24], round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
done

#!/bin/bash
# tuning aid: benchmark frame, closed box and teapot with the shadow walks inside the shade kernel (0) or in k_st_shadow (1)
for r in "$@"; do
  export GI_DEFER_SHADOWS=$r
  timeout -k 5 150 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-others > gpurun_out/defer_$r.json 2> gpurun_out/defer_$r.err
  timeout -k 5 150 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-others --scene cornell --width 512 --height 512 --spp 64 --photons 0 > gpurun_out/defer_c_$r.json 2>> gpurun_out/defer_$r.err
  timeout -k 5 150 python3 bench.py --steps 1 --warmup 1 --no-cpu --no-others --scene teapot --width 1920 --height 1080 --spp 64 --photons 200000 > gpurun_out/defer_t_$r.json 2>> gpurun_out/defer_$r.err
  python3 -c "
import json
for f in ('gpurun_out/defer_$r.json','gpurun_out/defer_c_$r.json','gpurun_out/defer_t_$r.json'):
    d=json.load(open(f)); print('defer $r', d['config']['workload'][:24], round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
done

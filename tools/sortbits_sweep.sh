#!/bin/bash
# tuning aid: how many low bits of the coherence key the sort of the continuing rays may ignore (GI_SORT_LO_BIT)
for r in "$@"; do
  export GI_SORT_LO_BIT=$r
  timeout -k 5 150 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-others > gpurun_out/sb_$r.json 2> gpurun_out/sb_$r.err
  timeout -k 5 150 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-others --scene cornell --width 512 --height 512 --spp 64 --photons 0 > gpurun_out/sb_c_$r.json 2>> gpurun_out/sb_$r.err
  python3 -c "
import json
for f in ('gpurun_out/sb_$r.json','gpurun_out/sb_c_$r.json'):
    d=json.load(open(f)); print('sort_lo_bit $r', d['config']['workload'][:24], round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
done

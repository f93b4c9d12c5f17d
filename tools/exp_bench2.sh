#!/bin/bash
# tuning aid: like exp_bench.sh, on the benchmark frame and the closed box
for so in "$@"; do
  if [ "$so" = base ]; then unset GI_LIB_PATH; else export GI_EXPERIMENTAL=1 GI_LIB_PATH=$PWD/exp/$so; fi
  timeout -k 5 150 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-others > gpurun_out/exp_$so.json 2> gpurun_out/exp_$so.err
  timeout -k 5 150 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-others --scene cornell --width 512 --height 512 --spp 64 --photons 0 > gpurun_out/exp_c_$so.json 2>> gpurun_out/exp_$so.err
  python3 -c "
import json
for f in ('gpurun_out/exp_$so.json','gpurun_out/exp_c_$so.json'):
    d=json.load(open(f)); print('$so', d['config']['workload'][:22], round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
done

#!/bin/bash
# tuning aid: the benchmark frame with experimental builds of the library (exp/*.so, GI_LIB_PATH); prints stage times per variant
for so in "$@"; do
  if [ "$so" = base ]; then unset GI_LIB_PATH; else export GI_EXPERIMENTAL=1 GI_LIB_PATH=$PWD/exp/$so; fi
  timeout -k 5 150 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-others > gpurun_out/exp_$so.json 2> gpurun_out/exp_$so.err
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/exp_$so.json')); print('$so', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"
done

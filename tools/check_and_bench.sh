#!/bin/bash
# tuning aid for the GPU box: the whole GPU test suite, then the three BASELINE frames (configs 3, 2, 4) with their stage times and executed-work counters
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/gpu_tests.log 2>&1
grep -E "^E |^FAILED|passed|failed" gpurun_out/gpu_tests.log | head -20
show='import json,sys; d=json.load(sys.stdin); e=d["roofline"].get("executed_work",{}).get("per_sample",{}); print(sys.argv[1], round(d["value"],1), {k:round(v,1) for k,v in d["roofline"]["stage_ms"].items()}, {k:round(v,2) for k,v in e.items() if k.startswith(("trace_","shadow_")) and not k.endswith(("rays","walks"))})'
timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others | python -c "$show" c3
timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --scene cornell --width 512 --height 512 --spp 64 --photons 0 | python -c "$show" c2
timeout -k 5 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-others --scene teapot | python -c "$show" c4

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3_t18.log 2>&1
grep -E "^E |passed|failed" gpurun_out/r3_t18.log | head
run() {
  tag=$1
  timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others | python -c "import json,sys; d=json.load(sys.stdin); e=d['roofline'].get('executed_work',{}).get('per_sample',{}); print('$tag c3', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()}, {k:round(e[k],2) for k in ('trace_records','trace_content_boxes','trace_leaves','trace_tris','trace_entity_boxes','shadow_records','shadow_leaves','shadow_entity_boxes','shadow_tris') if k in e})"
  timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --scene cornell --width 512 --height 512 --spp 64 --photons 0 | python -c "import json,sys; d=json.load(sys.stdin); e=d['roofline'].get('executed_work',{}).get('per_sample',{}); print('$tag c2', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()}, {k:round(e[k],2) for k in ('trace_records','trace_content_boxes','trace_leaves','trace_tris','trace_entity_boxes','shadow_records','shadow_leaves','shadow_entity_boxes','shadow_tris') if k in e})"
  timeout -k 5 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-others --scene teapot | python -c "import json,sys; d=json.load(sys.stdin); e=d['roofline'].get('executed_work',{}).get('per_sample',{}); print('$tag c4', round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()}, {k:round(e[k],2) for k in ('trace_records','trace_content_boxes','trace_leaves','trace_tris','trace_entity_boxes','shadow_records','shadow_leaves','shadow_entity_boxes','shadow_tris') if k in e})"
}
run now

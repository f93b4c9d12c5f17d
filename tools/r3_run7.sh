cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 5 300 python tools/count_probe.py caustics 1920 1080 256 200000 2>&1 | tail -2
timeout -k 5 300 python tools/count_probe.py cornell 512 512 64 0 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "work_counters or full_size or baseline" 2>&1 | tail -3
for i in 1 2; do timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --no-executed | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"; done
timeout -k 5 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --no-executed --scene cornell --width 512 --height 512 --spp 64 --photons 0 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), {k:round(v,1) for k,v in d['roofline']['stage_ms'].items()})"

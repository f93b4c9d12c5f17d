cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for env in "X=1" "GI_SORT_SHADE=0" "GI_GATHER_WAVE_BELOW=0" "GI_ENTITY_BOXES=0" "GI_FLAT_CANDIDATES=0"; do
  echo "== $env"
  env $env python -m pytest tests/test_gpu_parity.py -q -m gpu -k "pool_refills or small_pool" 2>&1 | tail -1
done

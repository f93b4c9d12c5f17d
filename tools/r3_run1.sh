set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rng or pool or baseline_configs or 256_spp or rccl or small_pool" > gpurun_out/r3_t1.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t1.log
tail -5 gpurun_out/r3_t1.log
GI_EXPERIMENTAL=1 GI_LIB_PATH=$PWD/exp/libgi_div.so timeout -k 5 300 python tools/divergence_probe.py > gpurun_out/r3_div0.log 2>&1
cat gpurun_out/r3_div0.log

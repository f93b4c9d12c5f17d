cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "render_matches_oracle or schedule_knobs or fog or textured or two_lights or alpha" 2>&1 | tail -2
bash tools/exp_bench2.sh base dw2.so
